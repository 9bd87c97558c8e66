// Register-resident EGC aggregate+combine kernel family for power-of-two layer shapes (gfx950).
//
// Same contract as the generic kernels in egc_aggregate.hip (and the same reference call sites:
// layers.py:109-138,191-225; optimized_layers.py:183-278), specialised for
//     B*L == 4 * LPR with LPR in {16, 32, 64},  L a power of two >= 4,  weightings laid out [h][b][a],
//     A <= 4, weight nonlinearity in {none, sigmoid, hardtanh}
// -- which covers the north-star shape (d=128, H=8, B=4, A=4: LPR=16, L=16).
//
// Work decomposition -- ONE launch, two roles selected by blockIdx:
//   * short rows (<= EGC_LONG_ROW_THRESHOLD entries): one LANE GROUP per row.  A basis row is LPR
//     16-byte slots, so a wavefront holds G = 64/LPR rows at once (4 at the north-star shape); every
//     wave-instruction gathers one neighbour row for each of its G rows, the running aggregates of a row
//     never leave its lane group (no cross-group merge), and the per-row fixed work (self-loop term,
//     finalisation, combine, store) is paid once per G rows.  With ~15 neighbours per row this is what
//     keeps both the instruction count and the number of gathers in flight per wavefront healthy.
//   * long rows: leading blocks reduce EGC_LONG_ROW_CHUNK-entry chunks with all G groups of a wavefront
//     splitting the entries; the wavefront that completes a row's last chunk (agent-scope release ->
//     arrival counter -> acquire; cdna guide Guideline 16) merges the partial records in chunk order
//     (deterministic) and finishes the row with the same epilogue.
// Epilogue (no LDS round trip for the aggregates): a lane holds 4 columns (one basis b, channels
// l..l+3) of every aggregator; for head h it forms sum_a w[h][b][a] * agg_a, a butterfly over the lanes
// that share l sums over b, and the lane with b == h mod B keeps the result -> each lane ends up with
// ceil(H/B) 16-byte pieces of the output row.  The weightings row is staged through LDS once per row.
// Instruction diet: wave-instructions whose G neighbour slots are all valid fold without masks
// (out-of-range buffer offsets return 0, neutral for the sums); only ragged tails and rows containing
// self-entries take the masked fold; cross-lane moves are ds_bpermute with precomputed byte addresses;
// v_max/v_min are emitted raw (no canonicalisation).
#include "egc_aggregate_dev.h"

namespace egc {

constexpr int AMAX = 4;     // aggregators supported by the register-resident combine
constexpr int FU = 4;       // neighbour-row loads in flight per lane group
constexpr int HPB_MAX = 4;  // ceil(H / B) supported

// Which optional running aggregates a layer needs (template mask: unused ones cost no registers).
constexpr int NEED_SQ = 1;   // sum of squares  (var, std)
constexpr int NEED_MN = 2;   // running minimum (min)

__device__ inline float vmax_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline float vmin_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline f4 f4_vmax(f4 a, f4 b) {
  return f4{vmax_raw(a.x, b.x), vmax_raw(a.y, b.y), vmax_raw(a.z, b.z), vmax_raw(a.w, b.w)};
}
__device__ inline f4 f4_vmin(f4 a, f4 b) {
  return f4{vmin_raw(a.x, b.x), vmin_raw(a.y, b.y), vmin_raw(a.z, b.z), vmin_raw(a.w, b.w)};
}

template <int NEED>
struct FAcc {
  f4 sum, mx, ws;
  f4 sq, mn;  // only touched when NEED says so; dead otherwise
  __device__ inline void init() {
    sum = 0.f; ws = 0.f; mx = -INFINITY;
    if constexpr (NEED & NEED_SQ) sq = 0.f;
    if constexpr (NEED & NEED_MN) mn = INFINITY;
  }
};

__device__ inline f4 f4_sqr_rn(f4 v) {
  return f4{__fmul_rn(v.x, v.x), __fmul_rn(v.y, v.y), __fmul_rn(v.z, v.z), __fmul_rn(v.w, v.w)};
}

// Every lane of the wave-instruction holds a valid, non-excluded neighbour slot.
template <int NEED>
__device__ inline void fold_plain(FAcc<NEED>& acc, f4 v, float w) {
  acc.sum += v;
  acc.mx = f4_vmax(acc.mx, v);
  acc.ws = f4_fma(f4{w, w, w, w}, v, acc.ws);
  if constexpr (NEED & NEED_SQ) acc.sq += f4_sqr_rn(v);
  if constexpr (NEED & NEED_MN) acc.mn = f4_vmin(acc.mn, v);
}

template <int NEED>
__device__ inline void fold_masked(FAcc<NEED>& acc, f4 v, float w, bool in_x, bool in_y) {
  const f4 vx = in_x ? v : f4{0.f, 0.f, 0.f, 0.f};
  acc.sum += vx;
  acc.mx = f4_vmax(acc.mx, in_x ? v : f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY});
  const float wy = in_y ? w : 0.f;
  acc.ws = f4_fma(f4{wy, wy, wy, wy}, v, acc.ws);
  if constexpr (NEED & NEED_SQ) acc.sq += f4_sqr_rn(vx);
  if constexpr (NEED & NEED_MN) acc.mn = f4_vmin(acc.mn, in_x ? v : f4{INFINITY, INFINITY, INFINITY, INFINITY});
}

__device__ inline float bperm(int byte_addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}
__device__ inline int bperm(int byte_addr, int v) { return __builtin_amdgcn_ds_bpermute(byte_addr, v); }
__device__ inline f4 bperm(int byte_addr, f4 v) {
  return f4{bperm(byte_addr, v.x), bperm(byte_addr, v.y), bperm(byte_addr, v.z), bperm(byte_addr, v.w)};
}

// Merge the G lane groups (every lane ends with the aggregates of its slot over all entries).
template <int LPR_LOG2, int NEED>
__device__ inline void all_reduce_groups(FAcc<NEED>& acc, int lane) {
#pragma unroll
  for (int off = 1 << LPR_LOG2; off < 64; off <<= 1) {
    const int addr = (lane ^ off) << 2;
    acc.sum += bperm(addr, acc.sum);
    acc.ws += bperm(addr, acc.ws);
    acc.mx = f4_vmax(acc.mx, bperm(addr, acc.mx));
    if constexpr (NEED & NEED_SQ) acc.sq += bperm(addr, acc.sq);
    if constexpr (NEED & NEED_MN) acc.mn = f4_vmin(acc.mn, bperm(addr, acc.mn));
  }
}

struct FastRsrc {
  __amdgpu_buffer_rsrc_t bases, out;
};

// ---------------------------------------------------------------------------------------------
// Epilogue shared by both roles.  Per lane group: `row` (same in all lanes of the group), the group's
// merged aggregates, its entry count `deg` and self-entry count `nself`; `store` masks the output.
// ---------------------------------------------------------------------------------------------
template <int LPR_LOG2, int HPB, int NEED>
__device__ inline void finish_group(const AggArgs& a, const FastRsrc& R, int lane, int row, bool row_ok, FAcc<NEED>& acc,
                                    int deg, int nself, float dis_i, f4 vself, bool has_self, const f4 (&wpre)[2],
                                    bool store, float* lds_w, const float* lds_bias) {
  constexpr int LPR = 1 << LPR_LOG2;
  const int g = lane >> LPR_LOG2;
  const int q = lane & (LPR - 1);
  const int b = q >> a.lpb_log2;
  const int l4 = q & ((1 << a.lpb_log2) - 1);

  // (1) the row's weightings (nonlinearity applied) -> this group's LDS strip, 32 bytes per lane
  float* wl = lds_w + g * a.w_lds_stride;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c0 = (q + k * LPR) * 4;
    if (c0 < a.W) {
      f4 t = wpre[k];
      if (a.act == EGC_ACT_SIGMOID) {
        t = f4{1.0f / (1.0f + expf(-t.x)), 1.0f / (1.0f + expf(-t.y)), 1.0f / (1.0f + expf(-t.z)),
               1.0f / (1.0f + expf(-t.w))};
      } else if (a.act == EGC_ACT_HARDTANH) {
        t = f4{fminf(fmaxf(t.x, -1.f), 1.f), fminf(fmaxf(t.y, -1.f), 1.f), fminf(fmaxf(t.z, -1.f), 1.f),
               fminf(fmaxf(t.w, -1.f), 1.f)};
      }
      *reinterpret_cast<f4*>(wl + c0) = t;
    }
  }

  // (2) self-loop term and aggregator finalisation
  int cnt = deg;
  if (a.x_looped) cnt = deg - nself + (has_self ? 1 : 0);
  if ((a.x_looped || a.y_looped) && has_self) {
    if (a.x_looped && a.y_looped) fold_plain<NEED>(acc, vself, dis_i * dis_i);  // vself is 0 where !has_self
    else fold_masked<NEED>(acc, vself, dis_i * dis_i, a.x_looped != 0, a.y_looped != 0);
  }
  const float cntf = (float)max(cnt, 1);
  const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
  f4 mean = zero, var = zero;
  if constexpr (NEED & NEED_SQ) {
    // exact divisions: var of identical neighbours must cancel to exactly 0 (see egc_aggregate_dev.h)
    mean = f4_div(acc.sum, cntf);
    var = f4_var(f4_div(acc.sq, cntf), mean);
  } else if (a.need_mean) {
    // no var/std in this layer: one reciprocal instead of four IEEE divisions (<= 1 ulp from sum / cnt)
    const float inv = 1.0f / cntf;
    mean = acc.sum * f4{inv, inv, inv, inv};
  }
  f4 val[AMAX];
#pragma unroll
  for (int t = 0; t < AMAX; ++t) {
    val[t] = zero;
    if (t < a.A) {  // wave-uniform
      switch (a.aggr[t]) {
        case EGC_AGGR_SUM: val[t] = acc.sum; break;
        case EGC_AGGR_MEAN: val[t] = mean; break;
        case EGC_AGGR_MAX: val[t] = cnt > 0 ? acc.mx : zero; break;
        case EGC_AGGR_MIN: if constexpr (NEED & NEED_MN) val[t] = cnt > 0 ? acc.mn : zero; break;
        case EGC_AGGR_VAR: val[t] = var; break;
        case EGC_AGGR_STD: val[t] = f4_std(var); break;
        default: val[t] = acc.ws; break;  // EGC_AGGR_SYMNORM
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // weight strip visible to the whole wavefront

  // (3) combine: for head h = hb*B + bb every lane forms its (b, l..l+3) share, the butterfly sums over
  //     b, and the lane with b == bb keeps the head's 4 channels
  f4 o[HPB];
#pragma unroll
  for (int hb = 0; hb < HPB; ++hb) {
    o[hb] = zero;
    for (int bb = 0; bb < a.B; ++bb) {
      const int h = hb * a.B + bb;
      if (h >= a.H) break;  // wave-uniform
      const float* wp = wl + (h * a.B + b) * a.A;
      f4 part;
      if (a.A == 4) {  // wave-uniform
        const f4 wv = *reinterpret_cast<const f4*>(wp);
        part = val[0] * f4{wv.x, wv.x, wv.x, wv.x};
        part = f4_fma(f4{wv.y, wv.y, wv.y, wv.y}, val[1], part);
        part = f4_fma(f4{wv.z, wv.z, wv.z, wv.z}, val[2], part);
        part = f4_fma(f4{wv.w, wv.w, wv.w, wv.w}, val[3], part);
      } else {
        part = zero;
#pragma unroll
        for (int t = 0; t < AMAX - 1; ++t)
          if (t < a.A) {
            const float w = wp[t];
            part = f4_fma(f4{w, w, w, w}, val[t], part);
          }
      }
      for (int off = 1 << a.lpb_log2; off < LPR; off <<= 1) part += bperm((lane ^ off) << 2, part);
      if (b == bb) o[hb] = part;
    }
  }
  // (4) bias + store: lane (b, l4) owns out[row, (hb*B + b)*L + 4*l4 ..+3]
  const unsigned orow = (unsigned)row * (unsigned)a.F_out * 4u;
#pragma unroll
  for (int hb = 0; hb < HPB; ++hb) {
    const int h = hb * a.B + b;
    const bool mine = store && row_ok && h < a.H;
    const int oc = h * a.L + 4 * l4;
    const f4 r = o[hb] + *reinterpret_cast<const f4*>(lds_bias + (mine ? oc : 0));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, r), R.out, mine ? orow + (unsigned)oc * 4u : OOB, 0, 0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // strip reads done before the next row overwrites it
}

// Row-only operands of the epilogue (weightings row as 2 x 16 bytes per lane, own basis slot).
template <int LPR_LOG2>
__device__ inline void load_row_operands(const AggArgs& a, const FastRsrc& R, int lane, int row, bool row_ok,
                                         f4 (&wpre)[2], f4& vself, bool& has_self) {
  constexpr int LPR = 1 << LPR_LOG2;
  const int q = lane & (LPR - 1);
  const int nloop = a.loops_all ? a.n_nodes : (*a.max_index + 1);
  has_self = row_ok && row < nloop;
  const bool want_self = (a.x_looped || a.y_looped) && has_self;
  vself = load_slot(R.bases, want_self ? (unsigned)row * (unsigned)a.ldb * 4u + (unsigned)q * 16u : OOB);
  const float* wrow = a.weightings + (int64_t)(row_ok ? row : 0) * a.W;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c0 = (q + k * LPR) * 4;
    wpre[k] = f4{0.f, 0.f, 0.f, 0.f};
    if (c0 + 3 < a.W) wpre[k] = *reinterpret_cast<const f4*>(wrow + c0);
    else if (c0 < a.W) {  // W not a multiple of 4: ragged last piece
      wpre[k].x = wrow[c0];
      if (c0 + 1 < a.W) wpre[k].y = wrow[c0 + 1];
      if (c0 + 2 < a.W) wpre[k].z = wrow[c0 + 2];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
template <int LPR_LOG2, int HPB, int NEED>
__global__ void __launch_bounds__(256) agg_fast_kernel(AggArgs a) {
  constexpr int LPR = 1 << LPR_LOG2, G = 64 / LPR;
  extern __shared__ float smem[];
  if ((int)blockIdx.x < a.chunk_blocks && (int)blockIdx.x * 4 >= a.plan[1]) return;  // unused chunk slots
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int g = lane >> LPR_LOG2;
  const int q = lane & (LPR - 1);
  const unsigned slot_off = (unsigned)q * 16u;
  const unsigned row_bytes = (unsigned)a.ldb * 4u;
  // per-wavefront LDS: [bias F_out][G weight strips]
  float* lds_bias = smem + wave * a.lds_floats_per_wave;
  float* lds_w = lds_bias + ((a.F_out + 3) & ~3);
  for (int o = lane; o < a.F_out; o += 64) lds_bias[o] = a.bias != nullptr ? a.bias[o] : 0.f;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  FastRsrc R;
  R.bases = bases_rsrc(a);
  R.out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (unsigned)a.n_nodes * (unsigned)a.F_out * 4u, 0x00020000);
  const bool xl = a.x_looped != 0, yl = a.y_looped != 0;

  if ((int)blockIdx.x < a.chunk_blocks) {
    // ---------------- long-row chunk role: the G groups split one chunk's entries ----------------
    const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (c >= a.plan[1]) return;
    const int cap_long = a.plan[2], cap_chunks = a.plan[3];
    const int* long_row = a.plan + 4;
    const int* long_chunk0 = long_row + cap_long;
    const int* chunk_slot = long_chunk0 + cap_long;
    const int* chunk_begin = chunk_slot + cap_chunks;
    const int slot = __builtin_amdgcn_readfirstlane(chunk_slot[c]);
    const int row = __builtin_amdgcn_readfirstlane(long_row[slot]);
    const int start = __builtin_amdgcn_readfirstlane(chunk_begin[c]);
    const int row_start = __builtin_amdgcn_readfirstlane(a.rowptr[row]);
    const int row_end = __builtin_amdgcn_readfirstlane(a.rowptr[row + 1]);
    const int end = min(start + EGC_LONG_ROW_CHUNK, row_end);
    const int deg = row_end - row_start;
    const int nch = (deg + EGC_LONG_ROW_CHUNK - 1) / EGC_LONG_ROW_CHUNK;
    const float dis_i = a.dis != nullptr ? a.dis[row] : 0.f;
    FAcc<NEED> acc;
    acc.init();
    int nself = 0;
    for (int base = start; base < end; base += 64) {
      const int p = base + lane;
      const bool pv = p < end;
      const int jj = pv ? a.col[p] : row;
      const float dd = a.dis != nullptr ? a.dis[jj] : 0.f;
      const int ns = __popcll(__ballot(pv && jj == row));
      nself += ns;
      const bool masked_all = ns > 0 && (xl || yl);
      const int cnt = min(64, end - base);
      for (int t0 = 0; t0 < cnt; t0 += FU * G) {
        f4 v[FU];
        const int n_in = min(cnt - t0, FU * G);
#pragma unroll
        for (int u = 0; u < FU; ++u) {
          const int j = bperm((g + t0 + u * G) << 2, jj);
          v[u] = load_slot(R.bases, (u * G + g < n_in) ? (unsigned)j * row_bytes + slot_off : OOB);
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) {
          if (u * G >= n_in) break;
          const float w = bperm((g + t0 + u * G) << 2, dd) * dis_i;
          if (!masked_all && (u + 1) * G <= n_in) {
            fold_plain<NEED>(acc, v[u], w);
          } else {
            const bool ok = u * G + g < n_in;
            const bool is_self = bperm((g + t0 + u * G) << 2, jj) == row;
            fold_masked<NEED>(acc, v[u], w, ok && !(xl && is_self), ok && !(yl && is_self));
          }
        }
      }
    }
    all_reduce_groups<LPR_LOG2, NEED>(acc, lane);
    if (nch > 1) {
      if (g == 0) {
        f4* rec = reinterpret_cast<f4*>(a.partial) + (int64_t)c * 5 * LPR;
        rec[0 * LPR + q] = acc.sum;
        rec[2 * LPR + q] = acc.mx;
        rec[4 * LPR + q] = acc.ws;
        if constexpr (NEED & NEED_SQ) rec[1 * LPR + q] = acc.sq;
        if constexpr (NEED & NEED_MN) rec[3 * LPR + q] = acc.mn;
      }
      if (lane == 0) a.partial_nself[c] = nself;
      // publish: stores drained -> agent-scope release -> arrival counter
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int arrived = 0;
      if (lane == 0) arrived = __hip_atomic_fetch_add(&a.counters[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      arrived = __builtin_amdgcn_readfirstlane(arrived);
      if (arrived != nch - 1) return;
      // last arriver: acquire, reset the counter for the next launch, merge in chunk order
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(&a.counters[slot], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int c0 = __builtin_amdgcn_readfirstlane(long_chunk0[slot]);
      acc.init();
      for (int k0 = g; k0 < nch; k0 += G) {
        const f4* rec = reinterpret_cast<const f4*>(a.partial) + (int64_t)(c0 + k0) * 5 * LPR;
        acc.sum += rec[0 * LPR + q];
        acc.mx = f4_vmax(acc.mx, rec[2 * LPR + q]);
        acc.ws += rec[4 * LPR + q];
        if constexpr (NEED & NEED_SQ) acc.sq += rec[1 * LPR + q];
        if constexpr (NEED & NEED_MN) acc.mn = f4_vmin(acc.mn, rec[3 * LPR + q]);
      }
      all_reduce_groups<LPR_LOG2, NEED>(acc, lane);
      nself = 0;
      for (int k = lane; k < nch; k += 64) nself += a.partial_nself[c0 + k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) nself += bperm((lane ^ off) << 2, nself);
    }
    // every group now holds the whole row: all run the epilogue, group 0 stores
    f4 wpre[2], vself;
    bool has_self;
    load_row_operands<LPR_LOG2>(a, R, lane, row, true, wpre, vself, has_self);
    finish_group<LPR_LOG2, HPB, NEED>(a, R, lane, row, true, acc, deg, nself, dis_i, vself, has_self, wpre, g == 0,
                                      lds_w, lds_bias);
    return;
  }

  // ---------------- short-row role: one lane group per row, G rows per wavefront at a time ----------------
  const int Q = a.rows_per_wave;  // row groups (of G rows) per wavefront; Q * G <= 63
  const int gw = ((int)blockIdx.x - a.chunk_blocks) * 4 + wave;
  const int r0 = __builtin_amdgcn_readfirstlane(gw * Q * G);
  if (r0 >= a.n_nodes) return;
  const int rp = a.rowptr[min(r0 + lane, a.n_nodes)];  // lanes 0..Q*G hold this wavefront's row pointers
  const int grp_addr = (g << LPR_LOG2) << 2;            // ds_bpermute byte address of the group's lane 0
  // stage the first row group's column indices (lane q of group g <- entry q of row r0 + g)
  int start_n = bperm(g << 2, rp);
  int nd_n;  // entries of the group's row handled here (0 for long or out-of-range rows)
  {
    const int deg = bperm((g + 1) << 2, rp) - start_n;
    nd_n = (r0 + g < a.n_nodes && deg <= EGC_LONG_ROW_THRESHOLD) ? deg : 0;
  }
  int jj_n = q < nd_n ? a.col[start_n + q] : 0;
  for (int k = 0; k < Q; ++k) {
    const int rbase = r0 + k * G;
    if (rbase >= a.n_nodes) break;
    const int row = rbase + g;
    const bool row_ok = row < a.n_nodes;
    const int start = start_n, nd = nd_n;
    int jj = jj_n;
    if (k + 1 < Q) {  // prefetch the next row group's bounds and first LPR column indices
      start_n = bperm(((k + 1) * G + g) << 2, rp);
      const int deg = bperm(((k + 1) * G + g + 1) << 2, rp) - start_n;
      nd_n = (rbase + G + g < a.n_nodes && deg <= EGC_LONG_ROW_THRESHOLD) ? deg : 0;
      jj_n = q < nd_n ? a.col[start_n + q] : 0;
    }
    // wave-uniform trip counts: entries still valid in every group / in any group
    int maxd = nd, mind = nd;
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
      maxd = max(maxd, bperm((lane ^ off) << 2, maxd));
      mind = min(mind, bperm((lane ^ off) << 2, mind));
    }
    maxd = __builtin_amdgcn_readfirstlane(maxd);
    mind = __builtin_amdgcn_readfirstlane(mind);
    // row-only operands, issued ahead of the gathers
    f4 wpre[2], vself;
    bool has_self;
    const float dis_i = (a.dis != nullptr && row_ok) ? a.dis[row] : 0.f;
    load_row_operands<LPR_LOG2>(a, R, lane, row, row_ok, wpre, vself, has_self);

    FAcc<NEED> acc;
    acc.init();
    int nself = 0;
    for (int ts = 0; ts < maxd; ts += LPR) {
      if (ts > 0) jj = (ts + q < nd) ? a.col[start + ts + q] : 0;  // rows of more than LPR entries
      const bool pv = ts + q < nd;
      const float dd = (a.dis != nullptr && pv) ? a.dis[jj] : 0.f;
      bool masked_all = false;
      if (xl || yl) {  // self-entries are excluded from LOOPED sets: count them per group
        const unsigned long long sb = __ballot(pv && jj == row);
        masked_all = sb != 0;
        nself += __popcll((sb >> (g << LPR_LOG2)) & ((LPR == 64) ? ~0ull : ((1ull << LPR) - 1ull)));
      }
      const int cnt = min(LPR, maxd - ts);  // wave-uniform
      for (int t0 = 0; t0 < cnt; t0 += FU) {
        f4 v[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
          const int j = bperm(grp_addr + ((t0 + u) << 2), jj);
          v[u] = load_slot(R.bases, (ts + t0 + u < nd) ? (unsigned)j * row_bytes + slot_off : OOB);
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) {
          const int e = ts + t0 + u;  // wave-uniform entry index inside each group's row
          if (e >= maxd) break;
          const float w = bperm(grp_addr + ((t0 + u) << 2), dd) * dis_i;
          if (!masked_all && e < mind) {
            fold_plain<NEED>(acc, v[u], w);
          } else {
            const bool ok = e < nd;
            const bool is_self = bperm(grp_addr + ((t0 + u) << 2), jj) == row;
            fold_masked<NEED>(acc, v[u], w, ok && !(xl && is_self), ok && !(yl && is_self));
          }
        }
      }
    }
    // Opaque copy of the lane id: keeps the compiler from hoisting the epilogue's lane arithmetic out
    // of the row loop, where it would stay live across the gathers and cost occupancy.
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int deg_all = bperm((k * G + g + 1) << 2, rp) - start;
    const bool is_short = deg_all <= EGC_LONG_ROW_THRESHOLD;
    finish_group<LPR_LOG2, HPB, NEED>(a, R, ln, row, row_ok, acc, nd, nself, dis_i, vself, has_self, wpre, is_short,
                                      lds_w, lds_bias);
  }
}

bool fast_path_supported(const AggArgs& a, int layout, int chunks) {
  if (chunks != 1 || layout != EGC_LAYOUT_HBA || a.act == EGC_ACT_SOFTMAX) return false;
  if (a.slots != 16 && a.slots != 32 && a.slots != 64) return false;
  if (a.ldb != a.B * a.L) return false;
  if (a.L < 4 || (a.L & (a.L - 1)) != 0 || (a.B & (a.B - 1)) != 0) return false;
  if (a.A < 1 || a.A > AMAX) return false;
  if ((a.H + a.B - 1) / a.B > HPB_MAX) return false;
  if (a.W > 8 * a.slots) return false;  // weightings row staged as 2 x 16 bytes per lane of a group
  // the buffer descriptor addresses `out` with 32-bit byte offsets
  if ((uint64_t)a.n_nodes * (uint64_t)a.F_out * 4ull > (uint64_t)OOB) return false;
  return true;
}

template <int LPR_LOG2, int HPB, int NEED>
static int launch_one(const AggArgs& a, unsigned grid, size_t lds, hipStream_t stream) {
  agg_fast_kernel<LPR_LOG2, HPB, NEED><<<grid, 256, lds, stream>>>(a);
  EGC_LAUNCH_CHECK("agg_fast_kernel");
  return EGC_OK;
}

template <int LPR_LOG2, int HPB>
static int launch_need(const AggArgs& a, int need, unsigned grid, size_t lds, hipStream_t stream) {
  if (need == 0) return launch_one<LPR_LOG2, HPB, 0>(a, grid, lds, stream);
  return launch_one<LPR_LOG2, HPB, NEED_SQ | NEED_MN>(a, grid, lds, stream);
}

template <int LPR_LOG2>
static int launch_a(const AggArgs& a, int need, unsigned grid, size_t lds, hipStream_t stream) {
  const int hpb = (a.H + a.B - 1) / a.B;
  if (hpb <= 1) return launch_need<LPR_LOG2, 1>(a, need, grid, lds, stream);
  if (hpb <= 2) return launch_need<LPR_LOG2, 2>(a, need, grid, lds, stream);
  return launch_need<LPR_LOG2, 4>(a, need, grid, lds, stream);
}

int launch_fast(AggArgs a, int64_t n_nodes, const PlanCaps& caps, hipStream_t stream) {
  const int G = 64 / a.slots;
  int lg = 0;
  while ((4 << lg) < a.L) ++lg;
  a.lpb_log2 = lg;
  if (a.rows_per_wave <= 0) a.rows_per_wave = 4;
  if (a.rows_per_wave * G > 60) a.rows_per_wave = 60 / G;
  a.chunk_blocks = (int)ceil_div(caps.cap_chunks, 4);
  a.need_mean = a.need_var = 0;
  int need = 0;
  for (int t = 0; t < a.A; ++t) {
    if (a.aggr[t] == EGC_AGGR_MEAN || a.aggr[t] == EGC_AGGR_VAR || a.aggr[t] == EGC_AGGR_STD) a.need_mean = 1;
    if (a.aggr[t] == EGC_AGGR_VAR || a.aggr[t] == EGC_AGGR_STD) { a.need_var = 1; need |= NEED_SQ; }
    if (a.aggr[t] == EGC_AGGR_MIN) need |= NEED_MN;
  }
  a.w_lds_stride = (a.W + 3) & ~3;
  a.lds_floats_per_wave = ((a.F_out + 3) & ~3) + G * a.w_lds_stride;
  const size_t lds = (size_t)4 * a.lds_floats_per_wave * sizeof(float);
  if (lds > 64 * 1024) return EGC_ERR_UNSUPPORTED;
  const int64_t row_blocks = ceil_div(n_nodes, (int64_t)4 * a.rows_per_wave * G);
  const unsigned grid = (unsigned)(a.chunk_blocks + row_blocks);
  switch (a.slots) {
    case 16: return launch_a<4>(a, need, grid, lds, stream);
    case 32: return launch_a<5>(a, need, grid, lds, stream);
    case 64: return launch_a<6>(a, need, grid, lds, stream);
    default: return EGC_ERR_UNSUPPORTED;
  }
}

}  // namespace egc
