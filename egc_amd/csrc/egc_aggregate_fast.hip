// Register-resident EGC aggregate+combine kernel family for power-of-two layer shapes (gfx950).
//
// Same contract as the generic kernels in egc_aggregate.hip (and the same reference call sites:
// layers.py:109-138,191-225; optimized_layers.py:183-278), specialised for
//     B*L == 4 * LPR with LPR in {16, 32, 64},  L a power of two >= 4,  weightings laid out [h][b][a],
//     A <= 4, weight nonlinearity in {none, sigmoid, hardtanh}
// -- which covers the north-star shape (d=128, H=8, B=4, A=4: LPR=16, L=16).  What changes:
//   * ONE launch.  Leading blocks reduce long-row chunks; the wavefront that completes a long row's
//     last chunk (agent-scope release -> arrival counter -> acquire) merges the partials in chunk order
//     and finishes the row; the remaining blocks take `rows_per_wave` consecutive short rows each.
//   * No LDS round trip in the epilogue.  After the cross-group merge every lane group holds the whole
//     aggregated row, so group g combines heads g, g+G, ...: a lane multiplies its 4 columns of every
//     aggregator by w[h][b][0..A) (one 16-byte load when A == 4), and a butterfly over the lanes that
//     share the same l sums over b.  out is written as 16-byte pieces of one contiguous row.
//   * Instruction diet (the path is issue-sensitive: ~15 neighbours per row).  Wave-instructions whose
//     4 (G) neighbour slots are all valid are folded without masks -- out-of-range buffer offsets return
//     0, which is neutral for the sums; only the ragged tail and rows that contain self-entries take the
//     masked fold.  Cross-lane traffic uses ds_bpermute with precomputed byte addresses; row-dependent
//     addresses ride in the scalar offset of buffer loads/stores.
#include "egc_aggregate_dev.h"

namespace egc {

constexpr int AMAX = 4;     // aggregators supported by the register-resident combine
constexpr int FU = 4;       // neighbour-row loads in flight per lane group

// Which optional running aggregates a layer needs (template mask: unused ones cost no registers).
constexpr int NEED_SQ = 1;   // sum of squares  (var, std)
constexpr int NEED_MN = 2;   // running minimum (min)

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
  acc.mx = f4_max(acc.mx, v);
  acc.ws = f4_fma(f4{w, w, w, w}, v, acc.ws);
  if constexpr (NEED & NEED_SQ) acc.sq += f4_sqr_rn(v);
  if constexpr (NEED & NEED_MN) acc.mn = f4_min(acc.mn, v);
}

template <int NEED>
__device__ inline void fold_masked(FAcc<NEED>& acc, f4 v, float w, bool in_x, bool in_y) {
  const f4 vx = in_x ? v : f4{0.f, 0.f, 0.f, 0.f};
  acc.sum += vx;
  acc.mx = f4_max(acc.mx, in_x ? v : f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY});
  const float wy = in_y ? w : 0.f;
  acc.ws = f4_fma(f4{wy, wy, wy, wy}, v, acc.ws);
  if constexpr (NEED & NEED_SQ) acc.sq += f4_sqr_rn(vx);
  if constexpr (NEED & NEED_MN) acc.mn = f4_min(acc.mn, in_x ? v : f4{INFINITY, INFINITY, INFINITY, INFINITY});
}

__device__ inline float bperm(int byte_addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}
__device__ inline int bperm(int byte_addr, int v) { return __builtin_amdgcn_ds_bpermute(byte_addr, v); }
__device__ inline f4 bperm(int byte_addr, f4 v) {
  return f4{bperm(byte_addr, v.x), bperm(byte_addr, v.y), bperm(byte_addr, v.z), bperm(byte_addr, v.w)};
}

// Per-lane constants of one wavefront (everything here is row-invariant).
template <int LPR_LOG2>
struct LaneCtx {
  static constexpr int LPR = 1 << LPR_LOG2, G = 64 / LPR;
  int lane, g, q;
  unsigned slot_off;    // byte offset of this lane's 16-byte slot inside a basis row
  int grp_addr;         // g * 4: ds_bpermute byte address of staged entry g
  __device__ inline void init(int lane_) {
    lane = lane_;
    g = lane >> LPR_LOG2;
    q = lane & (LPR - 1);
    slot_off = (unsigned)q * 16u;
    grp_addr = g << 2;
  }
};

// Merge the G lane groups (every lane ends with the aggregates of its slot over all entries).
template <int LPR_LOG2, int NEED>
__device__ inline void all_reduce_groups(FAcc<NEED>& acc, int lane) {
#pragma unroll
  for (int off = 1 << LPR_LOG2; off < 64; off <<= 1) {
    const int addr = (lane ^ off) << 2;
    acc.sum += bperm(addr, acc.sum);
    acc.ws += bperm(addr, acc.ws);
    acc.mx = f4_max(acc.mx, bperm(addr, acc.mx));
    if constexpr (NEED & NEED_SQ) acc.sq += bperm(addr, acc.sq);
    if constexpr (NEED & NEED_MN) acc.mn = f4_min(acc.mn, bperm(addr, acc.mn));
  }
}

// Fold `cnt` (<= 64) staged entries: lane k holds entry k's source id `jj` and deg^-1/2 `dd`.
// `masked_all` forces the masked fold (row contains self-entries that one of the edge sets excludes).
template <int LPR_LOG2, int NEED>
__device__ inline void reduce_staged(const AggArgs& a, __amdgpu_buffer_rsrc_t rsrc, const LaneCtx<LPR_LOG2>& L, int row,
                                     int cnt, int jj, float dd, float dis_i, bool masked_all, FAcc<NEED>& acc) {
  constexpr int G = LaneCtx<LPR_LOG2>::G;
  constexpr int BATCH = FU * G;
  const unsigned row_bytes = (unsigned)a.ldb * 4u;
  for (int t0 = 0; t0 < cnt; t0 += BATCH) {
    f4 v[FU];
    const int n_in = min(cnt - t0, BATCH);  // wave-uniform
#pragma unroll
    for (int u = 0; u < FU; ++u) {
      const int j = bperm(L.grp_addr + ((t0 + u * G) << 2), jj);
      const bool ok = u * G + L.g < n_in;
      v[u] = load_slot(rsrc, ok ? (unsigned)j * row_bytes + L.slot_off : OOB);
    }
#pragma unroll
    for (int u = 0; u < FU; ++u) {
      if (u * G >= n_in) break;  // wave-uniform: nothing staged for this instruction
      const float w = bperm(L.grp_addr + ((t0 + u * G) << 2), dd) * dis_i;
      if (!masked_all && (u + 1) * G <= n_in) {
        fold_plain<NEED>(acc, v[u], w);
      } else {
        const bool ok = u * G + L.g < n_in;
        const bool is_self = bperm(L.grp_addr + ((t0 + u * G) << 2), jj) == row;
        fold_masked<NEED>(acc, v[u], w, ok && !(a.x_looped && is_self), ok && !(a.y_looped && is_self));
      }
    }
  }
}

// Reduce CSR entries [start, end) of `row`; jj0 = prefetched col[start + lane] (or -1 to load it here).
template <int LPR_LOG2, int NEED>
__device__ inline void reduce_entries(const AggArgs& a, __amdgpu_buffer_rsrc_t rsrc, const LaneCtx<LPR_LOG2>& L, int row,
                                      int start, int end, int jj0, bool have_jj0, float dis_i, FAcc<NEED>& acc,
                                      int& nself) {
  for (int base = start; base < end; base += 64) {
    const int p = base + L.lane;
    const bool pv = p < end;
    int jj;
    if (have_jj0 && base == start) jj = jj0; else jj = pv ? a.col[p] : row;
    const float dd = a.dis != nullptr ? a.dis[pv ? jj : row] : 0.f;
    const int ns = __popcll(__ballot(pv && jj == row));
    nself += ns;
    const bool masked_all = ns > 0 && (a.x_looped || a.y_looped);
    reduce_staged<LPR_LOG2, NEED>(a, rsrc, L, row, min(64, end - base), jj, dd, dis_i, masked_all, acc);
  }
}

// Self-loop term, aggregator finalisation, in-register combine, bias, store.  `acc` is merged over groups.
template <int LPR_LOG2, int HPG, int NEED>
__device__ inline void finish_fast(const AggArgs& a, __amdgpu_buffer_rsrc_t rsrc, __amdgpu_buffer_rsrc_t rsrc_w,
                                   __amdgpu_buffer_rsrc_t rsrc_out, int lane_opaque, int row, FAcc<NEED>& acc, int deg,
                                   int nself, float dis_i, const float* lds_bias) {
  constexpr int LPR = 1 << LPR_LOG2, G = 64 / LPR;
  const int g = lane_opaque >> LPR_LOG2;
  const int q = lane_opaque & (LPR - 1);
  const int b = q >> a.lpb_log2;
  const int l4 = q & ((1 << a.lpb_log2) - 1);

  // operands that depend only on the row: issue their loads first
  const int nloop = a.loops_all ? a.n_nodes : (*a.max_index + 1);
  const bool has_self = row < nloop;
  const bool want_self = (a.x_looped || a.y_looped) && has_self;
  const f4 vself = __builtin_bit_cast(
      f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, want_self ? (unsigned)q * 16u : OOB,
                                                 (unsigned)row * (unsigned)a.ldb * 4u, 0));
  float wv[HPG][AMAX];
  const unsigned wrow_off = (unsigned)row * (unsigned)a.W * 4u;  // scalar
#pragma unroll
  for (int hh = 0; hh < HPG; ++hh) {
    const int h = g + hh * G;
    const unsigned woff = h < a.H ? (unsigned)((h * a.B + b) * a.A) * 4u : OOB;
    if (a.A == 4) {  // wave-uniform
      const f4 t = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, woff, wrow_off, 0));
      wv[hh][0] = t.x; wv[hh][1] = t.y; wv[hh][2] = t.z; wv[hh][3] = t.w;
    } else {
#pragma unroll
      for (int t = 0; t < AMAX; ++t)
        wv[hh][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                  rsrc_w, (t < a.A && woff != OOB) ? woff + 4u * t : OOB, wrow_off, 0));
    }
  }

  all_reduce_groups<LPR_LOG2, NEED>(acc, lane_opaque);

  int cnt = deg;
  if (a.x_looped) cnt = deg - nself + (has_self ? 1 : 0);
  if (want_self) fold_masked<NEED>(acc, vself, dis_i * dis_i, a.x_looped != 0, a.y_looped != 0);

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
  if (a.act != EGC_ACT_NONE) {
#pragma unroll
    for (int hh = 0; hh < HPG; ++hh)
#pragma unroll
      for (int t = 0; t < AMAX; ++t) {
        float w = wv[hh][t];
        if (a.act == EGC_ACT_SIGMOID) w = 1.0f / (1.0f + expf(-w));
        else w = fminf(fmaxf(w, -1.0f), 1.0f);
        // lanes whose load was out of range hold 0 and must keep contributing 0
        wv[hh][t] = (t < a.A && g + hh * G < a.H) ? w : 0.f;
      }
  }
  // this lane's share of sum_{a} w[h][b][a] * agg[a][b*L + l] for its (b, l..l+3) and each of its heads
  f4 part[HPG];
#pragma unroll
  for (int hh = 0; hh < HPG; ++hh) part[hh] = zero;
#pragma unroll
  for (int t = 0; t < AMAX; ++t) {
    if (t < a.A) {  // wave-uniform
      f4 val;
      switch (a.aggr[t]) {
        case EGC_AGGR_SUM: val = acc.sum; break;
        case EGC_AGGR_MEAN: val = mean; break;
        case EGC_AGGR_MAX: val = cnt > 0 ? acc.mx : zero; break;
        case EGC_AGGR_MIN: if constexpr (NEED & NEED_MN) val = cnt > 0 ? acc.mn : zero; else val = zero; break;
        case EGC_AGGR_VAR: val = var; break;
        case EGC_AGGR_STD: val = f4_std(var); break;
        default: val = acc.ws; break;  // EGC_AGGR_SYMNORM
      }
#pragma unroll
      for (int hh = 0; hh < HPG; ++hh) {
        const float w = wv[hh][t];
        part[hh] = f4_fma(f4{w, w, w, w}, val, part[hh]);
      }
    }
  }
  // sum over b: lanes q, q ^ LPB, q ^ 2 LPB, ... share the same l
  for (int off = 1 << a.lpb_log2; off < LPR; off <<= 1) {
    const int addr = (lane_opaque ^ off) << 2;
#pragma unroll
    for (int hh = 0; hh < HPG; ++hh) part[hh] += bperm(addr, part[hh]);
  }
  const unsigned orow_off = (unsigned)row * (unsigned)a.F_out * 4u;  // scalar
#pragma unroll
  for (int hh = 0; hh < HPG; ++hh) {
    const int h = g + hh * G;
    const int o = h * a.L + 4 * l4;
    const bool mine = h < a.H && b == (hh & (a.B - 1));
    const f4 r = part[hh] + *reinterpret_cast<const f4*>(lds_bias + (mine ? o : 0));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, r), rsrc_out, mine ? (unsigned)o * 4u : OOB, orow_off, 0);
  }
}

template <int LPR_LOG2, int HPG, int NEED>
__global__ void __launch_bounds__(256) agg_fast_kernel(AggArgs a) {
  constexpr int LPR = 1 << LPR_LOG2, G = 64 / LPR;
  extern __shared__ float smem[];
  if ((int)blockIdx.x < a.chunk_blocks && (int)blockIdx.x * 4 >= a.plan[1]) return;  // unused chunk slots
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  LaneCtx<LPR_LOG2> L;
  L.init(threadIdx.x & 63);
  // per-wavefront copy of the bias row (row-invariant, read as one ds_read_b128 per head)
  float* lds_bias = smem + wave * a.lds_floats_per_wave;
  for (int o = L.lane; o < a.F_out; o += 64) lds_bias[o] = a.bias != nullptr ? a.bias[o] : 0.f;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const __amdgpu_buffer_rsrc_t rsrc = bases_rsrc(a);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.weightings, 0, (unsigned)a.n_nodes * (unsigned)a.W * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_out =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (unsigned)a.n_nodes * (unsigned)a.F_out * 4u, 0x00020000);

  if ((int)blockIdx.x < a.chunk_blocks) {
    // ---------------- long-row chunk role ----------------
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
    const int row_end = __builtin_amdgcn_readfirstlane(a.rowptr[row + 1]);
    const int end = min(start + EGC_LONG_ROW_CHUNK, row_end);
    const float dis_i = a.dis != nullptr ? a.dis[row] : 0.f;
    FAcc<NEED> acc;
    acc.init();
    int nself = 0;
    reduce_entries<LPR_LOG2, NEED>(a, rsrc, L, row, start, end, 0, false, dis_i, acc, nself);
    all_reduce_groups<LPR_LOG2, NEED>(acc, L.lane);
    if (L.g == 0) {
      f4* rec = reinterpret_cast<f4*>(a.partial) + (int64_t)c * 5 * LPR;
      rec[0 * LPR + L.q] = acc.sum;
      rec[2 * LPR + L.q] = acc.mx;
      rec[4 * LPR + L.q] = acc.ws;
      if constexpr (NEED & NEED_SQ) rec[1 * LPR + L.q] = acc.sq;
      if constexpr (NEED & NEED_MN) rec[3 * LPR + L.q] = acc.mn;
    }
    if (L.lane == 0) a.partial_nself[c] = nself;
    // publish: stores drained -> agent-scope release -> arrival counter (cdna guide, Guideline 16)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int deg = row_end - __builtin_amdgcn_readfirstlane(a.rowptr[row]);
    const int nch = (deg + EGC_LONG_ROW_CHUNK - 1) / EGC_LONG_ROW_CHUNK;
    int arrived = 0;
    if (L.lane == 0) arrived = __hip_atomic_fetch_add(&a.counters[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived != nch - 1) return;
    // last arriver: acquire, reset the counter for the next launch, merge in chunk order, finish the row
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (L.lane == 0) __hip_atomic_store(&a.counters[slot], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int c0 = __builtin_amdgcn_readfirstlane(long_chunk0[slot]);
    acc.init();
    for (int k0 = L.g; k0 < nch; k0 += G) {
      const f4* rec = reinterpret_cast<const f4*>(a.partial) + (int64_t)(c0 + k0) * 5 * LPR;
      acc.sum += rec[0 * LPR + L.q];
      acc.mx = f4_max(acc.mx, rec[2 * LPR + L.q]);
      acc.ws += rec[4 * LPR + L.q];
      if constexpr (NEED & NEED_SQ) acc.sq += rec[1 * LPR + L.q];
      if constexpr (NEED & NEED_MN) acc.mn = f4_min(acc.mn, rec[3 * LPR + L.q]);
    }
    nself = 0;
    for (int k = L.lane; k < nch; k += 64) nself += a.partial_nself[c0 + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nself += bperm((L.lane ^ off) << 2, nself);
    finish_fast<LPR_LOG2, HPG, NEED>(a, rsrc, rsrc_w, rsrc_out, L.lane, row, acc, deg, nself, dis_i, lds_bias);
    return;
  }

  // ---------------- short-row role: rows_per_wave consecutive rows per wavefront ----------------
  const int R = a.rows_per_wave;
  const int gw = ((int)blockIdx.x - a.chunk_blocks) * 4 + wave;
  const int r0 = __builtin_amdgcn_readfirstlane(gw * R);
  if (r0 >= a.n_nodes) return;
  const int r1 = min(r0 + R, a.n_nodes);
  const int rp = a.rowptr[min(r0 + L.lane, a.n_nodes)];  // lanes 0..R hold the block's row pointers
  int start = __builtin_amdgcn_readlane(rp, 0);
  int end = __builtin_amdgcn_readlane(rp, 1);
  int jj_n = (start + L.lane < end) ? a.col[start + L.lane] : r0;
  for (int row = r0; row < r1; ++row) {
    const int s = start, e = end;
    const int jj = jj_n;
    if (row + 1 < r1) {  // prefetch the next row's first 64 column indices
      start = e;
      end = __builtin_amdgcn_readlane(rp, row + 2 - r0);
      jj_n = (start + L.lane < end) ? a.col[start + L.lane] : row + 1;
    }
    const int deg = e - s;
    if (deg > EGC_LONG_ROW_THRESHOLD) continue;
    const float dis_i = a.dis != nullptr ? a.dis[row] : 0.f;
    FAcc<NEED> acc;
    acc.init();
    int nself = 0;
    reduce_entries<LPR_LOG2, NEED>(a, rsrc, L, row, s, e, jj, true, dis_i, acc, nself);
    // Opaque copy of the lane id: keeps the compiler from hoisting the epilogue's lane arithmetic out
    // of the row loop, where it would stay live across the gathers and cost occupancy.
    int ln = L.lane;
    asm volatile("" : "+v"(ln));
    finish_fast<LPR_LOG2, HPG, NEED>(a, rsrc, rsrc_w, rsrc_out, ln, row, acc, deg, nself, dis_i, lds_bias);
  }
}

bool fast_path_supported(const AggArgs& a, int layout, int chunks) {
  if (chunks != 1 || layout != EGC_LAYOUT_HBA || a.act == EGC_ACT_SOFTMAX) return false;
  if (a.slots != 16 && a.slots != 32 && a.slots != 64) return false;
  if (a.ldb != a.B * a.L) return false;
  if (a.L < 4 || (a.L & (a.L - 1)) != 0 || (a.B & (a.B - 1)) != 0) return false;
  if (a.A < 1 || a.A > AMAX) return false;
  const int G = 64 / a.slots;
  if ((a.H + G - 1) / G > 4) return false;
  // buffer descriptors address weightings / out with 32-bit byte offsets
  if ((uint64_t)a.n_nodes * (uint64_t)a.W * 4ull > (uint64_t)OOB) return false;
  if ((uint64_t)a.n_nodes * (uint64_t)a.F_out * 4ull > (uint64_t)OOB) return false;
  return true;
}

template <int LPR_LOG2, int HPG, int NEED>
static int launch_one(const AggArgs& a, unsigned grid, size_t lds, hipStream_t stream) {
  agg_fast_kernel<LPR_LOG2, HPG, NEED><<<grid, 256, lds, stream>>>(a);
  EGC_LAUNCH_CHECK("agg_fast_kernel");
  return EGC_OK;
}

template <int LPR_LOG2, int HPG>
static int launch_need(const AggArgs& a, int need, unsigned grid, size_t lds, hipStream_t stream) {
  if (need == 0) return launch_one<LPR_LOG2, HPG, 0>(a, grid, lds, stream);
  return launch_one<LPR_LOG2, HPG, NEED_SQ | NEED_MN>(a, grid, lds, stream);
}

template <int LPR_LOG2>
static int launch_a(const AggArgs& a, int need, unsigned grid, size_t lds, hipStream_t stream) {
  if (a.hpg <= 1) return launch_need<LPR_LOG2, 1>(a, need, grid, lds, stream);
  if (a.hpg <= 2) return launch_need<LPR_LOG2, 2>(a, need, grid, lds, stream);
  return launch_need<LPR_LOG2, 4>(a, need, grid, lds, stream);
}

int launch_fast(AggArgs a, int64_t n_nodes, const PlanCaps& caps, hipStream_t stream) {
  const int G = 64 / a.slots;
  int lg = 0;
  while ((4 << lg) < a.L) ++lg;
  a.lpb_log2 = lg;
  a.hpg = (a.H + G - 1) / G;
  if (a.rows_per_wave <= 0) a.rows_per_wave = 4;
  if (a.rows_per_wave > 32) a.rows_per_wave = 32;
  a.chunk_blocks = (int)ceil_div(caps.cap_chunks, 4);
  a.need_mean = a.need_var = 0;
  int need = 0;
  for (int t = 0; t < a.A; ++t) {
    if (a.aggr[t] == EGC_AGGR_MEAN || a.aggr[t] == EGC_AGGR_VAR || a.aggr[t] == EGC_AGGR_STD) a.need_mean = 1;
    if (a.aggr[t] == EGC_AGGR_VAR || a.aggr[t] == EGC_AGGR_STD) { a.need_var = 1; need |= NEED_SQ; }
    if (a.aggr[t] == EGC_AGGR_MIN) need |= NEED_MN;
  }
  a.lds_floats_per_wave = (a.F_out + 3) & ~3;
  const size_t lds = (size_t)4 * a.lds_floats_per_wave * sizeof(float);
  const int64_t row_blocks = ceil_div(n_nodes, (int64_t)4 * a.rows_per_wave);
  const unsigned grid = (unsigned)(a.chunk_blocks + row_blocks);
  switch (a.slots) {
    case 16: return launch_a<4>(a, need, grid, lds, stream);
    case 32: return launch_a<5>(a, need, grid, lds, stream);
    case 64: return launch_a<6>(a, need, grid, lds, stream);
    default: return EGC_ERR_UNSUPPORTED;
  }
}

}  // namespace egc
