// Fused multi-aggregator neighbourhood reduction + per-node combine for the EGC layer (gfx950).
//
// Reference behaviour replaced, in ONE pass over a destination-keyed CSR (no [E, B*L] temporaries):
//   x_j gather                         MessagePassing.__collect__ via propagate (layers.py:191-193)
//   symnorm message scaling            layers.py:195-199, optimized_layers.py:226-230
//   scatter / spmm per aggregator      layers.py:201-225, optimized_layers.py:215-278
//   var / std epilogue                 layers.py:202-216, optimized_layers.py:237-244
//   stack, weight nonlinearity, weighted sum / bmm, bias
//                                      layers.py:109-138, optimized_layers.py:183-208
//
// Mapping to the hardware.  A basis row (B*L floats, leading dimension ldb, 16-byte slots) is read by
// `lpr` lanes with one buffer_load_dwordx4 each, so a wavefront-instruction gathers G = 64/lpr
// neighbour rows at once (4 rows of 256 B at the north-star shape).  Each lane keeps running
// sum / sum-of-squares / max / min / symnorm-weighted-sum for its 4 columns in registers; the G
// lane groups are merged with xor-shuffles; the finished row goes through LDS once for the
// [H, A*B] x [A*B, L] combine and is written as one contiguous F_out row.  Invalid lanes use an
// out-of-range buffer offset (hardware returns 0, no memory traffic) instead of divergent branches.
//
// Degree skew: rows with more than EGC_LONG_ROW_THRESHOLD entries are cut into
// EGC_LONG_ROW_CHUNK-entry chunks (plan from egc_csr_prepare); one wavefront reduces one chunk to a
// partial record, and a merge kernel folds a row's partials in chunk order (deterministic) before the
// same epilogue.
#include <stdlib.h>

#include <algorithm>

#include "egc_aggregate_dev.h"
#include "egc_pack_map.h"

namespace egc {

// EGC_STDVAR_REFERENCE set (not "" / "0") and the layer has a var / std aggregator; read on every call
bool stdvar_reference(const egc_layer* layer) {
  const char* e = getenv("EGC_STDVAR_REFERENCE");
  if (layer == nullptr || e == nullptr || e[0] == '\0' || (e[0] == '0' && e[1] == '\0')) return false;
  for (int t = 0; t < layer->num_aggrs && t < EGC_MAX_AGGRS; ++t)
    if (layer->aggrs[t] == EGC_AGGR_VAR || layer->aggrs[t] == EGC_AGGR_STD) return true;
  return false;
}

// Reduce CSR entries [start, end) of `row` into per-lane partial aggregates.
// Lane (g, q): group g = lane >> lpr_log2 takes entries g, g+G, ...; q = slot inside the basis row.
// The variance's shift (Acc::sh): the row's first entry, for every partial accumulator of the row.
template <int CHUNKS>
__device__ inline void set_shift(const AggArgs& a, __amdgpu_buffer_rsrc_t rsrc, int row, int lane, Acc<CHUNKS>& acc) {
  bool need = false;                       // (a.need_var belongs to the register-resident family's launchers)
  for (int t = 0; t < a.A; ++t) need = need || a.aggr[t] == EGC_AGGR_VAR || a.aggr[t] == EGC_AGGR_STD;
  if (!need && a.stats == nullptr) return;
  if (a.var_ref) return;                   // (the reference's formula: squares about zero)
  const int rs = __builtin_amdgcn_readfirstlane(a.rowptr[row]), re = __builtin_amdgcn_readfirstlane(a.rowptr[row + 1]);
  if (re <= rs) return;
  const int first = __builtin_amdgcn_readfirstlane(a.col[rs]);
  const int q = lane & ((1 << a.lpr_log2) - 1);
#pragma unroll
  for (int k = 0; k < CHUNKS; ++k) {
    const int s = k * 64 + q;
    acc.sh[k] = load_slot(rsrc, s < a.slots ? (unsigned)first * (unsigned)a.ldb * 4u + (unsigned)s * 16u : OOB);
  }
}

template <int CHUNKS, int U>
__device__ inline void accumulate_range(const AggArgs& a, __amdgpu_buffer_rsrc_t rsrc, int row, int start, int end,
                                        int lane, Acc<CHUNKS>& acc, int& nself) {
  const int G = 64 >> a.lpr_log2;
  const int g = lane >> a.lpr_log2;
  const int q = lane & ((1 << a.lpr_log2) - 1);
  const float dis_i = a.dis != nullptr ? a.dis[row] : 0.f;
  const bool xl = a.x_looped != 0, yl = a.y_looped != 0;
  unsigned slot_off[CHUNKS];
#pragma unroll
  for (int k = 0; k < CHUNKS; ++k) {
    const int s = k * 64 + q;
    slot_off[k] = s < a.slots ? (unsigned)s * 16u : OOB;
  }
  const unsigned row_bytes = (unsigned)a.ldb * 4u;

  for (int base = start; base < end; base += 64) {
    // stage up to 64 (col, dis[col]) pairs of this row, one per lane, coalesced
    const int p = base + lane;
    const bool pv = p < end;
    const int jj = pv ? a.col[p] : row;
    // source-side deg^-1/2: streamed per entry when the graph carries it (one dependent round trip less), else gathered
    const float dd = a.edis != nullptr ? (pv ? a.edis[p] : 0.f) : (a.dis != nullptr ? a.dis[jj] : 0.f);
    nself += __popcll(__ballot(pv && jj == row));
    const int cnt = min(64, end - base);
    for (int t0 = 0; t0 < cnt; t0 += U * G) {
      int j[U];
      float w[U];
      bool valid[U];
      f4 v[U][CHUNKS];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = t0 + u * G + g;
        valid[u] = idx < cnt;
        j[u] = __shfl(jj, idx & 63);
        w[u] = __shfl(dd, idx & 63) * dis_i;
        const unsigned rb = (unsigned)j[u] * row_bytes;
#pragma unroll
        for (int k = 0; k < CHUNKS; ++k)
          v[u][k] = load_slot(rsrc, (valid[u] && slot_off[k] != OOB) ? rb + slot_off[k] : OOB);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool is_self = j[u] == row;
        const bool in_x = valid[u] && !(xl && is_self);
        const bool in_y = valid[u] && !(yl && is_self);
#pragma unroll
        for (int k = 0; k < CHUNKS; ++k)
          fold(acc.sum[k], acc.sq[k], acc.mx[k], acc.mn[k], acc.ws[k], v[u][k], in_x, in_y, w[u], acc.sh[k]);
      }
    }
  }
}

// Merge the G lane groups so that every lane holds the aggregates of its slot over all entries.
template <int CHUNKS>
__device__ inline void reduce_groups(const AggArgs& a, Acc<CHUNKS>& acc) {
  if (CHUNKS > 1) return;  // lpr == 64: a single group
  for (int off = 1 << a.lpr_log2; off < 64; off <<= 1) {
    acc.sum[0] += f4_shfl_xor(acc.sum[0], off);
    acc.sq[0] += f4_shfl_xor(acc.sq[0], off);
    acc.ws[0] += f4_shfl_xor(acc.ws[0], off);
    acc.mx[0] = f4_max(acc.mx[0], f4_shfl_xor(acc.mx[0], off));
    acc.mn[0] = f4_min(acc.mn[0], f4_shfl_xor(acc.mn[0], off));
  }
}

// Row-only operands of the epilogue, requested BEFORE the row's gather so that their latency (a dependent chain of
// global loads per one-row wavefront otherwise) hides under it: the row's own basis slots (self-loop term) and its
// weightings row (W <= 128: two floats per lane; wider rows are read in the epilogue).
constexpr int ROW_PRE_W = 2;
template <int CHUNKS>
struct RowPre {
  f4 vself[CHUNKS];
  float w[ROW_PRE_W];
  bool has_self, has_w;
};
template <int CHUNKS>
__device__ inline void preload_row(const AggArgs& a, __amdgpu_buffer_rsrc_t rsrc, int row, int lane, RowPre<CHUNKS>& pre) {
  const int q = lane & ((1 << a.lpr_log2) - 1);
  const int nloop = a.loops_all ? a.n_nodes : (*a.max_index + 1);
  pre.has_self = row < nloop;
  const bool want = (a.x_looped || a.y_looped) && pre.has_self;
#pragma unroll
  for (int k = 0; k < CHUNKS; ++k) {
    const int s = k * 64 + q;
    pre.vself[k] = load_slot(rsrc, (want && s < a.slots) ? (unsigned)row * (unsigned)a.ldb * 4u + (unsigned)s * 16u : OOB);
  }
  pre.has_w = a.W <= 64 * ROW_PRE_W;
  const float* wrow_g = a.weightings + (int64_t)row * a.ldw;
#pragma unroll
  for (int i = 0; i < ROW_PRE_W; ++i) {
    const int k = lane + 64 * i;
    pre.w[i] = (pre.has_w && k < a.W) ? __builtin_nontemporal_load(wrow_g + k) : 0.f;
  }
}

// Self-loop term, aggregator finalisation, weight nonlinearity, combine, bias, store.
// `acc` must already be merged over lane groups; `deg` / `nself` are the row's entry and self-entry counts.
template <int CHUNKS>
__device__ inline void finish_row(const AggArgs& a, __amdgpu_buffer_rsrc_t rsrc, int row, Acc<CHUNKS>& acc, int deg,
                                  int nself, int lane, float* lds, const RowPre<CHUNKS>& pre) {
  const int q = lane & ((1 << a.lpr_log2) - 1);
  const int g = lane >> a.lpr_log2;
  const bool has_self = pre.has_self;
  int cnt = deg;
  if (a.x_looped) cnt = deg - nself + (has_self ? 1 : 0);

  if ((a.x_looped || a.y_looped) && has_self) {
    const float dis_i = a.dis != nullptr ? a.dis[row] : 0.f;
    const float wself = dis_i * dis_i;
#pragma unroll
    for (int k = 0; k < CHUNKS; ++k)
      fold(acc.sum[k], acc.sq[k], acc.mx[k], acc.mn[k], acc.ws[k], pre.vself[k], a.x_looped != 0, a.y_looped != 0, wself, acc.sh[k]);
  }

  float* lds_agg = lds;
  float* lds_w = lds + a.A * a.ldb;

  // (1) finalise every aggregator for this lane's slot(s) and park it in LDS as [A][ldb]
  const float cntf = (float)max(cnt, 1);
  if (g == 0) {
    if (a.stats != nullptr && lane == 0) a.cnt_out[row] = cnt;
#pragma unroll
    for (int k = 0; k < CHUNKS; ++k) {
      const int s = k * 64 + q;
      if (s < a.slots) {
        if (a.stats != nullptr) {  // training forward: keep the raw aggregates for the backward
          float* st = a.stats + ((int64_t)row * a.stat_k) * a.ldb + 4 * s;
          if (a.stat_slot[STAT_SUM] >= 0) *reinterpret_cast<f4*>(st + a.stat_slot[STAT_SUM] * a.ldb) = acc.sum[k];
          if (a.stat_slot[STAT_SQ] >= 0) {   // the backward's record keeps the VARIANCE as this forward forms it (a
            // reconstructed plain sum of squares, from which the backward re-derived var = sq / cnt - mean^2, gave (nearly) tied
            // neighbourhoods a relu mask and a std that were not the forward's)
            const float nc = -(float)cnt;
            const f4 ds = f4_fma(f4{nc, nc, nc, nc}, acc.sh[k], acc.sum[k]);
            *reinterpret_cast<f4*>(st + a.stat_slot[STAT_SQ] * a.ldb) = f4_var(f4_div(acc.sq[k], cntf), f4_div(ds, cntf));
          }
          if (a.stat_slot[STAT_MX] >= 0) *reinterpret_cast<f4*>(st + a.stat_slot[STAT_MX] * a.ldb) = acc.mx[k];
          if (a.stat_slot[STAT_MN] >= 0) *reinterpret_cast<f4*>(st + a.stat_slot[STAT_MN] * a.ldb) = acc.mn[k];
          if (a.stat_slot[STAT_WS] >= 0) *reinterpret_cast<f4*>(st + a.stat_slot[STAT_WS] * a.ldb) = acc.ws[k];
        }
        const f4 mean = f4_div(acc.sum[k], cntf);
        const float ncnt = -(float)cnt;
        const f4 dsum = f4_fma(f4{ncnt, ncnt, ncnt, ncnt}, acc.sh[k], acc.sum[k]);   // sum of (x - sh), one rounding
        const f4 var = f4_var(f4_div(acc.sq[k], cntf), f4_div(dsum, cntf));
        const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < a.A; ++t) {
          f4 val;
          switch (a.aggr[t]) {
            case EGC_AGGR_SUM: val = acc.sum[k]; break;
            case EGC_AGGR_MEAN: val = mean; break;
            case EGC_AGGR_MAX: val = cnt > 0 ? acc.mx[k] : zero; break;
            case EGC_AGGR_MIN: val = cnt > 0 ? acc.mn[k] : zero; break;
            case EGC_AGGR_VAR: val = var; break;
            case EGC_AGGR_STD: val = f4_std(var); break;
            default: val = acc.ws[k]; break;  // EGC_AGGR_SYMNORM
          }
          *reinterpret_cast<f4*>(lds_agg + t * a.ldb + 4 * s) = val;
        }
      }
    }
  }
  // (2) the node's weightings row, nonlinearity applied, into LDS
  auto w_act = [&](float w) {
    if (a.act == EGC_ACT_SIGMOID) w = 1.0f / (1.0f + expf(-w));
    else if (a.act == EGC_ACT_HARDTANH) w = fminf(fmaxf(w, -1.0f), 1.0f);
    return w;
  };
  if (pre.has_w) {
#pragma unroll
    for (int i = 0; i < ROW_PRE_W; ++i)
      if (lane + 64 * i < a.W) lds_w[lane + 64 * i] = w_act(pre.w[i]);
  } else {
    const float* wrow_g = a.weightings + (int64_t)row * a.ldw;
    for (int k = lane; k < a.W; k += 64) lds_w[k] = w_act(wrow_g[k]);
  }
  const int AB = a.A * a.B;
  if (a.act == EGC_ACT_SOFTMAX) {
    // softmax over the joint B*A axis of each head (layers.py:112-117)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int h = lane; h < a.H; h += 64) {
      float* wh = lds_w + h * AB;
      float m = -INFINITY;
      for (int k = 0; k < AB; ++k) m = fmaxf(m, wh[k]);
      float s = 0.f;
      for (int k = 0; k < AB; ++k) {
        const float e = expf(wh[k] - m);
        wh[k] = e;
        s += e;
      }
      for (int k = 0; k < AB; ++k) wh[k] = wh[k] / s;
    }
  }
  // LDS is per-wavefront here and LDS ops of one wavefront complete in order.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // (3) combine: out[h*L + l] = sum_{a,b} w[h][a][b] * agg[a][b*L + l] (+ bias)
  //     Two outputs per lane and four bases per step: sixteen independent LDS reads are in flight before the first
  //     fmaf needs one (one read per fmaf exposed the LDS latency A * B times per output); every output still adds
  //     its terms in the order (a, b), so the result does not depend on the grouping.
  float* orow = a.out + (int64_t)row * a.F_out;
  for (int o0 = lane; o0 < a.F_out; o0 += 128) {
    const bool two = o0 + 64 < a.F_out;
    const int o1 = two ? o0 + 64 : o0;
    const int h0 = a.L == 1 ? o0 : (int)__umulhi((unsigned)o0, a.magic_L);
    const int h1 = a.L == 1 ? o1 : (int)__umulhi((unsigned)o1, a.magic_L);
    const float* wh0 = lds_w + h0 * AB;
    const float* wh1 = lds_w + h1 * AB;
    const float* ag0 = lds_agg + (o0 - h0 * a.L);
    const float* ag1 = lds_agg + (o1 - h1 * a.L);
    float z0 = 0.f, z1 = 0.f;
    for (int t = 0; t < a.A; ++t) {
      const int wo = t * a.sa, ao = t * a.ldb;
      for (int b0 = 0; b0 < a.B; b0 += 4) {
        float w0[4], w1[4], v0[4], v1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int b = min(b0 + u, a.B - 1);     // (wave-uniform; the clamped repeats are not added)
          w0[u] = wh0[wo + b * a.sb];
          w1[u] = wh1[wo + b * a.sb];
          v0[u] = ag0[ao + b * a.Ls];
          v1[u] = ag1[ao + b * a.Ls];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (b0 + u < a.B) {
            z0 = fmaf(w0[u], v0[u], z0);
            z1 = fmaf(w1[u], v1[u], z1);
          }
      }
    }
    auto put = [&](int o, float z) {
      if (a.bias != nullptr) z += a.bias[o];
      if (a.post_scale != nullptr) z = fmaf(z, a.post_scale[o], a.post_shift[o]);
      if (a.post_relu) z = fmaxf(z, 0.f);
      if (a.residual != nullptr) z += a.residual[(int64_t)row * a.F_out + o];
      __builtin_nontemporal_store(z, &orow[o]);  // written once, read by a later kernel: keep it out of the L2 write-back at kernel end
    };
    put(o0, z0);
    if (two) put(o1, z1);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// One wavefront per CSR row (rows above the long-row threshold are left to the chunk/merge kernels).
template <int CHUNKS, int U>
__global__ void __launch_bounds__(256) agg_rows_kernel(AggArgs a) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = blockDim.x >> 6;
  const int row = __builtin_amdgcn_readfirstlane(a.row_begin + blockIdx.x * wpb + wave);
  if (row >= a.row_end) return;
  const int start = __builtin_amdgcn_readfirstlane(a.rowptr[row]);
  const int end = __builtin_amdgcn_readfirstlane(a.rowptr[row + 1]);
  const int deg = end - start;
  if (deg > EGC_LONG_ROW_THRESHOLD) return;
  const __amdgpu_buffer_rsrc_t rsrc = bases_rsrc(a);
  RowPre<CHUNKS> pre;
  preload_row<CHUNKS>(a, rsrc, row, lane, pre);
  Acc<CHUNKS> acc;
  acc.init();
  set_shift<CHUNKS>(a, rsrc, row, lane, acc);
  int nself = 0;
  accumulate_range<CHUNKS, U>(a, rsrc, row, start, end, lane, acc, nself);
  reduce_groups<CHUNKS>(a, acc);
  finish_row<CHUNKS>(a, rsrc, row, acc, deg, nself, lane, smem + wave * a.lds_floats_per_wave, pre);
}

// One wavefront per long-row chunk -> partial record.
template <int CHUNKS, int U>
__global__ void __launch_bounds__(256) agg_chunks_kernel(AggArgs a) {
  const int lane = threadIdx.x & 63;
  const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const int n_chunks = a.plan[1];
  if (c >= n_chunks) return;
  const int cap_long = a.plan[2], cap_chunks = a.plan[3];
  const int* long_row = a.plan + 4;
  const int* chunk_slot = long_row + 2 * cap_long;
  const int* chunk_begin = chunk_slot + cap_chunks;
  const int row = __builtin_amdgcn_readfirstlane(long_row[chunk_slot[c]]);
  if (row < a.row_begin || row >= a.row_end) return;
  const int start = __builtin_amdgcn_readfirstlane(chunk_begin[c]);
  const int end = min(start + EGC_LONG_ROW_CHUNK, __builtin_amdgcn_readfirstlane(a.rowptr[row + 1]));
  const __amdgpu_buffer_rsrc_t rsrc = bases_rsrc(a);
  Acc<CHUNKS> acc;
  acc.init();
  set_shift<CHUNKS>(a, rsrc, row, lane, acc);
  int nself = 0;
  accumulate_range<CHUNKS, U>(a, rsrc, row, start, end, lane, acc, nself);
  reduce_groups<CHUNKS>(a, acc);
  const int q = lane & ((1 << a.lpr_log2) - 1);
  if ((lane >> a.lpr_log2) == 0) {
    f4* rec = reinterpret_cast<f4*>(a.partial) + (int64_t)c * 5 * a.slots;
#pragma unroll
    for (int k = 0; k < CHUNKS; ++k) {
      const int s = k * 64 + q;
      if (s < a.slots) {
        rec[0 * a.slots + s] = acc.sum[k];
        rec[1 * a.slots + s] = acc.sq[k];
        rec[2 * a.slots + s] = acc.mx[k];
        rec[3 * a.slots + s] = acc.mn[k];
        rec[4 * a.slots + s] = acc.ws[k];
      }
    }
  }
  if (lane == 0) a.partial_nself[c] = nself;
}

// One wavefront per long row: fold its partial records in chunk order, then the common epilogue.
template <int CHUNKS>
__global__ void __launch_bounds__(256) agg_merge_kernel(AggArgs a) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slot = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + wave);
  if (slot >= a.plan[0]) return;
  const int cap_long = a.plan[2];
  const int* long_row = a.plan + 4;
  const int* long_chunk0 = long_row + cap_long;
  const int row = __builtin_amdgcn_readfirstlane(long_row[slot]);
  if (row < a.row_begin || row >= a.row_end) return;
  const int c0 = __builtin_amdgcn_readfirstlane(long_chunk0[slot]);
  const int deg = __builtin_amdgcn_readfirstlane(a.rowptr[row + 1] - a.rowptr[row]);
  const int nch = (deg + EGC_LONG_ROW_CHUNK - 1) / EGC_LONG_ROW_CHUNK;
  const int G = 64 >> a.lpr_log2;
  const int g = lane >> a.lpr_log2;
  const int q = lane & ((1 << a.lpr_log2) - 1);
  Acc<CHUNKS> acc;
  acc.init();
  set_shift<CHUNKS>(a, bases_rsrc(a), row, lane, acc);
  for (int k0 = g; k0 < nch; k0 += G) {
    const f4* rec = reinterpret_cast<const f4*>(a.partial) + (int64_t)(c0 + k0) * 5 * a.slots;
#pragma unroll
    for (int k = 0; k < CHUNKS; ++k) {
      const int s = k * 64 + q;
      if (s < a.slots) {
        acc.sum[k] += rec[0 * a.slots + s];
        acc.sq[k] += rec[1 * a.slots + s];
        acc.mx[k] = f4_max(acc.mx[k], rec[2 * a.slots + s]);
        acc.mn[k] = f4_min(acc.mn[k], rec[3 * a.slots + s]);
        acc.ws[k] += rec[4 * a.slots + s];
      }
    }
  }
  reduce_groups<CHUNKS>(a, acc);
  int nself = 0;
  for (int k = lane; k < nch; k += 64) nself += a.partial_nself[c0 + k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) nself += __shfl_xor(nself, off);
  const __amdgpu_buffer_rsrc_t rsrc = bases_rsrc(a);
  RowPre<CHUNKS> pre;
  preload_row<CHUNKS>(a, rsrc, row, lane, pre);
  finish_row<CHUNKS>(a, rsrc, row, acc, deg, nself, lane, smem + wave * a.lds_floats_per_wave, pre);
}

// One wavefront per segment: lanes stride over the columns, rows are summed in order (deterministic).
// out[block, c] = sum over this block's rows of x[r, c] (c < cols, cols a multiple of 4): thread = (16-byte column
// group, row lane); rows strided over the row lanes, float4 accumulators, LDS reduction over the row lanes.  Bias gradients of the training step (column sums of grad_out and of d weightings).
__global__ void __launch_bounds__(256) column_sums_kernel(const float* __restrict__ x, int64_t n_rows, int ld, int cols,
                                                          int rows_per_block, float* __restrict__ out) {
  __shared__ f4 red[256];
  const int cg = cols >> 2;                 // 16-byte column groups (<= 256)
  const int rl = 256 / cg;                  // row lanes
  const int g = threadIdx.x % cg, lane_r = threadIdx.x / cg;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(r0 + rows_per_block, n_rows);
  f4 acc = f4{0.f, 0.f, 0.f, 0.f};
  if (lane_r < rl)
    for (int64_t r = r0 + lane_r; r < r1; r += rl) acc += __builtin_nontemporal_load(reinterpret_cast<const f4*>(x + r * ld) + g);
  red[threadIdx.x] = acc;
  __syncthreads();
  if (lane_r == 0) {
    for (int k = 1; k < rl; ++k) acc += red[k * cg + g];
    reinterpret_cast<f4*>(out + (int64_t)blockIdx.x * cols)[g] = acc;   // this block's partial row
  }
}

__global__ void __launch_bounds__(256) segment_mean_kernel(const float* __restrict__ x, const int64_t* __restrict__ seg_ptr,
                                                           int64_t n_segments, int width, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= n_segments) return;
  const int64_t r0 = seg_ptr[g], r1 = seg_ptr[g + 1];
  const float cnt = (float)(r1 > r0 ? r1 - r0 : 1);  // scatter-mean divides the sum by the count
  for (int c = lane; c < width; c += 64) {
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += x[r * width + c];
    out[g * width + c] = s / cnt;
  }
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int validate_layer(const egc_layer* L) {
  if (L == nullptr) return EGC_ERR_INVALID;
  if (L->in_channels <= 0 || L->out_channels <= 0 || L->num_heads <= 0 || L->num_bases <= 0) return EGC_ERR_INVALID;
  if (L->num_aggrs <= 0 || L->num_aggrs > EGC_MAX_AGGRS) return EGC_ERR_INVALID;
  if (L->out_channels % L->num_heads != 0) return EGC_ERR_INVALID;
  for (int t = 0; t < L->num_aggrs; ++t)
    if (L->aggrs[t] < EGC_AGGR_SUM || L->aggrs[t] > EGC_AGGR_SYMNORM) return EGC_ERR_INVALID;
  if (L->agg_set != EGC_SET_RAW && L->agg_set != EGC_SET_LOOPED) return EGC_ERR_INVALID;
  if (L->sym_set != EGC_SET_RAW && L->sym_set != EGC_SET_LOOPED) return EGC_ERR_INVALID;
  if (L->weight_layout != EGC_LAYOUT_HBA && L->weight_layout != EGC_LAYOUT_HAB) return EGC_ERR_INVALID;
  if (L->weight_act < EGC_ACT_NONE || L->weight_act > EGC_ACT_HARDTANH) return EGC_ERR_INVALID;
  const int len = L->out_channels / L->num_heads;
  if (L->basis_stride < 0 || (L->basis_stride > len && (L->basis_stride & 3) != 0)) return EGC_ERR_INVALID;
  const int64_t fg = (int64_t)L->num_bases * layer_basis_stride(L);
  const int64_t w = (int64_t)L->num_heads * L->num_bases * L->num_aggrs;
  if (fg > EGC_MAX_BASIS_WIDTH || w > EGC_MAX_WEIGHT_WIDTH || L->out_channels > EGC_MAX_OUT_CHANNELS)
    return EGC_ERR_UNSUPPORTED;
  return EGC_OK;
}

struct WsLayout {
  size_t counter_bytes, partial_bytes, nself_bytes, queue_bytes, total;
};
static WsLayout ws_layout(const egc_layer* L, int64_t n_nodes, int64_t n_edges, int64_t n_chunks = -1) {
  const int ldb = egc_bases_ld(L);
  PlanCaps c = plan_caps(n_nodes, n_edges);
  WsLayout w;
  w.counter_bytes = align256((size_t)c.cap_long * sizeof(int));
  // One chunk record = up to 7 slots-of-16-bytes (5 aggregates + 2 arg positions) per LANE of the lane group that
  // publishes it.  The register-resident kernels lay a record out by their lane-group size (16 / 32 / 64 lanes),
  // which exceeds the row's slot count whenever that is not 16, 32 or 64 -- size the buffer by the larger of the two.
  const int slots = ldb / 4;
  // (65..128 slots: the two-slots-per-lane kernel publishes 5 aggregates x 2 sets x 64 lanes = 640 slots-of-16-bytes per chunk)
  const int rec_lanes = slots <= 16 ? 16 : slots <= 32 ? 32 : slots <= 64 ? 64 : slots <= 128 ? 128 : slots;
  // chunk slots that can be written: the plan's capacity, or the host-known chunk count of this graph
  const int64_t rec_chunks = (n_chunks >= 0 && n_chunks <= c.cap_chunks) ? n_chunks : c.cap_chunks;
  w.partial_bytes = align256((size_t)rec_chunks * 7 * rec_lanes * 16);
  w.nself_bytes = align256((size_t)rec_chunks * sizeof(int));
  w.queue_bytes = align256((size_t)FUSEDW_QUEUE_INTS * sizeof(int));  // (reserved: keeps the workspace layout of rounds 2-4)
  // order: [counters][queue] -- the part that must be zero before the first use, its size a function of (n_nodes,
  // n_edges) alone -- then [partials][nself], records that are written before they are read
  w.total = w.counter_bytes + w.queue_bytes + w.partial_bytes + w.nself_bytes;
  return w;
}

template <int CHUNKS>
static int launch_all(const AggArgs& a, int64_t n_nodes, const PlanCaps& caps, int wpb, size_t lds_bytes,
                      hipStream_t stream, bool wide_rows = false) {
  // neighbour rows in flight per lane group.  With two or more slots per lane the staging registers of four rows
  // cost a wavefront per SIMD (130 VGPRs -> 3 wavefronts; two rows: 4), and these one-row wavefronts are bound by
  // their chain of dependent memory round trips, i.e. by occupancy: 887 -> 802 us at 300/H4/B4 on the arxiv graph.
  constexpr int U = CHUNKS == 1 ? 4 : 2;
  const int threads = wpb * 64;
  if (wide_rows)   // 65..128 slots per row: the two-slots-per-lane register kernel, long-row chunks included (egc_aggregate_fast.hip)
    return launch_wide_rows(a, caps, stream);
  // long-row chunks first (they are the longest work items), then the per-row kernel, then the merge
  agg_chunks_kernel<CHUNKS, U><<<(unsigned)ceil_div(caps.cap_chunks, 4), 256, 0, stream>>>(a);
  EGC_LAUNCH_CHECK("agg_chunks_kernel");
  agg_rows_kernel<CHUNKS, U><<<(unsigned)ceil_div(a.row_end - a.row_begin, wpb), threads, lds_bytes, stream>>>(a);
  EGC_LAUNCH_CHECK("agg_rows_kernel");
  agg_merge_kernel<CHUNKS><<<(unsigned)ceil_div(caps.cap_long, wpb), threads, lds_bytes, stream>>>(a);
  EGC_LAUNCH_CHECK("agg_merge_kernel");
  return EGC_OK;
}


}  // namespace egc

using namespace egc;

extern "C" {

int32_t egc_bases_ld(const egc_layer* layer) {
  if (layer == nullptr || layer->num_heads <= 0) return -1;
  const int fg = layer->num_bases * layer_basis_stride(layer);
  return (fg + 3) & ~3;
}

size_t egc_aggregate_workspace_bytes_for(const egc_layer* layer, const egc_graph* graph) {
  if (graph == nullptr || validate_layer(layer) != EGC_OK || graph->n_nodes < 0 || graph->n_edges < 0) return 0;
  return ws_layout(layer, graph->n_nodes, graph->n_edges, graph->n_chunks).total;
}

size_t egc_aggregate_workspace_zero_bytes(const egc_layer* layer, int64_t n_nodes, int64_t n_edges) {
  if (validate_layer(layer) != EGC_OK || n_nodes < 0 || n_edges < 0) return 0;
  const WsLayout w = ws_layout(layer, n_nodes, n_edges);
  return w.counter_bytes + w.queue_bytes;
}

size_t egc_aggregate_workspace_bytes(const egc_layer* layer, int64_t n_nodes, int64_t n_edges) {
  if (validate_layer(layer) != EGC_OK || n_nodes < 0 || n_edges < 0) return 0;
  return ws_layout(layer, n_nodes, n_edges).total;
}

int egc_column_sums_f32(const float* x, int64_t n_rows, int32_t ld, int32_t cols, float* partials, int32_t n_partials,
                        egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_rows < 0 || cols <= 0 || ld < cols || partials == nullptr || n_partials <= 0) return EGC_ERR_INVALID;
  if ((cols & 3) != 0 || (ld & 3) != 0 || cols > 1024 || (reinterpret_cast<uintptr_t>(x) & 15) != 0 ||
      (reinterpret_cast<uintptr_t>(partials) & 15) != 0)
    return EGC_ERR_UNSUPPORTED;
  if (n_rows > 0 && x == nullptr) return EGC_ERR_INVALID;
  const int rows_per_block = (int)std::max<int64_t>(ceil_div(n_rows, (int64_t)n_partials), 1);  // empty blocks write zeros
  column_sums_kernel<<<(unsigned)n_partials, 256, 0, stream>>>(x, n_rows, ld, cols, rows_per_block, partials);
  EGC_LAUNCH_CHECK("column_sums_kernel");
  return EGC_OK;
}

int egc_segment_mean_f32(const float* x, const int64_t* seg_ptr, int64_t n_segments, int32_t width, float* out,
                         egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_segments < 0 || width <= 0 || n_segments >= ((int64_t)1 << 31)) return EGC_ERR_INVALID;
  if (n_segments == 0) return EGC_OK;
  // x may be NULL when every segment is empty (a zero-row tensor has no storage); the kernel then reads nothing
  if (seg_ptr == nullptr || out == nullptr) return EGC_ERR_INVALID;
  segment_mean_kernel<<<(unsigned)ceil_div(n_segments, 4), 256, 0, stream>>>(x, seg_ptr, n_segments, width, out);
  EGC_LAUNCH_CHECK("segment_mean_kernel");
  return EGC_OK;
}

int64_t egc_train_stats_floats(const egc_layer* layer) {
  if (validate_layer(layer) != EGC_OK) return 0;
  int slot[5];
  const int64_t k = stat_layout(layer->aggrs, layer->num_aggrs, slot), ldb = egc_bases_ld(layer);
  // raw aggregates, then the 8-bit in-row arg positions of max / min (ldb bytes each per row; egc_aggregate_dev.h)
  return k * ldb + ((slot[STAT_MX] >= 0 ? 1 : 0) + (slot[STAT_MN] >= 0 ? 1 : 0)) * (ldb / 4);
}

// where the arg8 tables of a training call sit inside its `stats` buffer
static void arg8_tables(const egc_layer* layer, int64_t n, float* stats, bool want_max, bool want_min, unsigned** a8max,
                        unsigned** a8min) {
  int slot[5];
  const int64_t k = stat_layout(layer->aggrs, layer->num_aggrs, slot), ldb = egc_bases_ld(layer);
  unsigned char* bytes = reinterpret_cast<unsigned char*>(stats + n * k * ldb);
  const bool has_max = slot[STAT_MX] >= 0, has_min = slot[STAT_MN] >= 0;
  *a8max = (has_max && want_max) ? reinterpret_cast<unsigned*>(bytes) : nullptr;
  *a8min = (has_min && want_min) ? reinterpret_cast<unsigned*>(bytes + (has_max ? n * ldb : 0)) : nullptr;
}

static int aggregate_combine_impl(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb,
                                  const float* weightings, const float* bias, const egc_post* post, float* out,
                                  float* stats, int32_t* cnt_out, int64_t row_begin, int64_t row_end, void* workspace,
                                  size_t workspace_bytes, egc_stream_t stream_, int32_t* arg_max = nullptr,
                                  int32_t* arg_min = nullptr, bool* arg_done = nullptr, int32_t ldw = 0);

int egc_aggregate_combine_f32(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb,
                              const float* weightings, const float* bias, float* out, int32_t* arg_max,
                              int32_t* arg_min, void* workspace, size_t workspace_bytes, egc_stream_t stream) {
  // the arg-extremum indices need the extrema themselves: egc_aggregate_combine_train_f32 keeps them
  if (arg_max != nullptr || arg_min != nullptr) return EGC_ERR_UNSUPPORTED;
  return aggregate_combine_impl(graph, layer, bases, ldb, weightings, bias, nullptr, out, nullptr, nullptr, 0, -1, workspace,
                                workspace_bytes, stream);
}

int egc_aggregate_combine_rows_f32(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb,
                                   const float* weightings, const float* bias, float* out, int64_t row_begin,
                                   int64_t row_end, void* workspace, size_t workspace_bytes, egc_stream_t stream) {
  if (row_end < 0) return EGC_ERR_INVALID;
  return aggregate_combine_impl(graph, layer, bases, ldb, weightings, bias, nullptr, out, nullptr, nullptr, row_begin,
                                row_end, workspace, workspace_bytes, stream);
}

int egc_aggregate_combine_post_f32(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb,
                                   const float* weightings, const float* bias, const egc_post* post, float* out,
                                   void* workspace, size_t workspace_bytes, egc_stream_t stream) {
  if (post != nullptr && ((post->scale == nullptr) != (post->shift == nullptr))) return EGC_ERR_INVALID;
  return aggregate_combine_impl(graph, layer, bases, ldb, weightings, bias, post, out, nullptr, nullptr, 0, -1, workspace,
                                workspace_bytes, stream);
}

int egc_aggregate_combine_strided_f32(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb,
                                      const float* weightings, int32_t ldw, const float* bias, const egc_post* post,
                                      float* out, void* workspace, size_t workspace_bytes, egc_stream_t stream) {
  if (post != nullptr && ((post->scale == nullptr) != (post->shift == nullptr))) return EGC_ERR_INVALID;
  return aggregate_combine_impl(graph, layer, bases, ldb, weightings, bias, post, out, nullptr, nullptr, 0, -1, workspace,
                                workspace_bytes, stream, nullptr, nullptr, nullptr, ldw);
}

int egc_aggregate_combine_train_f32(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb,
                                    const float* weightings, const float* bias, float* out, float* stats,
                                    int32_t* cnt, int32_t* arg_max, int32_t* arg_min, void* workspace,
                                    size_t workspace_bytes, egc_stream_t stream) {
  if (graph == nullptr || layer == nullptr || stats == nullptr || cnt == nullptr) return EGC_ERR_INVALID;
  bool arg_done = false;
  int st = aggregate_combine_impl(graph, layer, bases, ldb, weightings, bias, nullptr, out, stats, cnt, 0, -1, workspace,
                                  workspace_bytes, stream, arg_max, arg_min, &arg_done);
  if (st != EGC_OK || arg_done) return st;
  unsigned *a8max, *a8min;
  arg8_tables(layer, graph->n_nodes, stats, arg_max != nullptr, arg_min != nullptr, &a8max, &a8min);
  return egc::arg_extrema(graph, layer, bases, ldb, stats, cnt, arg_max, arg_min, a8max, a8min, (hipStream_t)stream);
}

int egc_aggregate_combine_train_rows_f32(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb,
                                         const float* weightings, const float* bias, float* out, float* stats,
                                         int32_t* cnt, int32_t* arg_max, int32_t* arg_min, int64_t row_begin,
                                         int64_t row_end, void* workspace, size_t workspace_bytes, egc_stream_t stream) {
  if (graph == nullptr || layer == nullptr || stats == nullptr || cnt == nullptr || row_end < 0) return EGC_ERR_INVALID;
  bool arg_done = false;
  int st = aggregate_combine_impl(graph, layer, bases, ldb, weightings, bias, nullptr, out, stats, cnt, row_begin, row_end,
                                  workspace, workspace_bytes, stream, arg_max, arg_min, &arg_done);
  // kernels that do not track the arg positions themselves leave them to one pass over ALL rows: it runs with the range
  // that ends at the last row (the ranges of a split call are issued in ascending order)
  if (st != EGC_OK || arg_done || row_end < graph->n_nodes) return st;
  unsigned *a8max, *a8min;
  arg8_tables(layer, graph->n_nodes, stats, arg_max != nullptr, arg_min != nullptr, &a8max, &a8min);
  return egc::arg_extrema(graph, layer, bases, ldb, stats, cnt, arg_max, arg_min, a8max, a8min, (hipStream_t)stream);
}

static int aggregate_combine_impl(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb,
                                  const float* weightings, const float* bias, const egc_post* post, float* out,
                                  float* stats, int32_t* cnt_out, int64_t row_begin, int64_t row_end, void* workspace,
                                  size_t workspace_bytes, egc_stream_t stream_, int32_t* arg_max, int32_t* arg_min,
                                  bool* arg_done, int32_t ldw) {
  hipStream_t stream = (hipStream_t)stream_;
  if (graph == nullptr) return EGC_ERR_INVALID;
  int st = validate_layer(layer);
  if (st != EGC_OK) return st;
  const int64_t n = graph->n_nodes, e = graph->n_edges;
  if (n < 0 || e < 0 || n >= ((int64_t)1 << 31) - 1 || e >= ((int64_t)1 << 31) - 1) return EGC_ERR_INVALID;
  if (n == 0) return EGC_OK;
  if (graph->rowptr == nullptr || graph->plan == nullptr || (e > 0 && graph->col == nullptr)) return EGC_ERR_INVALID;
  if (bases == nullptr || weightings == nullptr || out == nullptr) return EGC_ERR_INVALID;
  if (ldb != egc_bases_ld(layer)) return EGC_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(bases) & 15) != 0) return EGC_ERR_INVALID;
  const int64_t n_src = graph->n_src_rows > 0 ? graph->n_src_rows : n;  // owned rows + halo rows
  // a rectangular adjacency (relation between two node types; fewer sources than rows is possible) has no
  // self loops and no symmetric normalisation: nothing then indexes the source tables by a ROW id
  if (n_src < n && (layer->agg_set == EGC_SET_LOOPED || layer_uses_symnorm(layer))) return EGC_ERR_INVALID;
  if ((uint64_t)n_src * (uint64_t)ldb * 4ull > (uint64_t)OOB) return EGC_ERR_UNSUPPORTED;  // 32-bit buffer offsets

  AggArgs a;
  a.rowptr = graph->rowptr;
  a.col = graph->col;
  a.dis = nullptr;
  a.edis = nullptr;
  if (layer_uses_symnorm(layer)) {
    a.dis = layer->sym_set == EGC_SET_LOOPED ? graph->dis_looped : graph->dis_raw;
    a.edis = layer->sym_set == EGC_SET_LOOPED ? graph->edge_dis_looped : graph->edge_dis_raw;
    if (a.dis == nullptr) return EGC_ERR_INVALID;
  }
  a.loops_all = layer->loops_all_nodes != 0;
  a.max_index = graph->max_index;
  if (!a.loops_all && graph->max_index == nullptr) return EGC_ERR_INVALID;
  a.plan = graph->plan;
  a.bases = bases;
  a.weightings = weightings;
  a.bias = bias;
  a.out = out;
  a.n_nodes = (int)n;
  a.l4_off = 0;
  a.wide_p0 = a.wide_p1 = 0;
  a.row_begin = (int)row_begin;
  a.row_end = row_end < 0 ? (int)n : (int)row_end;
  if (a.row_begin < 0 || a.row_end > (int)n) return EGC_ERR_INVALID;
  if (a.row_begin >= a.row_end) return EGC_OK;
  a.ldb = ldb;
  a.slots = ldb / 4;
  a.F_out = layer->out_channels;
  a.H = layer->num_heads;
  a.B = layer->num_bases;
  a.A = layer->num_aggrs;
  a.L = layer->out_channels / layer->num_heads;
  a.Ls = layer_basis_stride(layer);
  a.W = a.H * a.B * a.A;
  a.ldw = ldw > 0 ? ldw : a.W;  // row stride of `weightings` (a column block of a wider array when > W)
  if (a.ldw != a.W && (a.ldw < a.W || (a.ldw & 3) != 0 || (reinterpret_cast<uintptr_t>(weightings) & 15) != 0))
    return EGC_ERR_INVALID;
  for (int t = 0; t < EGC_MAX_AGGRS; ++t) a.aggr[t] = t < a.A ? layer->aggrs[t] : 0;
  a.x_looped = layer->agg_set == EGC_SET_LOOPED;
  a.y_looped = layer->sym_set == EGC_SET_LOOPED;
  if (layer->weight_layout == EGC_LAYOUT_HAB) { a.sa = a.B; a.sb = 1; } else { a.sa = 1; a.sb = a.A; }
  a.act = layer->weight_act;
  a.magic_L = a.L > 1 ? (unsigned)(((uint64_t)1 << 32) / (uint64_t)a.L) + 1u : 0u;  // L == 1: h = o in-kernel
  a.bases_bytes = (unsigned)((uint64_t)n_src * ldb * 4ull);
  a.post_scale = post != nullptr ? post->scale : nullptr;
  a.post_shift = post != nullptr ? post->shift : nullptr;
  a.residual = post != nullptr ? post->residual : nullptr;
  a.post_relu = post != nullptr && post->relu != 0;
  a.stats = stats;
  a.cnt_out = cnt_out;
  a.stat_k = stat_layout(a.aggr, a.A, a.stat_slot);
  a.arg_max = a.stat_slot[STAT_MX] >= 0 ? arg_max : nullptr;
  a.arg_min = a.stat_slot[STAT_MN] >= 0 ? arg_min : nullptr;
  a.arg8_max = a.arg8_min = nullptr;
  if (stats != nullptr) arg8_tables(layer, n, stats, a.arg_max != nullptr, a.arg_min != nullptr, &a.arg8_max, &a.arg8_min);
  a.self_pos = (int)e;

  // EGC_STDVAR_REFERENCE=1 and a var / std layer: the variance by the reference's own float32 formula, mean(x^2) - mean(x)^2
  // (layers.py:203-214, optimized_layers.py:237-244; SURVEY.md 8a note 5) -- squares about zero instead of about the row's
  // first entry -- on the general kernels, one row per wavefront (a row's entries are then summed one after the other in
  // CSR = input order, as the reference's scatter sums them).  Speed is not the point of this mode.
  a.var_ref = stdvar_reference(layer) ? 1 : 0;
  int chunks = 1;
  if (a.slots <= 64) {
    int lg = 0;
    while ((1 << lg) < a.slots) ++lg;
    a.lpr_log2 = a.var_ref ? 6 : lg;
  } else {
    a.lpr_log2 = 6;
    chunks = (a.slots + 63) / 64;
  }
  a.lds_floats_per_wave = a.A * ldb + ((a.W + 3) & ~3);
  int wpb = 4;
  if ((size_t)wpb * a.lds_floats_per_wave * sizeof(float) > 40 * 1024) wpb = 1;
  const size_t lds_bytes = (size_t)wpb * a.lds_floats_per_wave * sizeof(float);
  if (lds_bytes > 64 * 1024) return EGC_ERR_UNSUPPORTED;

  WsLayout w = ws_layout(layer, n, e, graph->n_chunks);
  if (workspace == nullptr || workspace_bytes < w.total) return EGC_ERR_WORKSPACE;
  a.counters = (int*)workspace;
  a.queue = (int*)((char*)workspace + w.counter_bytes);
  a.partial = (float*)((char*)workspace + w.counter_bytes + w.queue_bytes);
  a.partial_nself = (int*)((char*)workspace + w.counter_bytes + w.queue_bytes + w.partial_bytes);
  a.x = nullptr;
  a.F_in = layer->in_channels;
  a.M = 0;
  a.wfrag = nullptr;
  a.wbias4 = nullptr;
  PlanCaps caps = plan_caps(n, e);
  a.rows_per_wave = 0;
  a.n_chunks_hint = (graph->n_chunks >= 0 && graph->n_chunks <= caps.cap_chunks) ? (int)graph->n_chunks : -1;
  const bool force_generic = getenv("EGC_FORCE_GENERIC") != nullptr || a.var_ref != 0;
  if (!force_generic && fast_path_supported(a, layer->weight_layout, chunks)) {
    if (arg_done != nullptr) *arg_done = true;  // the register-resident kernels track the arg positions themselves
    return launch_fast(a, n, caps, stream);
  }

  switch (chunks) {
    case 1: return launch_all<1>(a, n, caps, wpb, lds_bytes, stream);
    case 2: return launch_all<2>(a, n, caps, wpb, lds_bytes, stream,
                                 !force_generic && getenv("EGC_NO_WIDE") == nullptr && wide_path_supported(a, layer->weight_layout));
    case 3: return launch_all<3>(a, n, caps, wpb, lds_bytes, stream);
    case 4: return launch_all<4>(a, n, caps, wpb, lds_bytes, stream);
    default: return EGC_ERR_UNSUPPORTED;
  }
}

int egc_layer_forward_f32(const egc_graph* graph, const egc_layer* layer, const float* x, const float* wcat,
                          const float* bcat, const float* bias, float* bases, int32_t ldb, float* weightings,
                          float* out, void* workspace, size_t workspace_bytes, egc_stream_t stream) {
  if (graph == nullptr) return EGC_ERR_INVALID;
  int st = validate_layer(layer);
  if (st != EGC_OK) return st;
  const int fg = layer->num_bases * layer_basis_stride(layer);  // padded bases are GEMM columns too (zero weights)
  const int w = layer->num_heads * layer->num_bases * layer->num_aggrs;
  st = egc_basis_transform_f32(x, wcat, bcat, graph->n_nodes, layer->in_channels, fg, w, bases, ldb, weightings, stream);
  if (st != EGC_OK) return st;
  return egc_aggregate_combine_f32(graph, layer, bases, ldb, weightings, bias, out, nullptr, nullptr, workspace,
                                   workspace_bytes, stream);
}

int egc_layer_forward_packed(const egc_graph* graph, const egc_layer* layer, const float* x, const void* packed,
                             const float* bcat, const float* bias, float* bases, int32_t ldb, float* weightings,
                             float* out, void* workspace, size_t workspace_bytes, egc_stream_t stream) {
  if (graph == nullptr) return EGC_ERR_INVALID;
  int st = validate_layer(layer);
  if (st != EGC_OK) return st;
  const int fg = layer->num_bases * layer_basis_stride(layer);  // padded bases are GEMM columns too (zero weights)
  const int w = layer->num_heads * layer->num_bases * layer->num_aggrs;
  st = egc_basis_transform_packed_ex(x, packed, bcat, graph->n_nodes, layer->in_channels, fg, w, egc_layer_gemm_flags(layer),
                                     bases, ldb, weightings, stream);
  if (st != EGC_OK) return st;
  return egc_aggregate_combine_f32(graph, layer, bases, ldb, weightings, bias, out, nullptr, nullptr, workspace,
                                   workspace_bytes, stream);
}

// ---- batches of small graphs: tiles of whole graphs (egc_aggregate_tile.hip) ----
static int tile_layer_args(const egc_layer* layer, AggArgs& a, bool two_sets_ok = false) {
  int st = validate_layer(layer);
  if (st != EGC_OK) return st;
  a = AggArgs{};
  a.ldb = egc_bases_ld(layer);
  a.slots = a.ldb / 4;
  a.F_out = layer->out_channels;
  a.H = layer->num_heads;
  a.B = layer->num_bases;
  a.A = layer->num_aggrs;
  a.L = layer->out_channels / layer->num_heads;
  a.Ls = layer_basis_stride(layer);
  a.W = a.H * a.B * a.A;
  a.ldw = a.W;
  for (int t = 0; t < EGC_MAX_AGGRS; ++t) a.aggr[t] = t < a.A ? layer->aggrs[t] : 0;
  a.x_looped = layer->agg_set == EGC_SET_LOOPED;
  a.y_looped = layer->sym_set == EGC_SET_LOOPED;
  a.loops_all = layer->loops_all_nodes != 0;
  if (layer->weight_layout == EGC_LAYOUT_HAB) { a.sa = a.B; a.sb = 1; } else { a.sa = 1; a.sb = a.A; }
  a.act = layer->weight_act;
  a.lpr_log2 = 4;
  int chunks = a.slots <= 64 ? 1 : (a.slots + 63) / 64;
  a.n_nodes = 1;
  if (!fast_path_supported(a, layer->weight_layout, chunks) && !(two_sets_ok && wide_path_supported(a, layer->weight_layout)))
    return EGC_ERR_UNSUPPORTED;      // (65 .. 128 slots: the one-launch kernel finishes such rows in two passes, egc_fused_tile_dev.h)
  return EGC_OK;
}

int32_t egc_batch_tile_nodes(const egc_layer* layer, int32_t max_tile_nodes, int32_t max_tile_edges, int32_t with_post) {
  AggArgs a;
  if (tile_layer_args(layer, a) != EGC_OK || max_tile_nodes < 1 || max_tile_edges < 0) return 0;
  const int lpr = a.slots <= 16 ? 16 : a.slots <= 32 ? 32 : 64;
  a.w_lds_stride = (a.W + 3) & ~3;
  a.bias_lds_floats = (a.H * a.Ls + 3) & ~3;
  a.lds_floats_per_wave = (with_post ? 2 : 1) * a.bias_lds_floats + (64 / lpr) * a.w_lds_stride;
  return tile_capacity(a, max_tile_nodes, max_tile_edges);
}

int egc_batch_plan(const int64_t* graph_ptr, const int64_t* edge_ptr, int64_t n_graphs, const int64_t* dst, int64_t n_edges,
                   int64_t n_nodes, int32_t slot, int32_t* tiles, int32_t n_slots, int32_t* n_tiles, egc_stream_t stream) {
  if (graph_ptr == nullptr || tiles == nullptr || n_tiles == nullptr || n_graphs < 0 || n_edges < 0 || n_nodes < 0 || slot <= 0)
    return EGC_ERR_INVALID;
  if (n_nodes >= ((int64_t)1 << 31) - 1 || n_edges >= ((int64_t)1 << 31) - 1) return EGC_ERR_INVALID;
  if (n_slots != (int32_t)((n_nodes + slot - 1) / slot)) return EGC_ERR_INVALID;
  if (n_edges > 0 && dst == nullptr) return EGC_ERR_INVALID;
  if (n_slots == 0) return hipMemsetAsync(n_tiles, 0, sizeof(int32_t), (hipStream_t)stream) == hipSuccess ? EGC_OK : EGC_ERR_HIP;
  return launch_tile_plan(graph_ptr, n_graphs, dst, n_edges, n_nodes, slot, n_slots, reinterpret_cast<int4*>(tiles), n_tiles,
                          edge_ptr, (hipStream_t)stream);
}

int egc_aggregate_combine_batch_f32(const int32_t* tiles, const int32_t* n_tiles, int32_t n_tiles_bound, int32_t lds_nodes,
                                    int32_t max_tile_nodes, int32_t max_tile_edges, const int64_t* src, const int64_t* dst,
                                    int64_t n_nodes, const int32_t* max_index, const egc_layer* layer, const float* bases,
                                    int32_t ldb, const float* weightings, int32_t ldw, const float* bias, const egc_post* post,
                                    float* out, int32_t* status, int32_t* host_flag, egc_stream_t stream) {
  AggArgs a;
  int st = tile_layer_args(layer, a);
  if (st != EGC_OK) return st;
  if (n_nodes < 0 || n_tiles_bound < 0 || n_nodes >= ((int64_t)1 << 31) - 1) return EGC_ERR_INVALID;
  if (n_nodes == 0 || n_tiles_bound == 0) return EGC_OK;
  if (tiles == nullptr || n_tiles == nullptr || bases == nullptr || weightings == nullptr || out == nullptr || status == nullptr)
    return EGC_ERR_INVALID;
  if (ldb != a.ldb || (reinterpret_cast<uintptr_t>(bases) & 15) != 0) return EGC_ERR_INVALID;
  if (post != nullptr && ((post->scale == nullptr) != (post->shift == nullptr))) return EGC_ERR_INVALID;
  if (!a.loops_all && max_index == nullptr) return EGC_ERR_INVALID;
  if ((uint64_t)n_nodes * (uint64_t)ldb * 4ull > (uint64_t)OOB) return EGC_ERR_UNSUPPORTED;
  if ((uint64_t)n_nodes * (uint64_t)a.F_out * 4ull > (uint64_t)OOB) return EGC_ERR_UNSUPPORTED;   // out / residual descriptors
  a.ldw = ldw > 0 ? ldw : a.W;
  if (a.ldw != a.W && (a.ldw < a.W || (a.ldw & 3) != 0 || (reinterpret_cast<uintptr_t>(weightings) & 15) != 0)) return EGC_ERR_INVALID;
  a.n_nodes = (int)n_nodes;
  a.row_begin = 0;
  a.row_end = (int)n_nodes;
  a.bases = bases;
  a.weightings = weightings;
  a.bias = bias;
  a.out = out;
  a.dis = layer_uses_symnorm(layer) ? bases : nullptr;   // (a flag here: the deg^-1/2 tables are built per tile, in LDS)
  a.bases_bytes = (unsigned)((uint64_t)n_nodes * ldb * 4ull);
  a.post_scale = post != nullptr ? post->scale : nullptr;
  a.post_shift = post != nullptr ? post->shift : nullptr;
  a.residual = post != nullptr ? post->residual : nullptr;
  a.post_relu = post != nullptr && post->relu != 0;
  a.self_pos = 0;
  return launch_tile_simple(a, reinterpret_cast<const int4*>(tiles), n_tiles, n_tiles_bound, lds_nodes, max_tile_nodes,
                            max_tile_edges, src, dst, max_index, status, host_flag, (hipStream_t)stream);
}

// ---- batches of small graphs, the whole layer in one launch (egc_fused_tile.hip) ----
int32_t egc_batch_fused_tile_nodes(const egc_layer* layer, int32_t max_tile_edges, int32_t with_post) {
  AggArgs a;
  if (tile_layer_args(layer, a, true) != EGC_OK) return 0;
  return fused_tile_capacity(a, layer->in_channels, max_tile_edges, with_post != 0);
}

int32_t egc_batch_fused_tile_quantum(const egc_layer* layer) {
  AggArgs a;
  if (tile_layer_args(layer, a, true) != EGC_OK) return 0;
  return fused_tile_quantum(a, layer->in_channels);
}

int64_t egc_batch_fused_pack_bytes(const egc_layer* layer) {
  AggArgs a;
  if (tile_layer_args(layer, a, true) != EGC_OK || !fused_tile_shape(a, layer->in_channels)) return 0;
  return (int64_t)fused_tile_pack_bytes(a, layer->in_channels);
}

int egc_batch_fused_pack(const egc_layer* layer, const float* wcat, const float* bcat, void* packed, int64_t packed_bytes,
                         egc_stream_t stream) {
  AggArgs a;
  int st = tile_layer_args(layer, a, true);
  if (st != EGC_OK) return st;
  if (!fused_tile_shape(a, layer->in_channels)) return EGC_ERR_UNSUPPORTED;
  if (wcat == nullptr || packed == nullptr || packed_bytes < (int64_t)fused_tile_pack_bytes(a, layer->in_channels)) return EGC_ERR_INVALID;
  return fused_tile_pack(a, wcat, bcat, layer->in_channels, a.B * a.Ls, a.W, a.ldb, packed, (hipStream_t)stream);
}

int egc_layer_forward_batch_fused_f32(const int64_t* graph_ptr, const int64_t* edge_ptr, int64_t n_graphs, const int64_t* src,
                                      const int64_t* dst, int64_t n_edges, int64_t n_nodes, const int32_t* max_index,
                                      const egc_layer* layer, const float* x, const void* packed, const float* bias,
                                      const egc_post* post, float* out, int32_t tile_nodes, int32_t max_tile_edges,
                                      int32_t* status, int32_t* host_flag, egc_stream_t stream) {
  AggArgs a;
  int st = tile_layer_args(layer, a, true);
  if (st != EGC_OK) return st;
  if (n_nodes < 0 || n_graphs < 0 || n_edges < 0 || n_nodes >= ((int64_t)1 << 31) - 1 || n_edges >= ((int64_t)1 << 31) - 1)
    return EGC_ERR_INVALID;
  if (n_nodes == 0 || n_graphs == 0) return EGC_OK;
  if (graph_ptr == nullptr || x == nullptr || packed == nullptr || out == nullptr || status == nullptr) return EGC_ERR_INVALID;
  if (n_edges > 0 && (src == nullptr || dst == nullptr)) return EGC_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (reinterpret_cast<uintptr_t>(packed) & 15) != 0) return EGC_ERR_INVALID;
  if (post != nullptr && ((post->scale == nullptr) != (post->shift == nullptr))) return EGC_ERR_INVALID;
  if (!a.loops_all && max_index == nullptr) return EGC_ERR_INVALID;
  if ((uint64_t)n_nodes * (uint64_t)a.F_out * 4ull > (uint64_t)OOB) return EGC_ERR_UNSUPPORTED;
  a.n_nodes = (int)n_nodes;
  a.row_begin = 0;
  a.row_end = (int)n_nodes;
  a.bases = nullptr;          // never in memory
  a.weightings = nullptr;
  a.bases_bytes = 0;
  a.bias = bias;
  a.out = out;
  a.dis = layer_uses_symnorm(layer) ? x : nullptr;   // (a flag: the deg^-1/2 tables are built per tile, in LDS)
  a.post_scale = post != nullptr ? post->scale : nullptr;
  a.post_shift = post != nullptr ? post->shift : nullptr;
  a.residual = post != nullptr ? post->residual : nullptr;
  a.post_relu = post != nullptr && post->relu != 0;
  a.self_pos = 0;
  return launch_fused_tile(a, graph_ptr, edge_ptr, n_graphs, src, dst, n_edges, max_index, x, layer->in_channels, packed,
                           tile_nodes, max_tile_edges, status, host_flag, (hipStream_t)stream);
}

// ---- the same batches, the layer's BACKWARD in one launch (egc_fused_tile.hip, MODE 1) ----
int32_t egc_batch_fused_bwd_tile_nodes(const egc_layer* layer, int32_t max_tile_edges) {
  AggArgs a;
  if (tile_layer_args(layer, a) != EGC_OK) return 0;
  return fused_tile_bwd_capacity(a, layer->in_channels, max_tile_edges);
}

int64_t egc_batch_fused_bwd_pack_bytes(const egc_layer* layer) {
  AggArgs a;
  if (tile_layer_args(layer, a) != EGC_OK || !fused_tile_bwd_shape(a, layer->in_channels)) return 0;
  return (int64_t)fused_tile_bwd_pack_bytes();
}

int egc_batch_fused_bwd_pack(const egc_layer* layer, const float* wcat, void* packed_t, int64_t packed_bytes, egc_stream_t stream) {
  AggArgs a;
  int st = tile_layer_args(layer, a);
  if (st != EGC_OK) return st;
  if (!fused_tile_bwd_shape(a, layer->in_channels)) return EGC_ERR_UNSUPPORTED;
  if (wcat == nullptr || packed_t == nullptr || packed_bytes < (int64_t)fused_tile_bwd_pack_bytes()) return EGC_ERR_INVALID;
  return fused_tile_bwd_pack(a, wcat, layer->in_channels, packed_t, (hipStream_t)stream);
}

int egc_batch_fused_train_pack(const egc_layer* layer, const float* wcat, const float* bcat, void* packed, int64_t packed_bytes,
                               void* packed_t, int64_t packed_t_bytes, egc_stream_t stream) {
  AggArgs a, ab;
  int st = tile_layer_args(layer, a, true);
  if (st == EGC_OK) st = tile_layer_args(layer, ab);
  if (st != EGC_OK) return st;
  if (!fused_tile_shape(a, layer->in_channels) || !fused_tile_bwd_shape(ab, layer->in_channels)) return EGC_ERR_UNSUPPORTED;
  if (wcat == nullptr || packed == nullptr || packed_t == nullptr || packed_bytes < (int64_t)fused_tile_pack_bytes(a, layer->in_channels) ||
      packed_t_bytes < (int64_t)fused_tile_bwd_pack_bytes())
    return EGC_ERR_INVALID;
  return fused_tile_train_pack(ab, wcat, bcat, layer->in_channels, a.B * a.Ls, a.W, a.ldb, packed, packed_t, (hipStream_t)stream);
}

int egc_batch_fused_train_pack_params(const egc_layer* layer, const float* const* bases_parts, int32_t n_parts, const float* comb_weight,
                                      const float* comb_bias, const float* bcat, int32_t num_heads, int32_t num_aggrs, int32_t num_bases,
                                      int32_t basis_len, int32_t basis_stride, int32_t permute_hab, void* packed, int64_t packed_bytes,
                                      void* packed_t, int64_t packed_t_bytes, egc_stream_t stream) {
  AggArgs a, ab;
  int st = tile_layer_args(layer, a, true);
  if (st == EGC_OK) st = tile_layer_args(layer, ab);
  if (st != EGC_OK) return st;
  if (!fused_tile_shape(a, layer->in_channels) || !fused_tile_bwd_shape(ab, layer->in_channels)) return EGC_ERR_UNSUPPORTED;
  if (bases_parts == nullptr || comb_weight == nullptr || packed == nullptr || packed_t == nullptr ||
      (n_parts != 1 && n_parts != num_bases) || n_parts > PACK_MAX_PARTS || (comb_bias != nullptr && bcat != nullptr) ||
      packed_bytes < (int64_t)fused_tile_pack_bytes(a, layer->in_channels) || packed_t_bytes < (int64_t)fused_tile_bwd_pack_bytes())
    return EGC_ERR_INVALID;
  // the parameters must describe the layer the planes are packed for
  if (num_bases != a.B || basis_stride != a.Ls || basis_len <= 0 || basis_len > basis_stride || num_heads != a.H ||
      num_heads * num_bases * num_aggrs != a.W)
    return EGC_ERR_INVALID;
  PackPtrs ptrs;
  for (int i = 0; i < PACK_MAX_PARTS; ++i) ptrs.part[i] = i < n_parts ? const_cast<float*>(bases_parts[i]) : nullptr;
  for (int i = 0; i < n_parts; ++i)
    if (ptrs.part[i] == nullptr) return EGC_ERR_INVALID;
  const PackDims d{layer->in_channels, num_heads, num_aggrs, num_bases, basis_len, basis_stride, n_parts, permute_hab != 0};
  return fused_tile_train_pack_params(ab, ptrs, comb_weight, comb_bias, bcat, d, a.B * a.Ls, a.W, a.ldb, packed, packed_t,
                                      (hipStream_t)stream);
}

int egc_layer_backward_batch_fused_f32(const int64_t* graph_ptr, const int64_t* edge_ptr, int64_t n_graphs, const int64_t* src,
                                       const int64_t* dst, int64_t n_edges, int64_t n_nodes, const int32_t* max_index,
                                       const egc_layer* layer, const float* x, const void* packed, const void* packed_t,
                                       const float* grad_out, float* d_x, const float* d_x_add, float* d_cat, int32_t ld_dcat,
                                       int32_t tile_nodes, int32_t max_tile_edges, int32_t* status, int32_t* host_flag,
                                       egc_stream_t stream) {
  AggArgs a;
  int st = tile_layer_args(layer, a);
  if (st != EGC_OK) return st;
  if (n_nodes < 0 || n_graphs < 0 || n_edges < 0 || n_nodes >= ((int64_t)1 << 31) - 1 || n_edges >= ((int64_t)1 << 31) - 1)
    return EGC_ERR_INVALID;
  if (n_nodes == 0 || n_graphs == 0) return EGC_OK;
  if (graph_ptr == nullptr || x == nullptr || packed == nullptr || packed_t == nullptr || grad_out == nullptr || d_x == nullptr ||
      status == nullptr)
    return EGC_ERR_INVALID;
  if (n_edges > 0 && (src == nullptr || dst == nullptr)) return EGC_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (reinterpret_cast<uintptr_t>(packed) & 15) != 0 ||
      (reinterpret_cast<uintptr_t>(packed_t) & 15) != 0 || (reinterpret_cast<uintptr_t>(grad_out) & 15) != 0 ||
      (reinterpret_cast<uintptr_t>(d_x_add) & 3) != 0)
    return EGC_ERR_INVALID;
  if (d_cat != nullptr && (ld_dcat < a.ldb + a.W || (ld_dcat & 3) != 0 || (reinterpret_cast<uintptr_t>(d_cat) & 15) != 0)) return EGC_ERR_INVALID;
  if (!a.loops_all && max_index == nullptr) return EGC_ERR_INVALID;
  if ((uint64_t)n_nodes * (uint64_t)a.F_out * 4ull > (uint64_t)OOB) return EGC_ERR_UNSUPPORTED;
  a.n_nodes = (int)n_nodes;
  a.row_begin = 0;
  a.row_end = (int)n_nodes;
  a.bases = nullptr;
  a.weightings = nullptr;
  a.bases_bytes = 0;
  a.dis = layer_uses_symnorm(layer) ? x : nullptr;   // (a flag: the deg^-1/2 tables are built per tile, in LDS)
  a.self_pos = 0;
  return launch_fused_tile_bwd(a, graph_ptr, edge_ptr, n_graphs, src, dst, n_edges, max_index, x, layer->in_channels, packed, packed_t,
                               grad_out, d_x, d_x_add, d_cat, ld_dcat, tile_nodes, max_tile_edges, status, host_flag, (hipStream_t)stream);
}

int32_t egc_layer_gemm_flags(const egc_layer* layer) {
  // 0 for every layer since round 4: the variance is accumulated about the row's first entry (FAcc::sh), which takes the
  // cancellation -- and with it the amplification of what the 22-bit operand split drops -- out of std / var (44 fuzz seeds,
  // 5,280 configurations: every std / var layer within 7e-7 of float64 with either GEMM, where the float32 restatement
  // itself is up to 3.7e-4 off; profiles/r04_stdvar_shift.md).  EGC_GEMM_STDVAR_24BIT=1 brings the old choice back.
  if (layer == nullptr) return 0;
  if (egc::stdvar_reference(layer)) return EGC_GEMM_24BIT;     // (the reference-formula mode: 24-bit operands, no one-launch path)
  const char* e = getenv("EGC_GEMM_STDVAR_24BIT");
  if (e == nullptr || e[0] == '\0' || (e[0] == '0' && e[1] == '\0')) return 0;
  for (int t = 0; t < layer->num_aggrs && t < EGC_MAX_AGGRS; ++t)
    if (layer->aggrs[t] == EGC_AGGR_VAR || layer->aggrs[t] == EGC_AGGR_STD) return EGC_GEMM_24BIT;
  return 0;
}

}  // extern "C"
