// Backward of the fused EGC aggregate+combine (gfx950).  Not in the reference as code: PyTorch autograd
// derives it implicitly through experiments/layers.py:103-138 / optimized_layers.py:186-208 (gather ->
// scatter per aggregator -> stack -> weighted sum).  SURVEY.md 8(f) rank 1.
//
// Given g = dL/d out [N, F_out], per destination row i with aggregates agg_a:
//   d w'[h,b,a] = sum_l g[h,l] * agg_a[b,l]                (then through the weight nonlinearity)
//   d agg_a[b,l] = sum_h w'[h,b,a] * g[h,l]
// and per source row j (all aggregators that are linear in the messages collapse into three
// per-destination tables that are summed over j's OUT-neighbours, i.e. a gather over the transposed CSR):
//   d bases[j] = sum_{i in out(j)} T[i]  +  dis_j * sum_i S[i]  +  bases[j] * sum_i V[i]   (+ self-loop terms)
//     T[i] = d agg_sum + d agg_mean / cnt_i - 2 mean_i dvar_i / cnt_i
//     S[i] = dis_i * d agg_symnorm
//     V[i] = 2 dvar_i / cnt_i,     dvar_i = d agg_var + d agg_std * [var_i > 0] / (2 std_i)
//   max / min: the gradient goes to the FIRST entry (input order; the self-loop of a LOOPED set is last)
//   attaining the extremum -- torch_scatter's arg semantics -- with one float atomic per (row, column).
//
// Nothing is gathered twice: the training forward (egc_aggregate_combine_train_f32) keeps every row's raw
// running aggregates (`stats`, `cnt`), so the destination side is ROW-LOCAL, and the source side is a sum-only
// SpMM of the tables over the transposed CSR.  Three kernels:
//   arg_extrema_kernel   (part of the training forward, only with max / min): lane group per row, compares each
//                        gathered slot with the row's extremum and keeps the first CSR position that matches
//   bwd_dst_kernel       one wavefront per destination row, lane = basis column: d w' and the tables T/S/V,
//                        plus X/N = d agg_max / d agg_min, which the source side routes by the arg positions
//   bwd_src_kernel       lane group per source row (several rows per wavefront, FU loads in flight); rows
//                        longer than EGC_LONG_ROW_THRESHOLD are cut into chunks (plan of the transposed graph)
//                        whose partial sums arrive by float atomics
// Any H, B, L, A and every weight nonlinearity; deterministic except for the order of float atomics on hub rows.
#include <stdlib.h>

#include "egc_aggregate_dev.h"

namespace egc {

constexpr int ARG_INIT = 0x7f7f7f7f;  // memset pattern of the arg buffers: "no entry found yet"

// ---------------------------------------------------------------------------------------------
// arg-extrema (training forward)
// ---------------------------------------------------------------------------------------------
struct ArgArgs {
  const int* rowptr;
  const int* col;
  const float* bases;
  const float* stats;
  const int* plan;
  int* arg_max;
  int* arg_min;
  int n_nodes, n_edges, chunk_blocks;
  int ldb, slots, lpr_log2, stat_k, slot_mx, slot_mn;
  int x_looped;
  unsigned bases_bytes;
};

// Lane group per destination row (G rows per wavefront): the group walks its row in order and keeps, per
// column, the first position whose gathered value equals the row's extremum -- no atomics.  Rows longer than
// EGC_LONG_ROW_THRESHOLD are cut into the plan's chunks (leading blocks, one wavefront per chunk, the G groups
// splitting its entries); their candidates meet in an atomicMin on the position.
constexpr int ARG_FU = 4;
template <int NS>
__global__ void __launch_bounds__(256) arg_extrema_kernel(ArgArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int LPR = 1 << a.lpr_log2, G = 64 >> a.lpr_log2;
  const int g = lane >> a.lpr_log2, q = lane & (LPR - 1);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)a.bases, 0, a.bases_bytes, 0x00020000);
  int row, start, end, first, step;
  bool atomic;
  if ((int)blockIdx.x < a.chunk_blocks) {
    const int c = blockIdx.x * 4 + wave;
    if (c >= a.plan[1]) return;
    const int cap_long = a.plan[2], cap_chunks = a.plan[3];
    const int* long_row = a.plan + 4;
    const int* chunk_slot = long_row + 2 * cap_long;
    const int* chunk_begin = chunk_slot + cap_chunks;
    row = long_row[chunk_slot[c]];
    start = chunk_begin[c];
    end = min(start + EGC_LONG_ROW_CHUNK, a.rowptr[row + 1]);
    first = g; step = G;
    atomic = true;
  } else {
    row = (((int)blockIdx.x - a.chunk_blocks) * 4 + wave) * G + g;
    if (row >= a.n_nodes) return;
    start = a.rowptr[row]; end = a.rowptr[row + 1];
    if (end - start > EGC_LONG_ROW_THRESHOLD) return;  // the chunk blocks own this row (arg pre-set to ARG_INIT)
    first = 0; step = 1;
    atomic = false;
  }
  f4 mx[NS], mn[NS];
  int4 ax[NS], an[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int s = q + k * LPR;
    const float* st = a.stats + ((int64_t)row * a.stat_k) * a.ldb + 4 * s;
    const f4 nanv = f4{NAN, NAN, NAN, NAN};  // never equal: idle slots / unused extremum
    mx[k] = (a.arg_max != nullptr && s < a.slots) ? *reinterpret_cast<const f4*>(st + a.slot_mx * a.ldb) : nanv;
    mn[k] = (a.arg_min != nullptr && s < a.slots) ? *reinterpret_cast<const f4*>(st + a.slot_mn * a.ldb) : nanv;
    ax[k] = an[k] = int4{ARG_INIT, ARG_INIT, ARG_INIT, ARG_INIT};
  }
  for (int p0 = start + first; p0 < end; p0 += step * ARG_FU) {
    int j[ARG_FU];
#pragma unroll
    for (int u = 0; u < ARG_FU; ++u) {
      const int p = p0 + u * step;
      j[u] = p < end ? a.col[p] : -1;
      if (a.x_looped && j[u] == row) j[u] = -1;  // excluded from the LOOPED set; its self-loop is appended LAST
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int s = q + k * LPR;
      f4 v[ARG_FU];
#pragma unroll
      for (int u = 0; u < ARG_FU; ++u)
        v[u] = load_slot(rb, (j[u] >= 0 && s < a.slots) ? (unsigned)j[u] * (unsigned)a.ldb * 4u + (unsigned)s * 16u : OOB);
#pragma unroll
      for (int u = 0; u < ARG_FU; ++u) {
        if (j[u] < 0) continue;
        const int p = p0 + u * step;   // increasing: min() keeps the first hit
        ax[k].x = min(ax[k].x, v[u].x == mx[k].x ? p : ARG_INIT); ax[k].y = min(ax[k].y, v[u].y == mx[k].y ? p : ARG_INIT);
        ax[k].z = min(ax[k].z, v[u].z == mx[k].z ? p : ARG_INIT); ax[k].w = min(ax[k].w, v[u].w == mx[k].w ? p : ARG_INIT);
        an[k].x = min(an[k].x, v[u].x == mn[k].x ? p : ARG_INIT); an[k].y = min(an[k].y, v[u].y == mn[k].y ? p : ARG_INIT);
        an[k].z = min(an[k].z, v[u].z == mn[k].z ? p : ARG_INIT); an[k].w = min(an[k].w, v[u].w == mn[k].w ? p : ARG_INIT);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int s = q + k * LPR;
    if (s >= a.slots) continue;
    const int64_t o = (int64_t)row * a.ldb + 4 * s;
    if (!atomic) {
      if (a.arg_max != nullptr) *reinterpret_cast<int4*>(a.arg_max + o) = ax[k];
      if (a.arg_min != nullptr) *reinterpret_cast<int4*>(a.arg_min + o) = an[k];
    } else {
      if (a.arg_max != nullptr) {
        if (ax[k].x != ARG_INIT) atomicMin(&a.arg_max[o], ax[k].x);
        if (ax[k].y != ARG_INIT) atomicMin(&a.arg_max[o + 1], ax[k].y);
        if (ax[k].z != ARG_INIT) atomicMin(&a.arg_max[o + 2], ax[k].z);
        if (ax[k].w != ARG_INIT) atomicMin(&a.arg_max[o + 3], ax[k].w);
      }
      if (a.arg_min != nullptr) {
        if (an[k].x != ARG_INIT) atomicMin(&a.arg_min[o], an[k].x);
        if (an[k].y != ARG_INIT) atomicMin(&a.arg_min[o + 1], an[k].y);
        if (an[k].z != ARG_INIT) atomicMin(&a.arg_min[o + 2], an[k].z);
        if (an[k].w != ARG_INIT) atomicMin(&a.arg_min[o + 3], an[k].w);
      }
    }
  }
}

// ARG_INIT -> n_edges (the appended self-loop attains the extremum) or -1 (empty row)
__global__ void __launch_bounds__(256) arg_finalize_kernel(int* arg, const int* cnt, int64_t total, int ldb, int n_edges) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= total) return;
  if (arg[k] == ARG_INIT) arg[k] = cnt[k / ldb] > 0 ? n_edges : -1;
}

// arg8 tables (egc_aggregate_dev.h) from int32 arg rows: the generic training forward's last step
__global__ void __launch_bounds__(256) arg8_kernel(const int* __restrict__ arg, const int* __restrict__ rowptr,
                                                   unsigned* __restrict__ arg8, int64_t quads, int quads_per_row,
                                                   int n_edges) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread: 4 columns -> one packed dword
  if (k >= quads) return;
  const int row = (int)(k / quads_per_row);
  const int start = rowptr[row];
  const int4 a = *reinterpret_cast<const int4*>(arg + 4 * k);
  arg8[k] = arg8_pack(a, start, n_edges);
}

int arg_extrema(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb, const float* stats,
                const int32_t* cnt, int32_t* arg_max, int32_t* arg_min, unsigned* arg8_max, unsigned* arg8_min,
                hipStream_t stream) {
  int slot[5];
  const int k = stat_layout(layer->aggrs, layer->num_aggrs, slot);
  if (slot[STAT_MX] < 0) arg_max = nullptr;
  if (slot[STAT_MN] < 0) arg_min = nullptr;
  if (arg_max == nullptr && arg_min == nullptr) return EGC_OK;
  const int64_t n = graph->n_nodes, e = graph->n_edges;
  if (n == 0) return EGC_OK;
  const size_t bytes = (size_t)n * ldb * sizeof(int32_t);
  if (arg_max != nullptr) EGC_HIP_TRY(hipMemsetAsync(arg_max, 0x7f, bytes, stream));
  if (arg_min != nullptr) EGC_HIP_TRY(hipMemsetAsync(arg_min, 0x7f, bytes, stream));
  const int64_t n_src = graph->n_src_rows > 0 ? graph->n_src_rows : n;
  ArgArgs a;
  a.rowptr = graph->rowptr; a.col = graph->col; a.bases = bases; a.stats = stats;
  a.arg_max = arg_max; a.arg_min = arg_min;
  a.n_nodes = (int)n; a.n_edges = (int)e;
  a.ldb = ldb; a.slots = ldb / 4; a.stat_k = k; a.slot_mx = slot[STAT_MX]; a.slot_mn = slot[STAT_MN];
  a.x_looped = layer->agg_set == EGC_SET_LOOPED;
  a.bases_bytes = (unsigned)((uint64_t)n_src * ldb * 4ull);
  int lg = 4;
  while ((1 << lg) < a.slots && lg < 6) ++lg;
  a.lpr_log2 = lg;
  a.plan = graph->plan;
  if (e > 0) {
    if (graph->plan == nullptr) return EGC_ERR_INVALID;
    const PlanCaps caps = plan_caps(n, e);
    const int64_t n_chunks = (graph->n_chunks >= 0 && graph->n_chunks <= caps.cap_chunks) ? graph->n_chunks : caps.cap_chunks;
    a.chunk_blocks = (int)ceil_div(n_chunks, 4);
    const unsigned grid = (unsigned)(a.chunk_blocks + ceil_div(n, (int64_t)4 * (64 >> lg)));
    switch ((a.slots + (1 << lg) - 1) >> lg) {
      case 1: arg_extrema_kernel<1><<<grid, 256, 0, stream>>>(a); break;
      case 2: arg_extrema_kernel<2><<<grid, 256, 0, stream>>>(a); break;
      case 3: arg_extrema_kernel<3><<<grid, 256, 0, stream>>>(a); break;
      case 4: arg_extrema_kernel<4><<<grid, 256, 0, stream>>>(a); break;
      default: return EGC_ERR_UNSUPPORTED;
    }
    EGC_LAUNCH_CHECK("arg_extrema_kernel");
  }
  const int64_t total = n * ldb;
  if (arg_max != nullptr) arg_finalize_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, stream>>>(arg_max, cnt, total, ldb, (int)e);
  if (arg_min != nullptr) arg_finalize_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, stream>>>(arg_min, cnt, total, ldb, (int)e);
  EGC_LAUNCH_CHECK("arg_finalize_kernel");
  const int64_t quads = total / 4;
  if (arg_max != nullptr && arg8_max != nullptr)
    arg8_kernel<<<(unsigned)ceil_div(quads, 256), 256, 0, stream>>>(arg_max, graph->rowptr, arg8_max, quads, ldb / 4, (int)e);
  if (arg_min != nullptr && arg8_min != nullptr)
    arg8_kernel<<<(unsigned)ceil_div(quads, 256), 256, 0, stream>>>(arg_min, graph->rowptr, arg8_min, quads, ldb / 4, (int)e);
  EGC_LAUNCH_CHECK("arg8_kernel");
  return EGC_OK;
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
struct BwdArgs {
  const int* col;            // destination-side CSR entries (arg positions -> source ids)
  const float* dis;          // deg^-1/2 of the symnorm edge set or nullptr
  const int* max_index;
  // transposed graph (rows = sources, entries = destinations) with its own long-row plan
  const int* t_rowptr;
  const int* t_col;
  const int* t_plan;
  const float* bases;        // [n_src_rows, ldb]
  const float* weightings;   // [N, W] pre-activation
  const float* grad_out;     // [N, F_out]
  const float* stats;        // [N, stat_k, ldb] raw aggregates of the forward
  const int* cnt;            // [N] entries of the row's aggregation set
  const int* arg_max;        // [N, ldb] CSR position / n_edges (self loop) / -1, or nullptr
  const int* arg_min;
  const int* rowptr;         // destination-side row pointers (arg positions relative to their row)
  unsigned char* arg8_max;   // [N, ldb] the same positions as one byte each, relative to the row's first entry
  unsigned char* arg8_min;   //          (ARG8_NONE: self-loop / empty row, ARG8_FAR: >= ARG8_NONE entries into the row)
  float* d_bases;            // [n_src_rows, ld_db >= ldb]
  int overwrite_db;          // square graph: every d_bases row is WRITTEN (long source rows zeroed by the destination
                             // kernel, short ones stored by the source kernel); else the host zero-fills and rows are added to
  float* d_weightings;       // [N, ld_dw >= W]
  int ld_db, ld_dw;          // row strides (floats) of the two gradient arrays
  float* tab_t;              // [N, ldb]
  int need_t;                // some aggregator is linear in the plain messages (sum / mean / var / std): T is not all zero
  float* tab_s;              // [N, ldb] or nullptr
  float* tab_v;              // [N, ldb] or nullptr
  float* tab_x;              // [N, ldb] d agg_max, or nullptr
  float* tab_n;              // [N, ldb] d agg_min, or nullptr
  const int* t_edge_id;      // transposed entry -> position of the same edge in the destination-side CSR
  const unsigned* rec_x;     // [n_edges][16] extremum-gradient records per destination-CSR entry (records_from_columns), or nullptr
  const unsigned* rec_n;
  unsigned rec_bytes;
  int dst_group_floats;      // bwd_dst_fast_kernel: LDS floats per lane group
  const int* d_plan;         // long-row plan of the destination-side graph (records of hub rows)
  int dst_row_blocks;        // bwd_dst_fast_kernel: blocks of the row role; the blocks behind them build the records of hub-row chunks
  int rec_fused;             // bwd_dst_fast_kernel builds every record itself (row role: short rows; trailing blocks: hub-row chunks)
  int n_nodes, n_src_rows, n_edges;
  int ldb, slots, F_g, F_out, W, H, B, A, L, Ls;  // F_g = B * Ls: bases columns incl. per-basis padding
  int aggr[EGC_MAX_AGGRS];
  int stat_slot[5], stat_k;
  int x_looped, y_looped, loops_all;
  int act;
  int lds_floats_per_wave;
  int lpr_log2, chunk_blocks;
  unsigned tab_bytes;
};

// One wavefront per destination row.  LDS per wavefront: agg [A][ldb], g [F_out], w' [W], d w' [W].
__global__ void __launch_bounds__(256) bwd_dst_kernel(BwdArgs a) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row = blockIdx.x * (blockDim.x >> 6) + wave;
  if (row >= a.n_nodes) return;
  if (a.overwrite_db && a.t_rowptr[row + 1] - a.t_rowptr[row] > EGC_LONG_ROW_THRESHOLD)   // its chunks add by atomics
    for (int c = lane; c < a.ldb; c += 64) a.d_bases[(int64_t)row * a.ld_db + c] = 0.f;
  float* lds_agg = smem + wave * a.lds_floats_per_wave;
  float* lds_g = lds_agg + a.A * a.ldb;
  float* lds_w = lds_g + ((a.F_out + 3) & ~3);
  float* lds_dagg = lds_w + ((a.W + 3) & ~3);  // d w' scratch [W]
  const float dis_i = a.dis != nullptr ? a.dis[row] : 0.f;
  const int AB = a.A * a.B;

  // g row and activated weights -> LDS
  for (int k = lane; k < a.F_out; k += 64) lds_g[k] = a.grad_out[(int64_t)row * a.F_out + k];
  for (int k = lane; k < a.W; k += 64) {
    float w = a.weightings[(int64_t)row * a.W + k];
    if (a.act == EGC_ACT_SIGMOID) w = 1.0f / (1.0f + expf(-w));
    else if (a.act == EGC_ACT_HARDTANH) w = fminf(fmaxf(w, -1.0f), 1.0f);
    lds_w[k] = w;
  }
  if (a.act == EGC_ACT_SOFTMAX) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int h = lane; h < a.H; h += 64) {
      float* wh = lds_w + h * AB;
      float m = -INFINITY;
      for (int k = 0; k < AB; ++k) m = fmaxf(m, wh[k]);
      float s = 0.f;
      for (int k = 0; k < AB; ++k) { const float e = expf(wh[k] - m); wh[k] = e; s += e; }
      for (int k = 0; k < AB; ++k) wh[k] = wh[k] / s;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  const int cnt = a.cnt[row];
  const float cntf = (float)max(cnt, 1);
  const float* st = a.stats + ((int64_t)row * a.stat_k) * a.ldb;
  for (int c0 = 0; c0 < a.F_g; c0 += 64) {
    const int c = c0 + lane;
    if (c >= a.F_g) continue;
    const int b = c / a.Ls, l = c - b * a.Ls;
    if (l >= a.L) {  // padding column of a padded basis: nothing flows through it
      for (int t = 0; t < a.A; ++t) lds_agg[t * a.ldb + c] = 0.f;
      if (a.need_t) __builtin_nontemporal_store((float)(0.f), &a.tab_t[(int64_t)row * a.ldb + c]);
      if (a.tab_s != nullptr) __builtin_nontemporal_store((float)(0.f), &a.tab_s[(int64_t)row * a.ldb + c]);
      if (a.tab_v != nullptr) __builtin_nontemporal_store((float)(0.f), &a.tab_v[(int64_t)row * a.ldb + c]);
      if (a.tab_x != nullptr) __builtin_nontemporal_store((float)(0.f), &a.tab_x[(int64_t)row * a.ldb + c]);
      if (a.tab_n != nullptr) __builtin_nontemporal_store((float)(0.f), &a.tab_n[(int64_t)row * a.ldb + c]);
      continue;
    }
    const float sum = a.stat_slot[STAT_SUM] >= 0 ? st[a.stat_slot[STAT_SUM] * a.ldb + c] : 0.f;
    const float sq = a.stat_slot[STAT_SQ] >= 0 ? st[a.stat_slot[STAT_SQ] * a.ldb + c] : 0.f;
    const float mx = a.stat_slot[STAT_MX] >= 0 ? st[a.stat_slot[STAT_MX] * a.ldb + c] : 0.f;
    const float mn = a.stat_slot[STAT_MN] >= 0 ? st[a.stat_slot[STAT_MN] * a.ldb + c] : 0.f;
    const float ws = a.stat_slot[STAT_WS] >= 0 ? st[a.stat_slot[STAT_WS] * a.ldb + c] : 0.f;
    const float mean = sum / cntf;
    const float var = sq;       // the record's STAT_SQ slot IS the forward's variance (same mask, same std as the forward's)
    const float sd = sqrtf(fmaxf(var, 0.f) + 1e-5f);
    for (int t = 0; t < a.A; ++t) {
      float val;
      switch (a.aggr[t]) {
        case EGC_AGGR_SUM: val = sum; break;
        case EGC_AGGR_MEAN: val = mean; break;
        case EGC_AGGR_MAX: val = cnt > 0 ? mx : 0.f; break;
        case EGC_AGGR_MIN: val = cnt > 0 ? mn : 0.f; break;
        case EGC_AGGR_VAR: val = var; break;
        case EGC_AGGR_STD: val = sd; break;
        default: val = ws; break;
      }
      lds_agg[t * a.ldb + c] = val;
    }
    // d agg_t[c] = sum_h w'[h][b][t] * g[h*L + l]
    float d_t = 0.f, d_s = 0.f, d_v = 0.f;
    for (int t = 0; t < a.A; ++t) {
      float d = 0.f;
      for (int h = 0; h < a.H; ++h) d = fmaf(lds_w[h * AB + b * a.A + t], lds_g[h * a.L + l], d);
      switch (a.aggr[t]) {
        case EGC_AGGR_SUM: d_t += d; break;
        case EGC_AGGR_MEAN: d_t += d / cntf; break;
        case EGC_AGGR_MAX: __builtin_nontemporal_store((float)(cnt > 0 ? d : 0.f), &a.tab_x[(int64_t)row * a.ldb + c]); break;
        case EGC_AGGR_MIN: __builtin_nontemporal_store((float)(cnt > 0 ? d : 0.f), &a.tab_n[(int64_t)row * a.ldb + c]); break;
        case EGC_AGGR_VAR: d_v += d; break;
        case EGC_AGGR_STD: d_v += (var > 0.f) ? d / (2.0f * sd) : 0.f; break;
        default: d_s += d * dis_i; break;
      }
    }
    // var = E[x^2] - mean^2:  d/dx_j = 2 (x_j - mean) / cnt
    if (a.need_t) __builtin_nontemporal_store((float)(d_t - 2.0f * mean * d_v / cntf), &a.tab_t[(int64_t)row * a.ldb + c]);
    if (a.tab_s != nullptr) __builtin_nontemporal_store((float)(d_s), &a.tab_s[(int64_t)row * a.ldb + c]);
    if (a.tab_v != nullptr) __builtin_nontemporal_store((float)(2.0f * d_v / cntf), &a.tab_v[(int64_t)row * a.ldb + c]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- d w'[h][b][t] = sum_l g[h*L + l] * agg_t[b*L + l], then through the nonlinearity
  for (int k = lane; k < a.W; k += 64) {
    const int h = k / AB, r = k - h * AB, b = r / a.A, t = r - b * a.A;
    float d = 0.f;
    for (int l = 0; l < a.L; ++l) d = fmaf(lds_g[h * a.L + l], lds_agg[t * a.ldb + b * a.Ls + l], d);
    lds_dagg[k] = d;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int k = lane; k < a.W; k += 64) {
    const float w = lds_w[k];
    float d = lds_dagg[k];
    if (a.act == EGC_ACT_SIGMOID) d = d * w * (1.0f - w);
    else if (a.act == EGC_ACT_HARDTANH) {
      const float pre = a.weightings[(int64_t)row * a.W + k];
      d = (pre > -1.0f && pre < 1.0f) ? d : 0.f;
    } else if (a.act == EGC_ACT_SOFTMAX) {
      const int h = k / AB;
      float dot = 0.f;
      for (int q = 0; q < AB; ++q) dot = fmaf(lds_dagg[h * AB + q], lds_w[h * AB + q], dot);
      d = w * (d - dot);
    }
    __builtin_nontemporal_store(d, &a.d_weightings[(int64_t)row * a.ld_dw + k]);
  }
}

// ---------------------------------------------------------------------------------------------
// extremum-gradient records
// ---------------------------------------------------------------------------------------------
// The gradient of max / min goes, per (destination row, column), to ONE entry of the row: the one its arg position names.
// Seen from a transposed entry, that is a handful of the destination row's columns (F_g / degree on average) -- fetched
// as arg bytes plus 16-byte pieces of the X row it costs ~260 bytes of sectors per entry for ~17 bytes of payload.
// The destination side (inside bwd_dst_fast_kernel; bwd_records_kernel behind the LDS-based destination kernel) turns the
// (arg, X) rows into one 64-byte record per destination-CSR entry: the (value, column) pairs
// that entry receives.  The source kernel then reads ONE 64-byte line per entry, with no dependent loads before it
// (the arg-byte path needs rowptr[dst] -> arg bytes -> X pieces), and adds the values into a per-lane-group LDS row.
// Entries that receive more than REC_ITEMS columns (rows of a few entries) are marked REC_OVERFLOW; the source kernel
// compares the int32 arg row for them.
#ifndef EGC_REC_LOAD_AUX
#define EGC_REC_LOAD_AUX 0
#endif
constexpr int REC_ITEMS = 12;             // (value, column) pairs per record
constexpr unsigned REC_OVERFLOW = 0xffu;  // dword 15: count, or this
constexpr int REC_FIT_COLUMNS = 10;        // records are built where an entry receives at most this many columns on average (ldb N / E)
constexpr int REC_BALLOT_MAX = 8;         // entries of a one-row wavefront up to which the columns are ranked by ballots
// record = 16 dwords: [0..11] values, [12..14] their columns (one byte each: ldb <= 256), [15] count

struct RecArgs {
  const int* rowptr;
  const int* plan;     // long-row plan of the destination-side graph
  const int* arg;      // [N, ldb] CSR position / n_edges / -1
  const float* x;      // [N, ldb] d agg_max (d agg_min)
  unsigned* rec;       // [n_edges][16]
  int n_nodes, ldb, slots, chunk_blocks, group_u32;
  int short_rows;      // 0: the destination kernel writes the short rows' records itself (bwd_dst_fast_kernel)
};

// Four lanes per entry, 16 bytes of its record each: a store instruction writes whole 64-byte lines (a lane per
// record would touch 64 lines per instruction).  Parts 0..2: values o + 4 v .. + 3; part 3: the column bytes
// o .. o + 11 (three dwords cut out of four aligned ones) and the count.
template <int GS>
__device__ inline void records_store(const unsigned* cnt, const unsigned* off, const unsigned* vals, const unsigned char* cols,
                                     unsigned* rec0, int n_entries, int q) {
  const int v = q & 3;
  for (int e = q >> 2; e < n_entries; e += GS / 4) {
    const unsigned cn = cnt[e], o = off[e];
    const unsigned m = min(cn, (unsigned)REC_ITEMS);
    const unsigned* src = v < 3 ? vals + o + 4 * v : reinterpret_cast<const unsigned*>(cols) + (o >> 2);
    const unsigned d0 = src[0], d1 = src[1], d2 = src[2], d3 = src[3];
    uint4 w;
    if (v < 3) {
      const unsigned i0 = 4 * v;
      w = uint4{i0 < m ? d0 : 0u, i0 + 1 < m ? d1 : 0u, i0 + 2 < m ? d2 : 0u, i0 + 3 < m ? d3 : 0u};
    } else {
      const unsigned sh = o & 3u;
      auto keep = [&](unsigned first) -> unsigned {   // bytes first .. first + 3 of the list that exist
        return m >= first + 4 ? 0xffffffffu : m > first ? (1u << (8 * (m - first))) - 1u : 0u;
      };
      w = uint4{__builtin_amdgcn_alignbyte(d1, d0, sh) & keep(0), __builtin_amdgcn_alignbyte(d2, d1, sh) & keep(4),
                __builtin_amdgcn_alignbyte(d3, d2, sh) & keep(8), cn <= (unsigned)REC_ITEMS ? cn : REC_OVERFLOW};
    }
#ifdef EGC_REC_NT_STORE
    __builtin_nontemporal_store(w, reinterpret_cast<uint4*>(rec0 + (int64_t)e * 16) + v);
#else
    reinterpret_cast<uint4*>(rec0 + (int64_t)e * 16)[v] = w;
#endif
  }
}

// One lane group (GS lanes, NS slots of four columns per lane: slot q + k GS) builds the records of n_entries <= TS
// consecutive entries of a destination row; rel[k][c] = the entry (relative to the first of them) that column
// 4 (q + k GS) + c names, or anything outside [0, n_entries).  Counting sort of the columns by entry in LDS
// (lds: count [TS] | offset [TS] | items [ldb] x (value, column)), then one 64-byte store sequence per entry.
template <int GS_LOG2, int NS, int TS>
__device__ inline void records_from_columns(unsigned* lds, int ldb, unsigned* rec0, int n_entries, int (&rel)[NS][4],
                                            const float (&xv)[NS][4], int lane) {
  constexpr int GS = 1 << GS_LOG2, CPL = TS / GS;   // counters per lane
  static_assert(CPL == 1 || CPL == 2 || CPL == 4, "table size");
  const int q = lane & (GS - 1);
  unsigned* cnt = lds;
  unsigned* off = lds + TS;
  unsigned* vals = lds + 2 * TS;                                       // [ldb] values, sorted by entry
  unsigned char* cols = reinterpret_cast<unsigned char*>(vals + ldb);  // [ldb] their columns
  // A whole wavefront on ONE row of a few entries (the 64-lane groups of the reference's wide nets on molecule batches: two to
  // four entries per row): every column's place by ballots -- entry by entry, column slot by column slot, lanes ascending: the
  // order the returning LDS atomics below produce, so the records are the same bits -- with the counts and offsets falling out
  // of the same popcounts.  The atomics of such a row all land on two to four LDS words and run one lane per clock: 224 columns
  // were ~260 cycles of a CU's LDS pipe per row, 22 of the 31 us this function cost at 224 / H4 / B4 on molhiv b2048 (round 6).
  if constexpr (GS == 64 && NS == 1) {
    const int ne = __builtin_amdgcn_readfirstlane(n_entries);
    if (ne <= REC_BALLOT_MAX) {
      unsigned pos[4] = {0u, 0u, 0u, 0u};
      unsigned run = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (!((unsigned)rel[0][c] < (unsigned)ne)) rel[0][c] = -1;
      for (int e = 0; e < ne; ++e) {
        unsigned base = run;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool mine = rel[0][c] == e;
          const unsigned long long mask = __ballot(mine);
          const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
          if (mine) pos[c] = base + below;
          base += (unsigned)__builtin_popcountll(mask);
        }
        if (lane == 0) { cnt[e] = base - run; off[e] = run; }
        run = base;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (rel[0][c] >= 0) {
          vals[pos[c]] = __float_as_uint(xv[0][c]);
          cols[pos[c]] = (unsigned char)(4 * q + c);
        }
      records_store<GS>(cnt, off, vals, cols, rec0, ne, q);
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < CPL; ++i) cnt[CPL * q + i] = 0u;
  unsigned rank[NS][4];
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool ok = (unsigned)rel[k][c] < (unsigned)n_entries;
      if (!ok) rel[k][c] = -1;
      rank[k][c] = ok ? __hip_atomic_fetch_add(&cnt[rel[k][c]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
    }
  // exclusive prefix of the counts (LDS operations of one wavefront execute in order: no barrier)
  unsigned cl[CPL], t = 0;
#pragma unroll
  for (int i = 0; i < CPL; ++i) { cl[i] = cnt[CPL * q + i]; t += cl[i]; }
  unsigned incl = t;
#pragma unroll
  for (int d = 1; d < GS; d <<= 1) {
    const unsigned o = __shfl(incl, max(lane - d, 0));
    if (q >= d) incl += o;
  }
  unsigned run = incl - t;
#pragma unroll
  for (int i = 0; i < CPL; ++i) { off[CPL * q + i] = run; run += cl[i]; }
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (rel[k][c] >= 0) {
        const unsigned p = off[rel[k][c]] + rank[k][c];
        vals[p] = __float_as_uint(xv[k][c]);
        cols[p] = (unsigned char)(4 * (q + k * GS) + c);
      }
  records_store<GS>(cnt, off, vals, cols, rec0, n_entries, q);
}

// the same from the (arg, X) rows in memory
template <int GS_LOG2, int NS>
__device__ inline void build_records(const RecArgs& a, unsigned* lds, int row, int begin, int n_entries, bool valid, int lane) {
  constexpr int GS = 1 << GS_LOG2;
  const int q = lane & (GS - 1);
  int rel[NS][4];
  float xv[NS][4];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int s = q + k * GS;
    const bool live = valid && s < a.slots;
    const int64_t o = (int64_t)row * a.ldb + 4 * s;
    const int4 ar = live ? *reinterpret_cast<const int4*>(a.arg + o) : int4{-1, -1, -1, -1};
    const f4 x = live ? *reinterpret_cast<const f4*>(a.x + o) : f4{0.f, 0.f, 0.f, 0.f};
    // (arg < begin, self loop = n_edges, -1: not an entry of this range)
    rel[k][0] = ar.x - begin; rel[k][1] = ar.y - begin; rel[k][2] = ar.z - begin; rel[k][3] = ar.w - begin;
    xv[k][0] = x.x; xv[k][1] = x.y; xv[k][2] = x.z; xv[k][3] = x.w;
    if (!live) rel[k][0] = rel[k][1] = rel[k][2] = rel[k][3] = -1;
  }
  records_from_columns<GS_LOG2, NS, 4 * GS>(lds, a.ldb, a.rec + (int64_t)begin * 16, valid ? n_entries : 0, rel, xv, lane);
}

// Leading blocks: one wavefront per EGC_LONG_ROW_CHUNK-entry chunk of a long destination row (the row's columns over
// the 64 lanes); the other blocks: 16 lanes per short row, 16 rows per workgroup.  Every entry's record is written.
template <int NS>
__global__ void __launch_bounds__(256) bwd_records_kernel(RecArgs a) {
  extern __shared__ unsigned rec_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  unsigned* wl = rec_lds + wave * 4 * a.group_u32;
  if ((int)blockIdx.x < a.chunk_blocks) {
    const int c = blockIdx.x * 4 + wave;
    if (c >= a.plan[1]) return;
    const int cap_long = a.plan[2], cap_chunks = a.plan[3];
    const int* long_row = a.plan + 4;
    const int* chunk_slot = long_row + 2 * cap_long;
    const int* chunk_begin = chunk_slot + cap_chunks;
    const int row = long_row[chunk_slot[c]];
    const int begin = chunk_begin[c];
    const int n_entries = min(EGC_LONG_ROW_CHUNK, a.rowptr[row + 1] - begin);
    build_records<6, 1>(a, wl, row, begin, n_entries, true, lane);
    return;
  }
  if (!a.short_rows) return;
  const int g = lane >> 4;
  const int row = (((int)blockIdx.x - a.chunk_blocks) * 4 + wave) * 4 + g;
  bool valid = row < a.n_nodes;
  const int begin = valid ? a.rowptr[row] : 0;
  int n_entries = valid ? a.rowptr[row + 1] - begin : 0;
  if (n_entries > EGC_LONG_ROW_THRESHOLD) { valid = false; n_entries = 0; }   // the chunk blocks own this row
  build_records<4, NS>(a, wl + g * a.group_u32, valid ? row : 0, begin, n_entries, valid, lane);
}

// Register-resident form of bwd_dst_kernel for the layouts the forward's register kernels serve with shifts:
// every basis spans a power-of-two number P of 16-byte slots (L or the padded stride a multiple of 4), B a power
// of two, S = B P <= 64 slots, and P divides H.  One LANE GROUP per destination row (G = 64 / LPR rows per wavefront,
// as in the forward): lane q = b P + l4 owns columns 4 l4 .. 4 l4 + 3 of basis b.  The row's g and activated
// weights sit in per-group LDS strips; a lane forms  d agg_t = sum_h w'[h][b][t] g[h][l..l+3]  and its share of
// d w'[h][b][t] = sum_l g[h][l] agg_t[b][l]  in one pass over h, the shares meet in an xor butterfly over the P
// lanes of the basis, and lane l4 stores the heads l4 H/P .. (l4+1) H/P - 1.
constexpr int BWD_HMAX = 16;   // heads supported by the register form
// HT / AT: compile-time head and aggregator counts (the d = 128, H = 8 layers; everything else: bwd_dst_kernel)
// AGG: the aggregator codes packed 3 bits each (first in the low bits) with bit 31 set, or 0 = read them from
// the arguments.  With the list compiled in, the per-aggregator switches, the statistics layout and the weight
// nonlinearity (none) fold away -- the run-time form is mostly scalar branches.
constexpr unsigned BWD_STATIC = 0x80000000u;
constexpr unsigned bwd_agg_pack(int a0, int a1 = 0, int a2 = 0, int a3 = 0) {
  return BWD_STATIC | (unsigned)a0 | ((unsigned)a1 << 3) | ((unsigned)a2 << 6) | ((unsigned)a3 << 9);
}
template <unsigned AGG, int AT>
struct BwdAggs {
  static constexpr bool fixed = (AGG & BWD_STATIC) != 0;
  static constexpr int code(int t) { return (int)((AGG >> (3 * t)) & 7u); }
  static constexpr bool needs(int s) {  // stat_layout()'s rule for statistic s
    for (int t = 0; t < AT; ++t) {
      const int c = code(t);
      const bool sum = c == EGC_AGGR_SUM || c == EGC_AGGR_MEAN || c == EGC_AGGR_VAR || c == EGC_AGGR_STD;
      const bool sq = c == EGC_AGGR_VAR || c == EGC_AGGR_STD;
      if ((s == STAT_SUM && sum) || (s == STAT_SQ && sq) || (s == STAT_MX && c == EGC_AGGR_MAX) ||
          (s == STAT_MN && c == EGC_AGGR_MIN) || (s == STAT_WS && c == EGC_AGGR_SYMNORM))
        return true;
    }
    return false;
  }
  static constexpr int slot(int s) {
    if (!needs(s)) return -1;
    int k = 0;
    for (int u = 0; u < s; ++u) k += needs(u) ? 1 : 0;
    return k;
  }
  static __device__ inline int aggr(const BwdArgs& a, int t) { return fixed ? code(t) : a.aggr[t]; }
  static __device__ inline int stat_slot(const BwdArgs& a, int s) { return fixed ? slot(s) : a.stat_slot[s]; }
  static __device__ inline int act(const BwdArgs& a) { return fixed ? (int)EGC_ACT_NONE : a.act; }
};

#ifndef EGC_BWD_DST_MINW
#define EGC_BWD_DST_MINW 1     // wavefronts per SIMD the register allocation must leave room for (tuning constant)
#endif
template <int LPR_LOG2, int HT, int AT, unsigned AGG = 0>
__global__ void __launch_bounds__(256, EGC_BWD_DST_MINW) bwd_dst_fast_kernel(BwdArgs a) {
  using AG = BwdAggs<AGG, AT>;
  extern __shared__ float smem[];
  constexpr int LPR = 1 << LPR_LOG2, G = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> LPR_LOG2, q = lane & (LPR - 1);
  const int P = a.Ls >> 2;                      // lanes per basis: a power of two (shifts, xor butterfly) or not
  const bool p2 = (P & (P - 1)) == 0;           // (division, segmented reduction): uniform over the launch
  const int plog = 31 - __builtin_clz(P);
  const bool live = q < a.slots;
  const int bq = p2 ? q >> plog : q / P;
  const int b = min(bq, a.B - 1), l4 = q - bq * P;
  const int A = AT > 0 ? AT : a.A, H = HT > 0 ? HT : a.H;
  if (a.rec_fused && (int)blockIdx.x >= a.dst_row_blocks) {
    // ---- trailing blocks: the records of one 256-entry chunk of a hub row per wavefront (lane = slot of the row).  The
    // row's d agg_max / d agg_min are formed here again from g and w' (H fused multiply-adds per column, in the row role's
    // order: the same bits) instead of waiting for the row role's X table -- no second launch for the hub rows.
    const int c = ((int)blockIdx.x - a.dst_row_blocks) * 4 + wave;
    if (c >= a.d_plan[1]) return;
    const int cap_long = a.d_plan[2], cap_chunks = a.d_plan[3];
    const int* long_row = a.d_plan + 4;
    const int* chunk_slot = long_row + 2 * cap_long;
    const int* chunk_begin = chunk_slot + cap_chunks;
    const int crow = long_row[chunk_slot[c]];
    const int begin = chunk_begin[c];
    const int n_entries = min(EGC_LONG_ROW_CHUNK, a.rowptr[crow + 1] - begin);
    const bool clive = lane < a.slots;
    const int cbq = p2 ? lane >> plog : lane / P;
    const int cb = min(cbq, a.B - 1), cl = 4 * (lane - cbq * P);
    unsigned* wl = reinterpret_cast<unsigned*>(smem + wave * G * a.dst_group_floats);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const unsigned* rec = e == 0 ? a.rec_x : a.rec_n;
      if (rec == nullptr) continue;
      int ta = 0;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t < A && AG::aggr(a, t) == (e == 0 ? (int)EGC_AGGR_MAX : (int)EGC_AGGR_MIN)) ta = t;
      f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < (HT > 0 ? HT : BWD_HMAX); ++h) {
        if (h >= H) break;
        float w = clive ? a.weightings[(int64_t)crow * a.W + (h * a.B + cb) * A + ta] : 0.f;
        if (AG::act(a) == EGC_ACT_SIGMOID) w = 1.0f / (1.0f + expf(-w));
        else if (AG::act(a) == EGC_ACT_HARDTANH) w = fminf(fmaxf(w, -1.0f), 1.0f);
        const float* gp = a.grad_out + (int64_t)crow * a.F_out + h * a.L + cl;
        f4 gv = f4{0.f, 0.f, 0.f, 0.f};
        if (clive) {
          if (cl < a.L) gv.x = gp[0];
          if (cl + 1 < a.L) gv.y = gp[1];
          if (cl + 2 < a.L) gv.z = gp[2];
          if (cl + 3 < a.L) gv.w = gp[3];
        }
        acc = f4_fma(f4{w, w, w, w}, gv, acc);
      }
      const int* argp = e == 0 ? a.arg_max : a.arg_min;
      const int4 ar = clive ? *reinterpret_cast<const int4*>(argp + (int64_t)crow * a.ldb + 4 * lane) : int4{-1, -1, -1, -1};
      int rel[1][4] = {{ar.x - begin, ar.y - begin, ar.z - begin, ar.w - begin}};
      if (!clive) rel[0][0] = rel[0][1] = rel[0][2] = rel[0][3] = -1;
      const float xv[1][4] = {{acc.x, acc.y, acc.z, acc.w}};
      records_from_columns<6, 1, 4 * 64>(wl, a.ldb, const_cast<unsigned*>(rec) + (int64_t)begin * 16, n_entries, rel, xv, lane);
    }
    return;
  }
  const int row = (blockIdx.x * 4 + wave) * G + g;
  const bool row_ok = row < a.n_nodes;
  const int rr = row_ok ? row : 0;
  const int gpad = (a.H * a.Ls + 3) & ~3;
  float* lds_g = smem + (wave * G + g) * a.dst_group_floats;  // g in the padded head layout [h][Ls]
  float* lds_w = lds_g + gpad;                              // activated weights [h][b][a]
  constexpr int HM = HT > 0 ? HT : BWD_HMAX;

  // ---- stage the row's g and w' (LPR lanes, 16 bytes each per step)
  for (int o = 4 * q; o < H * a.Ls; o += 4 * LPR) {
    const int h = p2 ? (o >> 2) >> plog : (o >> 2) / P, l = o - h * a.Ls;   // 4 consecutive channels of one head (Ls = 4 P)
    f4 v = f4{0.f, 0.f, 0.f, 0.f};
    const float* gp = a.grad_out + (int64_t)rr * a.F_out + h * a.L + l;
    if (l + 3 < a.L) { v.x = gp[0]; v.y = gp[1]; v.z = gp[2]; v.w = gp[3]; }
    else { if (l < a.L) v.x = gp[0]; if (l + 1 < a.L) v.y = gp[1]; if (l + 2 < a.L) v.z = gp[2]; }
    *reinterpret_cast<f4*>(lds_g + o) = v;
  }
  for (int k = q; k < a.W; k += LPR) {
    float w = a.weightings[(int64_t)rr * a.W + k];
    if (AG::act(a) == EGC_ACT_SIGMOID) w = 1.0f / (1.0f + expf(-w));
    else if (AG::act(a) == EGC_ACT_HARDTANH) w = fminf(fmaxf(w, -1.0f), 1.0f);
    lds_w[k] = w;
  }
  // ---- this lane's slot of the saved aggregates
  const int cnt = row_ok ? a.cnt[rr] : 0;
  const float cntf = (float)max(cnt, 1);
  const float dis_i = a.dis != nullptr ? a.dis[rr] : 0.f;
  const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
  const float* st = a.stats + ((int64_t)rr * a.stat_k) * a.ldb + 4 * q;
  auto stat = [&](int s) -> f4 {
    return (AG::stat_slot(a, s) >= 0 && live) ? *reinterpret_cast<const f4*>(st + AG::stat_slot(a, s) * a.ldb) : zero;
  };
  const f4 sum = stat(STAT_SUM), sq = stat(STAT_SQ), mx = stat(STAT_MX), mn = stat(STAT_MN), ws = stat(STAT_WS);
  // exact divisions only where the forward needs them bit for bit (var / std); a reciprocal otherwise
  const bool has_sq = AG::stat_slot(a, STAT_SQ) >= 0;
  const float rcnt = 1.0f / cntf;
  const f4 mean = has_sq ? f4_div(sum, cntf) : sum * f4{rcnt, rcnt, rcnt, rcnt};
  const f4 var = has_sq ? sq : zero;       // the record's STAT_SQ slot IS the forward's variance
  const f4 sd = has_sq ? f4_std(var) : zero;
  f4 val[4], dagg[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    val[t] = dagg[t] = zero;
    if (t < A) {
      switch (AG::aggr(a, t)) {
        case EGC_AGGR_SUM: val[t] = sum; break;
        case EGC_AGGR_MEAN: val[t] = mean; break;
        case EGC_AGGR_MAX: val[t] = cnt > 0 ? mx : zero; break;
        case EGC_AGGR_MIN: val[t] = cnt > 0 ? mn : zero; break;
        case EGC_AGGR_VAR: val[t] = var; break;
        case EGC_AGGR_STD: val[t] = sd; break;
        default: val[t] = ws; break;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // strips of this group written (one wavefront: in order)

  // ---- one pass over the heads
  float dwp[HM][4];   // this lane's share of d w'[h][b][t]
#pragma unroll
  for (int h = 0; h < HM; ++h) {
    if (h < H) {
      const f4 gv = *reinterpret_cast<const f4*>(lds_g + h * a.Ls + 4 * l4);
      const float* wp = lds_w + (h * a.B + b) * A;
      f4 wv = zero;
      if (A == 4) wv = *reinterpret_cast<const f4*>(wp);
      else { wv.x = wp[0]; if (A > 1) wv.y = wp[1]; if (A > 2) wv.z = wp[2]; }
      dagg[0] = f4_fma(f4{wv.x, wv.x, wv.x, wv.x}, gv, dagg[0]);
      if (A > 1) dagg[1] = f4_fma(f4{wv.y, wv.y, wv.y, wv.y}, gv, dagg[1]);
      if (A > 2) dagg[2] = f4_fma(f4{wv.z, wv.z, wv.z, wv.z}, gv, dagg[2]);
      if (A > 3) dagg[3] = f4_fma(f4{wv.w, wv.w, wv.w, wv.w}, gv, dagg[3]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        dwp[h][t] = t < A ? fmaf(gv.w, val[t].w, fmaf(gv.z, val[t].z, fmaf(gv.y, val[t].y, gv.x * val[t].x))) : 0.f;
    }
  }

  // ---- tables for the source side
  f4 d_t = zero, d_s = zero, d_v = zero, d_x = zero, d_n = zero;
  const int64_t o = (int64_t)rr * a.ldb + 4 * q;
  const bool wr = row_ok && live;
  // the row's arg positions as bytes relative to its first entry (egc_aggregate_dev.h): requested here, used last
  unsigned a8x = 0xffffffffu, a8n = 0xffffffffu;
  if (a.rec_fused && wr) {
    if (a.rec_x != nullptr) a8x = reinterpret_cast<const unsigned*>(a.arg8_max)[o >> 2];
    if (a.rec_n != nullptr) a8n = reinterpret_cast<const unsigned*>(a.arg8_min)[o >> 2];
  }
  if (a.overwrite_db && wr && a.t_rowptr[rr + 1] - a.t_rowptr[rr] > EGC_LONG_ROW_THRESHOLD)   // its chunks add by atomics
    *reinterpret_cast<f4*>(a.d_bases + (int64_t)rr * a.ld_db + 4 * q) = zero;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t >= A) break;
    const f4 d = dagg[t];
    switch (AG::aggr(a, t)) {
      case EGC_AGGR_SUM: d_t += d; break;
      case EGC_AGGR_MEAN: d_t += d * f4{rcnt, rcnt, rcnt, rcnt}; break;
      case EGC_AGGR_MAX: d_x = cnt > 0 ? d : zero; if (wr) __builtin_nontemporal_store(d_x, reinterpret_cast<f4*>(a.tab_x + o)); break;
      case EGC_AGGR_MIN: d_n = cnt > 0 ? d : zero; if (wr) __builtin_nontemporal_store(d_n, reinterpret_cast<f4*>(a.tab_n + o)); break;
      case EGC_AGGR_VAR: d_v += d; break;
      case EGC_AGGR_STD:
        d_v += f4{var.x > 0.f ? d.x / (2.0f * sd.x) : 0.f, var.y > 0.f ? d.y / (2.0f * sd.y) : 0.f,
                  var.z > 0.f ? d.z / (2.0f * sd.z) : 0.f, var.w > 0.f ? d.w / (2.0f * sd.w) : 0.f};
        break;
      default: d_s += d * f4{dis_i, dis_i, dis_i, dis_i}; break;
    }
  }
  if (wr) {
    const f4 two_dv = d_v * f4{2.f * rcnt, 2.f * rcnt, 2.f * rcnt, 2.f * rcnt};
    if (a.need_t) __builtin_nontemporal_store(d_t - mean * two_dv, reinterpret_cast<f4*>(a.tab_t + o));  // else never read
    if (a.tab_s != nullptr) __builtin_nontemporal_store(d_s, reinterpret_cast<f4*>(a.tab_s + o));
    if (a.tab_v != nullptr) __builtin_nontemporal_store(two_dv, reinterpret_cast<f4*>(a.tab_v + o));
  }

  // ---- d w': sum over the P lanes of a basis.  P a power of two: xor butterfly (every lane ends with the sum), then
  // lane l4 keeps heads [l4 H/P, (l4+1) H/P).  Otherwise (the reference's 168 / 224 / 296 / 136-wide nets: P = 6 / 14 / 10 / 9)
  // the shares cross the group's LDS strip transposed: lane q writes its H A shares into rows of LPR + 1 floats, lane k of
  // the group then adds the P shares of weighting k = (h B + b) A + t in lane order and stores d w'[k] -- one coalesced store
  // per row.  (Round 6; before: a segmented shift-down reduction, 4 rounds x H A ds_bpermute + select + add, and a store per
  // (head, aggregator) from one lane per basis -- 20 of the 87 us of this kernel at 224 / H4 / B4, 16 of 66 at 296 / H8 / B4.)
  if (p2) {
    for (int off = 1; off < P; off <<= 1) {
#pragma unroll
      for (int h = 0; h < HM; ++h)
        if (h < H) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (t < A) dwp[h][t] += __shfl_xor(dwp[h][t], off);
        }
    }
    const int hpl = H >= P ? H / P : 1;   // heads per lane (host: P divides H, or H < P: lanes 0 .. H-1 keep one head each)
#pragma unroll
    for (int h = 0; h < HM; ++h) {
      if (h >= H) break;
      if (!(wr && h / hpl == l4)) continue;
      const int k0 = (h * a.B + b) * A;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t >= A) break;
        float d = dwp[h][t];
        const float w = lds_w[k0 + t];
        if (AG::act(a) == EGC_ACT_SIGMOID) d = d * w * (1.0f - w);
        else if (AG::act(a) == EGC_ACT_HARDTANH) {
          const float pre = a.weightings[(int64_t)row * a.W + k0 + t];
          d = (pre > -1.0f && pre < 1.0f) ? d : 0.f;
        }
        __builtin_nontemporal_store(d, &a.d_weightings[(int64_t)row * a.ld_dw + k0 + t]);
      }
    }
  } else {
    float* tr = lds_w + ((a.W + 3) & ~3);      // [H A][LPR + 1], behind the strips of g and w' (the host sized the group for it)
    constexpr int TP = LPR + 1;
    if (live) {
#pragma unroll
      for (int h = 0; h < HM; ++h)
        if (h < H) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (t < A) tr[(h * A + t) * TP + q] = dwp[h][t];
        }
    }
    const int blog = 31 - __builtin_clz(a.B);  // B is a power of two (host)
    for (int k = q; k < a.W; k += LPR) {
      const int hb = k / A, t = k - hb * A, hh = hb >> blog, bb = hb & (a.B - 1);
      const float* src = tr + (hh * A + t) * TP + bb * P;
      float d = 0.f;
      for (int i = 0; i < P; ++i) d += src[i];
      const float w = lds_w[k];
      if (AG::act(a) == EGC_ACT_SIGMOID) d = d * w * (1.0f - w);
      else if (AG::act(a) == EGC_ACT_HARDTANH) {
        const float pre = a.weightings[(int64_t)rr * a.W + k];
        d = (pre > -1.0f && pre < 1.0f) ? d : 0.f;
      }
      if (row_ok) __builtin_nontemporal_store(d, &a.d_weightings[(int64_t)row * a.ld_dw + k]);
    }
  }

  // ---- the records of the row's entries (short rows; the strips of g and w' are no longer needed: LDS reused)
  if (a.rec_fused) {
    const int begin = row_ok ? a.rowptr[rr] : 0;
    int n_entries = row_ok ? a.rowptr[rr + 1] - begin : 0;
    if (n_entries > EGC_LONG_ROW_THRESHOLD) n_entries = 0;   // the trailing blocks' chunk role
    unsigned* rl = reinterpret_cast<unsigned*>(lds_g);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const unsigned* rec = e == 0 ? a.rec_x : a.rec_n;
      if (rec == nullptr) continue;
      const unsigned a8 = e == 0 ? a8x : a8n;
      const f4 d = e == 0 ? d_x : d_n;
      int rel[1][4] = {{(int)(a8 & 0xffu), (int)((a8 >> 8) & 0xffu), (int)((a8 >> 16) & 0xffu), (int)(a8 >> 24)}};   // ARG8_NONE / ARG8_FAR >= 64
      const float xv[1][4] = {{d.x, d.y, d.z, d.w}};
      records_from_columns<LPR_LOG2, 1, 64>(rl, a.ldb, const_cast<unsigned*>(rec) + (int64_t)begin * 16, n_entries, rel, xv, lane);
    }
  }
}

// Sum of the destination tables over one source row's out-entries [start, end), entries t, t + step, ...:
// lane q of the group holds slot(s) q, q + LPR, ... (NS of them).
// FL: which tables exist and which edge sets are LOOPED, compiled in (bit 0 set) or read from the arguments (0).
// The run-time form makes the compiler clone the gather loop per flag combination (9,000 lines of ISA, SGPR
// spills); the shipped layer kinds get a lean kernel each.
constexpr unsigned SRC_STATIC = 1u, SRC_S = 2u, SRC_V = 4u, SRC_X = 8u, SRC_N = 16u, SRC_XL = 32u, SRC_YL = 64u, SRC_T = 128u,
                   SRC_REC = 256u;   // max / min gradients arrive as per-entry records (REC_*) instead of arg bytes + X sectors
template <unsigned FL>
struct SrcCfg {
  static constexpr bool fixed = (FL & SRC_STATIC) != 0;
  static __device__ inline bool has_t(const BwdArgs& a) { return fixed ? (FL & SRC_T) != 0 : a.need_t != 0; }
  static __device__ inline bool has_s(const BwdArgs& a) { return fixed ? (FL & SRC_S) != 0 : a.tab_s != nullptr; }
  static __device__ inline bool has_v(const BwdArgs& a) { return fixed ? (FL & SRC_V) != 0 : a.tab_v != nullptr; }
  static __device__ inline bool has_x(const BwdArgs& a) { return fixed ? (FL & SRC_X) != 0 : a.tab_x != nullptr; }
  static __device__ inline bool has_n(const BwdArgs& a) { return fixed ? (FL & SRC_N) != 0 : a.tab_n != nullptr; }
  static __device__ inline bool xl(const BwdArgs& a) { return fixed ? (FL & SRC_XL) != 0 : a.x_looped != 0; }
  static __device__ inline bool yl(const BwdArgs& a) { return fixed ? (FL & SRC_YL) != 0 : a.y_looped != 0; }
  static __device__ inline bool rec(const BwdArgs& a) { return fixed ? (FL & SRC_REC) != 0 : (a.rec_x != nullptr || a.rec_n != nullptr); }
};

#ifndef EGC_BWD_FU
#define EGC_BWD_FU 4
#endif
constexpr int BWD_FU = EGC_BWD_FU;
template <int NS, class SC>
__device__ inline void sum_tables(const BwdArgs& a, const __amdgpu_buffer_rsrc_t (&rt)[3],
                                  const __amdgpu_buffer_rsrc_t (&rx)[4], const __amdgpu_buffer_rsrc_t (&r8)[2],
                                  const __amdgpu_buffer_rsrc_t (&rr)[2], float* acc, int row,
                                  int start, int end, int first, int step,
                                  int q, int LPR, bool xl, bool yl, f4 (&at)[NS], f4 (&as)[NS], f4 (&av)[NS]) {
  // Every memory round trip on the critical path of a batch costs as much as the gathers themselves (the kernel is
  // bound by the batches in flight, and loads return in order): the indices of batch n + 1 are loaded beside the
  // gathers of batch n, the arg positions are requested BEFORE the table slots so that the dependent gather of the
  // extremum gradients leaves while the table slots are still in flight.
  const bool ext_any = SC::has_x(a) || SC::has_n(a);
  const bool rec = ext_any && SC::rec(a);   // records: one 64-byte line per entry, no dependent loads
  const bool ext = ext_any && !rec;         // arg bytes, then pieces of the X rows
  const int lane = threadIdx.x & 63;
  int dst_n[BWD_FU], pos_n[BWD_FU];
  auto load_indices = [&](int p0) {
#pragma unroll
    for (int u = 0; u < BWD_FU; ++u) {
      const int p = p0 + u * step;
      dst_n[u] = p < end ? a.t_col[p] : -1;
      pos_n[u] = (ext_any && p < end) ? a.t_edge_id[p] : -2;
    }
  };
  load_indices(start + first);
  for (int p0 = start + first; p0 < end; p0 += step * BWD_FU) {
    int dst[BWD_FU], pos[BWD_FU], rp[BWD_FU];
#pragma unroll
    for (int u = 0; u < BWD_FU; ++u) { dst[u] = dst_n[u]; pos[u] = pos_n[u]; }
    load_indices(p0 + step * BWD_FU);
#pragma unroll
    for (int u = 0; u < BWD_FU; ++u) rp[u] = (ext && dst[u] >= 0) ? a.rowptr[dst[u]] : 0;   // -> position inside the destination row
    unsigned rw[2][BWD_FU];
    if (rec) {   // dword q of the entry's record: the group's first 16 lanes fetch the 64-byte line
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (!(e == 0 ? SC::has_x(a) : SC::has_n(a))) continue;
#pragma unroll
        for (int u = 0; u < BWD_FU; ++u)
          rw[e][u] = __builtin_amdgcn_raw_buffer_load_b32(rr[e], (dst[u] >= 0 && q < 16) ? (unsigned)pos[u] * 64u + (unsigned)q * 4u : OOB, 0, EGC_REC_LOAD_AUX);
      }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int s = q + k * LPR;
      // max / min: the whole gradient of (destination, column) goes to the entry its arg position names
      unsigned a8[2][BWD_FU];
      if (ext) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (!(e == 0 ? SC::has_x(a) : SC::has_n(a))) continue;
#pragma unroll
          for (int u = 0; u < BWD_FU; ++u) {
            const bool lv = dst[u] >= 0 && s < a.slots;
            const unsigned off = (unsigned)dst[u] * (unsigned)a.ldb + (unsigned)s * 4u;
            a8[e][u] = __builtin_amdgcn_raw_buffer_load_b32(r8[e], lv ? off : OOB, 0, 0);
          }
        }
      }
      f4 vt[BWD_FU], vs[BWD_FU], vv[BWD_FU];
#pragma unroll
      for (int u = 0; u < BWD_FU; ++u) {
        const bool live = dst[u] >= 0 && s < a.slots;
        const bool is_self = dst[u] == row;
        const unsigned off = (unsigned)dst[u] * (unsigned)a.ldb * 4u + (unsigned)s * 16u;
        vt[u] = load_slot(rt[0], (SC::has_t(a) && live && !(xl && is_self)) ? off : OOB);
        if (SC::has_v(a)) vv[u] = load_slot(rt[2], (live && !(xl && is_self)) ? off : OOB);
        if (SC::has_s(a)) vs[u] = load_slot(rt[1], (live && !(yl && is_self)) ? off : OOB);
      }
      if (ext) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (!(e == 0 ? SC::has_x(a) : SC::has_n(a))) continue;
          f4 gv[BWD_FU];
          int4 ar[BWD_FU];
          bool mx[BWD_FU], my[BWD_FU], mz[BWD_FU], mw[BWD_FU], far[BWD_FU];
#pragma unroll
          for (int u = 0; u < BWD_FU; ++u) {
            // positions up to ARG8_NONE - 1 entries into the destination row compare bytes (an out-of-range load
            // returns 0 and a byte 0 would match position 0, hence the lv term); further in (hub rows) a byte
            // ARG8_FAR marks a candidate that the int32 row decides -- requested together with the gradient slot
            const bool lv = dst[u] >= 0 && s < a.slots;
            const unsigned rel = min((unsigned)(pos[u] - rp[u]), ARG8_FAR);
            const unsigned want = rel < ARG8_NONE ? rel : ARG8_FAR;
            far[u] = rel >= ARG8_NONE;
            mx[u] = lv && (a8[e][u] & 0xffu) == want;
            my[u] = lv && ((a8[e][u] >> 8) & 0xffu) == want;
            mz[u] = lv && ((a8[e][u] >> 16) & 0xffu) == want;
            mw[u] = lv && (a8[e][u] >> 24) == want;
            const bool any = mx[u] || my[u] || mz[u] || mw[u];
            const unsigned off = (unsigned)dst[u] * (unsigned)a.ldb * 4u + (unsigned)s * 16u;
            gv[u] = load_slot(rx[2 * e + 1], any ? off : OOB);
            ar[u] = __builtin_bit_cast(int4, load_slot(rx[2 * e], (any && far[u]) ? off : OOB));
          }
#pragma unroll
          for (int u = 0; u < BWD_FU; ++u) {
            const bool nx = !far[u] || ar[u].x == pos[u], ny = !far[u] || ar[u].y == pos[u];
            const bool nz = !far[u] || ar[u].z == pos[u], nw = !far[u] || ar[u].w == pos[u];
            at[k].x += (mx[u] && nx) ? gv[u].x : 0.f; at[k].y += (my[u] && ny) ? gv[u].y : 0.f;
            at[k].z += (mz[u] && nz) ? gv[u].z : 0.f; at[k].w += (mw[u] && nw) ? gv[u].w : 0.f;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < BWD_FU; ++u) {
        at[k] += vt[u];
        if (SC::has_v(a)) av[k] += vv[u];
        if (SC::has_s(a)) as[k] += vs[u];
      }
    }
    if (rec) {
      const int gl = lane & ~(LPR - 1);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (!(e == 0 ? SC::has_x(a) : SC::has_n(a))) continue;
        bool ovf[BWD_FU], any_ovf = false;
#pragma unroll
        for (int u = 0; u < BWD_FU; ++u) {
          const unsigned cn = __shfl(rw[e][u], gl + 15) & 0xffu;              // (a dead entry loaded zeros: count 0)
          const unsigned colw = __shfl(rw[e][u], gl + 12 + ((q & 15) >> 2));
          const unsigned col = (colw >> ((q & 3) * 8)) & 0xffu;
          ovf[u] = cn == REC_OVERFLOW;
          any_ovf = any_ovf || ovf[u];
          if (q < REC_ITEMS && (unsigned)q < cn && !ovf[u])
            __hip_atomic_fetch_add(&acc[col], __uint_as_float(rw[e][u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (__ballot(any_ovf) != 0ull) {   // entries that receive more than REC_ITEMS columns: the int32 arg row decides
#pragma unroll
          for (int k = 0; k < NS; ++k) {
            const int s = q + k * LPR;
            f4 gv[BWD_FU];
            int4 ar[BWD_FU];
#pragma unroll
            for (int u = 0; u < BWD_FU; ++u) {
              const bool lv = ovf[u] && dst[u] >= 0 && s < a.slots;
              const unsigned off = (unsigned)dst[u] * (unsigned)a.ldb * 4u + (unsigned)s * 16u;
              gv[u] = load_slot(rx[2 * e + 1], lv ? off : OOB);
              ar[u] = __builtin_bit_cast(int4, load_slot(rx[2 * e], lv ? off : OOB));
            }
#pragma unroll
            for (int u = 0; u < BWD_FU; ++u) {
              const bool lv = ovf[u] && dst[u] >= 0 && s < a.slots;
              at[k].x += (lv && ar[u].x == pos[u]) ? gv[u].x : 0.f; at[k].y += (lv && ar[u].y == pos[u]) ? gv[u].y : 0.f;
              at[k].z += (lv && ar[u].z == pos[u]) ? gv[u].z : 0.f; at[k].w += (lv && ar[u].w == pos[u]) ? gv[u].w : 0.f;
            }
          }
        }
      }
    }
  }
}

// d bases[j] += sum over out-neighbours of the tables (+ self-loop terms).  Leading blocks: one wavefront per
// EGC_LONG_ROW_CHUNK-entry chunk of a long row, partial sums by float atomics; the other blocks: one lane group per short row.
#ifndef EGC_SRC_WAVES
#define EGC_SRC_WAVES 1
#endif
template <int NS, unsigned FL = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NS == 1 ? EGC_SRC_WAVES : 1, 8))) bwd_src_kernel(BwdArgs a) {
  using SC = SrcCfg<FL>;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int LPR = 1 << a.lpr_log2, G = 64 >> a.lpr_log2;
  const int g = lane >> a.lpr_log2, q = lane & (LPR - 1);
  const bool xl = SC::xl(a), yl = SC::yl(a);
  const int nloop = a.loops_all ? a.n_nodes : (*a.max_index + 1);
  __amdgpu_buffer_rsrc_t rt[3];
  rt[0] = __builtin_amdgcn_make_buffer_rsrc((void*)a.tab_t, 0, a.tab_bytes, 0x00020000);
  rt[1] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.tab_s != nullptr ? a.tab_s : a.tab_t), 0, a.tab_bytes, 0x00020000);
  rt[2] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.tab_v != nullptr ? a.tab_v : a.tab_t), 0, a.tab_bytes, 0x00020000);
  // arg positions and extremum gradients: {arg_max, tab_x, arg_min, tab_n}
  __amdgpu_buffer_rsrc_t rx[4];
  rx[0] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.arg_max != nullptr ? (const void*)a.arg_max : (const void*)a.tab_t), 0, a.tab_bytes, 0x00020000);
  rx[1] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.tab_x != nullptr ? a.tab_x : a.tab_t), 0, a.tab_bytes, 0x00020000);
  rx[2] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.arg_min != nullptr ? (const void*)a.arg_min : (const void*)a.tab_t), 0, a.tab_bytes, 0x00020000);
  rx[3] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.tab_n != nullptr ? a.tab_n : a.tab_t), 0, a.tab_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t r8[2];   // the arg positions in 8 bits: a quarter of the bytes of rx[0] / rx[2]
  r8[0] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.arg8_max != nullptr ? (const void*)a.arg8_max : (const void*)a.tab_t), 0, a.tab_bytes / 4, 0x00020000);
  r8[1] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.arg8_min != nullptr ? (const void*)a.arg8_min : (const void*)a.tab_t), 0, a.tab_bytes / 4, 0x00020000);
  __amdgpu_buffer_rsrc_t rr[2];   // the per-entry records of max / min
  rr[0] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.rec_x != nullptr ? (const void*)a.rec_x : (const void*)a.tab_t), 0, a.rec_x != nullptr ? a.rec_bytes : 0u, 0x00020000);
  rr[1] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.rec_n != nullptr ? (const void*)a.rec_n : (const void*)a.tab_t), 0, a.rec_n != nullptr ? a.rec_bytes : 0u, 0x00020000);
  // records path: a row of F_g sums per lane group, filled by LDS float adds (G * ldb <= 256 floats per wavefront)
  __shared__ float src_acc[4 * 256];
  float* acc = src_acc + wave * 256 + g * a.ldb;
  const bool use_rec = (SC::has_x(a) || SC::has_n(a)) && SC::rec(a);
  if (use_rec) *reinterpret_cast<f4*>(src_acc + wave * 256 + lane * 4) = f4{0.f, 0.f, 0.f, 0.f};
  const f4 zero = f4{0.f, 0.f, 0.f, 0.f};

  int row, start, end, first, step;
  bool add_self, atomic;
  if ((int)blockIdx.x < a.chunk_blocks) {
    const int c = blockIdx.x * 4 + wave;
    if (c >= a.t_plan[1]) return;
    const int cap_long = a.t_plan[2], cap_chunks = a.t_plan[3];
    const int* long_row = a.t_plan + 4;
    const int* chunk_slot = long_row + 2 * cap_long;
    const int* chunk_begin = chunk_slot + cap_chunks;
    row = long_row[chunk_slot[c]];
    start = chunk_begin[c];
    end = min(start + EGC_LONG_ROW_CHUNK, a.t_rowptr[row + 1]);
    first = g; step = G;
    add_self = start == a.t_rowptr[row];  // the row's first chunk also carries the self-loop terms
    atomic = true;
  } else {
    row = (((int)blockIdx.x - a.chunk_blocks) * 4 + wave) * G + g;
    if (row >= a.n_src_rows) { row = -1; start = end = 0; } else { start = a.t_rowptr[row]; end = a.t_rowptr[row + 1]; }
    // long rows belong to the chunk blocks: this group must not even read-modify-write their d_bases row
    if (end - start > EGC_LONG_ROW_THRESHOLD) { row = -1; start = end = 0; }
    first = 0; step = 1;
    add_self = true;
    atomic = false;
  }
  f4 at[NS], as[NS], av[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) at[k] = as[k] = av[k] = zero;
  sum_tables<NS, SC>(a, rt, rx, r8, rr, acc, row, start, end, first, step, q, LPR, xl, yl, at, as, av);
  if (use_rec) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int s = q + k * LPR;
      if (s < a.slots) at[k] += *reinterpret_cast<const f4*>(acc + 4 * s);   // (LDS operations of a wavefront are in order)
    }
  }
  if (atomic) {  // merge the G groups of the chunk
    for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        at[k] += f4_shfl_xor(at[k], off);
        as[k] += f4_shfl_xor(as[k], off);
        av[k] += f4_shfl_xor(av[k], off);
      }
    }
    if (g != 0) return;
  }
  if (row < 0) return;
  const bool has_self = add_self && row < a.n_nodes && row < nloop;
  const float dis_j = a.dis != nullptr ? a.dis[row] : 0.f;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int s = q + k * LPR;
    if (s >= a.slots) continue;
    const int64_t o = (int64_t)row * a.ldb + 4 * s;
    f4 t = at[k], sv = as[k], vv = av[k];
    if (xl && has_self) {
      if (SC::has_t(a)) t += *reinterpret_cast<const f4*>(a.tab_t + o);
      if (SC::has_v(a)) vv += *reinterpret_cast<const f4*>(a.tab_v + o);
    }
    if (yl && has_self && SC::has_s(a)) sv += *reinterpret_cast<const f4*>(a.tab_s + o);
    if (xl && has_self) {
      // the appended self-loop attained the extremum: arg == n_edges, i.e. byte ARG8_NONE in the 8-bit table (an empty
      // row has it too, with a zero gradient) -- 4 bytes instead of 16 per slot, and the gradient slot only where some
      // column of it goes to the loop (about one slot in four at config 2)
      const unsigned off = (unsigned)row * (unsigned)a.ldb * 4u + (unsigned)s * 16u;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (!(e == 0 ? SC::has_x(a) : SC::has_n(a))) continue;
        const unsigned b8 = __builtin_amdgcn_raw_buffer_load_b32(r8[e], off >> 2, 0, 0);
        const bool sx = (b8 & 0xffu) == ARG8_NONE, sy = ((b8 >> 8) & 0xffu) == ARG8_NONE;
        const bool sz = ((b8 >> 16) & 0xffu) == ARG8_NONE, sw = (b8 >> 24) == ARG8_NONE;
        const f4 gx = load_slot(rx[2 * e + 1], (sx || sy || sz || sw) ? off : OOB);
        t.x += sx ? gx.x : 0.f; t.y += sy ? gx.y : 0.f; t.z += sz ? gx.z : 0.f; t.w += sw ? gx.w : 0.f;
      }
    }
    f4 d = t;
    if (SC::has_s(a)) d = f4_fma(f4{dis_j, dis_j, dis_j, dis_j}, sv, d);
    if (SC::has_v(a)) d = f4_fma(*reinterpret_cast<const f4*>(a.bases + o), vv, d);
    float* dst = a.d_bases + (int64_t)row * a.ld_db + 4 * s;
    if (atomic) {
      atomicAdd(dst, d.x); atomicAdd(dst + 1, d.y); atomicAdd(dst + 2, d.z); atomicAdd(dst + 3, d.w);
    } else if (a.overwrite_db) {
      *reinterpret_cast<f4*>(dst) = d;   // short rows are owned by this lane group: no fill, no read-modify-write
    } else {
      *reinterpret_cast<f4*>(dst) += d;  // host-zeroed array (rectangular graphs)
    }
  }
}

}  // namespace egc

using namespace egc;

extern "C" {

size_t egc_backward_workspace_bytes(const egc_layer* layer, int64_t n_nodes) {
  if (layer == nullptr || n_nodes < 0 || layer->num_heads <= 0) return 0;
  const int ldb = egc_bases_ld(layer);
  return (((size_t)5 * (size_t)n_nodes * ldb * sizeof(float) + 255) & ~(size_t)255) + 256;  // tables T, S, V, X, N
}

static int layer_extrema(const egc_layer* layer) {
  int mx = 0, mn = 0;
  for (int t = 0; t < layer->num_aggrs && t < EGC_MAX_AGGRS; ++t) {
    if (layer->aggrs[t] == EGC_AGGR_MAX) mx = 1;
    if (layer->aggrs[t] == EGC_AGGR_MIN) mn = 1;
  }
  return mx + mn;
}

// records of the extremum gradients (bwd_records_kernel): one 64-byte line per entry and extremum, behind the tables
// A record holds REC_ITEMS (value, column) pairs; an entry that receives more is marked REC_OVERFLOW and the source side reads
// the destination's whole X row and int32 arg row for it on top of the record.  An entry receives ldb / degree columns on
// average: 4.6 on the ogbn-arxiv graph at 64 basis columns, but 108 on a molhiv batch at 224 (two entries per row) -- there
// EVERY record overflowed, the source kernel fetched 282 MB per launch for 110 MB of payload (FETCH_SIZE, round 6) and the
// destination kernel spent a fifth of its time building records nobody could use.  Records only where they mostly fit.
static bool records_apply(const egc_layer* layer, int64_t n_nodes, int64_t n_edges) {
  const int ldb = egc_bases_ld(layer);
  if (!(layer_extrema(layer) > 0 && n_edges > 0 && ldb <= 256 && (uint64_t)n_edges * 64ull < (uint64_t)OOB &&
        getenv("EGC_BWD_NO_REC") == nullptr))
    return false;
  return (double)ldb * (double)std::max<int64_t>(n_nodes, 1) <= (double)REC_FIT_COLUMNS * (double)n_edges;
}

size_t egc_backward_workspace_bytes_for(const egc_layer* layer, const egc_graph* graph) {
  if (layer == nullptr || graph == nullptr || graph->n_nodes < 0 || graph->n_edges < 0) return 0;
  const size_t base = egc_backward_workspace_bytes(layer, graph->n_nodes);
  if (base == 0 || !records_apply(layer, graph->n_nodes, graph->n_edges)) return base;
  return base + (size_t)layer_extrema(layer) * (size_t)graph->n_edges * 64;
}

int egc_aggregate_combine_backward_f32(const egc_graph* graph, const egc_graph* t_graph, const egc_layer* layer,
                                       const float* bases, int32_t ldb, const float* weightings, const float* grad_out,
                                       const float* stats, const int32_t* cnt, const int32_t* arg_max,
                                       const int32_t* arg_min, float* d_bases, int32_t ld_d_bases, float* d_weightings,
                                       int32_t ld_d_weightings, void* workspace, size_t workspace_bytes,
                                       egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (graph == nullptr || t_graph == nullptr || layer == nullptr) return EGC_ERR_INVALID;
  if (layer->num_aggrs <= 0 || layer->num_aggrs > EGC_MAX_AGGRS || layer->out_channels % layer->num_heads != 0)
    return EGC_ERR_INVALID;
  const int64_t n = graph->n_nodes;
  const int64_t n_src = graph->n_src_rows > 0 ? graph->n_src_rows : n;
  if (n == 0) return EGC_OK;
  // Rectangular adjacencies (n_src != n) carry no self loops / symnorm; on a vertex partition the caller owns
  // the reverse exchange of the halo rows of d_bases.
  if (n_src < n && (layer->agg_set == EGC_SET_LOOPED || layer_uses_symnorm(layer))) return EGC_ERR_INVALID;
  if (t_graph->n_nodes != n_src || t_graph->rowptr == nullptr || t_graph->plan == nullptr) return EGC_ERR_INVALID;
  if (bases == nullptr || weightings == nullptr || grad_out == nullptr || d_bases == nullptr || d_weightings == nullptr ||
      stats == nullptr || cnt == nullptr)
    return EGC_ERR_INVALID;
  if (ldb != egc_bases_ld(layer)) return EGC_ERR_INVALID;
  if (layer->weight_layout != EGC_LAYOUT_HBA) return EGC_ERR_UNSUPPORTED;  // the host packs [h][b][a]
  if (workspace == nullptr || workspace_bytes < egc_backward_workspace_bytes(layer, n)) return EGC_ERR_WORKSPACE;
  if ((uint64_t)n * (uint64_t)ldb * 4ull > (uint64_t)OOB) return EGC_ERR_UNSUPPORTED;  // 32-bit buffer offsets

  BwdArgs a;
  a.col = graph->col;
  a.max_index = graph->max_index;
  a.t_rowptr = t_graph->rowptr;
  a.t_col = t_graph->col;
  a.t_plan = t_graph->plan;
  a.bases = bases;
  a.weightings = weightings;
  a.grad_out = grad_out;
  a.stats = stats;
  a.cnt = cnt;
  a.arg_max = arg_max;
  a.arg_min = arg_min;
  a.d_bases = d_bases;
  a.d_weightings = d_weightings;
  a.ld_db = ld_d_bases > 0 ? ld_d_bases : ldb;
  a.ld_dw = ld_d_weightings > 0 ? ld_d_weightings : layer->num_heads * layer->num_bases * layer->num_aggrs;
  // 16-byte row accesses of d_bases; both arrays at least as wide as what is written
  if (a.ld_db < ldb || (a.ld_db & 3) != 0 || (reinterpret_cast<uintptr_t>(d_bases) & 15) != 0 ||
      a.ld_dw < layer->num_heads * layer->num_bases * layer->num_aggrs)
    return EGC_ERR_INVALID;
  a.n_nodes = (int)n;
  a.n_src_rows = (int)n_src;
  a.overwrite_db = n_src == n;
  a.n_edges = (int)graph->n_edges;
  a.ldb = ldb;
  a.slots = ldb / 4;
  a.H = layer->num_heads;
  a.B = layer->num_bases;
  a.A = layer->num_aggrs;
  a.L = layer->out_channels / layer->num_heads;
  a.Ls = layer_basis_stride(layer);
  a.F_g = a.B * a.Ls;
  a.F_out = layer->out_channels;
  a.W = a.H * a.B * a.A;
  bool sym = false, var = false;
  for (int t = 0; t < EGC_MAX_AGGRS; ++t) {
    a.aggr[t] = t < a.A ? layer->aggrs[t] : 0;
    if (t < a.A && layer->aggrs[t] == EGC_AGGR_SYMNORM) sym = true;
    if (t < a.A && (layer->aggrs[t] == EGC_AGGR_VAR || layer->aggrs[t] == EGC_AGGR_STD)) var = true;
    if (t < a.A && layer->aggrs[t] == EGC_AGGR_MAX && arg_max == nullptr) return EGC_ERR_INVALID;
    if (t < a.A && layer->aggrs[t] == EGC_AGGR_MIN && arg_min == nullptr) return EGC_ERR_INVALID;
  }
  a.stat_k = stat_layout(a.aggr, a.A, a.stat_slot);
  a.x_looped = layer->agg_set == EGC_SET_LOOPED;
  a.y_looped = layer->sym_set == EGC_SET_LOOPED;
  a.loops_all = layer->loops_all_nodes != 0;
  if (!a.loops_all && graph->max_index == nullptr) return EGC_ERR_INVALID;
  a.act = layer->weight_act;
  a.dis = nullptr;
  if (sym) {
    a.dis = a.y_looped ? graph->dis_looped : graph->dis_raw;
    if (a.dis == nullptr) return EGC_ERR_INVALID;
  }
  float* ws = (float*)workspace;
  a.tab_t = ws;
  a.need_t = 0;
  for (int t = 0; t < a.A; ++t)
    if (a.aggr[t] == EGC_AGGR_SUM || a.aggr[t] == EGC_AGGR_MEAN || a.aggr[t] == EGC_AGGR_VAR || a.aggr[t] == EGC_AGGR_STD)
      a.need_t = 1;
  a.tab_s = sym ? ws + (size_t)n * ldb : nullptr;
  a.tab_v = var ? ws + (size_t)2 * n * ldb : nullptr;
  a.tab_x = a.stat_slot[STAT_MX] >= 0 ? ws + (size_t)3 * n * ldb : nullptr;
  a.tab_n = a.stat_slot[STAT_MN] >= 0 ? ws + (size_t)4 * n * ldb : nullptr;
  a.t_edge_id = t_graph->edge_id;
  if ((a.tab_x != nullptr || a.tab_n != nullptr) && a.t_edge_id == nullptr) return EGC_ERR_INVALID;
  a.rowptr = graph->rowptr;
  {  // the 8-bit in-row arg positions the training forward left behind the raw aggregates (egc_aggregate_dev.h)
    unsigned char* bytes = reinterpret_cast<unsigned char*>(const_cast<float*>(stats) + (size_t)n * a.stat_k * ldb);
    a.arg8_max = a.tab_x != nullptr ? bytes : nullptr;
    a.arg8_min = a.tab_n != nullptr ? bytes + (a.tab_x != nullptr ? (size_t)n * ldb : 0) : nullptr;
    if ((a.arg8_max != nullptr || a.arg8_min != nullptr) && a.rowptr == nullptr) return EGC_ERR_INVALID;
  }
  a.tab_bytes = (unsigned)((uint64_t)n * ldb * 4ull);
  // per-entry records when the caller's workspace holds them (egc_backward_workspace_bytes_for)
  a.rec_x = a.rec_n = nullptr;
  a.rec_bytes = 0;
  const size_t tables_bytes = egc_backward_workspace_bytes(layer, n);   // (a multiple of 256: 64-byte aligned records)
  if ((a.tab_x != nullptr || a.tab_n != nullptr) && records_apply(layer, graph->n_nodes, graph->n_edges) && graph->plan != nullptr &&
      workspace_bytes >= tables_bytes + (size_t)layer_extrema(layer) * (size_t)graph->n_edges * 64) {
    unsigned char* r = reinterpret_cast<unsigned char*>(workspace) + tables_bytes;
    if (a.tab_x != nullptr) { a.rec_x = reinterpret_cast<const unsigned*>(r); r += (size_t)graph->n_edges * 64; }
    if (a.tab_n != nullptr) a.rec_n = reinterpret_cast<const unsigned*>(r);
    a.rec_bytes = (unsigned)((uint64_t)graph->n_edges * 64ull);
  }
  a.rec_fused = 0;
  a.d_plan = graph->plan;
  a.dst_row_blocks = 0;
  a.lds_floats_per_wave = a.A * ldb + ((a.F_out + 3) & ~3) + 2 * ((a.W + 3) & ~3);
  int wpb = 4;
  if ((size_t)wpb * a.lds_floats_per_wave * sizeof(float) > 48 * 1024) wpb = 1;
  const size_t lds = (size_t)wpb * a.lds_floats_per_wave * sizeof(float);
  if (lds > 64 * 1024) return EGC_ERR_UNSUPPORTED;
  {
    // register-resident form when the layout allows (see bwd_dst_fast_kernel), else the LDS-based kernel
    const int P = a.Ls / 4;
    const bool p2 = (P & (P - 1)) == 0;
    const bool fast = getenv("EGC_BWD_GENERIC") == nullptr && (a.Ls & 3) == 0 && a.ldb == a.B * a.Ls && P >= 1 && P <= 16 &&
                      (a.B & (a.B - 1)) == 0 && a.slots <= 64 && a.A <= 4 && a.H <= BWD_HMAX &&
                      (!p2 || a.H % P == 0 || a.H < P) && a.act != EGC_ACT_SOFTMAX &&
                      // only the compiled head / aggregator counts: with run-time counts the LDS kernel is faster
                      ((a.slots <= 16 && a.H == 8 && (a.A == 1 || a.A == 3 || a.A == 4)) ||
                       // the other trained nets of the reference (hyperparameters.md): zinc EGC-M 124/H4/B4, CIFAR EGC-M
                       // 128/H4/B4 (32 slots); zinc / CIFAR EGC-S 168/H8/B4, arxiv EGC-S 184/H8/B4 (24 slots); arxiv EGC-M
                       // 136/H4/B4 (36), molhiv EGC-M 224/H4/B4 (56); molhiv EGC-S 296/H8/B4 (40)
                       (a.slots > 16 && a.slots <= 32 && ((a.H == 4 && a.A == 3) || (a.H == 8 && a.A == 1))) ||
                       (a.slots > 32 && ((a.H == 4 && a.A == 3) || (a.H == 8 && a.A == 1))));
    if (fast) {
      const int lpr = a.slots <= 16 ? 16 : a.slots <= 32 ? 32 : 64;
      const int G = 64 / lpr;
      a.rec_fused = a.rec_bytes != 0 && getenv("EGC_BWD_REC_SEPARATE") == nullptr;
      // per lane group: the strips of g and w'; afterwards the record builder's count | offset | values | column bytes
      // (64 entries per group in the row role; 256 per WAVEFRONT in the hub-chunk role of the trailing blocks)
      // (+ the transposed d w' shares of a basis that is not a power-of-two number of lanes wide: [H A][lpr + 1])
      a.dst_group_floats = std::max(((a.H * a.Ls + 3) & ~3) + ((a.W + 3) & ~3) + (p2 ? 0 : a.H * a.A * (lpr + 1)),
                                    a.rec_fused ? std::max(128 + a.ldb + a.ldb / 4 + 4, (512 + a.ldb + a.ldb / 4 + 4 + G - 1) / G) : 0);
      const size_t flds = (size_t)4 * G * a.dst_group_floats * sizeof(float);
      a.dst_row_blocks = (int)ceil_div(n, (int64_t)4 * G);
      a.d_plan = graph->plan;
      int rec_chunk_blocks = 0;
      if (a.rec_fused) {
        const PlanCaps dc = plan_caps(n, graph->n_edges);
        rec_chunk_blocks = (int)ceil_div((graph->n_chunks >= 0 && graph->n_chunks <= dc.cap_chunks) ? graph->n_chunks : dc.cap_chunks, 4);
      }
      const unsigned fgrid = (unsigned)(a.dst_row_blocks + rec_chunk_blocks);
      if (flds > 64 * 1024) return EGC_ERR_UNSUPPORTED;
      unsigned packed = BWD_STATIC;
      for (int t = 0; t < a.A; ++t) packed |= (unsigned)a.aggr[t] << (3 * t);
      constexpr int S = EGC_AGGR_SUM, M = EGC_AGGR_MEAN, X = EGC_AGGR_MAX, Y = EGC_AGGR_SYMNORM;
      // the reference's own batched nets with their aggregator lists compiled in (round 6: these layers train on this path --
      // DESIGN.md section 3.7 -- and the run-time form's per-aggregator switches are most of its instructions): molhiv EGC-M
      // 224 / H4 / B4 add, mean, max; molhiv EGC-S 296 / H8 / B4 and zinc / cifar EGC-S 168 / H8 / B4 symadd
      const bool plain = a.act == EGC_ACT_NONE;
      if (a.slots > 32 && a.H == 4 && plain && packed == bwd_agg_pack(S, M, X)) bwd_dst_fast_kernel<6, 4, 3, bwd_agg_pack(S, M, X)><<<fgrid, 256, flds, stream>>>(a);
      else if (a.slots > 32 && a.H == 8 && plain && packed == bwd_agg_pack(Y)) bwd_dst_fast_kernel<6, 8, 1, bwd_agg_pack(Y)><<<fgrid, 256, flds, stream>>>(a);
      else if (a.slots > 16 && a.slots <= 32 && a.H == 8 && plain && packed == bwd_agg_pack(Y)) bwd_dst_fast_kernel<5, 8, 1, bwd_agg_pack(Y)><<<fgrid, 256, flds, stream>>>(a);
      // arxiv EGC-M 136 / H4 / B4 symadd, max, mean (36 slots); zinc EGC-M 124 / H4 / B4 add, std, max and cifar EGC-M 128 / H4 / B4 symadd, std, max (32)
      else if (a.slots > 32 && a.H == 4 && plain && packed == bwd_agg_pack(Y, X, M)) bwd_dst_fast_kernel<6, 4, 3, bwd_agg_pack(Y, X, M)><<<fgrid, 256, flds, stream>>>(a);
      else if (a.slots > 16 && a.slots <= 32 && a.H == 4 && plain && packed == bwd_agg_pack(S, EGC_AGGR_STD, X)) bwd_dst_fast_kernel<5, 4, 3, bwd_agg_pack(S, EGC_AGGR_STD, X)><<<fgrid, 256, flds, stream>>>(a);
      else if (a.slots > 16 && a.slots <= 32 && a.H == 4 && plain && packed == bwd_agg_pack(Y, EGC_AGGR_STD, X)) bwd_dst_fast_kernel<5, 4, 3, bwd_agg_pack(Y, EGC_AGGR_STD, X)><<<fgrid, 256, flds, stream>>>(a);
      else if (a.slots > 32 && a.H == 4) bwd_dst_fast_kernel<6, 4, 3><<<fgrid, 256, flds, stream>>>(a);
      else if (a.slots > 32) bwd_dst_fast_kernel<6, 8, 1><<<fgrid, 256, flds, stream>>>(a);
      else if (a.slots > 16 && a.H == 4) bwd_dst_fast_kernel<5, 4, 3><<<fgrid, 256, flds, stream>>>(a);
      else if (a.slots > 16) bwd_dst_fast_kernel<5, 8, 1><<<fgrid, 256, flds, stream>>>(a);
      else if (a.A == 4 && a.act == EGC_ACT_NONE && packed == bwd_agg_pack(S, M, X, Y))      // EGConv north star, compiled in
        bwd_dst_fast_kernel<4, 8, 4, bwd_agg_pack(S, M, X, Y)><<<fgrid, 256, flds, stream>>>(a);
      else if (a.A == 3 && a.act == EGC_ACT_NONE && packed == bwd_agg_pack(Y, X, M))    // EfficientGraphConv EGC-M
        bwd_dst_fast_kernel<4, 8, 3, bwd_agg_pack(Y, X, M)><<<fgrid, 256, flds, stream>>>(a);
      else if (a.A == 1 && a.act == EGC_ACT_NONE && packed == bwd_agg_pack(Y))          // EGC-S
        bwd_dst_fast_kernel<4, 8, 1, bwd_agg_pack(Y)><<<fgrid, 256, flds, stream>>>(a);
      else if (a.A == 4) bwd_dst_fast_kernel<4, 8, 4><<<fgrid, 256, flds, stream>>>(a);
      else if (a.A == 3) bwd_dst_fast_kernel<4, 8, 3><<<fgrid, 256, flds, stream>>>(a);
      else bwd_dst_fast_kernel<4, 8, 1><<<fgrid, 256, flds, stream>>>(a);
      EGC_LAUNCH_CHECK("bwd_dst_fast_kernel");
    } else {
      bwd_dst_kernel<<<(unsigned)ceil_div(n, wpb), wpb * 64, lds, stream>>>(a);
      EGC_LAUNCH_CHECK("bwd_dst_kernel");
    }
  }

  if (a.rec_bytes != 0) {   // (arg, X) rows -> per-entry records
    RecArgs r;
    r.rowptr = graph->rowptr; r.plan = graph->plan;
    r.n_nodes = (int)n; r.ldb = ldb; r.slots = a.slots;
    r.group_u32 = 128 + ldb + ldb / 4 + 4;   // count | offset | values | column bytes (+ the over-read of the last list)
    const PlanCaps dcaps = plan_caps(n, graph->n_edges);
    const int64_t dchunks = (graph->n_chunks >= 0 && graph->n_chunks <= dcaps.cap_chunks) ? graph->n_chunks : dcaps.cap_chunks;
    r.chunk_blocks = (int)ceil_div(dchunks, 4);
    r.short_rows = a.rec_fused ? 0 : 1;
    // (with the records built inside bwd_dst_fast_kernel -- rows AND hub chunks -- there is nothing left to launch)
    const unsigned rgrid = a.rec_fused ? 0u : (unsigned)(r.chunk_blocks + ceil_div(n, (int64_t)16));
    const size_t rlds = (size_t)16 * r.group_u32 * sizeof(unsigned);
    for (int e = 0; e < 2; ++e) {
      r.arg = e == 0 ? a.arg_max : a.arg_min;
      r.x = e == 0 ? a.tab_x : a.tab_n;
      r.rec = const_cast<unsigned*>(e == 0 ? a.rec_x : a.rec_n);
      if (r.rec == nullptr || rgrid == 0) continue;
      switch ((a.slots + 15) / 16) {
        case 1: bwd_records_kernel<1><<<rgrid, 256, rlds, stream>>>(r); break;
        case 2: bwd_records_kernel<2><<<rgrid, 256, rlds, stream>>>(r); break;
        case 3: bwd_records_kernel<3><<<rgrid, 256, rlds, stream>>>(r); break;
        default: bwd_records_kernel<4><<<rgrid, 256, rlds, stream>>>(r); break;
      }
      EGC_LAUNCH_CHECK("bwd_records_kernel");
    }
  }

  int lg = 0;
  while ((1 << lg) < a.slots && lg < 6) ++lg;
  if (lg < 4) lg = 4;
  a.lpr_log2 = lg;
  const int G = 64 >> lg;
  const int ns = (a.slots + (1 << lg) - 1) >> lg;  // slots per lane
  const PlanCaps caps = plan_caps(n_src, t_graph->n_edges);
  const int64_t n_chunks = (t_graph->n_chunks >= 0 && t_graph->n_chunks <= caps.cap_chunks) ? t_graph->n_chunks : caps.cap_chunks;
  a.chunk_blocks = (int)ceil_div(n_chunks, 4);
  const unsigned grid = (unsigned)(a.chunk_blocks + ceil_div(n_src, (int64_t)4 * G));
  unsigned fl = SRC_STATIC | (a.need_t ? SRC_T : 0u) | (a.tab_s != nullptr ? SRC_S : 0u) | (a.tab_v != nullptr ? SRC_V : 0u) |
                (a.tab_x != nullptr ? SRC_X : 0u) | (a.tab_n != nullptr ? SRC_N : 0u) | (a.x_looped ? SRC_XL : 0u) |
                (a.y_looped ? SRC_YL : 0u) | (a.rec_bytes != 0 ? SRC_REC : 0u);
  if (getenv("EGC_BWD_GENERIC") != nullptr) fl = 0;
  // the same layer kinds on low-degree batches, where no records are built (records_apply): the arg-byte form, compiled in
  if (ns == 1 && fl == (SRC_STATIC | SRC_T | SRC_X | SRC_YL)) {                                // add+mean+max (molhiv EGC-M)
    bwd_src_kernel<1, SRC_STATIC | SRC_T | SRC_X | SRC_YL><<<grid, 256, 0, stream>>>(a);
  } else if (ns == 1 && fl == (SRC_STATIC | SRC_T | SRC_S | SRC_X | SRC_YL)) {                 // symadd+max+mean
    bwd_src_kernel<1, SRC_STATIC | SRC_T | SRC_S | SRC_X | SRC_YL><<<grid, 256, 0, stream>>>(a);
  } else if (ns == 1 && fl == (SRC_STATIC | SRC_T | SRC_V | SRC_X | SRC_YL)) {                 // add+std+max (zinc EGC-M)
    bwd_src_kernel<1, SRC_STATIC | SRC_T | SRC_V | SRC_X | SRC_YL><<<grid, 256, 0, stream>>>(a);
  } else if (ns == 1 && fl == (SRC_STATIC | SRC_T | SRC_S | SRC_V | SRC_X | SRC_YL)) {         // symadd+std+max (CIFAR EGC-M)
    bwd_src_kernel<1, SRC_STATIC | SRC_T | SRC_S | SRC_V | SRC_X | SRC_YL><<<grid, 256, 0, stream>>>(a);
  } else if (ns == 1 && fl == (SRC_STATIC | SRC_T | SRC_S | SRC_X | SRC_XL | SRC_YL)) {        // EGConv sum+mean+max+symnorm
    bwd_src_kernel<1, SRC_STATIC | SRC_T | SRC_S | SRC_X | SRC_XL | SRC_YL><<<grid, 256, 0, stream>>>(a);
  } else
  if (ns == 1 && fl == (SRC_STATIC | SRC_T | SRC_S | SRC_X | SRC_XL | SRC_YL | SRC_REC)) {  // EGConv sum+mean+max+symnorm (north star)
    bwd_src_kernel<1, SRC_STATIC | SRC_T | SRC_S | SRC_X | SRC_XL | SRC_YL | SRC_REC><<<grid, 256, 0, stream>>>(a);
  } else if (ns == 1 && fl == (SRC_STATIC | SRC_T | SRC_S | SRC_X | SRC_YL | SRC_REC)) {     // EfficientGraphConv symadd+max+mean
    bwd_src_kernel<1, SRC_STATIC | SRC_T | SRC_S | SRC_X | SRC_YL | SRC_REC><<<grid, 256, 0, stream>>>(a);
  } else if (ns == 1 && fl == (SRC_STATIC | SRC_S | SRC_YL)) {                     // EfficientGraphConv symadd (EGC-S)
    bwd_src_kernel<1, SRC_STATIC | SRC_S | SRC_YL><<<grid, 256, 0, stream>>>(a);
  } else if (ns == 1 && fl == (SRC_STATIC | SRC_S | SRC_XL | SRC_YL)) {            // EGConv symnorm
    bwd_src_kernel<1, SRC_STATIC | SRC_S | SRC_XL | SRC_YL><<<grid, 256, 0, stream>>>(a);
  } else if (ns == 1 && fl == (SRC_STATIC | SRC_T | SRC_X | SRC_REC)) {                      // relational EGC: mean+max, raw
    bwd_src_kernel<1, SRC_STATIC | SRC_T | SRC_X | SRC_REC><<<grid, 256, 0, stream>>>(a);
  } else if (ns == 1 && fl == (SRC_STATIC | SRC_T | SRC_V | SRC_X | SRC_YL | SRC_REC)) {     // EfficientGraphConv add+std+max (zinc EGC-M)
    bwd_src_kernel<1, SRC_STATIC | SRC_T | SRC_V | SRC_X | SRC_YL | SRC_REC><<<grid, 256, 0, stream>>>(a);
  } else if (ns == 1 && fl == (SRC_STATIC | SRC_T | SRC_S | SRC_V | SRC_X | SRC_YL | SRC_REC)) {  // symadd+std+max (CIFAR EGC-M)
    bwd_src_kernel<1, SRC_STATIC | SRC_T | SRC_S | SRC_V | SRC_X | SRC_YL | SRC_REC><<<grid, 256, 0, stream>>>(a);
  } else if (ns == 1 && fl == (SRC_STATIC | SRC_T | SRC_X | SRC_YL | SRC_REC)) {             // add+mean+max (molhiv EGC-M)
    bwd_src_kernel<1, SRC_STATIC | SRC_T | SRC_X | SRC_YL | SRC_REC><<<grid, 256, 0, stream>>>(a);
  } else
  switch (ns) {
    case 1: bwd_src_kernel<1><<<grid, 256, 0, stream>>>(a); break;
    case 2: bwd_src_kernel<2><<<grid, 256, 0, stream>>>(a); break;
    case 3: bwd_src_kernel<3><<<grid, 256, 0, stream>>>(a); break;
    case 4: bwd_src_kernel<4><<<grid, 256, 0, stream>>>(a); break;
    default: return EGC_ERR_UNSUPPORTED;
  }
  EGC_LAUNCH_CHECK("bwd_src_kernel");
  return EGC_OK;
}

}  // extern "C"
