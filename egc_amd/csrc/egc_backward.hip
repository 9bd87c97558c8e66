// Backward of the fused EGC aggregate+combine (gfx950).  Not in the reference as code: PyTorch autograd
// derives it implicitly through experiments/layers.py:103-138 / optimized_layers.py:186-208 (gather ->
// scatter per aggregator -> stack -> weighted sum).  SURVEY.md 8(f) rank 1.
//
// Given g = dL/d out [N, F_out], per destination row i with aggregates agg_a (recomputed by re-gathering,
// nothing but `bases` and the pre-activation `weightings` is kept from the forward):
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
// Two launches: `bwd_dst_kernel` (one wavefront per destination row, lane = basis column) and
// `bwd_src_kernel` (one wavefront per source row over the transposed CSR).  Written for generality and
// correctness first (any H, B, L, A; all nonlinearities); it is not yet tuned like the forward.
#include "egc_aggregate_dev.h"

namespace egc {

struct BwdArgs {
  // destination-side graph
  const int* rowptr;
  const int* col;
  const float* dis;       // deg^-1/2 of the symnorm edge set or nullptr
  const int* max_index;
  // transposed graph (rows = sources, entries = destinations)
  const int* t_rowptr;
  const int* t_col;
  const float* bases;        // [n_src_rows, ldb]
  const float* weightings;   // [N, W] pre-activation
  const float* grad_out;     // [N, F_out]
  float* d_bases;            // [n_src_rows, ldb]  (zero-initialised by the host)
  float* d_weightings;       // [N, W]
  float* tab_t;              // [N, ldb]
  float* tab_s;              // [N, ldb] or nullptr
  float* tab_v;              // [N, ldb] or nullptr
  int n_nodes, n_src_rows;
  int ldb, F_g, F_out, W, H, B, A, L;
  int aggr[EGC_MAX_AGGRS];
  int x_looped, y_looped, loops_all;
  int act;
  int lds_floats_per_wave;
};

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// One wavefront per destination row.  LDS per wavefront: agg [A][ldb], g [F_out], w' [W], d w' [W].
__global__ void __launch_bounds__(256) bwd_dst_kernel(BwdArgs a) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row = blockIdx.x * (blockDim.x >> 6) + wave;
  if (row >= a.n_nodes) return;
  float* lds_agg = smem + wave * a.lds_floats_per_wave;
  float* lds_g = lds_agg + a.A * a.ldb;
  float* lds_w = lds_g + ((a.F_out + 3) & ~3);
  float* lds_dagg = lds_w + ((a.W + 3) & ~3);  // d w' scratch [W]

  const int start = a.rowptr[row], end = a.rowptr[row + 1];
  const bool xl = a.x_looped != 0, yl = a.y_looped != 0;
  const int nloop = a.loops_all ? a.n_nodes : (*a.max_index + 1);
  const bool has_self = row < nloop;
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

  // count of self entries (excluded from LOOPED sets)
  int nself = 0;
  if (xl || yl) {
    for (int p = start + lane; p < end; p += 64) nself += (a.col[p] == row);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nself += __shfl_xor(nself, off);
  }
  const int deg = end - start;
  const int cnt = xl ? deg - nself + (has_self ? 1 : 0) : deg;
  const float cntf = (float)max(cnt, 1);

  for (int c0 = 0; c0 < a.F_g; c0 += 64) {
    const int c = c0 + lane;
    const bool cv = c < a.F_g;
    // ---- re-gather: aggregates of this column + first-attaining sources for max / min
    float sum = 0.f, sq = 0.f, ws = 0.f, mx = -INFINITY, mn = INFINITY;
    int amx = -1, amn = -1;
    for (int p = start; p < end; ++p) {
      const int j = a.col[p];
      const bool is_self = j == row;
      const float v = cv ? a.bases[(int64_t)j * a.ldb + c] : 0.f;
      if (!(xl && is_self)) {
        sum += v;
        sq += __fmul_rn(v, v);
        if (v > mx) { mx = v; amx = j; }
        if (v < mn) { mn = v; amn = j; }
      }
      if (!(yl && is_self) && a.dis != nullptr) ws = fmaf(a.dis[j] * dis_i, v, ws);
    }
    const float vself = cv ? a.bases[(int64_t)row * a.ldb + c] : 0.f;
    if (xl && has_self) {
      sum += vself;
      sq += __fmul_rn(vself, vself);
      if (vself > mx) { mx = vself; amx = row; }
      if (vself < mn) { mn = vself; amn = row; }
    }
    if (yl && has_self) ws = fmaf(dis_i * dis_i, vself, ws);
    const float mean = sum / cntf;
    const float var = __fsub_rn(sq / cntf, __fmul_rn(mean, mean));
    const float sd = sqrtf(fmaxf(var, 0.f) + 1e-5f);
    if (cv) {
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
      const int b = c / a.L, l = c - b * a.L;
      float d_t = 0.f, d_s = 0.f, d_v = 0.f;
      for (int t = 0; t < a.A; ++t) {
        float d = 0.f;
        for (int h = 0; h < a.H; ++h) d = fmaf(lds_w[h * AB + b * a.A + t], lds_g[h * a.L + l], d);
        switch (a.aggr[t]) {
          case EGC_AGGR_SUM: d_t += d; break;
          case EGC_AGGR_MEAN: d_t += d / cntf; break;
          case EGC_AGGR_MAX: if (cnt > 0 && amx >= 0) atomicAdd(&a.d_bases[(int64_t)amx * a.ldb + c], d); break;
          case EGC_AGGR_MIN: if (cnt > 0 && amn >= 0) atomicAdd(&a.d_bases[(int64_t)amn * a.ldb + c], d); break;
          case EGC_AGGR_VAR: d_v += d; break;
          case EGC_AGGR_STD: d_v += (var > 0.f) ? d / (2.0f * sd) : 0.f; break;
          default: d_s += d * dis_i; break;
        }
      }
      // var = E[x^2] - mean^2:  d/dx_j = 2 (x_j - mean) / cnt
      a.tab_t[(int64_t)row * a.ldb + c] = d_t - 2.0f * mean * d_v / cntf;
      if (a.tab_s != nullptr) a.tab_s[(int64_t)row * a.ldb + c] = d_s;
      if (a.tab_v != nullptr) a.tab_v[(int64_t)row * a.ldb + c] = 2.0f * d_v / cntf;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- d w'[h][b][t] = sum_l g[h*L + l] * agg_t[b*L + l], then through the nonlinearity
  for (int k = lane; k < a.W; k += 64) {
    const int h = k / AB, r = k - h * AB, b = r / a.A, t = r - b * a.A;
    float d = 0.f;
    for (int l = 0; l < a.L; ++l) d = fmaf(lds_g[h * a.L + l], lds_agg[t * a.ldb + b * a.L + l], d);
    lds_dagg[k] = d;  // re-used as scratch for d w'
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
    a.d_weightings[(int64_t)row * a.W + k] = d;
  }
}

// One wavefront per source row j: d bases[j] += sum over out-neighbours of the tables (+ self-loop terms).
__global__ void __launch_bounds__(256) bwd_src_kernel(BwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= a.n_src_rows) return;
  const bool owned = row < a.n_nodes;
  const int start = a.t_rowptr[row], end = a.t_rowptr[row + 1];
  const bool xl = a.x_looped != 0, yl = a.y_looped != 0;
  const int nloop = a.loops_all ? a.n_nodes : (*a.max_index + 1);
  const bool has_self = owned && row < nloop;
  const float dis_j = a.dis != nullptr ? a.dis[row] : 0.f;
  for (int c0 = 0; c0 < a.F_g; c0 += 64) {
    const int c = c0 + lane;
    if (c >= a.F_g) continue;
    float st = 0.f, ss = 0.f, sv = 0.f;
    for (int p = start; p < end; ++p) {
      const int i = a.t_col[p];
      const bool is_self = i == row;
      if (!(xl && is_self)) {
        st += a.tab_t[(int64_t)i * a.ldb + c];
        if (a.tab_v != nullptr) sv += a.tab_v[(int64_t)i * a.ldb + c];
      }
      if (a.tab_s != nullptr && !(yl && is_self)) ss += a.tab_s[(int64_t)i * a.ldb + c];
    }
    if (xl && has_self) {
      st += a.tab_t[(int64_t)row * a.ldb + c];
      if (a.tab_v != nullptr) sv += a.tab_v[(int64_t)row * a.ldb + c];
    }
    if (yl && has_self && a.tab_s != nullptr) ss += a.tab_s[(int64_t)row * a.ldb + c];
    const float own = a.bases[(int64_t)row * a.ldb + c];
    a.d_bases[(int64_t)row * a.ldb + c] += st + dis_j * ss + own * sv;
  }
}

}  // namespace egc

using namespace egc;

extern "C" {

size_t egc_backward_workspace_bytes(const egc_layer* layer, int64_t n_nodes) {
  if (layer == nullptr || n_nodes < 0 || layer->num_heads <= 0) return 0;
  const int ldb = egc_bases_ld(layer);
  return (size_t)3 * (size_t)n_nodes * ldb * sizeof(float) + 256;
}

int egc_aggregate_combine_backward_f32(const egc_graph* graph, const int32_t* t_rowptr, const int32_t* t_col,
                                       const egc_layer* layer, const float* bases, int32_t ldb,
                                       const float* weightings, const float* grad_out, float* d_bases,
                                       float* d_weightings, void* workspace, size_t workspace_bytes,
                                       egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (graph == nullptr || layer == nullptr || t_rowptr == nullptr) return EGC_ERR_INVALID;
  if (layer->num_aggrs <= 0 || layer->num_aggrs > EGC_MAX_AGGRS || layer->out_channels % layer->num_heads != 0)
    return EGC_ERR_INVALID;
  const int64_t n = graph->n_nodes;
  const int64_t n_src = graph->n_src_rows > 0 ? graph->n_src_rows : n;
  if (n == 0) return EGC_OK;
  // t_rowptr has n_src + 1 entries.  Rectangular adjacencies (n_src != n) carry no self loops / symnorm; on a
  // vertex partition the caller owns the reverse exchange of the halo rows of d_bases.
  if (n_src < n && (layer->agg_set == EGC_SET_LOOPED || layer_uses_symnorm(layer))) return EGC_ERR_INVALID;
  if (bases == nullptr || weightings == nullptr || grad_out == nullptr || d_bases == nullptr || d_weightings == nullptr)
    return EGC_ERR_INVALID;
  if (ldb != egc_bases_ld(layer)) return EGC_ERR_INVALID;
  if (layer->weight_layout != EGC_LAYOUT_HBA) return EGC_ERR_UNSUPPORTED;  // the host packs [h][b][a]
  if (workspace == nullptr || workspace_bytes < egc_backward_workspace_bytes(layer, n)) return EGC_ERR_WORKSPACE;

  BwdArgs a;
  a.rowptr = graph->rowptr;
  a.col = graph->col;
  a.max_index = graph->max_index;
  a.t_rowptr = t_rowptr;
  a.t_col = t_col;
  a.bases = bases;
  a.weightings = weightings;
  a.grad_out = grad_out;
  a.d_bases = d_bases;
  a.d_weightings = d_weightings;
  a.n_nodes = (int)n;
  a.n_src_rows = (int)n_src;
  a.ldb = ldb;
  a.H = layer->num_heads;
  a.B = layer->num_bases;
  a.A = layer->num_aggrs;
  a.L = layer->out_channels / layer->num_heads;
  a.F_g = a.B * a.L;
  a.F_out = layer->out_channels;
  a.W = a.H * a.B * a.A;
  bool sym = false, var = false;
  for (int t = 0; t < EGC_MAX_AGGRS; ++t) {
    a.aggr[t] = t < a.A ? layer->aggrs[t] : 0;
    if (t < a.A && layer->aggrs[t] == EGC_AGGR_SYMNORM) sym = true;
    if (t < a.A && (layer->aggrs[t] == EGC_AGGR_VAR || layer->aggrs[t] == EGC_AGGR_STD)) var = true;
  }
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
  a.tab_s = sym ? ws + (size_t)n * ldb : nullptr;
  a.tab_v = var ? ws + (size_t)2 * n * ldb : nullptr;
  a.lds_floats_per_wave = a.A * ldb + ((a.F_out + 3) & ~3) + 2 * ((a.W + 3) & ~3);
  int wpb = 4;
  if ((size_t)wpb * a.lds_floats_per_wave * sizeof(float) > 48 * 1024) wpb = 1;
  const size_t lds = (size_t)wpb * a.lds_floats_per_wave * sizeof(float);
  if (lds > 64 * 1024) return EGC_ERR_UNSUPPORTED;
  bwd_dst_kernel<<<(unsigned)ceil_div(n, wpb), wpb * 64, lds, stream>>>(a);
  EGC_LAUNCH_CHECK("bwd_dst_kernel");
  bwd_src_kernel<<<(unsigned)ceil_div(n_src, 4), 256, 0, stream>>>(a);
  EGC_LAUNCH_CHECK("bwd_src_kernel");
  return EGC_OK;
}

}  // extern "C"
