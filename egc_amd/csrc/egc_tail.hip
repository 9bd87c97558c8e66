// The caller-side tail of an EGC layer in TRAINING mode: the reference's graph nets run
//     h = conv(x, edge_index);  h = bn(h);  h = relu(h);  x = x + h
// every step (zinc/models.py:66-72, mol/pna_style_models.py:71-78, cifar/models.py:67-74).  In eval mode the whole
// tail folds into the store of the aggregate kernel (egc_post); with batch statistics it cannot -- the statistics
// need every row of h first -- but it still is two passes instead of PyTorch's five (statistics, normalise, relu,
// add; and as many again backward):
//     forward   column_moments  (sum h, sum h^2 per channel, float64 accumulation)         reads h
//               affine_act_residual   out = act(h * scale + shift) + residual              reads h, residual; writes out
//     backward  tail_backward_moments (sum g, sum g h per channel; g = dout * [pre-activation > 0])   reads dout, h
//               tail_backward   dh = A g + B h + C  (the BatchNorm backward with its two sums folded into A, B, C)
// All four are plain streaming kernels: 16 bytes per lane, channel constants from LDS, HBM-bound.
#include <algorithm>

#include "egc_common.h"
#include "egc_pack_map.h"

namespace egc {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ inline d4 to_d4(f4 v) { return d4{(double)v.x, (double)v.y, (double)v.z, (double)v.w}; }

// out[block][0][c] = sum over the block's rows of a[r][c] (* mask), out[block][1][c] = sum of a[r][c] * b[r][c]
// (b == a: the second moment).  MASKED: a is multiplied by [b * scale + shift > 0] first (the ReLU mask recomputed
// from the saved pre-BatchNorm activations instead of stored).
// dropout: keep[r][c] (one byte per element, 0 = dropped), survivors scaled by keep_scale = 1 / (1 - p)
__device__ inline f4 keep4(const unsigned char* __restrict__ keep, int64_t quad, float keep_scale) {
  const unsigned m = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(keep) + quad);
  return f4{(m & 0xffu) ? keep_scale : 0.f, (m & 0xff00u) ? keep_scale : 0.f, (m & 0xff0000u) ? keep_scale : 0.f,
            (m & 0xff000000u) ? keep_scale : 0.f};
}

// Small inputs (few partial blocks): the per-column step between the statistics pass and the elementwise pass rides in
// the SAME launch -- the block that arrives last (agent-scope counter; the partials travel as agent-scope stores and
// loads: no cache write-back fence) adds the partials in the finalize kernels' fixed order and finishes every column.
// One launch less per block and direction -- and NOT faster where it was meant to be: the chain write-through partials ->
// arrival atomic -> last block's reads costs 17-18 us per launch against ~6 + ~4.5 us for the two kernels, and a replayed
// ZINC-shaped step takes 426-429 us with it against 372-373 us without (include/egc_hip.h); opt-in.
struct FinArgs {
  int* sync;                    // arrival counter: zero on entry, zero again on exit
  double eps, momentum;
  const float* gamma;
  const float* beta;
  double* stats;                // forward: written [3][cols]
  float* affine;
  float* running_mean;
  float* running_var;
  const int64_t* n_tracked;
  const double* bstats;         // backward: the forward's stats
  float* outv;                  // backward: [5][cols]
  float* dh_sums;               // backward: column sums of dh [cols], or nullptr
};

template <bool COHERENT>
__device__ inline void moment_totals(const double* __restrict__ partials, int n_partials, int cols, int col, int pl,
                                     double (&red)[2][4][64 + 1], double& t1, double& t2);
__device__ inline void bn_forward_column(double s1, double s2, int col, int cols, double n_rows, const float* gamma, const float* beta,
                                         double eps, double* stats, float* affine, float* running_mean, float* running_var,
                                         double momentum, double n_tracked);
__device__ inline void bn_backward_column(double s1, double sgh, int col, int cols, double n_rows, const double* stats,
                                          const float* gamma, float* outv, float* dh_sums);

template <bool MASKED, int FIN = 0>   // FIN: 0 partials only, 1 + forward finalize, 2 + backward finalize
__global__ void __launch_bounds__(256) column_moments_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             int64_t n_rows, int cols, int rows_per_block,
                                                             double* __restrict__ out, int relu,
                                                             const unsigned char* __restrict__ keep, float keep_scale,
                                                             int64_t* __restrict__ count_inc,
                                                             const int64_t* __restrict__ n_valid, FinArgs fin = FinArgs()) {
  __shared__ d4 red[2][256];
  // rows [*n_valid, n_rows) are padding (a batch padded to the static shape of a hipGraph recording): not counted
  if (n_valid != nullptr) n_rows = min(n_rows, max(*n_valid, (int64_t)0));
  // BatchNorm's num_batches_tracked: bumped here, one launch BEFORE the finalize kernel reads it
  if (count_inc != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    if (FIN != 0) __hip_atomic_fetch_add(count_inc, (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read by the last block
    else *count_inc += 1;
  }
  const int cg = cols >> 2;                 // 16-byte column groups (<= 256)
  const int rl = 256 / cg;                  // row lanes
  const int g = threadIdx.x % cg, lane_r = threadIdx.x / cg;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(r0 + rows_per_block, n_rows);
  d4 s1 = d4{0, 0, 0, 0}, s2 = d4{0, 0, 0, 0};
  f4 sc = f4{0.f, 0.f, 0.f, 0.f}, sh = sc;
  if (MASKED && relu) { sc = reinterpret_cast<const f4*>(scale)[g]; sh = reinterpret_cast<const f4*>(shift)[g]; }
  auto fold = [&](f4 va, f4 vb, int64_t r) {
    if (MASKED) {
      if (keep != nullptr) va = va * keep4(keep, r * cg + g, keep_scale);
      if (relu) {
        const f4 pre = vb * sc + sh;
        va = f4{pre.x > 0.f ? va.x : 0.f, pre.y > 0.f ? va.y : 0.f, pre.z > 0.f ? va.z : 0.f, pre.w > 0.f ? va.w : 0.f};
      }
    }
    const d4 da = to_d4(va);
    s1 += da;
    s2 += da * to_d4(vb);
  };
  if (lane_r < rl) {
    // four rows of loads in flight per thread (the float64 adds are a dependent chain; the loads need not wait for it)
    int64_t r = r0 + lane_r;
    for (; r + 3 * (int64_t)rl < r1; r += 4 * (int64_t)rl) {
      f4 va[4], vb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        va[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a + (r + u * (int64_t)rl) * cols) + g);
        vb[u] = MASKED ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(b + (r + u * (int64_t)rl) * cols) + g) : va[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) fold(va[u], vb[u], r + u * (int64_t)rl);
    }
    for (; r < r1; r += rl) {
      const f4 va = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a + r * cols) + g);
      const f4 vb = MASKED ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(b + r * cols) + g) : va;
      fold(va, vb, r);
    }
  }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  if (lane_r == 0) {
    for (int k = 1; k < rl; ++k) { s1 += red[0][k * cg + g]; s2 += red[1][k * cg + g]; }
    double* o = out + (int64_t)blockIdx.x * 2 * cols;
    if (FIN != 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __hip_atomic_store(o + 4 * g + i, s1[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + cols + 4 * g + i, s2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      reinterpret_cast<d4*>(o)[g] = s1;
      reinterpret_cast<d4*>(o + cols)[g] = s2;
    }
  }
  if constexpr (FIN != 0) {
    __shared__ int last_block;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this block's partials have left
    __syncthreads();
    if (threadIdx.x == 0)
      last_block = __hip_atomic_fetch_add(fin.sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!last_block) return;
    if (threadIdx.x == 0) __hip_atomic_store(fin.sync, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifndef EGC_BN_NO_ACQ
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    const double nr = fmax((double)n_rows, 1.0);
    // One thread per column; at most 64 partial blocks (host), added in the finalize kernels' order -- eight runs of eight
    // partials, then the eight run totals: the same bits as the two-launch form.  All loads of a column are independent.
    const int np = (int)gridDim.x;
    for (int col = threadIdx.x; col < cols; col += 256) {
      double v1[64], v2[64];
#pragma unroll
      for (int p = 0; p < 64; ++p) {
        const bool ok = p < np;
        // (plain loads: the acquire fence above dropped this XCD's stale lines, the writers' stores were write-through)
        v1[p] = ok ? out[(int64_t)p * 2 * cols + col] : 0.0;
        v2[p] = ok ? out[(int64_t)p * 2 * cols + cols + col] : 0.0;
      }
      double t1 = 0.0, t2 = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        double u1 = 0.0, u2 = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { u1 += v1[8 * k + j]; u2 += v2[8 * k + j]; }
        t1 += u1; t2 += u2;
      }
      if (FIN == 1) {
        double tracked = 1.0;
        if (fin.momentum < 0.0 && fin.running_mean != nullptr)
          tracked = (double)__hip_atomic_load(fin.n_tracked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bn_forward_column(t1, t2, col, cols, nr, fin.gamma, fin.beta, fin.eps, fin.stats, fin.affine, fin.running_mean,
                          fin.running_var, fin.momentum, tracked);
      } else {
        bn_backward_column(t1, t2, col, cols, nr, fin.bstats, fin.gamma, fin.outv, fin.dh_sums);
      }
    }
  }
}

// Column totals of the partial moments: FIN_COLS columns x FIN_LANES partial lanes per workgroup.  Each lane adds its
// partials in index order (four independent loads in flight: one load per dependent float64 add left this launch at
// 23 us for 1,024 partials), the lanes of a column are then added eight at a time in lane order -- a fixed order: the
// result does not depend on timing.
constexpr int FIN_COLS = 4, FIN_LANES = 64;
template <bool COHERENT>
__device__ inline void moment_totals(const double* __restrict__ partials, int n_partials, int cols, int col, int pl,
                                     double (&red)[2][FIN_COLS][FIN_LANES + 1], double& t1, double& t2) {
  auto ld = [&](int64_t i) -> double {      // partials of the same launch come through agent-scope loads
    return COHERENT ? __hip_atomic_load(partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : partials[i];
  };
  const int cl = threadIdx.x % FIN_COLS;
  double s1 = 0.0, s2 = 0.0;
  if (col < cols) {
    int p = pl;
    for (; p + 3 * FIN_LANES < n_partials; p += 4 * FIN_LANES) {
      double a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = ld((int64_t)(p + u * FIN_LANES) * 2 * cols + col);
        b[u] = ld((int64_t)(p + u * FIN_LANES) * 2 * cols + cols + col);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s1 += a[u]; s2 += b[u]; }
    }
    for (; p < n_partials; p += FIN_LANES) {
      s1 += ld((int64_t)p * 2 * cols + col);
      s2 += ld((int64_t)p * 2 * cols + cols + col);
    }
  }
  red[0][cl][pl] = s1;
  red[1][cl][pl] = s2;
  __syncthreads();
  if (pl < 8) {      // lanes 8 k .. 8 k + 7 -> lane k
    double u1 = 0.0, u2 = 0.0;
    for (int k = 0; k < 8; ++k) { u1 += red[0][cl][pl * 8 + k]; u2 += red[1][cl][pl * 8 + k]; }
    s1 = u1; s2 = u2;
  }
  __syncthreads();
  if (pl < 8) { red[0][cl][pl] = s1; red[1][cl][pl] = s2; }
  __syncthreads();
  t1 = t2 = 0.0;
  if (pl == 0)
    for (int k = 0; k < 8; ++k) { t1 += red[0][cl][k]; t2 += red[1][cl][k]; }
}

__device__ inline void bn_forward_column(double s1, double s2, int col, int cols, double n_rows, const float* gamma, const float* beta,
                                         double eps, double* stats, float* affine, float* running_mean, float* running_var,
                                         double momentum, double n_tracked) {
  const double mean = s1 / n_rows;
  double var = s2 / n_rows - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const double rstd = 1.0 / sqrt(var + eps);
  const double g = gamma != nullptr ? (double)gamma[col] : 1.0, b = beta != nullptr ? (double)beta[col] : 0.0;
  stats[col] = mean;
  stats[cols + col] = var;
  stats[2 * cols + col] = rstd;
  affine[col] = (float)(g * rstd);
  affine[cols + col] = (float)(b - mean * g * rstd);
  if (running_mean != nullptr) {
    const double m = momentum >= 0.0 ? momentum : 1.0 / n_tracked;
    running_mean[col] = (float)((1.0 - m) * (double)running_mean[col] + m * mean);
    running_var[col] = (float)((1.0 - m) * (double)running_var[col] + m * var * (n_rows > 1.0 ? n_rows / (n_rows - 1.0) : 1.0));
  }
}

__device__ inline void bn_backward_column(double s1, double sgh, int col, int cols, double n_rows, const double* stats,
                                          const float* gamma, float* outv, float* dh_sums) {
  const double mean = stats[col], rstd = stats[2 * cols + col];
  const double s2 = (sgh - mean * s1) * rstd;                 // sum g * h_hat
  const double a = (gamma != nullptr ? (double)gamma[col] : 1.0) * rstd;
  const float cg = (float)a, ch = (float)(-(a / n_rows) * rstd * s2), c1 = (float)(-(a / n_rows) * (s1 - mean * rstd * s2));
  outv[col] = (float)s2;                                      // d gamma
  outv[cols + col] = (float)s1;                               // d beta
  outv[2 * cols + col] = cg;                                  // dh = a g - (a / n) (s1 + (h - mean) rstd s2)
  outv[3 * cols + col] = ch;
  outv[4 * cols + col] = c1;
  // The column sum of the dh the elementwise pass will write (the gradient of a bias in front of the BatchNorm: zero but for
  // rounding -- autograd's value is the float sum of its own dh): sum_rows (cg g + ch h + c1) with the three coefficients AS
  // ROUNDED above, from the sums this step already holds (sum g = s1, sum h = n mean) -- no pass over dh.
  if (dh_sums != nullptr) dh_sums[col] = (float)((double)cg * s1 + (double)ch * (n_rows * mean) + n_rows * (double)c1);
}

// Everything between the statistics pass and the elementwise pass of the training-mode BatchNorm tail, per column:
// batch mean / biased variance / 1/std (float64), the affine pair the elementwise kernel applies, and the running
// statistics of the module (nn.BatchNorm1d: unbiased variance, momentum or -- momentum < 0 -- the cumulative average
// 1 / *n_tracked, the count already incremented by the caller).  Was ~20 five-microsecond torch kernels per layer.
__global__ void __launch_bounds__(256) bn_forward_finalize_kernel(const double* __restrict__ partials, int n_partials, int cols,
                                                                  double n_rows, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, double eps,
                                                                  double* __restrict__ stats, float* __restrict__ affine,
                                                                  float* __restrict__ running_mean,
                                                                  float* __restrict__ running_var, double momentum,
                                                                  const int64_t* __restrict__ n_tracked,
                                                                  const int64_t* __restrict__ n_valid) {
  __shared__ double red[2][FIN_COLS][FIN_LANES + 1];
  if (n_valid != nullptr) n_rows = fmax(fmin(n_rows, (double)*n_valid), 1.0);
  const int col = blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS, pl = threadIdx.x / FIN_COLS;
  double s1, s2;
  moment_totals<false>(partials, n_partials, cols, col, pl, red, s1, s2);
  if (pl != 0 || col >= cols) return;
  bn_forward_column(s1, s2, col, cols, n_rows, gamma, beta, eps, stats, affine, running_mean, running_var, momentum,
                    (momentum < 0.0 && running_mean != nullptr) ? (double)(*n_tracked) : 1.0);
}

// Backward counterpart: from the partial sums (sum g, sum g h) of the masked upstream gradient, the gradients of the
// affine parameters and the three per-column coefficients of dh = c_g g + c_h h + c_1.
__global__ void __launch_bounds__(256) bn_backward_finalize_kernel(const double* __restrict__ partials, int n_partials, int cols,
                                                                   double n_rows, const double* __restrict__ stats,
                                                                   const float* __restrict__ gamma, float* __restrict__ outv,
                                                                   const int64_t* __restrict__ n_valid, float* __restrict__ dh_sums) {
  __shared__ double red[2][FIN_COLS][FIN_LANES + 1];
  if (n_valid != nullptr) n_rows = fmax(fmin(n_rows, (double)*n_valid), 1.0);
  const int col = blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS, pl = threadIdx.x / FIN_COLS;
  double s1, sgh;
  moment_totals<false>(partials, n_partials, cols, col, pl, red, s1, sgh);
  if (pl != 0 || col >= cols) return;
  bn_backward_column(s1, sgh, col, cols, n_rows, stats, gamma, outv, dh_sums);
}

// out = act(h * scale + shift) + residual, 16 bytes per thread
__global__ void __launch_bounds__(256) affine_act_residual_kernel(const float* __restrict__ h, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift,
                                                                  const float* __restrict__ residual, int relu,
                                                                  int64_t quads, int cg, float* __restrict__ out,
                                                                  const unsigned char* __restrict__ keep, float keep_scale,
                                                                  const int64_t* __restrict__ n_valid) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= quads) return;
  const bool p2 = (cg & (cg - 1)) == 0;              // (a 64-bit modulo per 16 bytes is most of this kernel's arithmetic)
  const int g = p2 ? (int)(k & (cg - 1)) : (int)(k % cg);
  if (n_valid != nullptr && (p2 ? k >> __builtin_ctz(cg) : k / cg) >= *n_valid) {      // padding rows stay zero
    __builtin_nontemporal_store(f4{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f4*>(out) + k);
    return;
  }
  f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(h) + k) * reinterpret_cast<const f4*>(scale)[g] +
         reinterpret_cast<const f4*>(shift)[g];
  if (relu) v = f4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
  if (keep != nullptr) v = v * keep4(keep, k, keep_scale);
  if (residual != nullptr) v += __builtin_nontemporal_load(reinterpret_cast<const f4*>(residual) + k);
  __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out) + k);
}

// dh = A * g + B * h + C with g = dout * [h * scale + shift > 0] (relu) or dout
__global__ void __launch_bounds__(256) tail_backward_kernel(const float* __restrict__ dout, const float* __restrict__ h,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            int relu, const float* __restrict__ ca, const float* __restrict__ cb,
                                                            const float* __restrict__ cc, int64_t quads, int cg,
                                                            float* __restrict__ dh, const unsigned char* __restrict__ keep,
                                                            float keep_scale, const int64_t* __restrict__ n_valid) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= quads) return;
  const bool p2 = (cg & (cg - 1)) == 0;
  const int g = p2 ? (int)(k & (cg - 1)) : (int)(k % cg);
  if (n_valid != nullptr && (p2 ? k >> __builtin_ctz(cg) : k / cg) >= *n_valid) {      // padding rows carry no gradient (the per-channel constant
    __builtin_nontemporal_store(f4{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f4*>(dh) + k);   // term would otherwise reach
    return;                                             // the layer's bias and weight gradients)
  }
  const f4 hv = __builtin_nontemporal_load(reinterpret_cast<const f4*>(h) + k);
  f4 gv = __builtin_nontemporal_load(reinterpret_cast<const f4*>(dout) + k);
  if (keep != nullptr) gv = gv * keep4(keep, k, keep_scale);
  if (relu) {
    const f4 pre = hv * reinterpret_cast<const f4*>(scale)[g] + reinterpret_cast<const f4*>(shift)[g];
    gv = f4{pre.x > 0.f ? gv.x : 0.f, pre.y > 0.f ? gv.y : 0.f, pre.z > 0.f ? gv.z : 0.f, pre.w > 0.f ? gv.w : 0.f};
  }
  const f4 r = reinterpret_cast<const f4*>(ca)[g] * gv + reinterpret_cast<const f4*>(cb)[g] * hv + reinterpret_cast<const f4*>(cc)[g];
  __builtin_nontemporal_store(r, reinterpret_cast<f4*>(dh) + k);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// The GEMM operand of a layer from its parameters, and the parameter gradients from the operand's gradient.
//   wcat [F_in][B Ls + W] = [the B basis matrices side by side, each padded from L to Ls columns | comb_weight^T],
//   bcat [W] = comb_bias   (W = H B A).  The bases come as ONE [F_in][B L] matrix (EGConv.bases_weight) or as B matrices
//   [F_in][L] (EfficientGraphConv.bases_weight.{0..B-1}); with `permute` the Linear's rows [h][a][b] (EGConv's
//   comb_weight, optimized_layers.py:195-202) become columns [h][b][a], otherwise the rows are [h][b][a] already.
// GRAD: the same index map read the other way (wcat / bcat are the gradients, the parameters' gradients are written).
template <bool GRAD>
__global__ void __launch_bounds__(256) weights_pack_kernel(PackPtrs bases, float* __restrict__ comb_w,
                                                           float* __restrict__ comb_b, float* __restrict__ wcat,
                                                           float* __restrict__ bcat, PackDims d) {
  const int F_g = d.B * d.Ls, W = d.H * d.B * d.A, cols = F_g + W;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < (int64_t)d.F_in * cols) {
    const int k = (int)(idx / cols), c = (int)(idx - (int64_t)k * cols);
    float* p = pack_param_ptr(bases, comb_w, d, k, c);
    if (p != nullptr) {
      if (GRAD) *p = wcat[idx]; else wcat[idx] = *p;
    } else if (!GRAD) {
      wcat[idx] = 0.f;   // padding column of a padded basis
    }
  }
  if (idx < W && bcat != nullptr && comb_b != nullptr) {
    const int j = (int)idx;
    if (GRAD) comb_b[pack_comb_row(d, j)] = bcat[j]; else bcat[j] = comb_b[pack_comb_row(d, j)];
  }
}

}  // namespace
}  // namespace egc

using namespace egc;

extern "C" {

int egc_column_moments_f64(const float* a, const float* b, const float* scale, const float* shift, int32_t relu,
                           const uint8_t* keep, float keep_scale, int64_t n_rows, int32_t cols, double* partials,
                           int32_t n_partials, int64_t* count_inc, const int64_t* n_valid, egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_rows < 0 || cols <= 0 || partials == nullptr || n_partials <= 0) return EGC_ERR_INVALID;
  if ((cols & 3) != 0 || cols > 1024 || !aligned16(a) || !aligned16(b) || !aligned16(scale) || !aligned16(shift) ||
      (reinterpret_cast<uintptr_t>(partials) & 31) != 0)
    return EGC_ERR_UNSUPPORTED;
  if (n_rows > 0 && a == nullptr) return EGC_ERR_INVALID;
  const bool masked = b != nullptr;
  if (masked && relu && (scale == nullptr || shift == nullptr)) return EGC_ERR_INVALID;
  if (keep != nullptr && (!masked || (reinterpret_cast<uintptr_t>(keep) & 3) != 0)) return EGC_ERR_INVALID;
  const int rows_per_block = (int)std::max<int64_t>(ceil_div(n_rows, (int64_t)n_partials), 1);  // empty blocks write zeros
  if (masked)
    column_moments_kernel<true><<<(unsigned)n_partials, 256, 0, stream>>>(a, b, scale, shift, n_rows, cols,
                                                                          rows_per_block, partials, relu, keep, keep_scale, count_inc, n_valid);
  else
    column_moments_kernel<false><<<(unsigned)n_partials, 256, 0, stream>>>(a, nullptr, nullptr, nullptr, n_rows, cols, rows_per_block,
                                                                           partials, 0, nullptr, 1.f, count_inc, n_valid);
  EGC_LAUNCH_CHECK("column_moments_kernel");
  return EGC_OK;
}

constexpr int BN_FUSE_MAX_PARTIALS = 64;   // beyond: the finalize as its own launch (many blocks share the column totals)

int egc_bn_forward_stats_f32(const float* h, int64_t n_rows, int32_t cols, double* partials, int32_t n_partials,
                             int64_t* count_inc, const int64_t* n_valid, const float* gamma, const float* beta, double eps,
                             double* stats, float* affine, float* running_mean, float* running_var, double momentum,
                             const int64_t* n_tracked, int32_t* sync, egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (sync == nullptr || n_partials > BN_FUSE_MAX_PARTIALS) {
    const int st = egc_column_moments_f64(h, nullptr, nullptr, nullptr, 0, nullptr, 1.f, n_rows, cols, partials, n_partials, count_inc,
                                          n_valid, stream_);
    if (st != EGC_OK) return st;
    return egc_bn_forward_finalize(partials, n_partials, cols, n_rows, gamma, beta, eps, stats, affine, running_mean, running_var,
                                   momentum, n_tracked, n_valid, stream_);
  }
  if (n_rows <= 0 || cols <= 0 || partials == nullptr || n_partials <= 0 || h == nullptr || stats == nullptr || affine == nullptr)
    return EGC_ERR_INVALID;
  if ((cols & 3) != 0 || cols > 1024 || !aligned16(h) || (reinterpret_cast<uintptr_t>(partials) & 31) != 0) return EGC_ERR_UNSUPPORTED;
  if ((running_mean == nullptr) != (running_var == nullptr)) return EGC_ERR_INVALID;
  if (running_mean != nullptr && momentum < 0.0 && n_tracked == nullptr) return EGC_ERR_INVALID;
  if (running_mean != nullptr && n_rows < 2) return EGC_ERR_INVALID;
  FinArgs f = FinArgs();
  f.sync = sync; f.eps = eps; f.momentum = momentum; f.gamma = gamma; f.beta = beta; f.stats = stats; f.affine = affine;
  f.running_mean = running_mean; f.running_var = running_var; f.n_tracked = n_tracked;
  const int rows_per_block = (int)std::max<int64_t>(ceil_div(n_rows, (int64_t)n_partials), 1);
  column_moments_kernel<false, 1><<<(unsigned)n_partials, 256, 0, stream>>>(h, nullptr, nullptr, nullptr, n_rows, cols, rows_per_block,
                                                                              partials, 0, nullptr, 1.f, count_inc, n_valid, f);
  EGC_LAUNCH_CHECK("column_moments_kernel (with the forward finalize)");
  return EGC_OK;
}

static int bn_backward_finalize_impl(const double* partials, int32_t n_partials, int32_t cols, int64_t n_rows, const double* stats,
                                     const float* gamma, float* out5, const int64_t* n_valid, float* dh_sums, hipStream_t stream) {
  if (partials == nullptr || n_partials <= 0 || cols <= 0 || n_rows <= 0 || stats == nullptr || out5 == nullptr)
    return EGC_ERR_INVALID;
  bn_backward_finalize_kernel<<<(unsigned)ceil_div(cols, FIN_COLS), FIN_COLS * FIN_LANES, 0, stream>>>(partials, n_partials, cols, (double)n_rows, stats,
                                                                              gamma, out5, n_valid, dh_sums);
  EGC_LAUNCH_CHECK("bn_backward_finalize_kernel");
  return EGC_OK;
}

int egc_bn_backward_stats_f32(const float* dout, const float* h, const float* scale, const float* shift, int32_t relu,
                              const uint8_t* keep, float keep_scale, int64_t n_rows, int32_t cols, double* partials,
                              int32_t n_partials, const int64_t* n_valid, const double* stats, const float* gamma, float* out5,
                              int32_t* sync, egc_stream_t stream_) {
  return egc_bn_backward_stats_sums_f32(dout, h, scale, shift, relu, keep, keep_scale, n_rows, cols, partials, n_partials, n_valid, stats,
                                        gamma, out5, nullptr, sync, stream_);
}

int egc_bn_backward_stats_sums_f32(const float* dout, const float* h, const float* scale, const float* shift, int32_t relu,
                                   const uint8_t* keep, float keep_scale, int64_t n_rows, int32_t cols, double* partials,
                                   int32_t n_partials, const int64_t* n_valid, const double* stats, const float* gamma, float* out5,
                                   float* dh_col_sums, int32_t* sync, egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (sync == nullptr || n_partials > BN_FUSE_MAX_PARTIALS) {
    const int st = egc_column_moments_f64(dout, h, scale, shift, relu, keep, keep_scale, n_rows, cols, partials, n_partials, nullptr,
                                          n_valid, stream_);
    if (st != EGC_OK) return st;
    return bn_backward_finalize_impl(partials, n_partials, cols, n_rows, stats, gamma, out5, n_valid, dh_col_sums, stream);
  }
  if (n_rows <= 0 || cols <= 0 || partials == nullptr || n_partials <= 0 || dout == nullptr || h == nullptr || stats == nullptr ||
      out5 == nullptr)
    return EGC_ERR_INVALID;
  if ((cols & 3) != 0 || cols > 1024 || !aligned16(dout) || !aligned16(h) || !aligned16(scale) || !aligned16(shift) ||
      (reinterpret_cast<uintptr_t>(partials) & 31) != 0)
    return EGC_ERR_UNSUPPORTED;
  if (relu && (scale == nullptr || shift == nullptr)) return EGC_ERR_INVALID;
  if (keep != nullptr && (reinterpret_cast<uintptr_t>(keep) & 3) != 0) return EGC_ERR_INVALID;
  FinArgs f = FinArgs();
  f.sync = sync; f.gamma = gamma; f.bstats = stats; f.outv = out5; f.dh_sums = dh_col_sums;
  const int rows_per_block = (int)std::max<int64_t>(ceil_div(n_rows, (int64_t)n_partials), 1);
  column_moments_kernel<true, 2><<<(unsigned)n_partials, 256, 0, stream>>>(dout, h, scale, shift, n_rows, cols, rows_per_block, partials,
                                                                             relu, keep, keep_scale, nullptr, n_valid, f);
  EGC_LAUNCH_CHECK("column_moments_kernel (with the backward finalize)");
  return EGC_OK;
}

int egc_bn_forward_finalize(const double* partials, int32_t n_partials, int32_t cols, int64_t n_rows, const float* gamma,
                            const float* beta, double eps, double* stats, float* affine, float* running_mean,
                            float* running_var, double momentum, const int64_t* n_tracked, const int64_t* n_valid,
                            egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (partials == nullptr || n_partials <= 0 || cols <= 0 || n_rows <= 0 || stats == nullptr || affine == nullptr)
    return EGC_ERR_INVALID;
  if ((running_mean == nullptr) != (running_var == nullptr)) return EGC_ERR_INVALID;
  if (running_mean != nullptr && momentum < 0.0 && n_tracked == nullptr) return EGC_ERR_INVALID;
  if (running_mean != nullptr && n_rows < 2) return EGC_ERR_INVALID;   // (nn.BatchNorm1d raises on one row in training mode)
  // (with a device-side row count the check is the caller's: a count below 2 leaves the variance term unscaled)
  bn_forward_finalize_kernel<<<(unsigned)ceil_div(cols, FIN_COLS), FIN_COLS * FIN_LANES, 0, stream>>>(partials, n_partials, cols, (double)n_rows, gamma,
                                                                             beta, eps, stats, affine, running_mean,
                                                                             running_var, momentum, n_tracked, n_valid);
  EGC_LAUNCH_CHECK("bn_forward_finalize_kernel");
  return EGC_OK;
}

int egc_bn_backward_finalize(const double* partials, int32_t n_partials, int32_t cols, int64_t n_rows, const double* stats,
                             const float* gamma, float* out5, const int64_t* n_valid, egc_stream_t stream_) {
  return bn_backward_finalize_impl(partials, n_partials, cols, n_rows, stats, gamma, out5, n_valid, nullptr, (hipStream_t)stream_);
}

int egc_affine_act_residual_f32(const float* h, const float* scale, const float* shift, const float* residual,
                                int32_t relu, const uint8_t* keep, float keep_scale, int64_t n_rows, int32_t cols,
                                float* out, const int64_t* n_valid, egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_rows < 0 || cols <= 0 || scale == nullptr || shift == nullptr) return EGC_ERR_INVALID;
  if ((cols & 3) != 0 || !aligned16(h) || !aligned16(scale) || !aligned16(shift) || !aligned16(residual) || !aligned16(out))
    return EGC_ERR_UNSUPPORTED;
  if (n_rows == 0) return EGC_OK;
  if (h == nullptr || out == nullptr) return EGC_ERR_INVALID;
  const int64_t quads = n_rows * (cols / 4);
  if ((reinterpret_cast<uintptr_t>(keep) & 3) != 0) return EGC_ERR_UNSUPPORTED;
  affine_act_residual_kernel<<<(unsigned)ceil_div(quads, 256), 256, 0, stream>>>(h, scale, shift, residual, relu, quads, cols / 4, out,
                                                                               keep, keep_scale, n_valid);
  EGC_LAUNCH_CHECK("affine_act_residual_kernel");
  return EGC_OK;
}

int egc_affine_act_backward_f32(const float* dout, const float* h, const float* scale, const float* shift, int32_t relu,
                                const uint8_t* keep, float keep_scale, const float* coef_g, const float* coef_h,
                                const float* coef_1, int64_t n_rows, int32_t cols, float* dh, const int64_t* n_valid,
                                egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_rows < 0 || cols <= 0 || scale == nullptr || shift == nullptr || coef_g == nullptr || coef_h == nullptr ||
      coef_1 == nullptr)
    return EGC_ERR_INVALID;
  if ((cols & 3) != 0 || !aligned16(dout) || !aligned16(h) || !aligned16(scale) || !aligned16(shift) || !aligned16(coef_g) ||
      !aligned16(coef_h) || !aligned16(coef_1) || !aligned16(dh))
    return EGC_ERR_UNSUPPORTED;
  if (n_rows == 0) return EGC_OK;
  if (dout == nullptr || h == nullptr || dh == nullptr) return EGC_ERR_INVALID;
  const int64_t quads = n_rows * (cols / 4);
  if ((reinterpret_cast<uintptr_t>(keep) & 3) != 0) return EGC_ERR_UNSUPPORTED;
  tail_backward_kernel<<<(unsigned)ceil_div(quads, 256), 256, 0, stream>>>(dout, h, scale, shift, relu, coef_g, coef_h, coef_1, quads,
                                                                         cols / 4, dh, keep, keep_scale, n_valid);
  EGC_LAUNCH_CHECK("tail_backward_kernel");
  return EGC_OK;
}

int egc_weights_pack_f32(const float* const* bases_parts, int32_t n_parts, const float* comb_weight, const float* comb_bias,
                         int32_t f_in, int32_t num_heads, int32_t num_aggrs, int32_t num_bases, int32_t basis_len,
                         int32_t basis_stride, int32_t permute_hab, float* wcat, float* bcat, int32_t grad,
                         egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (f_in <= 0 || num_heads <= 0 || num_aggrs <= 0 || num_bases <= 0 || basis_len <= 0 || basis_stride < basis_len ||
      bases_parts == nullptr || comb_weight == nullptr || wcat == nullptr || (n_parts != 1 && n_parts != num_bases))
    return EGC_ERR_INVALID;
  if (n_parts > PACK_MAX_PARTS) return EGC_ERR_UNSUPPORTED;
  PackPtrs ptrs;
  for (int i = 0; i < PACK_MAX_PARTS; ++i) ptrs.part[i] = i < n_parts ? const_cast<float*>(bases_parts[i]) : nullptr;
  for (int i = 0; i < n_parts; ++i)
    if (ptrs.part[i] == nullptr) return EGC_ERR_INVALID;
  const PackDims d{f_in, num_heads, num_aggrs, num_bases, basis_len, basis_stride, n_parts, permute_hab != 0};
  const int64_t total = (int64_t)f_in * (num_bases * basis_stride + num_heads * num_bases * num_aggrs);
  const unsigned grid = (unsigned)ceil_div(total, 256);
  float* cw = const_cast<float*>(comb_weight);
  float* cb = const_cast<float*>(comb_bias);
  if (grad) weights_pack_kernel<true><<<grid, 256, 0, stream>>>(ptrs, cw, cb, wcat, bcat, d);
  else weights_pack_kernel<false><<<grid, 256, 0, stream>>>(ptrs, cw, cb, wcat, bcat, d);
  EGC_LAUNCH_CHECK("weights_pack_kernel");
  return EGC_OK;
}

}  // extern "C"
