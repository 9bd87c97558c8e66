// The index map between a layer's parameters and its GEMM operand (egc_weights_pack_f32, egc_tail.hip), shared with the
// weight-gradient reduction (egc_gemm_xt.hip), which can write the parameters' gradients through it directly:
//   wcat [F_in][B Ls + W] = [the B basis matrices side by side, each padded from L to Ls columns | comb_weight^T],
//   bcat [W] = comb_bias   (W = H B A).  The bases come as ONE [F_in][B L] matrix (EGConv.bases_weight) or as B matrices
//   [F_in][L] (EfficientGraphConv.bases_weight.{0..B-1}); with `permute` the Linear's rows [h][a][b] (EGConv's
//   comb_weight, optimized_layers.py:195-202) become columns [h][b][a], otherwise the rows are [h][b][a] already.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace egc {

constexpr int PACK_MAX_PARTS = 32;
struct PackDims { int F_in, H, A, B, L, Ls, n_parts, permute; };
struct PackPtrs { float* part[PACK_MAX_PARTS]; };

// row of the combination Linear (weight row / bias element) behind operand column j of the weightings block
__device__ inline int pack_comb_row(const PackDims& d, int j) {
  if (!d.permute) return j;
  const int h = j / (d.B * d.A), r = j - h * d.B * d.A, b = r / d.A, a = r - b * d.A;
  return (h * d.A + a) * d.B + b;
}

// the parameter element behind wcat[k][c], or nullptr for the padding columns of a padded basis
__device__ inline float* pack_param_ptr(const PackPtrs& bases, float* comb_w, const PackDims& d, int k, int c) {
  const int F_g = d.B * d.Ls;
  if (c < F_g) {
    const int b = c / d.Ls, l = c - b * d.Ls;
    if (l >= d.L) return nullptr;
    return d.n_parts == 1 ? bases.part[0] + (int64_t)k * d.B * d.L + b * d.L + l : bases.part[b] + (int64_t)k * d.L + l;
  }
  return comb_w + (int64_t)pack_comb_row(d, c - F_g) * d.F_in + k;
}

}  // namespace egc
