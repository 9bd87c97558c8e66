// Basis transform + weightings Linear as one fp32-in / fp32-acc MFMA GEMM on gfx950:
//     [bases | weightings] = x[N,F_in] @ [bases_weight | comb.weight^T]  (+ comb.bias)
//
// Reference behaviour replaced: torch.matmul(x, bases_weight) (experiments/layers.py:97-101,
// optimized_layers.py:180) and comb_weights(x) (layers.py:110, optimized_layers.py:182).
// v_mfma_f32_32x32x2_f32 is exact fp32 (an fmaf chain in k order), so the result stays within
// GEMM-reordering distance of the reference's MKL/rocBLAS fp32 result.
//
// Tiling: 128 x 64 block tile, 4 wavefronts stacked along M, each owning a 32 x 64 strip as two
// 32x32 accumulators; K walked in steps of 32 through LDS.  The output column space is "virtual":
//   [0, F_g)        -> bases[:, v]                 (source column v of wcat)
//   [F_g, ldb)      -> bases pad columns, written as 0 (keeps the gather kernel's float4 rows clean)
//   [ldb, ldb+W)    -> weightings[:, v-ldb] + bcat  (source column v-ldb+F_g of wcat)
#include "egc_common.h"

namespace egc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BN = 64;
constexpr int KT = 32;

template <bool A_VEC4>
__global__ void __launch_bounds__(256) basis_gemm_kernel(const float* __restrict__ x, const float* __restrict__ wcat,
                                                         const float* __restrict__ bcat, int64_t M, int K, int F_g,
                                                         int W, float* __restrict__ bases, int ldb,
                                                         float* __restrict__ weightings) {
  __shared__ float As[KT][BM + 1];  // k-major; +1 keeps the transposing stores conflict-free
  __shared__ float Bs[KT][BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int v0 = blockIdx.y * BN;
  const int wc = F_g + W;   // row stride of wcat
  const int vn = ldb + W;   // width of the virtual column space

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

  // B-tile source mapping is k-invariant: precompute per thread (8 elements: k = i*4 + tid/64 ... )
  const int bn = tid & 63;          // column inside the tile
  const int bk = tid >> 6;          // first k row handled (rows bk, bk+4, ..., bk+28)
  const int v = v0 + bn;
  const int src_col = (v < F_g) ? v : ((v < ldb || v >= vn) ? -1 : v - ldb + F_g);

  for (int k0 = 0; k0 < K; k0 += KT) {
    if (A_VEC4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = (tid >> 3) + 32 * i;
        const int k4 = (tid & 7) * 4;
        const int64_t gm = m0 + m;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gm < M && k0 + k4 < K) val = *reinterpret_cast<const float4*>(x + gm * K + k0 + k4);
        As[k4 + 0][m] = val.x;
        As[k4 + 1][m] = val.y;
        As[k4 + 2][m] = val.z;
        As[k4 + 3][m] = val.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int idx = tid + 256 * i;
        const int m = idx >> 5;
        const int k = idx & 31;
        const int64_t gm = m0 + m;
        As[k][m] = (gm < M && k0 + k < K) ? x[gm * K + k0 + k] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = bk + 4 * i;
      const int gk = k0 + k;
      Bs[k][bn] = (src_col >= 0 && gk < K) ? wcat[(int64_t)gk * wc + src_col] : 0.f;
    }
    __syncthreads();
    const int kh = lane >> 5;
    const int ml = 32 * wave + (lane & 31);
    const int nl = lane & 31;
#pragma unroll
    for (int kk = 0; kk < KT / 2; ++kk) {
      const float a = As[2 * kk + kh][ml];
      const float b0 = Bs[2 * kk + kh][nl];
      const float b1 = Bs[2 * kk + kh][32 + nl];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
    }
    __syncthreads();
  }

  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int vc = v0 + 32 * half + (lane & 31);
    if (vc >= vn) continue;
    const bool to_bases = vc < ldb;
    const float badd = (!to_bases && bcat != nullptr) ? bcat[vc - ldb] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int64_t gm = m0 + 32 * wave + row;
      if (gm >= M) continue;
      const float val = half == 0 ? acc0[r] : acc1[r];
      if (to_bases) bases[gm * ldb + vc] = val;
      else weightings[gm * (int64_t)W + (vc - ldb)] = val + badd;
    }
  }
}

}  // namespace egc

using namespace egc;

extern "C" int egc_basis_transform_f32(const float* x, const float* wcat, const float* bcat, int64_t n_nodes,
                                       int32_t f_in, int32_t f_g, int32_t w_cols, float* bases, int32_t ldb,
                                       float* weightings, egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_nodes < 0 || f_in <= 0 || f_g <= 0 || w_cols < 0 || ldb < f_g) return EGC_ERR_INVALID;
  if (n_nodes == 0) return EGC_OK;
  if (x == nullptr || wcat == nullptr || bases == nullptr || (w_cols > 0 && weightings == nullptr)) return EGC_ERR_INVALID;
  const int64_t mblocks = ceil_div(n_nodes, BM);
  const int nblocks = (int)ceil_div(ldb + w_cols, BN);
  if (mblocks >= ((int64_t)1 << 31)) return EGC_ERR_INVALID;
  dim3 grid((unsigned)mblocks, (unsigned)nblocks);
  const bool vec4 = (f_in % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (vec4)
    basis_gemm_kernel<true><<<grid, 256, 0, stream>>>(x, wcat, bcat, n_nodes, f_in, f_g, w_cols, bases, ldb, weightings);
  else
    basis_gemm_kernel<false><<<grid, 256, 0, stream>>>(x, wcat, bcat, n_nodes, f_in, f_g, w_cols, bases, ldb, weightings);
  EGC_LAUNCH_CHECK("basis_gemm_kernel");
  return EGC_OK;
}
