// Shared between the split-precision GEMM translation units (egc_gemm_bf16x3.hip, egc_gemm_f16x2.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace egc {

constexpr int GEMM_KT = 32;  // k per packed staging step

// Shapes served by the fp16x2 register-stationary kernel: everything else uses the bf16x3 planes.
// (a wavefront's 32 columns lie in `bases` or in `weightings`: ldb % 32 == 0 -- or there are no weightings at all, as in the d x
// GEMM of the 168- and 184-wide nets, [d bases | d weightings] (128 columns) x wcat^T -> 168 / 184: round 6, 117.7 -> ~60 us at
// CIFAR b2048 against the three-plane kernel those shapes took before)
inline bool f16x2_shape(int f_in, int ldb, int NV, int w_cols) {
  return f_in > 96 && f_in <= 128 && f_in % 4 == 0 && NV == 192 && (ldb % 32 == 0 || w_cols == 0);
}

size_t f16x2_pack_bytes(int KS, int NV);
// (rs, cs): floats between consecutive k / consecutive columns of the source: (f_g + w_cols, 1) for wcat [f_in][f_g + w_cols],
// (1, ld) for its transpose stored [f_g + w_cols][ld]
int f16x2_pack(const float* wcat, int64_t rs, int64_t cs, int f_in, int f_g, int w_cols, int ldb, int NV, int KS, void* packed,
               hipStream_t stream);
int f16x2_launch(const float* x, const void* packed, const float* bcat, int64_t M, int K, int W, float* bases, int ldb,
                 float* weightings, int NV, hipStream_t stream);

// Shapes served by the long-k fp16x2 kernel (egc_gemm_f16x2k.hip): 128 < F_in <= 384, at most 16 column tiles of 16.
bool f16x2k_shape(int f_in, int f_g, int ldb, int w_cols);
size_t f16x2k_pack_bytes(int f_in, int f_g, int ldb, int w_cols);
int f16x2k_pack(const float* wcat, int64_t rs, int64_t cs, int f_in, int f_g, int ldb, int w_cols, void* packed,
                hipStream_t stream);
// addend (or nullptr): [M][ldb] added to the bases columns in the store
int f16x2k_launch(const float* x, const void* packed, const float* bcat, int64_t M, int K, int f_g, int ldb, int W,
                  float* bases, float* weightings, hipStream_t stream, const float* addend = nullptr);

}  // namespace egc
