// Shared host/device helpers for libegc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "egc_hip.h"

#define EGC_WAVE 64

namespace egc {

// Records the message returned by egc_last_error().
void set_last_error(const char* what, hipError_t err);

#define EGC_HIP_TRY(expr)                          \
  do {                                             \
    hipError_t _e = (expr);                        \
    if (_e != hipSuccess) {                        \
      ::egc::set_last_error(#expr, _e);            \
      return EGC_ERR_HIP;                          \
    }                                              \
  } while (0)

#define EGC_LAUNCH_CHECK(name)                     \
  do {                                             \
    hipError_t _e = hipGetLastError();             \
    if (_e != hipSuccess) {                        \
      ::egc::set_last_error(name, _e);             \
      return EGC_ERR_HIP;                          \
    }                                              \
  } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// floats between consecutive bases of a `bases` row (egc_layer.basis_stride; 0 = contiguous)
static inline int layer_basis_stride(const egc_layer* L) {
  const int len = L->out_channels / L->num_heads;
  return L->basis_stride > len ? L->basis_stride : len;
}

static inline bool layer_uses_symnorm(const egc_layer* L) {
  for (int t = 0; t < L->num_aggrs; ++t)
    if (L->aggrs[t] == EGC_AGGR_SYMNORM) return true;
  return false;
}

// Long-row plan layout (int32 words), shared by egc_csr_prepare and the aggregate kernels:
//   [0] n_long   [1] n_chunks   [2] cap_long   [3] cap_chunks
//   [4 .. 4+cap_long)                 long_row[s]      row id of long-row slot s
//   [.. +cap_long)                    long_chunk0[s]   first chunk slot of that row
//   [.. +cap_chunks)                  chunk_slot[c]    long-row slot the chunk belongs to
//   [.. +cap_chunks)                  chunk_begin[c]   first CSR entry of the chunk
struct PlanCaps {
  int64_t cap_long;
  int64_t cap_chunks;
};
static inline PlanCaps plan_caps(int64_t n_nodes, int64_t n_edges) {
  PlanCaps c;
  c.cap_long = n_edges / (EGC_LONG_ROW_THRESHOLD + 1) + 1;
  if (c.cap_long > n_nodes + 1) c.cap_long = n_nodes + 1;
  c.cap_chunks = n_edges / EGC_LONG_ROW_CHUNK + c.cap_long;
  return c;
}

// egc_backward.hip: CSR positions of the first entries attaining each row's max / min (training forward)
int arg_extrema(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb, const float* stats,
                const int32_t* cnt, int32_t* arg_max, int32_t* arg_min, unsigned* arg8_max, unsigned* arg8_min,
                hipStream_t stream);

}  // namespace egc
