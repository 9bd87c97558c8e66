// Graph preparation for the EGC hot path on gfx950:
//   egc_coo_to_csr   -- PyG edge_index (COO, int64) -> CSR by destination, stable inside a row
//   egc_csr_prepare  -- degree statistics (deg^-1/2 for symnorm) + long-row work plan
//
// Reference behaviour replaced: the gather/scatter index handling of MessagePassing.propagate
// (experiments/layers.py:191-193, optimized_layers.py:191-193), ToSparseTensor's sort by
// (col*N+row) (experiments/utils.py:95-113) and the degree pass of gcn_norm
// (layers.py:173-178, optimized_layers.py:131-137).
#include <algorithm>
#include <cstring>
#include <string>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "egc_common.h"

namespace egc {

static thread_local std::string g_last_error;

void set_last_error(const char* what, hipError_t err) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(err);
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------

// keys[e] = dst[e] as u32; running max of every node id seen (src and dst) -> *max_index.
// One atomic per BLOCK (the launch caps the grid): thousands of same-address atomics cost more than the read.
__global__ void __launch_bounds__(256) coo_keys_kernel(const int64_t* __restrict__ src,
                                                       const int64_t* __restrict__ dst, int64_t n_edges,
                                                       uint32_t* __restrict__ keys, int32_t* __restrict__ max_index) {
  __shared__ int wave_max[4];
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int m = -1;
  for (; e < n_edges; e += (int64_t)gridDim.x * blockDim.x) {
    int s = (int)src[e], d = (int)dst[e];
    keys[e] = (uint32_t)d;
    m = max(m, max(s, d));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
    if (m >= 0) atomicMax(max_index, m);
  }
}

// col[p] = src[edge_id[p]]
__global__ void __launch_bounds__(256) gather_col_kernel(const int64_t* __restrict__ src,
                                                         const uint32_t* __restrict__ edge_id, int64_t n_edges,
                                                         int32_t* __restrict__ col) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; p < n_edges; p += (int64_t)gridDim.x * blockDim.x) col[p] = (int32_t)src[edge_id[p]];
}

// rowptr[i] = first position p with sorted_dst[p] >= i  (i in [0, n_nodes])
__global__ void __launch_bounds__(256) rowptr_kernel(const uint32_t* __restrict__ sorted_dst, int64_t n_edges,
                                                     int64_t n_nodes, int32_t* __restrict__ rowptr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_nodes) return;
  int64_t lo = 0, hi = n_edges;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if ((int64_t)sorted_dst[mid] < i) lo = mid + 1; else hi = mid;
  }
  rowptr[i] = (int32_t)lo;
}

// 16 lanes per row, 64 rows per block: non-self in-degree, deg^-1/2 arrays, long-row plan entries.  Long rows
// reserve their plan entries with LDS atomics inside the block and ONE pair of global atomics per block.
constexpr int PREP_ROWS = 64;
constexpr int PREP_HUGE = 2048;  // rows longer than this are counted by all 1024 threads of the block
__global__ void __launch_bounds__(16 * PREP_ROWS) prepare_kernel(int64_t n_nodes, const int32_t* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ col,
                                                                 float* __restrict__ dis_raw, float* __restrict__ dis_looped,
                                                                 int32_t* __restrict__ plan, int cap_long, int cap_chunks) {
  __shared__ int s_long, s_chunks, s_base_long, s_base_chunk;
  __shared__ int s_huge_n, s_huge_row[PREP_ROWS], s_huge_cnt[PREP_ROWS];  // hub rows: counted by the whole block
  const int sl = threadIdx.x & 15;
  const int64_t row = (int64_t)blockIdx.x * PREP_ROWS + (threadIdx.x >> 4);
  const bool live = row < n_nodes;
  if (threadIdx.x == 0) { s_long = 0; s_chunks = 0; s_huge_n = 0; }
  __syncthreads();
  const int start = live ? rowptr[row] : 0, end = live ? rowptr[row + 1] : 0;
  const int deg = end - start;
  if (dis_looped != nullptr) {
    if (deg > PREP_HUGE) {  // a hub row would keep its 16 lanes busy long after the rest of the grid has finished
      if (sl == 0) {
        const int h = atomicAdd(&s_huge_n, 1);
        s_huge_row[h] = (int)(threadIdx.x >> 4);
        s_huge_cnt[h] = 0;
      }
    } else {
      int nonself = 0;
      for (int p = start + sl; p < end; p += 16) nonself += (col[p] != (int)row);
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) nonself += __shfl_xor(nonself, off);
      if (live && sl == 0) dis_looped[row] = 1.0f / sqrtf((float)(nonself + 1));
    }
    __syncthreads();
    for (int h = 0; h < s_huge_n; ++h) {
      const int64_t hr = (int64_t)blockIdx.x * PREP_ROWS + s_huge_row[h];
      const int hs = rowptr[hr], he = rowptr[hr + 1];
      int nonself = 0;
      for (int p = hs + (int)threadIdx.x; p < he; p += 16 * PREP_ROWS) nonself += (col[p] != (int)hr);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) nonself += __shfl_xor(nonself, off);
      if ((threadIdx.x & 63) == 0 && nonself != 0) atomicAdd(&s_huge_cnt[h], nonself);
      __syncthreads();
      if (threadIdx.x == 0) dis_looped[hr] = 1.0f / sqrtf((float)(s_huge_cnt[h] + 1));
    }
  }
  if (live && sl == 0 && dis_raw != nullptr) dis_raw[row] = deg > 0 ? 1.0f / sqrtf((float)deg) : 0.0f;
  const bool is_long = live && deg > EGC_LONG_ROW_THRESHOLD;
  const int nch = is_long ? (deg + EGC_LONG_ROW_CHUNK - 1) / EGC_LONG_ROW_CHUNK : 0;
  int slot = 0, c0 = 0;
  if (is_long && sl == 0) {
    slot = atomicAdd(&s_long, 1);
    c0 = atomicAdd(&s_chunks, nch);
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_long > 0) {
    s_base_long = atomicAdd(&plan[0], s_long);
    s_base_chunk = atomicAdd(&plan[1], s_chunks);
  }
  __syncthreads();
  if (is_long) {
    slot = __shfl(slot, threadIdx.x & 48) + s_base_long;   // lane 0 of this 16-lane group
    c0 = __shfl(c0, threadIdx.x & 48) + s_base_chunk;
    int32_t* long_row = plan + 4;
    int32_t* long_chunk0 = long_row + cap_long;
    int32_t* chunk_slot = long_chunk0 + cap_long;
    int32_t* chunk_begin = chunk_slot + cap_chunks;
    if (slot < cap_long && c0 + nch <= cap_chunks) {  // always true by construction of the caps
      if (sl == 0) {
        long_row[slot] = (int32_t)row;
        long_chunk0[slot] = c0;
      }
      for (int k = sl; k < nch; k += 16) {
        chunk_slot[c0 + k] = slot;
        chunk_begin[c0 + k] = start + k * EGC_LONG_ROW_CHUNK;
      }
    }
  }
}

// out[p] = dis[col[p]] for the two deg^-1/2 tables (either may be absent)
__global__ void __launch_bounds__(256) edge_dis_kernel(int64_t n_edges, const int32_t* __restrict__ col,
                                                       const float* __restrict__ dis_raw, const float* __restrict__ dis_looped,
                                                       float* __restrict__ out_raw, float* __restrict__ out_looped) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; p < n_edges; p += (int64_t)gridDim.x * blockDim.x) {
    const int j = col[p];
    if (out_raw != nullptr) out_raw[p] = dis_raw[j];
    if (out_looped != nullptr) out_looped[p] = dis_looped[j];
  }
}

__global__ void plan_header_kernel(int32_t* plan, int cap_long, int cap_chunks) {
  plan[0] = 0;
  plan[1] = 0;
  plan[2] = cap_long;
  plan[3] = cap_chunks;
}

__global__ void init_scalar_kernel(int32_t* p, int32_t v) { *p = v; }

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static unsigned key_bits(int64_t n_nodes) {
  unsigned b = 1;
  while (((int64_t)1 << b) < n_nodes && b < 32) ++b;
  return b;
}

static hipError_t sort_temp_bytes(int64_t n_nodes, int64_t n_edges, size_t* bytes) {
  *bytes = 0;
  if (n_edges == 0) return hipSuccess;
  return rocprim::radix_sort_pairs(nullptr, *bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                   rocprim::counting_iterator<uint32_t>(0), (uint32_t*)nullptr,
                                   (size_t)n_edges, 0u, key_bits(n_nodes), (hipStream_t)0);
}

}  // namespace egc

using namespace egc;

extern "C" {

const char* egc_last_error(void) { return g_last_error.c_str(); }

const char* egc_version(void) { return "egc_hip 0.1.0 gfx950"; }

int64_t egc_plan_ints(int64_t n_nodes, int64_t n_edges) {
  if (n_nodes < 0 || n_edges < 0) return -1;
  PlanCaps c = plan_caps(n_nodes, n_edges);
  return 4 + 2 * c.cap_long + 2 * c.cap_chunks;
}

size_t egc_coo_to_csr_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
  if (n_nodes < 0 || n_edges < 0) return 0;
  size_t temp = 0;
  if (sort_temp_bytes(n_nodes, n_edges, &temp) != hipSuccess) return 0;
  // keys_in + keys_out + rocprim temp
  return 2 * align256((size_t)n_edges * sizeof(uint32_t)) + align256(temp) + 256;
}

int egc_coo_to_csr(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_nodes, int32_t* rowptr,
                   int32_t* col, int32_t* edge_id, int32_t* max_index, void* workspace, size_t workspace_bytes,
                   egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_nodes < 0 || n_edges < 0 || n_nodes >= ((int64_t)1 << 31) - 1 || n_edges >= ((int64_t)1 << 31) - 1)
    return EGC_ERR_INVALID;
  if (rowptr == nullptr || max_index == nullptr) return EGC_ERR_INVALID;
  if (n_edges > 0 && (src == nullptr || dst == nullptr || col == nullptr || edge_id == nullptr)) return EGC_ERR_INVALID;

  init_scalar_kernel<<<1, 1, 0, stream>>>(max_index, -1);
  EGC_LAUNCH_CHECK("init_scalar_kernel");
  if (n_edges == 0) {
    EGC_HIP_TRY(hipMemsetAsync(rowptr, 0, (size_t)(n_nodes + 1) * sizeof(int32_t), stream));
    return EGC_OK;
  }
  size_t temp = 0;
  EGC_HIP_TRY(sort_temp_bytes(n_nodes, n_edges, &temp));
  const size_t kbytes = align256((size_t)n_edges * sizeof(uint32_t));
  if (workspace == nullptr || workspace_bytes < 2 * kbytes + align256(temp)) return EGC_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  uint32_t* keys_in = (uint32_t*)ws;
  uint32_t* keys_out = (uint32_t*)(ws + kbytes);
  void* sort_temp = ws + 2 * kbytes;

  const int threads = 256;
  const int blocks = (int)std::min<int64_t>(ceil_div(n_edges, threads), 256 * 8);
  const int key_blocks = std::min(blocks, 512);
  coo_keys_kernel<<<key_blocks, threads, 0, stream>>>(src, dst, n_edges, keys_in, max_index);
  EGC_LAUNCH_CHECK("coo_keys_kernel");
  // Stable LSD radix sort of (dst, input position): the value array IS edge_id.
  EGC_HIP_TRY(rocprim::radix_sort_pairs(sort_temp, temp, (const uint32_t*)keys_in, keys_out,
                                        rocprim::counting_iterator<uint32_t>(0), (uint32_t*)edge_id,
                                        (size_t)n_edges, 0u, key_bits(n_nodes), stream));
  gather_col_kernel<<<blocks, threads, 0, stream>>>(src, (const uint32_t*)edge_id, n_edges, col);
  EGC_LAUNCH_CHECK("gather_col_kernel");
  rowptr_kernel<<<(int)ceil_div(n_nodes + 1, threads), threads, 0, stream>>>(keys_out, n_edges, n_nodes, rowptr);
  EGC_LAUNCH_CHECK("rowptr_kernel");
  return EGC_OK;
}

int egc_csr_edge_dis(int64_t n_edges, const int32_t* col, const float* dis_raw, const float* dis_looped,
                     float* edge_dis_raw, float* edge_dis_looped, egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_edges < 0) return EGC_ERR_INVALID;
  if (dis_raw == nullptr) edge_dis_raw = nullptr;
  if (dis_looped == nullptr) edge_dis_looped = nullptr;
  if (n_edges == 0 || (edge_dis_raw == nullptr && edge_dis_looped == nullptr)) return EGC_OK;
  if (col == nullptr) return EGC_ERR_INVALID;
  const int blocks = (int)std::min<int64_t>(ceil_div(n_edges, 256), 256 * 8);
  edge_dis_kernel<<<blocks, 256, 0, stream>>>(n_edges, col, dis_raw, dis_looped, edge_dis_raw, edge_dis_looped);
  EGC_LAUNCH_CHECK("edge_dis_kernel");
  return EGC_OK;
}

int egc_csr_prepare(int64_t n_nodes, int64_t n_edges, const int32_t* rowptr, const int32_t* col, float* dis_raw,
                    float* dis_looped, int32_t* plan, egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_nodes < 0 || n_edges < 0 || rowptr == nullptr || plan == nullptr) return EGC_ERR_INVALID;
  if (n_edges > 0 && col == nullptr) return EGC_ERR_INVALID;
  PlanCaps c = plan_caps(n_nodes, n_edges);
  plan_header_kernel<<<1, 1, 0, stream>>>(plan, (int)c.cap_long, (int)c.cap_chunks);
  EGC_LAUNCH_CHECK("plan_header_kernel");
  if (n_nodes == 0) return EGC_OK;
  prepare_kernel<<<(int)ceil_div(n_nodes, (int64_t)PREP_ROWS), 16 * PREP_ROWS, 0, stream>>>(
      n_nodes, rowptr, col, dis_raw, dis_looped, plan, (int)c.cap_long, (int)c.cap_chunks);
  EGC_LAUNCH_CHECK("prepare_kernel");
  return EGC_OK;
}

}  // extern "C"
