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
// With `status` (range-checked form): an edge whose source is outside [0, n_src) or whose destination is outside
// [0, n_nodes) gets the key n_nodes -- it sorts behind every row, rowptr[n_nodes] then counts the edges kept -- and the
// flags are raised (what the reference's PyG path answers with an exception at index_select, optimized_layers.py:191-193).
__global__ void __launch_bounds__(256) coo_keys_kernel(const int64_t* __restrict__ src,
                                                       const int64_t* __restrict__ dst, int64_t n_edges,
                                                       uint32_t* __restrict__ keys, int32_t* __restrict__ max_index,
                                                       int64_t n_nodes, int64_t n_src, int32_t* __restrict__ status,
                                                       int32_t* __restrict__ host_flag) {
  __shared__ int wave_max[4];
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int m = -1;
  bool bad = false;
  for (; e < n_edges; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s64 = src[e], d64 = dst[e];
    if (status != nullptr && (s64 < 0 || s64 >= n_src || d64 < 0 || d64 >= n_nodes)) {
      keys[e] = (uint32_t)n_nodes;
      bad = true;
      continue;
    }
    int s = (int)s64, d = (int)d64;
    keys[e] = (uint32_t)d;
    m = max(m, max(s, d));
  }
  if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) {
    atomicOr(status, 1);
    if (host_flag != nullptr) *(volatile int32_t*)host_flag = 1;
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
// col[p] = src[edge_id[p]]; positions behind the kept entries (dropped edges of the range-checked form) get a harmless 0
__global__ void __launch_bounds__(256) gather_col_kernel(const int64_t* __restrict__ src,
                                                         const uint32_t* __restrict__ edge_id, int64_t n_edges,
                                                         int32_t* __restrict__ col, const int32_t* __restrict__ n_kept) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t kept = n_kept != nullptr ? (int64_t)*n_kept : n_edges;
  for (; p < n_edges; p += (int64_t)gridDim.x * blockDim.x) col[p] = p < kept ? (int32_t)src[edge_id[p]] : 0;
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

// The CSR back as a COO list with the roles swapped -- entry p of row i (source j = col[p]) becomes the edge
// (source i, destination j) -- i.e. the input of the TRANSPOSED graph's build (the backward's source-side pass).
// One launch (a binary search of the row pointers per entry) instead of the half-dozen torch kernels of
// diff / repeat_interleave / cast / stack, and no read-back of the entry count.
__global__ void __launch_bounds__(256) csr_transposed_coo_kernel(int n_nodes, int64_t n_edges, const int32_t* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ col, int64_t* __restrict__ out_src,
                                                                 int64_t* __restrict__ out_dst) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t kept = rowptr[n_nodes];        // < n_edges when a range-checked build dropped edges
  for (; p < n_edges; p += (int64_t)gridDim.x * blockDim.x) {
    if (p >= kept) {                           // no entry here: an id the transposed graph's build drops again
      out_src[p] = -1;
      out_dst[p] = -1;
      continue;
    }
    int lo = 0, hi = n_nodes;                  // largest row with rowptr[row] <= p
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)rowptr[mid] <= p) lo = mid; else hi = mid;
    }
    out_src[p] = lo;
    out_dst[p] = col[p];
  }
}


// =============================================================================================
// Fast graph build (egc_graph_build): COO -> stable CSR + degree tables + long-row plan + per-entry deg^-1/2 in
// FIVE launches, no library sort -- per-batch graphs of the reference's batched nets (zinc/models.py:60-74,
// mol/pna_style_models.py:64-79, cifar/models.py:61-75) pay this conversion every step, and at ~100 k edges the
// eight-plus launches of the radix-sort pipeline cost more than the layer they feed.
//   1 hist     in-degree (and non-self in-degree) per destination; range check of every id.  A tile of 2048 edges whose
//              destinations span <= 2048 rows (graph-contiguous batches) is counted in LDS and reaches memory as one
//              atomic per distinct row on consecutive addresses
//   2 sums     per-block sums of the degrees
//   3 scan     exclusive scan -> rowptr (every block re-adds the few block sums in front of it); deg^-1/2 tables
//   4 scatter  entry position = rowptr[dst] + a slot drawn from the row's counter (any order inside a row; per tile one
//              reservation per distinct row, slots inside the run from an LDS cursor)
//   5 rows     every row sorted by input position (the order torch_scatter's first-edge arg rule needs): 16 lanes per
//              short row, the whole block on long rows (LDS bitonic up to 4096 entries, in global memory beyond);
//              long-row plan; per-entry deg^-1/2
// Workspace: int32 deg[n + 1] | deg_ns[n] | block sums | counters -- zero on entry, left zero on exit (the scatter
// counts the degrees back down, the scan clears the rest).
// =============================================================================================
constexpr int BUILD_SCAN_ITEMS = 4096;   // elements per scan block (1024 threads x 4)
constexpr int BUILD_LDS_SORT = 4096;     // longest row sorted in LDS
constexpr int BUILD_WAVE_ROW = BUILD_LDS_SORT / 16;  // longest row sorted by ONE wavefront (16 per workgroup)
constexpr int BUILD_RANK_ROW = 1024;     // longest row rank-sorted by the workgroup (one entry per thread)

constexpr int BUILD_TILE = 2048;   // edges per workgroup of the histogram / scatter kernels (256 threads x 8)
constexpr int BUILD_WIN = 2048;    // destination window a tile may span to be counted in LDS

// Batches of small graphs are graph-contiguous: the 2048 edges of a tile point into a window of a few hundred
// destinations.  The tile is then counted in LDS and only the distinct rows go to memory, as atomics on CONSECUTIVE
// addresses (scattered atomics run at ~20 M/s per stream on MI355X, consecutive ones at the memory rate); a tile whose
// destinations span more than BUILD_WIN rows falls back to one atomic per edge.
struct TileEdges {
  int s[8], d[8];      // -1 = absent or out of range
  int dmin, dmax;      // window of the valid destinations (dmin > dmax: no valid edge)
  bool bad;
};

__device__ inline TileEdges load_tile(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t n_edges,
                                      int n_nodes, int n_src, int* s_min, int* s_max) {
  TileEdges t;
  t.bad = false;
  int lo = 0x7fffffff, hi = -1;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t e = (int64_t)blockIdx.x * BUILD_TILE + k * 256 + threadIdx.x;
    t.s[k] = t.d[k] = -1;
    if (e < n_edges) {
      const int64_t s = src[e], d = dst[e];
      if (s < 0 || s >= n_src || d < 0 || d >= n_nodes) { t.bad = true; continue; }   // dropped; reported through *status
      t.s[k] = (int)s;
      t.d[k] = (int)d;
      lo = min(lo, (int)d);
      hi = max(hi, (int)d);
    }
  }
  if (threadIdx.x == 0) { *s_min = 0x7fffffff; *s_max = -1; }
  __syncthreads();
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off)); }
  if ((threadIdx.x & 63) == 0) { atomicMin(s_min, lo); atomicMax(s_max, hi); }
  __syncthreads();
  t.dmin = *s_min;
  t.dmax = *s_max;
  return t;
}

__global__ void __launch_bounds__(256) build_hist_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                                         int64_t n_edges, int n_nodes, int n_src, int* __restrict__ deg,
                                                         int* __restrict__ deg_ns, int* __restrict__ maxp1,
                                                         int* __restrict__ status, int32_t* __restrict__ plan,
                                                         int cap_long, int cap_chunks, int32_t* __restrict__ host_flag) {
  __shared__ int s_cnt[BUILD_WIN], s_ns[BUILD_WIN];
  __shared__ int s_min, s_max, s_top;
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // header of the long-row plan: the scan kernel registers the long rows
    plan[0] = 0; plan[1] = 0; plan[2] = cap_long; plan[3] = cap_chunks;
  }
  const TileEdges t = load_tile(src, dst, n_edges, n_nodes, n_src, &s_min, &s_max);
  if (__ballot(t.bad) != 0 && (threadIdx.x & 63) == 0) {
    atomicOr(status, 1);
    if (host_flag != nullptr) *(volatile int32_t*)host_flag = 1;   // sticky, host-visible: read without a synchronisation
  }
  if (t.dmin > t.dmax) return;
  int m = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (t.d[k] >= 0) m = max(m, max(t.s[k], t.d[k]) + 1);
  if (threadIdx.x == 0) s_top = 0;
  const int span = t.dmax - t.dmin + 1;
  if (span <= BUILD_WIN) {
    for (int w = threadIdx.x; w < span; w += 256) { s_cnt[w] = 0; s_ns[w] = 0; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (t.d[k] >= 0) {
        atomicAdd(&s_cnt[t.d[k] - t.dmin], 1);
        if (t.s[k] != t.d[k]) atomicAdd(&s_ns[t.d[k] - t.dmin], 1);
      }
    __syncthreads();
    for (int w = threadIdx.x; w < span; w += 256) {
      const int c = s_cnt[w], cn = s_ns[w];
      if (c) atomicAdd(&deg[t.dmin + w], c);
      if (cn) atomicAdd(&deg_ns[t.dmin + w], cn);
    }
  } else {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (t.d[k] >= 0) {
        atomicAdd(&deg[t.d[k]], 1);
        if (t.s[k] != t.d[k]) atomicAdd(&deg_ns[t.d[k]], 1);
      }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) atomicMax(&s_top, m);
  __syncthreads();
  if (threadIdx.x == 0 && s_top > 0) atomicMax(maxp1, s_top);
}

__global__ void __launch_bounds__(1024) build_sums_kernel(const int* __restrict__ deg, int n_nodes, int* __restrict__ bsum) {
  __shared__ int ws[16];
  const int base = blockIdx.x * BUILD_SCAN_ITEMS + threadIdx.x * 4;
  int v = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) v += (base + k < n_nodes) ? deg[base + k] : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int k = 0; k < 16; ++k) t += ws[k];
    bsum[blockIdx.x] = t;
  }
}

__global__ void __launch_bounds__(1024) build_scan_kernel(const int* __restrict__ deg, int* __restrict__ deg_ns, int n_nodes,
                                                          int* __restrict__ bsum, int n_blocks, int* __restrict__ rowptr,
                                                          float* __restrict__ dis_raw, float* __restrict__ dis_looped,
                                                          int* __restrict__ maxp1, int32_t* __restrict__ max_index,
                                                          int32_t* __restrict__ plan, int cap_long, int cap_chunks) {
  __shared__ int ws[16];
  __shared__ int s_prefix;
  // prefix of the blocks in front of this one: a few hundred values at most, re-added by every block
  int p = 0;
  for (int k = threadIdx.x; k < (int)blockIdx.x; k += 1024) p += bsum[k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = p;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int k = 0; k < 16; ++k) t += ws[k];
    s_prefix = t;
  }
  __syncthreads();
  const int base = blockIdx.x * BUILD_SCAN_ITEMS + threadIdx.x * 4;
  int d[4], dn[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool ok = base + k < n_nodes;
    d[k] = ok ? deg[base + k] : 0;
    dn[k] = ok ? deg_ns[base + k] : 0;
  }
  int mine = d[0] + d[1] + d[2] + d[3];
  int incl = mine;  // inclusive scan over the wavefront, then over the 16 wavefronts
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if ((int)(threadIdx.x & 63) >= off) incl += t;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = incl;
  __syncthreads();
  int wave_off = 0;
  for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) wave_off += ws[k];
  int run = s_prefix + wave_off + incl - mine;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < n_nodes) {
      rowptr[base + k] = run;
      if (dis_raw != nullptr) dis_raw[base + k] = d[k] > 0 ? 1.0f / sqrtf((float)d[k]) : 0.0f;
      if (dis_looped != nullptr) dis_looped[base + k] = 1.0f / sqrtf((float)(dn[k] + 1));
      deg_ns[base + k] = 0;   // workspace left zero
      if (d[k] > EGC_LONG_ROW_THRESHOLD) {  // long row: a slot of the plan and a run of chunk slots (entries: rows kernel)
        const int slot = atomicAdd(&plan[0], 1);
        const int c0 = atomicAdd(&plan[1], (d[k] + EGC_LONG_ROW_CHUNK - 1) / EGC_LONG_ROW_CHUNK);
        if (slot < cap_long) { plan[4 + slot] = base + k; plan[4 + cap_long + slot] = c0; }
      }
    }
    run += d[k];
  }
  if (base <= n_nodes && n_nodes < base + 4) rowptr[n_nodes] = run - 0;   // (run has passed every element < n_nodes of this thread)
  __syncthreads();
  if (blockIdx.x == (unsigned)n_blocks - 1 && threadIdx.x == 0) {
    *max_index = *maxp1 - 1;
    *maxp1 = 0;
  }
}

__global__ void __launch_bounds__(256) build_scatter_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                                            int64_t n_edges, int n_nodes, int n_src, int* __restrict__ deg,
                                                            const int* __restrict__ rowptr, int* __restrict__ col,
                                                            int* __restrict__ edge_id, int* __restrict__ bsum, int n_blocks) {
  __shared__ int s_cnt[BUILD_WIN], s_base[BUILD_WIN];
  __shared__ int s_min, s_max;
  if (blockIdx.x == 0)
    for (int k = threadIdx.x; k < n_blocks; k += blockDim.x) bsum[k] = 0;
  const TileEdges t = load_tile(src, dst, n_edges, n_nodes, n_src, &s_min, &s_max);
  if (t.dmin > t.dmax) return;
  const int span = t.dmax - t.dmin + 1;
  if (span <= BUILD_WIN) {
    // the tile reserves a run of slots per distinct row with ONE atomic (the slots count the degree back down: the
    // counters end at zero), the edges of the tile take their slot inside the run from an LDS cursor
    for (int w = threadIdx.x; w < span; w += 256) s_cnt[w] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (t.d[k] >= 0) atomicAdd(&s_cnt[t.d[k] - t.dmin], 1);
    __syncthreads();
    for (int w = threadIdx.x; w < span; w += 256) {
      const int c = s_cnt[w];
      if (c) s_base[w] = rowptr[t.dmin + w] + atomicSub(&deg[t.dmin + w], c) - c;
      s_cnt[w] = 0;   // becomes the cursor
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (t.d[k] >= 0) {
        const int w = t.d[k] - t.dmin;
        const int pos = s_base[w] + atomicAdd(&s_cnt[w], 1);
        col[pos] = t.s[k];
        edge_id[pos] = (int)((int64_t)blockIdx.x * BUILD_TILE + k * 256 + threadIdx.x);
      }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (t.d[k] >= 0) {
        const int pos = rowptr[t.d[k]] + atomicSub(&deg[t.d[k]], 1) - 1;
        col[pos] = t.s[k];
        edge_id[pos] = (int)((int64_t)blockIdx.x * BUILD_TILE + k * 256 + threadIdx.x);
      }
  }
}

// Bitonic sort of np2 (a power of two) (key, value) pairs by one workgroup.  Comparator direction follows the GLOBAL
// index, so the same routine sorts a whole array (base = 0, np2 = its length) or finishes the small-stride steps of a
// bigger network inside one LDS-resident chunk (base = the chunk's first index).
__device__ inline void bitonic_steps(int2* kv, int n, int base, int k, int j_first) {
  for (int j = j_first; j > 0; j >>= 1) {
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
      const int l = t ^ j;
      if (l > t) {
        const int2 a = kv[t], b = kv[l];
        const bool up = ((base + t) & k) == 0;
        if ((a.x > b.x) == up) { kv[t] = b; kv[l] = a; }
      }
    }
    __syncthreads();
  }
}

// Rows longer than the LDS buffer: chunks of BUILD_LDS_SORT are sorted in LDS, the strides >= BUILD_LDS_SORT of the
// remaining merge stages run in global memory (one pass each), everything below again chunk by chunk in LDS.
__device__ inline void bitonic_big(int2* g, int np2, int2* lds) {
  constexpr int C = BUILD_LDS_SORT;
  for (int c0 = 0; c0 < np2; c0 += C) {
    for (int t = threadIdx.x; t < C; t += blockDim.x) lds[t] = g[c0 + t];
    __syncthreads();
    for (int k = 2; k <= C; k <<= 1) bitonic_steps(lds, C, c0, k, k >> 1);
    for (int t = threadIdx.x; t < C; t += blockDim.x) g[c0 + t] = lds[t];
    __syncthreads();
  }
  for (int k = 2 * C; k <= np2; k <<= 1) {
    for (int j = k >> 1; j >= C; j >>= 1) {
      for (int i = threadIdx.x; i < np2; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const int2 a = g[i], b = g[l];
          const bool up = (i & k) == 0;
          if ((a.x > b.x) == up) { g[i] = b; g[l] = a; }
        }
      }
      __syncthreads();
    }
    for (int c0 = 0; c0 < np2; c0 += C) {
      for (int t = threadIdx.x; t < C; t += blockDim.x) lds[t] = g[c0 + t];
      __syncthreads();
      bitonic_steps(lds, C, c0, k, C >> 1);
      for (int t = threadIdx.x; t < C; t += blockDim.x) g[c0 + t] = lds[t];
      __syncthreads();
    }
  }
}

// Rows of BUILD_LDS_SORT < d <= BUILD_MERGE_ROW entries, by one workgroup: chunks of BUILD_LDS_SORT sorted in LDS (no
// padding of the whole row to a power of two), then every entry's final place = its index in its own chunk + its
// lower bounds in the other chunks, found by binary search with ONE chunk's keys resident in LDS at a time (keys are
// input positions: unique).  g: scratch of 2 d int2 (sorted chunks, then d ranks).  Quadratic in the chunk count,
// hence the cap; longer rows take the bitonic network.
constexpr int BUILD_MERGE_ROW = 65536;
__device__ inline void merge_sort_big(int2* g, int d, int2* lds, int* edge_id_row, int* col_row, const float* dis_raw,
                                      const float* dis_looped, float* edis_raw_row, float* edis_looped_row) {
  constexpr int C = BUILD_LDS_SORT;
  int* rank = reinterpret_cast<int*>(g + d);
  int* keys = reinterpret_cast<int*>(lds);
  for (int c0 = 0; c0 < d; c0 += C) {
    const int len = min(C, d - c0);
    int np2 = 1;
    while (np2 < len) np2 <<= 1;
    for (int t = threadIdx.x; t < np2; t += blockDim.x) lds[t] = t < len ? int2{edge_id_row[c0 + t], col_row[c0 + t]} : int2{0x7fffffff, 0};
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1) bitonic_steps(lds, np2, 0, k, k >> 1);
    for (int t = threadIdx.x; t < len; t += blockDim.x) { g[c0 + t] = lds[t]; rank[c0 + t] = t; }
    __syncthreads();
  }
  for (int c0 = 0; c0 < d; c0 += C) {
    const int len = min(C, d - c0);
    for (int t = threadIdx.x; t < len; t += blockDim.x) keys[t] = g[c0 + t].x;
    __syncthreads();
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
      if (i >= c0 && i < c0 + len) continue;
      const int key = g[i].x;
      int lo = 0, hi = len;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
      }
      rank[i] += lo;
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    const int2 v = g[i];
    const int p = rank[i];
    edge_id_row[p] = v.x;
    col_row[p] = v.y;
    if (edis_raw_row != nullptr) edis_raw_row[p] = dis_raw[v.y];
    if (edis_looped_row != nullptr) edis_looped_row[p] = dis_looped[v.y];
  }
}

constexpr int ROWS_PER_BLOCK = 64;   // lane groups (of 16) per workgroup
constexpr int ROWS_PER_GROUP = 2;    // rows per lane group of the short-row pass
__global__ void __launch_bounds__(16 * ROWS_PER_BLOCK) build_rows_kernel(int n_nodes, int64_t n_edges, const int* __restrict__ rowptr,
                                                                         int* __restrict__ col, int* __restrict__ edge_id,
                                                                         const float* __restrict__ dis_raw,
                                                                         const float* __restrict__ dis_looped,
                                                                         float* __restrict__ edis_raw, float* __restrict__ edis_looped,
                                                                         int32_t* __restrict__ plan, int cap_long, int cap_chunks,
                                                                         int2* __restrict__ big_scratch, int* __restrict__ ws_status,
                                                                         int32_t* __restrict__ status) {
  __shared__ int s_ids[ROWS_PER_BLOCK][EGC_LONG_ROW_THRESHOLD];   // input positions of a short row
  __shared__ int2 s_sort[BUILD_LDS_SORT];                           // (input position, source) pairs of a long row
  const int sl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  if (blockIdx.x == 0 && threadIdx.x == 0) {   // the range-check flag leaves the (zero-on-exit) workspace
    *status = *ws_status;
    *ws_status = 0;
  }
  if (blockIdx.x == gridDim.x - 1) {           // dropped edges leave positions behind the last row: defined, harmless
    for (int64_t p = (int64_t)rowptr[n_nodes] + threadIdx.x; p < n_edges; p += blockDim.x) {
      col[p] = 0;
      edge_id[p] = 0x7fffffff;
      if (edis_raw != nullptr) edis_raw[p] = 0.0f;
      if (edis_looped != nullptr) edis_looped[p] = 0.0f;
    }
  }
  // ---- short rows: rank sort by input position, 16 lanes per row, up to 4 entries per lane; every lane group takes
  // ROWS_PER_GROUP rows with the loads of all of them issued before the first is ranked (the kernel is a chain of
  // dependent memory round trips -- row pointers, entries, deg^-1/2 of the sources -- so rows in flight are its speed)
  int start[ROWS_PER_GROUP], deg[ROWS_PER_GROUP];
  bool shortrow[ROWS_PER_GROUP];
#pragma unroll
  for (int r = 0; r < ROWS_PER_GROUP; ++r) {
    const int row = (blockIdx.x * ROWS_PER_BLOCK + rl) * ROWS_PER_GROUP + r;
    const bool live = row < n_nodes;
    start[r] = live ? rowptr[row] : 0;
    deg[r] = (live ? rowptr[row + 1] : 0) - start[r];
    shortrow[r] = live && deg[r] <= EGC_LONG_ROW_THRESHOLD;
  }
  int my_id[ROWS_PER_GROUP][4], my_col[ROWS_PER_GROUP][4];
#pragma unroll
  for (int r = 0; r < ROWS_PER_GROUP; ++r)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool ok = shortrow[r] && sl + 16 * k < deg[r];
      my_id[r][k] = ok ? edge_id[start[r] + sl + 16 * k] : 0x7fffffff;
      my_col[r][k] = ok ? col[start[r] + sl + 16 * k] : 0;
    }
#pragma unroll
  for (int r = 0; r < ROWS_PER_GROUP; ++r) {
    if (deg[r] > 16) {   // the row's strip: written and read by the 16 lanes of ONE wavefront, no workgroup barrier
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (shortrow[r] && sl + 16 * k < deg[r]) s_ids[rl][sl + 16 * k] = my_id[r][k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (shortrow[r]) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (sl + 16 * k < deg[r]) {
          int rank = 0;
          if (deg[r] <= 16) {   // one entry per lane: compare through the lanes, no LDS round trip
            for (int t = 0; t < deg[r]; ++t) rank += __shfl(my_id[r][0], (threadIdx.x & 48) + t) < my_id[r][k];
          } else {
            for (int t = 0; t < deg[r]; ++t) rank += s_ids[rl][t] < my_id[r][k];
          }
          const int p = start[r] + rank;
          col[p] = my_col[r][k];
          edge_id[p] = my_id[r][k];
          if (edis_raw != nullptr) edis_raw[p] = dis_raw[my_col[r][k]];
          if (edis_looped != nullptr) edis_looped[p] = dis_looped[my_col[r][k]];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();   // the strip is reused by the group's next row
  }
  // ---- long rows: registered in the plan by the scan kernel, spread over ALL workgroups by plan slot (hubs often
  // have consecutive ids: a workgroup that sorted the long rows among its own 64 would own every one of them) ----
  const int n_long = min(plan[0], cap_long);
  if (n_long == 0) return;
  const int32_t* long_row = plan + 4;
  const int32_t* long_chunk0 = long_row + cap_long;
  int32_t* chunk_slot = plan + 4 + 2 * cap_long;
  int32_t* chunk_begin = chunk_slot + cap_chunks;
  // up to BUILD_WAVE_ROW entries: one wavefront per row, rank sort in its strip of s_sort, no workgroup barrier
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int slot = blockIdx.x * 16 + wave; slot < n_long; slot += gridDim.x * 16) {
    const int r = long_row[slot];
    const int rs = rowptr[r], d = rowptr[r + 1] - rs;
    if (d > BUILD_WAVE_ROW) continue;
    int2* kv = s_sort + wave * BUILD_WAVE_ROW;
    int2 mine[BUILD_WAVE_ROW / 64];
#pragma unroll
    for (int k = 0; k < BUILD_WAVE_ROW / 64; ++k) {
      const int i = lane + 64 * k;
      mine[k] = i < d ? int2{edge_id[rs + i], col[rs + i]} : int2{0x7fffffff, 0};
      if (i < d) kv[i] = mine[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the strip is read by the other lanes of this wavefront
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < BUILD_WAVE_ROW / 64; ++k) {
      if (lane + 64 * k < d) {
        int rank = 0;
        for (int t = 0; t < d; ++t) rank += kv[t].x < mine[k].x;
        const int p = rs + rank;
        edge_id[p] = mine[k].x;
        col[p] = mine[k].y;
        if (edis_raw != nullptr) edis_raw[p] = dis_raw[mine[k].y];
        if (edis_looped != nullptr) edis_looped[p] = dis_looped[mine[k].y];
      }
    }
    __builtin_amdgcn_wave_barrier();   // strip reused by this wavefront's next row
  }
  // longer ones: the whole workgroup, one slot at a time -- rank sort up to BUILD_RANK_ROW entries, bitonic networks
  // beyond (in LDS up to BUILD_LDS_SORT, through global scratch above that); every slot's chunk entries on the way
  for (int slot = blockIdx.x; slot < n_long; slot += gridDim.x) {
    const int r = long_row[slot], c0 = long_chunk0[slot];
    const int rs = rowptr[r], re = rowptr[r + 1], d = re - rs;
    const int nch = (d + EGC_LONG_ROW_CHUNK - 1) / EGC_LONG_ROW_CHUNK;
    if (c0 + nch <= cap_chunks)
      for (int k = threadIdx.x; k < nch; k += blockDim.x) { chunk_slot[c0 + k] = slot; chunk_begin[c0 + k] = rs + k * EGC_LONG_ROW_CHUNK; }
    if (d <= BUILD_WAVE_ROW) continue;   // uniform over the workgroup
    int np2 = 1;
    while (np2 < d) np2 <<= 1;
    int2* kv = d <= BUILD_LDS_SORT ? s_sort : big_scratch + (int64_t)rs * 2;   // global scratch: 2 x its own range (padding)
    __syncthreads();
    if (d <= BUILD_RANK_ROW) {
      const int i = threadIdx.x;
      const int2 v = i < d ? int2{edge_id[rs + i], col[rs + i]} : int2{0x7fffffff, 0};
      if (i < d) kv[i] = v;
      __syncthreads();
      if (i < d) {
        int rank = 0;
        for (int t = 0; t < d; ++t) rank += kv[t].x < v.x;
        const int p = rs + rank;
        edge_id[p] = v.x;
        col[p] = v.y;
        if (edis_raw != nullptr) edis_raw[p] = dis_raw[v.y];
        if (edis_looped != nullptr) edis_looped[p] = dis_looped[v.y];
      }
      continue;
    }
    if (d <= BUILD_LDS_SORT || d > BUILD_MERGE_ROW)
      for (int i = threadIdx.x; i < np2; i += blockDim.x) kv[i] = i < d ? int2{edge_id[rs + i], col[rs + i]} : int2{0x7fffffff, 0};
    __syncthreads();
    if (d <= BUILD_LDS_SORT) {
      for (int k = 2; k <= np2; k <<= 1) bitonic_steps(kv, np2, 0, k, k >> 1);
    } else if (d <= BUILD_MERGE_ROW) {
      merge_sort_big(kv, d, s_sort, edge_id + rs, col + rs, dis_raw, dis_looped, edis_raw != nullptr ? edis_raw + rs : nullptr,
                     edis_looped != nullptr ? edis_looped + rs : nullptr);
      continue;
    } else {
      bitonic_big(kv, np2, s_sort);
    }
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
      const int2 v = kv[i];
      edge_id[rs + i] = v.x;
      col[rs + i] = v.y;
      if (edis_raw != nullptr) edis_raw[rs + i] = dis_raw[v.y];
      if (edis_looped != nullptr) edis_looped[rs + i] = dis_looped[v.y];
    }
  }
}

// out[k][:] = table[idx[k]][:] for rows of w4 16-byte pieces (the send pack of the halo exchange): one piece per lane,
// consecutive lanes on consecutive pieces of a row -- whole rows are read and written as contiguous runs
__global__ void __launch_bounds__(256) gather_rows_kernel(const float4* __restrict__ table, const int64_t* __restrict__ idx,
                                                          int64_t n_pieces, int w4, int64_t ld4, float4* __restrict__ out) {
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; g < n_pieces; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = g / w4;
    const int c = (int)(g - k * w4);
    out[g] = table[idx[k] * ld4 + c];
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

const char* egc_version(void) { return "egc_hip 0.3.0 gfx950"; }

int64_t egc_plan_ints(int64_t n_nodes, int64_t n_edges) {
  if (n_nodes < 0 || n_edges < 0) return -1;
  PlanCaps c = plan_caps(n_nodes, n_edges);
  return 4 + 2 * c.cap_long + 2 * c.cap_chunks;
}

size_t egc_coo_to_csr_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
  if (n_nodes < 0 || n_edges < 0) return 0;
  size_t temp = 0;
  if (sort_temp_bytes(n_nodes + 1, n_edges, &temp) != hipSuccess) return 0;   // (+1: the checked form's key for dropped edges)
  // keys_in + keys_out + rocprim temp
  return 2 * align256((size_t)n_edges * sizeof(uint32_t)) + align256(temp) + 256;
}

int egc_coo_to_csr(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_nodes, int32_t* rowptr,
                   int32_t* col, int32_t* edge_id, int32_t* max_index, void* workspace, size_t workspace_bytes,
                   egc_stream_t stream_) {
  return egc_coo_to_csr_checked(src, dst, n_edges, n_nodes, 0, rowptr, col, edge_id, max_index, nullptr, nullptr, workspace,
                                workspace_bytes, stream_);
}

int egc_coo_to_csr_checked(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_nodes, int64_t n_src_rows,
                           int32_t* rowptr, int32_t* col, int32_t* edge_id, int32_t* max_index, int32_t* status,
                           int32_t* host_flag, void* workspace, size_t workspace_bytes, egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_nodes < 0 || n_edges < 0 || n_nodes >= ((int64_t)1 << 31) - 1 || n_edges >= ((int64_t)1 << 31) - 1)
    return EGC_ERR_INVALID;
  if (rowptr == nullptr || max_index == nullptr) return EGC_ERR_INVALID;
  if (n_edges > 0 && (src == nullptr || dst == nullptr || col == nullptr || edge_id == nullptr)) return EGC_ERR_INVALID;

  init_scalar_kernel<<<1, 1, 0, stream>>>(max_index, -1);
  EGC_LAUNCH_CHECK("init_scalar_kernel");
  if (status != nullptr) {
    init_scalar_kernel<<<1, 1, 0, stream>>>(status, 0);
    EGC_LAUNCH_CHECK("init_scalar_kernel");
  }
  if (n_edges == 0) {
    EGC_HIP_TRY(hipMemsetAsync(rowptr, 0, (size_t)(n_nodes + 1) * sizeof(int32_t), stream));
    return EGC_OK;
  }
  const int64_t n_src = n_src_rows > 0 ? n_src_rows : n_nodes;
  const int64_t n_keys = status != nullptr ? n_nodes + 1 : n_nodes;   // the checked form sorts dropped edges under key n_nodes
  size_t temp = 0;
  EGC_HIP_TRY(sort_temp_bytes(n_keys, n_edges, &temp));
  const size_t kbytes = align256((size_t)n_edges * sizeof(uint32_t));
  if (workspace == nullptr || workspace_bytes < 2 * kbytes + align256(temp)) return EGC_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  uint32_t* keys_in = (uint32_t*)ws;
  uint32_t* keys_out = (uint32_t*)(ws + kbytes);
  void* sort_temp = ws + 2 * kbytes;

  const int threads = 256;
  const int blocks = (int)std::min<int64_t>(ceil_div(n_edges, threads), 256 * 8);
  const int key_blocks = std::min(blocks, 512);
  coo_keys_kernel<<<key_blocks, threads, 0, stream>>>(src, dst, n_edges, keys_in, max_index, n_nodes, n_src, status, host_flag);
  EGC_LAUNCH_CHECK("coo_keys_kernel");
  // Stable LSD radix sort of (dst, input position): the value array IS edge_id.
  EGC_HIP_TRY(rocprim::radix_sort_pairs(sort_temp, temp, (const uint32_t*)keys_in, keys_out,
                                        rocprim::counting_iterator<uint32_t>(0), (uint32_t*)edge_id,
                                        (size_t)n_edges, 0u, key_bits(n_keys), stream));
  rowptr_kernel<<<(int)ceil_div(n_nodes + 1, threads), threads, 0, stream>>>(keys_out, n_edges, n_nodes, rowptr);
  EGC_LAUNCH_CHECK("rowptr_kernel");
  gather_col_kernel<<<blocks, threads, 0, stream>>>(src, (const uint32_t*)edge_id, n_edges, col,
                                                    status != nullptr ? rowptr + n_nodes : nullptr);
  EGC_LAUNCH_CHECK("gather_col_kernel");
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

int egc_csr_transposed_coo(int64_t n_nodes, int64_t n_edges, const int32_t* rowptr, const int32_t* col, int64_t* out_src,
                           int64_t* out_dst, egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_nodes < 0 || n_edges < 0 || n_nodes >= (1ll << 31) || n_edges >= (1ll << 31)) return EGC_ERR_INVALID;
  if (n_edges == 0) return EGC_OK;
  if (n_nodes == 0 || rowptr == nullptr || col == nullptr || out_src == nullptr || out_dst == nullptr) return EGC_ERR_INVALID;
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n_edges, (int64_t)256), 8192);
  csr_transposed_coo_kernel<<<blocks, 256, 0, stream>>>((int)n_nodes, n_edges, rowptr, col, out_src, out_dst);
  EGC_LAUNCH_CHECK("csr_transposed_coo_kernel");
  return EGC_OK;
}

int egc_gather_rows_f32(const float* table, int64_t ld, const int64_t* idx, int64_t n_rows, int32_t width, float* out,
                        egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_rows < 0 || width <= 0 || (width & 3) != 0 || (ld & 3) != 0 || ld < width) return EGC_ERR_INVALID;
  if (n_rows == 0) return EGC_OK;
  if (table == nullptr || idx == nullptr || out == nullptr) return EGC_ERR_INVALID;
  if (((uintptr_t)table & 15) != 0 || ((uintptr_t)out & 15) != 0) return EGC_ERR_INVALID;
  const int64_t pieces = n_rows * (width / 4);
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(pieces, (int64_t)256), 256 * 16);
  gather_rows_kernel<<<blocks, 256, 0, stream>>>((const float4*)table, idx, pieces, width / 4, ld / 4, (float4*)out);
  EGC_LAUNCH_CHECK("gather_rows_kernel");
  return EGC_OK;
}

size_t egc_graph_build_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
  if (n_nodes < 0 || n_edges < 0) return 0;
  const int64_t nb = ceil_div(n_nodes + 1, (int64_t)BUILD_SCAN_ITEMS);
  // deg[n + 1] | deg_ns[n] | block sums | max id + 1: zero on entry, ALL of it zero again on exit -- so the next call
  // may lay a different graph's arrays over the same bytes
  return align256((size_t)(n_nodes + 1) * 4) + align256((size_t)n_nodes * 4 + 4) + align256((size_t)nb * 4) + 256;
}

size_t egc_graph_build_scratch_bytes(int64_t n_edges) {
  if (n_edges < 0) return 0;
  // sort area of the rows longer than the LDS sort: two (position, source) pairs per entry (power-of-two padding),
  // indexed by the row's own CSR range; any content on entry, garbage on exit -- hence not part of the workspace
  return align256((size_t)n_edges * 2 * sizeof(int2)) + 256;
}

int egc_graph_build(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_nodes, int64_t n_src_rows,
                    int32_t* rowptr, int32_t* col, int32_t* edge_id, int32_t* max_index, float* dis_raw,
                    float* dis_looped, float* edge_dis_raw, float* edge_dis_looped, int32_t* plan, int32_t* status,
                    int32_t* host_flag, void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                    egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_nodes < 0 || n_edges < 0 || n_nodes >= ((int64_t)1 << 30) || n_edges >= ((int64_t)1 << 30)) return EGC_ERR_INVALID;
  if (rowptr == nullptr || max_index == nullptr || plan == nullptr || status == nullptr) return EGC_ERR_INVALID;
  if (n_edges > 0 && (src == nullptr || dst == nullptr || col == nullptr || edge_id == nullptr)) return EGC_ERR_INVALID;
  if (workspace == nullptr || workspace_bytes < egc_graph_build_workspace_bytes(n_nodes, n_edges)) return EGC_ERR_WORKSPACE;
  if (scratch == nullptr || scratch_bytes < egc_graph_build_scratch_bytes(n_edges)) return EGC_ERR_WORKSPACE;
  if (dis_raw == nullptr) edge_dis_raw = nullptr;
  if (dis_looped == nullptr) edge_dis_looped = nullptr;
  const int64_t n_src = n_src_rows > 0 ? n_src_rows : n_nodes;
  const int nb = (int)ceil_div(n_nodes + 1, (int64_t)BUILD_SCAN_ITEMS);
  char* ws = (char*)workspace;
  int* deg = (int*)ws;
  int* deg_ns = (int*)(ws + align256((size_t)(n_nodes + 1) * 4));
  int* bsum = (int*)((char*)deg_ns + align256((size_t)n_nodes * 4 + 4));
  int* maxp1 = (int*)((char*)bsum + align256((size_t)nb * 4));
  int* ws_status = maxp1 + 1;
  int2* big = (int2*)scratch;
  PlanCaps c = plan_caps(n_nodes, n_edges);
  if (n_nodes == 0) {  // nothing to build: header and flag only (every edge is out of range)
    plan_header_kernel<<<1, 1, 0, stream>>>(plan, (int)c.cap_long, (int)c.cap_chunks);
    init_scalar_kernel<<<1, 1, 0, stream>>>(status, n_edges > 0 ? 1 : 0);
    if (n_edges > 0 && host_flag != nullptr) init_scalar_kernel<<<1, 1, 0, stream>>>(host_flag, 1);
    init_scalar_kernel<<<1, 1, 0, stream>>>(max_index, -1);
    EGC_HIP_TRY(hipMemsetAsync(rowptr, 0, sizeof(int32_t), stream));
    EGC_LAUNCH_CHECK("egc_graph_build(empty)");
    return EGC_OK;
  }
  const int eblocks = (int)std::max<int64_t>(1, ceil_div(n_edges, (int64_t)BUILD_TILE));
  build_hist_kernel<<<eblocks, 256, 0, stream>>>(src, dst, n_edges, (int)n_nodes, (int)n_src, deg, deg_ns, maxp1, ws_status, plan,
                                             (int)c.cap_long, (int)c.cap_chunks, host_flag);
  EGC_LAUNCH_CHECK("build_hist_kernel");
  build_sums_kernel<<<nb, 1024, 0, stream>>>(deg, (int)n_nodes, bsum);
  EGC_LAUNCH_CHECK("build_sums_kernel");
  build_scan_kernel<<<nb, 1024, 0, stream>>>(deg, deg_ns, (int)n_nodes, bsum, nb, rowptr, dis_raw, dis_looped, maxp1, max_index, plan,
                                             (int)c.cap_long, (int)c.cap_chunks);
  EGC_LAUNCH_CHECK("build_scan_kernel");
  build_scatter_kernel<<<eblocks, 256, 0, stream>>>(src, dst, n_edges, (int)n_nodes, (int)n_src, deg, rowptr, col, edge_id, bsum, nb);
  EGC_LAUNCH_CHECK("build_scatter_kernel");
  build_rows_kernel<<<(int)ceil_div(n_nodes, (int64_t)ROWS_PER_BLOCK * ROWS_PER_GROUP), 16 * ROWS_PER_BLOCK, 0, stream>>>(
      (int)n_nodes, n_edges, rowptr, col, edge_id, dis_raw, dis_looped, edge_dis_raw, edge_dis_looped, plan, (int)c.cap_long,
      (int)c.cap_chunks, big, ws_status, status);
  EGC_LAUNCH_CHECK("build_rows_kernel");
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
