// Batches of small graphs (BASELINE configs 3 / 4: molecules, superpixel graphs): the aggregate+combine launch on TILES
// OF WHOLE GRAPHS, with the graph preparation inside it.
//
// The reference's batched nets (zinc/models.py:60-74, mol/pna_style_models.py:64-79, cifar/models.py:61-75) call the layer
// with a PyG batch: the disjoint union of a few hundred to a few thousand small graphs, nodes numbered graph by graph,
// the edges of one graph contiguous in `edge_index`.  On such a batch the adjacency is block diagonal, so a workgroup
// that takes a run of consecutive graphs needs nothing from outside its own rows:
//
//   plan    (tile_plan_kernel, once per batch, shared by the layers of a net): the node range [0, N) is cut into slots of
//           S nodes; tile k = the graphs whose first node lies in slot k -> node range [n0, n1), edge range [e0, e1)
//           (wave-wide 64-ary searches in the graph offsets and in the destination row of edge_index);
//   layer   (agg_tile_kernel, one workgroup per tile): the tile's rows of `bases` are copied into LDS once (each row
//           is then gathered from LDS by every in-neighbour instead of from L2 / HBM: at 8 in-neighbours per node the
//           CIFAR batch reads 62 MB of basis rows instead of 556 MB of gathered ones); the tile's edges are turned into
//           a CSR in LDS (degree count by LDS atomics, wave scan, scatter) together with both deg^-1/2 tables --
//           the five launches of egc_graph_build per batch disappear; the lane groups then aggregate their rows
//           from LDS and run the register epilogue of the fast kernel family (egc_aggregate_fast_dev.h: finish_group),
//           weightings rows prefetched one pass ahead.
//
// Replaces, for such batches, the same reference call sites as egc_graph_build + egc_aggregate_combine_post_f32
// (MessagePassing.propagate's index handling and gather, torch_scatter's reductions, the combine and the caller's
// BatchNorm(eval) / ReLU / residual tail).  Inference form (no arg positions / statistics for a backward).
// Every edge is range-checked against ITS TILE: an edge that leaves its graph's tile (an edge list that is not grouped by
// graph, an id outside [0, N)), a tile with more nodes or edges than the LDS areas hold -> *status and the sticky
// host flag are raised (codes below) and the tile's rows are left unwritten.
// A row's entries sit in the LDS CSR in INPUT ORDER (round 6, as in egc_fused_tile_dev.h): the tile's edges are cut into eight
// contiguous spans, wavefront w holds span 2 (w % 4) + w / 4; the two wavefronts of a PAIR (w, w + 4) -- two adjacent spans --
// share one 16-bit counter per row (four counters: two words), which the scan turns into the pair's cursor; the scatter takes an
// entry's position from a returning LDS add to that cursor -- one wavefront's instructions execute in program order and serve
// equal addresses in ascending lane order (tools/src/lds_atomic_order.hip) -- in two passes, wavefronts 0-3 then 4-7, so a pair's
// second span follows its first.  The reference's CPU scatter sums a row in edge order (SURVEY.md 8a note 9), and two launches
// give the same bits.
#include <algorithm>

#include "egc_aggregate_fast_dev.h"

namespace egc {

#ifndef EGC_TILE_THREADS
#define EGC_TILE_THREADS 512
#endif
#ifndef EGC_TILE_WGS
#define EGC_TILE_WGS 2
#endif
constexpr int TILE_THREADS = EGC_TILE_THREADS;
constexpr int TILE_WAVES = TILE_THREADS / 64;
constexpr int TILE_MAX_NODES = 2048;    // cap of the per-tile CSR areas (local ids are 16-bit)
constexpr int TILE_WGS_PER_CU = EGC_TILE_WGS;      // what the LDS areas are sized for (tile_capacity)
constexpr int TILE_EDGE_REGS = 6;       // edges per thread kept in registers between the two CSR passes (3072 per tile)

struct TileArgs {
  const int4* tiles;       // (n0, n1, e0, e1) per non-empty tile, compacted (any order)
  const int* n_tiles;      // device scalar: how many
  int n_tiles_cap;         // records the list holds (a plan of malformed offsets counts more: reported, the excess ignored)
  const int64_t* src;      // edge_index[0]
  const int64_t* dst;      // edge_index[1]
  const int* max_index;    // device scalar (layers with loops_all == 0), or nullptr
  int32_t* status;
  int32_t* host_flag;
  int tlds;                // nodes whose basis rows fit the LDS area: larger tiles gather from memory instead
  int tmax, emax;          // capacity of the per-tile CSR areas: nodes / edges
  // byte offsets of the tile areas inside dynamic LDS (behind the per-wavefront epilogue strips)
  int off_bases, off_col, off_rowptr, off_cnt, off_ns, off_dis_raw, off_dis_looped;
};

// first index i in [0, n) with arr[i] >= key (n if none), by HALF a wavefront (lanes [32 h, 32 h + 32) share `key`): 32-ary
// narrowing, then one probe per lane -- the two halves of a wavefront run two searches side by side.  On an array that
// is not sorted the result is still a deterministic function of (arr, key): the tiles' edge ranges therefore always
// partition [0, E), and the tile kernel's per-edge range check reports what the search got wrong.
__device__ inline int64_t half_wave_lower_bound(const int64_t* __restrict__ arr, int64_t n, int64_t key, int lane) {
  const int l32 = lane & 31, sh = lane & 32;
  int64_t lo = 0, hi = n;   // answer in [lo, hi]; everything before lo is < key, arr[hi] (if hi < n) is >= key
  while (__ballot(hi - lo > 32) != 0) {            // (the other half may still be narrowing: keep probing in step)
    const bool live = hi - lo > 32;
    const int64_t step = live ? (hi - lo + 31) / 32 : 1;
    const int64_t i = lo + (int64_t)l32 * step;
    const bool ge = (live && i < hi) ? arr[i] >= key : true;
    const unsigned m = (unsigned)(__ballot(ge) >> sh);
    if (!live) continue;
    const int f = __ffs((int)m) - 1;               // first probe that is >= key
    if (f < 0) { lo = lo + 31 * step + 1; if (lo > hi) lo = hi; continue; }   // all probes < key: the answer lies behind the last
    const int64_t nhi = lo + (int64_t)f * step;
    lo = f > 0 ? lo + (int64_t)(f - 1) * step + 1 : lo;
    hi = nhi < hi ? nhi : hi;
  }
  const int64_t i = lo + l32;
  const bool ge = i < hi ? arr[i] >= key : true;
  const unsigned m = (unsigned)(__ballot(ge) >> sh);
  const int f = __ffs((int)m) - 1;
  return f < 0 ? hi : (lo + f < hi ? lo + f : hi);
}

// One wavefront per slot k: lanes 0-31 find where the slot's first graph starts (node n0, edge e0), lanes 32-63 the same
// for slot k + 1 (n1, e1); a non-empty tile [n0, n1) x [e0, e1) takes the next place of the compacted list.
__global__ void __launch_bounds__(256) tile_plan_kernel(const int64_t* __restrict__ ptr, int64_t n_graphs,
                                                        const int64_t* __restrict__ dst, int64_t n_edges, int64_t n_nodes,
                                                        int slot, int n_slots, int4* __restrict__ tiles, int* __restrict__ count,
                                                        const int64_t* __restrict__ edge_ptr) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n_slots) return;
  const int kb = k + (lane >> 5);                  // the boundary this half works on
  const int64_t g = half_wave_lower_bound(ptr, n_graphs + 1, (int64_t)kb * slot, lane);
  int64_t nd = g <= n_graphs ? ptr[g] : n_nodes;
  nd = nd < 0 ? 0 : (nd > n_nodes ? n_nodes : nd);
  if (kb == 0) nd = 0;
  if (kb >= n_slots) nd = n_nodes;                 // (ptr[G] == N for a well-formed batch; stragglers stay with the last tile)
  // the graph's first edge: given by the caller (PyG keeps the per-graph edge offsets of a collated batch), else found
  // in the destination row of edge_index (edges grouped by graph)
  int64_t ed = edge_ptr != nullptr ? (g <= n_graphs ? edge_ptr[g] : n_edges) : half_wave_lower_bound(dst, n_edges, nd, lane);
  ed = ed < 0 ? 0 : (ed > n_edges ? n_edges : ed);
  if (nd <= 0) ed = 0;
  if (nd >= n_nodes) ed = n_edges;
  const int n0 = (int)__shfl(nd, 0), e0 = (int)__shfl(ed, 0);
  int n1 = (int)__shfl(nd, 32), e1 = (int)__shfl(ed, 32);
  n1 = n1 < n0 ? n0 : n1;
  e1 = e1 < e0 ? e0 : e1;
  if (lane == 0 && (n1 > n0 || e1 > e0)) {
    const int i = atomicAdd(count, 1);
    if (i < n_slots) tiles[i] = int4{n0, n1, e0, e1};
  }
}

// The same tiles from the GRAPH side, when the caller supplies the graphs' edge offsets: one thread per graph; a graph
// whose first node lies in another slot than its predecessor's heads a tile, which runs to the next head.  No search:
// one round of loads (against the eight dependent probes of the search form).
__global__ void __launch_bounds__(256) tile_plan_graphs_kernel(const int64_t* __restrict__ ptr, const int64_t* __restrict__ edge_ptr,
                                                               int64_t n_graphs, int64_t n_edges, int64_t n_nodes, int slot,
                                                               int4* __restrict__ tiles, int* __restrict__ count, int n_slots) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= n_graphs) return;
  const int64_t p0 = ptr[g];
  const int64_t sg = p0 / slot;
  if (g > 0 && ptr[g - 1] / slot == sg) return;     // not a head
  int64_t end = g + 1;
  for (bool found = false; !found && end < n_graphs;) {
    int64_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = end + j < n_graphs ? ptr[end + j] : (int64_t)-1;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (!found) {
        if (end + j >= n_graphs || v[j] / slot != sg) { found = true; end += j; }
      }
    if (!found) end += 8;
  }
  if (end > n_graphs) end = n_graphs;
  auto clampn = [&](int64_t v) { return v < 0 ? (int64_t)0 : (v > n_nodes ? n_nodes : v); };
  auto clampe = [&](int64_t v) { return v < 0 ? (int64_t)0 : (v > n_edges ? n_edges : v); };
  const int64_t n0 = g == 0 ? 0 : clampn(p0);
  int64_t n1 = end >= n_graphs ? n_nodes : clampn(ptr[end]);
  const int64_t e0 = g == 0 ? 0 : clampe(edge_ptr[g]);
  int64_t e1 = end >= n_graphs ? n_edges : clampe(edge_ptr[end]);
  n1 = n1 < n0 ? n0 : n1;
  e1 = e1 < e0 ? e0 : e1;
  // A well-formed (non-decreasing) ptr yields at most one head per slot; offsets that jump back and forth make almost
  // every graph a head: the list holds n_slots records and no more (the tile kernel reports the batch: *count > n_slots).
  if (n1 > n0 || e1 > e0) {
    const int i = atomicAdd(count, 1);
    if (i < n_slots) tiles[i] = int4{(int)n0, (int)n1, (int)e0, (int)e1};
  }
}

template <int LPR_LOG2, class C>
__device__ inline void load_weightings_row(const AggArgs& a, int q, int row, bool row_ok, f4 (&wpre)[2]) {
  constexpr int LPR = 1 << LPR_LOG2;
  const int W = C::W(a);
  const float* wrow = a.weightings + (int64_t)(row_ok ? row : 0) * a.ldw;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c0 = (q + k * LPR) * 4;
    wpre[k] = f4{0.f, 0.f, 0.f, 0.f};
    if (row_ok && c0 + 3 < W) wpre[k] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(wrow + c0));
    else if (row_ok && c0 < W) {
      wpre[k].x = wrow[c0];
      if (c0 + 1 < W) wpre[k].y = wrow[c0 + 1];
      if (c0 + 2 < W) wpre[k].z = wrow[c0 + 2];
    }
  }
}

__device__ inline void tile_error(const TileArgs& t, int code) {
  atomicOr(t.status, code);
  if (t.host_flag != nullptr) *(volatile int32_t*)t.host_flag = 1;
}

// Persistent: the workgroups (TILE_WGS_PER_CU per CU, sized by their LDS) take tiles k = blockIdx.x, + gridDim.x, ...
// (two 8-wavefront workgroups per CU = 4 wavefronts per SIMD: 128 VGPRs each)
template <int LPR_LOG2, int HPB, int NEED, class C>
__global__ void __launch_bounds__(TILE_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) agg_tile_kernel(AggArgs a, TileArgs t) {
  constexpr int LPR = 1 << LPR_LOG2, G = 64 / LPR;
  extern __shared__ float smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int g = lane >> LPR_LOG2;
  const int q = lane & (LPR - 1);
  const int F_out = C::F_out(a);
  int n_tiles = *t.n_tiles;
  if (n_tiles > t.n_tiles_cap) {       // graph offsets that are not non-decreasing (egc_batch_plan): code 1, like a bad edge
    if (tid == 0 && blockIdx.x == 0) { atomicOr(t.status, 1); if (t.host_flag != nullptr) *(volatile int32_t*)t.host_flag = 1; }
    n_tiles = t.n_tiles_cap;
  }
  int k = blockIdx.x;
  if (k >= n_tiles) return;
  int4 tl = t.tiles[k];

  // ---- per-wavefront epilogue strips: [bias (x scale + shift)][scale][G weight strips], as agg_fast_kernel ----
  float* lds_bias = smem + wave * a.lds_floats_per_wave;
  const bool post = a.post_scale != nullptr;
  float* lds_scale = lds_bias + a.bias_lds_floats;
  float* lds_w = lds_bias + (post ? 2 : 1) * a.bias_lds_floats;
  for (int o = lane; o < C::H(a) * C::Ls(a); o += 64) {
    const int h = o / C::Ls(a), l = o - h * C::Ls(a);
    const int c = h * C::L(a) + l;
    const bool real = l < C::L(a);
    float bv = (a.bias != nullptr && real) ? a.bias[c] : 0.f;
    if (post) {
      const float sc = real ? a.post_scale[c] : 0.f;
      bv = fmaf(bv, sc, real ? a.post_shift[c] : 0.f);
      lds_scale[o] = sc;
    }
    lds_bias[o] = bv;
  }
  char* base = reinterpret_cast<char*>(smem);
  f4* lds_bases4 = reinterpret_cast<f4*>(base + t.off_bases);
  unsigned short* lds_col = reinterpret_cast<unsigned short*>(base + t.off_col);
  int* lds_rowptr = reinterpret_cast<int*>(base + t.off_rowptr);
  int* lds_cnt = reinterpret_cast<int*>(base + t.off_cnt);       // [2][tmax]: per row and PAIR of wavefronts a 16-bit count, then cursor
  int* lds_ns = reinterpret_cast<int*>(base + t.off_ns);         // non-self in-degree
  float* lds_dis_raw = reinterpret_cast<float*>(base + t.off_dis_raw);
  float* lds_dis_looped = reinterpret_cast<float*>(base + t.off_dis_looped);
  const int ldb4 = a.ldb >> 2;
  const unsigned row_bytes = (unsigned)a.ldb * 4u;
  const unsigned slot_off = (unsigned)q * 16u;
  FastRsrc R;
  R.bases = bases_rsrc(a);
  R.out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (unsigned)a.n_nodes * (unsigned)F_out * 4u, 0x00020000);
  R.res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual != nullptr ? a.residual : a.out), 0,
                                            (unsigned)a.n_nodes * (unsigned)F_out * 4u, 0x00020000);
  const bool looped_any = C::xl(a) || C::yl(a);
  const float* lds_dis = C::yl(a) ? lds_dis_looped : lds_dis_raw;
  const bool want_dis = a.dis != nullptr;       // (set by the host when the layer has a symnorm aggregator)
  const int max_index = (!C::loops_all(a) && t.max_index != nullptr) ? *t.max_index : 0x7fffffff;
  constexpr int RPP = TILE_WAVES * G;           // rows per pass
  const int grp_addr = (g << LPR_LOG2) << 2;

  // LDS-only barrier: __syncthreads() would also drain the LDS-DMA in flight (hipcc emits vmcnt(0) in front of it)
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // the edges of a tile, TILE_EDGE_REGS per thread, checked against the tile (-1 = absent or outside: reported)
  int es[TILE_EDGE_REGS], ed[TILE_EDGE_REGS];
  bool bad = false;
  int edges_of = -1;            // the tile whose edges es / ed hold (requested one tile ahead, during the rows phase)
  static_assert(TILE_WAVES == 8, "the CSR build pairs wavefront w with w + 4");
  auto span_len = [](int tE) -> int { return (((tE + TILE_WAVES - 1) / TILE_WAVES) + 63) & ~63; };
  const int my_span = 2 * (wave & 3) + (wave >> 2);
  auto request_edges = [&](const int4 rec, int which) {
    const int tn0 = rec.x, tT = rec.y - rec.x, te0 = rec.z, tE = rec.w - rec.z;
    bad = false;
    edges_of = which;
    if (tT <= 0 || tE > TILE_THREADS * TILE_EDGE_REGS) return;
    const int S = span_len(tE);
#pragma unroll
    for (int j = 0; j < TILE_EDGE_REGS; ++j) {
      const int i = my_span * S + 64 * j + lane;    // this wavefront's span of the tile's edges, in rounds of 64
      es[j] = ed[j] = -1;
      if (64 * j + lane < S && i < tE) {
        const int64_t s = t.src[(int64_t)te0 + i] - tn0, d = t.dst[(int64_t)te0 + i] - tn0;
        if (s < 0 || s >= tT || d < 0 || d >= tT) bad = true;
        else { es[j] = (int)s; ed[j] = (int)d; }
      }
    }
  };
  // this wavefront's pair p = wave % 4: 16 bits of a row's two counter words -- word p / 2, half p % 2
  int* my_cnt = lds_cnt + ((wave & 3) >> 1) * t.tmax;
  const int my_inc = (wave & 1) ? 0x10000 : 1;
  auto my_half = [&](int v) -> int { return (wave & 1) ? (int)((unsigned)v >> 16) : (v & 0xffff); };

  for (; k < n_tiles; k += gridDim.x) {
    const int4 cur = tl;
    const int n0 = tl.x, T = tl.y - tl.x, e0 = tl.z, Et = tl.w - tl.z;
    const bool has_next = k + (int)gridDim.x < n_tiles;
    if (has_next) tl = t.tiles[k + gridDim.x];   // the next tile's record, long before it is needed
    if (T <= 0 || T > t.tmax || Et > t.emax) {   // edges without rows / beyond the CSR areas: report; the rows become zeros, not
      if (tid == 0) tile_error(t, T <= 0 ? 1 : 2);   // whatever the allocation held (the caller may read `out` before it checks)
      if (T > 0) {
        const int64_t lo = (int64_t)n0 * F_out, hi = (int64_t)(n0 + T > a.n_nodes ? a.n_nodes : n0 + T) * F_out;
        for (int64_t i = lo + tid; i < hi; i += TILE_THREADS) a.out[i] = 0.f;
      }
      continue;
    }
    const bool in_lds = T <= t.tlds;             // the tile's basis rows fit LDS (else: gathered from memory)
    const bool in_regs = Et <= TILE_THREADS * TILE_EDGE_REGS;   // else: both CSR passes stream the edges from memory

    // ---- (A) the weightings of the first two passes; the edges unless they were requested during the previous tile's
    //      rows; counters zeroed ----
    f4 wn0[2], wn1[2];
    {
      const int r = wave * G + g;
      load_weightings_row<LPR_LOG2, C>(a, q, n0 + r, r < T, wn0);
      load_weightings_row<LPR_LOG2, C>(a, q, n0 + r + RPP, r + RPP < T, wn1);
    }
    if (in_regs && edges_of != k) request_edges(cur, k);
    for (int i = tid; i < T; i += TILE_THREADS) {
      lds_cnt[i] = 0;
      lds_cnt[t.tmax + i] = 0;
      lds_ns[i] = 0;
    }
    lds_barrier();

    // ---- (B) in-degrees; every edge checked against the tile ----
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < TILE_EDGE_REGS; ++j)
        if (ed[j] >= 0) {
          atomicAdd(&my_cnt[ed[j]], my_inc);
          if (es[j] != ed[j]) atomicAdd(&lds_ns[ed[j]], 1);
        }
    } else {
      bad = false;
      const int S = span_len(Et);
#pragma unroll 4
      for (int i0 = 0; i0 < S; i0 += 64) {
        const int i = my_span * S + i0 + lane;
        if (i >= Et) continue;
        const int64_t s = t.src[(int64_t)e0 + i] - n0, d = t.dst[(int64_t)e0 + i] - n0;
        if (s < 0 || s >= T || d < 0 || d >= T) { bad = true; continue; }
        atomicAdd(&my_cnt[(int)d], my_inc);
        if (s != d) atomicAdd(&lds_ns[(int)d], 1);
      }
    }
    if (__ballot(bad) != 0 && lane == 0) tile_error(t, 1);
    // the tile's basis rows by LDS-DMA (one flat copy), issued AFTER the edges have been consumed (the compiler drains
    // every outstanding load at the first use of an ordinary one) and waited for only in front of the rows phase: the
    // copy runs beside the scan and the scatter
    if (in_lds) {
      const int n4 = T * ldb4;
      const f4* gb4 = reinterpret_cast<const f4*>(a.bases) + (int64_t)n0 * ldb4;
      for (int i = tid; i < n4; i += TILE_THREADS)    // LDS destination = wave-uniform base + lane * 16: a flat copy qualifies
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb4 + i),
                                         (__attribute__((address_space(3))) void*)(lds_bases4 + i), 16, 0, 0);
    }
    lds_barrier();

    // ---- (C) exclusive scan -> rowptr, deg^-1/2 tables (wavefront 0) ----
    if (wave == 0) {
      const int per = (T + 63) >> 6;
      const int b0 = lane * per;
      int mine = 0;
      auto row_count = [&](int i) -> int {      // the row's in-degree: its four 16-bit counters
        const int v0 = lds_cnt[i], v1 = lds_cnt[t.tmax + i];
        return (v0 & 0xffff) + (int)((unsigned)v0 >> 16) + (v1 & 0xffff) + (int)((unsigned)v1 >> 16);
      };
      for (int j = 0; j < per; ++j) mine += (b0 + j < T) ? row_count(b0 + j) : 0;
      int incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
      }
      int run = incl - mine;
      for (int j = 0; j < per; ++j) {
        const int i = b0 + j;
        if (i < T) {
          lds_rowptr[i] = run;
          int c = 0;
#pragma unroll
          for (int p = 0; p < 2; ++p) {                                     // counts -> cursors: where each pair's entries start
            const int v = lds_cnt[p * t.tmax + i];
            const int lo = v & 0xffff, hi = (int)((unsigned)v >> 16);
            lds_cnt[p * t.tmax + i] = (run + c) | ((run + c + lo) << 16);
            c += lo + hi;
          }
          lds_dis_raw[i] = c > 0 ? 1.0f / sqrtf((float)c) : 0.0f;          // as prepare_kernel / build_scan_kernel
          lds_dis_looped[i] = 1.0f / sqrtf((float)(lds_ns[i] + 1));
          run += c;
        }
      }
      if (lane == 63) lds_rowptr[T] = incl;
    }
    lds_barrier();

    // ---- (D) scatter, in two passes: wavefronts 0-3 (the pairs' first spans), then 4-7 (their second spans) ----
    for (int pass = 0; pass < 2; ++pass) {
      if ((wave >> 2) == pass) {
        if (in_regs) {
#pragma unroll
          for (int j = 0; j < TILE_EDGE_REGS; ++j)
            if (ed[j] >= 0) lds_col[my_half(atomicAdd(&my_cnt[ed[j]], my_inc))] = (unsigned short)es[j];
        } else {
          const int S = span_len(Et);
#pragma unroll 4
          for (int i0 = 0; i0 < S; i0 += 64) {
            const int i = my_span * S + i0 + lane;
            if (i >= Et) continue;
            const int64_t s = t.src[(int64_t)e0 + i] - n0, d = t.dst[(int64_t)e0 + i] - n0;
            if (s < 0 || s >= T || d < 0 || d >= T) continue;
            lds_col[my_half(atomicAdd(&my_cnt[(int)d], my_inc))] = (unsigned short)s;
          }
        }
      }
      if (pass == 0) lds_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's pieces of the basis rows have landed ...
    lds_barrier();                                       // ... and so have everybody else's
    if (has_next) request_edges(tl, k + (int)gridDim.x); // the next tile's edges travel during this tile's rows

    // ---- (E) rows: one lane group per row, G rows per wavefront and pass; weightings two passes ahead ----
    for (int r0 = 0; r0 < T; r0 += RPP) {
      const int r = r0 + wave * G + g;             // local row of this lane group
      const bool row_ok = r < T;
      const int row = n0 + (row_ok ? r : 0);
      f4 wpre[2] = {wn0[0], wn0[1]};
      wn0[0] = wn1[0]; wn0[1] = wn1[1];
      if (r0 + 2 * RPP < T) load_weightings_row<LPR_LOG2, C>(a, q, n0 + r + 2 * RPP, r + 2 * RPP < T, wn1);
      const int start = row_ok ? lds_rowptr[r] : 0;
      const int nd = row_ok ? lds_rowptr[r + 1] - start : 0;
      int maxd = nd;
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1) maxd = max(maxd, bperm((lane ^ off) << 2, maxd));
      maxd = __builtin_amdgcn_readfirstlane(maxd);
      const float dis_i = (want_dis && row_ok) ? lds_dis[r] : 0.f;
      const bool has_self = row_ok && (C::loops_all(a) || row <= max_index);
      const bool want_self = looped_any && has_self && q < C::slots(a);
      f4 vself = f4{0.f, 0.f, 0.f, 0.f};
      if (in_lds) { if (want_self) vself = lds_bases4[r * ldb4 + q]; }
      else vself = load_slot(R.bases, want_self ? (unsigned)row * row_bytes + slot_off : OOB);

      FAcc<NEED> acc;
      acc.init();
      if constexpr (NEED & NEED_SQ) {   // the variance's shift: the row's first entry in the tile's CSR (FAcc::sh)
        const bool want = row_ok && nd > 0 && q < C::slots(a);
        const int first = want ? (int)lds_col[start] : 0;
        if (in_lds) { if (want) acc.sh = lds_bases4[first * ldb4 + q]; }
        else acc.sh = load_slot(R.bases, want ? (unsigned)(n0 + first) * row_bytes + slot_off : OOB);
      }
      int nself = 0;
      const int n_valid = q < C::slots(a) ? nd : 0;
      for (int ts = 0; ts < maxd; ts += LPR) {
        const bool pv = ts + q < nd;
        const int jj = pv ? (int)lds_col[start + ts + q] : 0;
        const float dd = (pv && want_dis) ? lds_dis[jj] : 0.f;
        if (looped_any) {
          const unsigned long long sb = __ballot(pv && jj == r);
          nself += __popcll((sb >> (g << LPR_LOG2)) & ((LPR == 64) ? ~0ull : ((1ull << LPR) - 1ull)));
        }
        const int cnt = min(LPR, maxd - ts);
        for (int t0 = 0; t0 < cnt; t0 += FU) {
          f4 v[FU];
          float w[FU];
          bool in_x[FU];
#pragma unroll
          for (int u = 0; u < FU; ++u) {
            const int addr = grp_addr + ((t0 + u) << 2);
            const int j = bperm(addr, jj);
            const bool is_self = j == r;
            in_x[u] = (ts + t0 + u < n_valid) && !(C::xl(a) && is_self);
            if (in_lds) v[u] = in_x[u] ? lds_bases4[j * ldb4 + q] : f4{0.f, 0.f, 0.f, 0.f};
            else v[u] = load_slot(R.bases, in_x[u] ? (unsigned)(n0 + j) * row_bytes + slot_off : OOB);
            w[u] = bperm(addr, dd) * dis_i;
            if (C::yl(a) && !C::xl(a)) w[u] = is_self ? 0.f : w[u];
          }
#pragma unroll
          for (int u = 0; u < FU; ++u) fold<NEED>(acc, v[u], w[u], in_x[u], start + ts + t0 + u);
        }
      }
      int ln = lane;
      asm volatile("" : "+v"(ln));
      finish_group<LPR_LOG2, HPB, NEED, C>(a, R, ln, row, row_ok, acc, nd, nself, dis_i, vself, has_self, wpre, true, lds_w,
                                           lds_bias, lds_scale);
    }
    lds_barrier();   // every wavefront is done with the tile's LDS areas
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct TileLds {
  size_t strips, total;
  int off_bases, off_col, off_rowptr, off_cnt, off_ns, off_dis_raw, off_dis_looped;
};

static TileLds tile_lds(const AggArgs& a, int tlds, int tmax, int emax) {
  TileLds L;
  auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  size_t at = up16((size_t)TILE_WAVES * a.lds_floats_per_wave * sizeof(float));
  L.strips = at;
  L.off_bases = (int)at; at += up16((size_t)tlds * a.ldb * 4);
  L.off_col = (int)at; at += up16((size_t)emax * 2);
  L.off_rowptr = (int)at; at += up16((size_t)(tmax + 1) * 4);
  L.off_cnt = (int)at; at += up16((size_t)tmax * 4 * 2);      // a 16-bit counter / cursor per pair of wavefronts and row
  L.off_ns = (int)at; at += up16((size_t)tmax * 4);
  L.off_dis_raw = (int)at; at += up16((size_t)tmax * 4);
  L.off_dis_looped = (int)at; at += up16((size_t)tmax * 4);
  L.total = at;
  return L;
}

constexpr size_t TILE_LDS_BUDGET = (160 * 1024) / TILE_WGS_PER_CU - 512;   // per workgroup (two per CU)

template <int LPR_LOG2, int HPB, int NEED, class C>
static int launch_tile_one(const AggArgs& a, const TileArgs& t, unsigned grid, size_t lds, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&agg_tile_kernel<LPR_LOG2, HPB, NEED, C>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_LDS_BUDGET);
    if (e != hipSuccess) { set_last_error("hipFuncSetAttribute(agg_tile_kernel)", e); return EGC_ERR_HIP; }
    attr_set = true;
  }
  agg_tile_kernel<LPR_LOG2, HPB, NEED, C><<<grid, TILE_THREADS, lds, stream>>>(a, t);
  EGC_LAUNCH_CHECK("agg_tile_kernel");
  return EGC_OK;
}

template <int LPR_LOG2>
static int launch_tile_rt(const AggArgs& a, const TileArgs& t, int need, unsigned grid, size_t lds, hipStream_t stream) {
  const int hpb = (a.H + a.B - 1) / a.B;
  if (need == 0) {
    if (hpb <= 1) return launch_tile_one<LPR_LOG2, 1, 0, RtCfg>(a, t, grid, lds, stream);
    if (hpb <= 2) return launch_tile_one<LPR_LOG2, 2, 0, RtCfg>(a, t, grid, lds, stream);
    return launch_tile_one<LPR_LOG2, 4, 0, RtCfg>(a, t, grid, lds, stream);
  }
  if (hpb <= 1) return launch_tile_one<LPR_LOG2, 1, NEED_SQ | NEED_MN, RtCfg>(a, t, grid, lds, stream);
  if (hpb <= 2) return launch_tile_one<LPR_LOG2, 2, NEED_SQ | NEED_MN, RtCfg>(a, t, grid, lds, stream);
  return launch_tile_one<LPR_LOG2, 4, NEED_SQ | NEED_MN, RtCfg>(a, t, grid, lds, stream);
}

// the layer configurations with their constants compiled in (the same idea as EGC_STATIC_CFG of the fast kernel)
template <class C, int LPR_LOG2, int HPB, int NEED>
static bool try_tile_static(const AggArgs& a, const TileArgs& t, int h, int b, int l, int na, unsigned agg, bool xl, bool yl,
                            bool loops_all, unsigned grid, size_t lds, hipStream_t stream, int* status) {
  unsigned packed = 0;
  for (int k = 0; k < a.A; ++k) packed |= (unsigned)a.aggr[k] << (3 * k);
  if (a.H != h || a.B != b || a.L != l || a.Ls != l || a.A != na || packed != agg || a.act != EGC_ACT_NONE ||
      (a.x_looped != 0) != xl || (a.y_looped != 0) != yl || (a.loops_all != 0) != loops_all)
    return false;
  *status = launch_tile_one<LPR_LOG2, HPB, NEED, C>(a, t, grid, lds, stream);
  return true;
}

// nodes whose basis rows fit the LDS area of one workgroup, next to CSR areas for (tmax nodes, emax edges)
int tile_capacity(const AggArgs& a_in, int tmax, int emax) {
  AggArgs a = a_in;
  int best = 0;
  for (int tlds = 16; tlds <= tmax; tlds += 16) {
    if (tile_lds(a, tlds, tmax, emax).total <= TILE_LDS_BUDGET) best = tlds; else break;
  }
  return best;
}

int launch_tile(AggArgs a, TileArgs t, int n_tiles, hipStream_t stream) {
  const int lpr = a.slots <= 16 ? 16 : a.slots <= 32 ? 32 : 64;
  const int G = 64 / lpr;
  a.lanes_pb = a.Ls / 4;
  a.magic_P = (unsigned)(((uint64_t)1 << 32) / (uint64_t)a.lanes_pb) + 1u;
  if ((a.lanes_pb & (a.lanes_pb - 1)) == 0) {
    int lg = 0;
    while ((4 << lg) < a.Ls) ++lg;
    a.lpb_log2 = lg;
  } else {
    a.lpb_log2 = -1;
  }
  a.need_mean = a.need_var = 0;
  int need = 0;
  for (int k = 0; k < a.A; ++k) {
    if (a.aggr[k] == EGC_AGGR_MEAN || a.aggr[k] == EGC_AGGR_VAR || a.aggr[k] == EGC_AGGR_STD) a.need_mean = 1;
    if (a.aggr[k] == EGC_AGGR_VAR || a.aggr[k] == EGC_AGGR_STD) { a.need_var = 1; need |= NEED_SQ; }
    if (a.aggr[k] == EGC_AGGR_MIN) need |= NEED_MN;
  }
  a.w_lds_stride = (a.W + 3) & ~3;
  a.bias_lds_floats = (a.H * a.Ls + 3) & ~3;
  a.lds_floats_per_wave = (a.post_scale != nullptr ? 2 : 1) * a.bias_lds_floats + G * a.w_lds_stride;
  const TileLds L = tile_lds(a, t.tlds, t.tmax, t.emax);
  if (L.total > TILE_LDS_BUDGET || t.tmax > TILE_MAX_NODES || t.tlds > t.tmax || t.tlds < 1 || t.emax > 65535) return EGC_ERR_UNSUPPORTED;   // (16-bit cursors)
  t.off_bases = L.off_bases; t.off_col = L.off_col; t.off_rowptr = L.off_rowptr; t.off_cnt = L.off_cnt; t.off_ns = L.off_ns;
  t.off_dis_raw = L.off_dis_raw; t.off_dis_looped = L.off_dis_looped;
  const unsigned grid = (unsigned)std::min(n_tiles, 256 * TILE_WGS_PER_CU);   // persistent: n_tiles = upper bound of the tile count
  if (getenv("EGC_NO_STATIC_CFG") == nullptr) {
    int status = EGC_OK;
    constexpr int S = EGC_AGGR_SUM, M = EGC_AGGR_MEAN, X = EGC_AGGR_MAX, Y = EGC_AGGR_SYMNORM;
    // EGConv EGC-M north star (configs 3 / 4 of BASELINE.json): d=128, H=8, B=4, sum+mean+max+symnorm, loops on every node
    if (try_tile_static<StCfg<8, 4, 16, 4, agg_pack(S, M, X, Y), EGC_ACT_NONE, true, true, true>, 4, 2, 0>(
            a, t, 8, 4, 16, 4, agg_pack(S, M, X, Y), true, true, true, grid, L.total, stream, &status)) return status;
    // EfficientGraphConv EGC-M / EGC-S at d=128 (symadd looped, the others raw)
    if (try_tile_static<StCfg<8, 4, 16, 3, agg_pack(Y, X, M), EGC_ACT_NONE, false, true, true>, 4, 2, 0>(
            a, t, 8, 4, 16, 3, agg_pack(Y, X, M), false, true, true, grid, L.total, stream, &status)) return status;
    if (try_tile_static<StCfg<8, 4, 16, 1, agg_pack(Y), EGC_ACT_NONE, false, true, true>, 4, 2, 0>(
            a, t, 8, 4, 16, 1, agg_pack(Y), false, true, true, grid, L.total, stream, &status)) return status;
  }
  switch (lpr) {
    case 16: return launch_tile_rt<4>(a, t, need, grid, L.total, stream);
    case 32: return launch_tile_rt<5>(a, t, need, grid, L.total, stream);
    default: return launch_tile_rt<6>(a, t, need, grid, L.total, stream);
  }
}

int launch_tile_simple(AggArgs a, const int4* tiles, const int* n_tiles_dev, int n_tiles_bound, int tlds, int tmax, int emax,
                       const int64_t* src, const int64_t* dst, const int* max_index, int32_t* status, int32_t* host_flag,
                       hipStream_t stream) {
  TileArgs t = {};
  t.tiles = tiles; t.n_tiles = n_tiles_dev; t.n_tiles_cap = n_tiles_bound; t.src = src; t.dst = dst; t.max_index = max_index; t.status = status;
  t.host_flag = host_flag;
  t.tlds = tlds; t.tmax = tmax; t.emax = emax;
  return launch_tile(a, t, n_tiles_bound, stream);
}

int launch_tile_plan(const int64_t* ptr, int64_t n_graphs, const int64_t* dst, int64_t n_edges, int64_t n_nodes, int slot,
                     int n_slots, int4* tiles, int* count, const int64_t* edge_ptr, hipStream_t stream) {
  EGC_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int), stream));
  if (edge_ptr != nullptr && n_graphs > 0) {
    tile_plan_graphs_kernel<<<(unsigned)((n_graphs + 255) / 256), 256, 0, stream>>>(ptr, edge_ptr, n_graphs, n_edges, n_nodes, slot,
                                                                                tiles, count, n_slots);
    EGC_LAUNCH_CHECK("tile_plan_graphs_kernel");
    return EGC_OK;
  }
  tile_plan_kernel<<<(unsigned)((n_slots + 3) / 4), 256, 0, stream>>>(ptr, n_graphs, dst, n_edges, n_nodes, slot, n_slots, tiles, count,
                                                                      edge_ptr);
  EGC_LAUNCH_CHECK("tile_plan_kernel");
  return EGC_OK;
}

}  // namespace egc
