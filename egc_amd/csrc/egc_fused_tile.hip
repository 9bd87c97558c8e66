// Batches of small graphs (BASELINE configs 1 / 3 / 4), the WHOLE layer in one launch: for a tile of whole graphs
//
//     x rows -> fp16x2 planes -> [bases | weightings] on the matrix cores -> LDS -> CSR of the tile's edges in LDS ->
//     multi-aggregator reduction from LDS -> per-head combine -> (+ bias, BatchNorm(eval) / ReLU / residual) -> out
//
// A tile of whole graphs is closed under "in-neighbour of": neither `bases` nor `weightings` ever exist in memory, the
// launch reads x and the edge list once and writes `out` once (SURVEY.md section 8 f2 + f3).  Reference call sites replaced:
// torch.matmul(x, bases_weight) and comb_weight(x) (layers.py:97-101,110; optimized_layers.py:180-182), gcn_norm /
// add_remaining_self_loops (optimized_layers.py:127-175), MessagePassing.propagate + the per-aggregator scatters
// (optimized_layers.py:186-249; layers.py:165-219), the bmm / broadcast combine and bias (optimized_layers.py:195-208;
// layers.py:127-138) and the callers' BatchNorm(eval) / ReLU / residual tail (zinc/models.py:66-73, cifar/models.py:64-71).
//
// Workgroup = 16 wavefronts, ONE per CU (the LDS image of a tile is what limits it), persistent over its share of the
// batch:
//   plan   inside the launch: workgroup b owns the graphs whose first node lies in [b N / nWG, (b + 1) N / nWG) (two
//          64-ary searches in the graph offsets, once); its tiles are greedy runs of those graphs of at most `tcap` nodes,
//          found one tile ahead by wavefront 15 (edge range: the caller's edge offsets, else two searches in the
//          destination row) -- no plan launch, no tile list in memory.
//   GEMM   wavefronts 0-11 each keep ONE 16-column tile of the packed weight planes (k = 128, two fp16 planes: 32 VGPRs)
//          for the whole launch; wavefronts 12-15 fetch x in 16-row chunks (FT_DEPTH chunks in flight, registers), split
//          every row into the two fp16 planes (row scale = power of two, egc_gemm_f16x2.hip's scheme) and stage them in a
//          two-deep LDS ring whose 16-byte pieces are XOR-swizzled by the row (conflict-free A-operand reads without
//          padding); v_mfma_f32_16x16x32_f16, three products per k-step (xh wh, xl wh, xh wl), fp32 accumulate; the D tile
//          is scaled, biased, passed through the weight nonlinearity and written to the LDS image of the tile:
//          bases [T][ldb] and weightings [T][H B 4].
//   CSR    as egc_aggregate_tile.hip: in-degrees by LDS atomics, wavefront scan, scatter; both deg^-1/2 tables.
//   rows   one lane group per row, neighbours gathered from LDS, register epilogue of the fast kernel family
//          (finish_group, weightings read from the LDS image).
// Tiles beyond the LDS image (a single graph with more than `tcap` nodes, more than `emax` edges) and edges that leave
// their tile raise *status and the sticky host flag; the host routes batches whose declared largest graph exceeds the
// capacity to the two-launch path.
#include <algorithm>

#include "egc_aggregate_fast_dev.h"

namespace egc {

typedef _Float16 ft_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 ft_h2 __attribute__((ext_vector_type(2)));
typedef float ft_f2 __attribute__((ext_vector_type(2)));
typedef unsigned int ft_u2 __attribute__((ext_vector_type(2)));
typedef unsigned short ft_u16;

constexpr int FT_THREADS = 1024;
constexpr int FT_WAVES = FT_THREADS / 64;
constexpr int FT_MFMA_WAVES = 12;       // 16-column tiles of the virtual column space [bases (ldb) | weightings (W)]
constexpr int FT_FIRST_HELPER = 12;     // wavefronts 12-15: x rows -> planes
constexpr int FT_HELPER_THREADS = (FT_WAVES - FT_FIRST_HELPER) * 64;
constexpr int FT_KP = 128;              // k extent of the register-resident weight tiles (F_in <= 128, zero beyond)
constexpr int FT_CHUNK = 16;            // rows per GEMM step (one MFMA tile)
#ifndef EGC_FT_DEPTH
#define EGC_FT_DEPTH 4
#endif
constexpr int FT_DEPTH = EGC_FT_DEPTH;  // x chunks in flight
constexpr int FT_EDGE_REGS = 3;         // edges per thread kept in registers (16:16 packed local ids)
constexpr int FT_MAX_NODES = 2048;      // local ids are 16-bit, the scan is one wavefront
constexpr int FT_NV = FT_MFMA_WAVES * 16;
constexpr int FT_PLANE_BYTES = FT_CHUNK * FT_KP * 2;     // one plane of one chunk
constexpr int FT_PLANES_BYTES = 2 * 2 * FT_PLANE_BYTES;  // [2 buffers][2 planes]

struct FusedTileArgs {
  const int64_t* ptr;        // node offsets of the graphs [G + 1]
  const int64_t* edge_ptr;   // their edge offsets [G + 1], or nullptr
  int64_t n_graphs;
  const int64_t* src;
  const int64_t* dst;
  int64_t n_edges;
  const int* max_index;      // device scalar (layers with loops_all == 0), or nullptr
  int32_t* status;
  int32_t* host_flag;
  const float* x;
  const ft_u16* packed;      // [12][4][2][64][8] fp16 weight fragments, float col_inv[192], float col_bias[192]
  int F_in;
  int n_ct;                  // column tiles in use = ceil((ldb + W) / 16)
  int tcap, emax;            // LDS image: rows of bases / weightings, entries of the CSR
  int wl_floats;             // floats per weightings row in LDS: H * B * 4
  int off_rec, off_planes, off_rowinv, off_bases, off_wt, off_col, off_rowptr, off_cnt, off_ns, off_dis_raw, off_dis_looped;
};

// packed[column tile][k-step of 32][plane][lane][8]: lane 16 (k % 32 / 8) + column % 16 holds k = 32 s + 8 (lane / 16) ..+7 of
// column 16 ct + lane % 16 -- the B operand of v_mfma_f32_16x16x32_f16 as it is loaded, one KiB per (ct, s, plane).
// Column scales and the two planes as pack_f16x2_kernel (egc_gemm_f16x2.hip).
__global__ void __launch_bounds__(64) ft_pack_kernel(const float* __restrict__ wcat, const float* __restrict__ bcat, int K,
                                                      int F_g, int W, int ldb, ft_u16* __restrict__ packed) {
  const int v = blockIdx.x;
  const int lane = threadIdx.x;
  const int ncol = F_g + W;
  const int src = (v < F_g) ? v : ((v < ldb || v >= ldb + W) ? -1 : v - ldb + F_g);
  unsigned amax = 0;
  if (src >= 0)
    for (int k = lane; k < K; k += 64) amax = max(amax, __float_as_uint(wcat[(int64_t)k * ncol + src]) & 0x7fffffffu);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, d));
  unsigned be = amax >> 23;
  be = be > 253u ? 253u : be;
  const float scale = __uint_as_float((254u - be) << 23);
  const float inv = __uint_as_float(be << 23);
  for (int k = lane; k < FT_KP; k += 64) {
    const float w = (src >= 0 && k < K) ? wcat[(int64_t)k * ncol + src] * scale : 0.f;
    const _Float16 h = (_Float16)w;
    const _Float16 l = (_Float16)((w - (float)h) * 2048.f);
    const int64_t base = ((((int64_t)(v >> 4) * 4 + (k >> 5)) * 2) * 64 + 16 * ((k & 31) >> 3) + (v & 15)) * 8 + (k & 7);
    packed[base] = __builtin_bit_cast(ft_u16, h);
    packed[base + 64 * 8] = __builtin_bit_cast(ft_u16, l);
  }
  if (lane == 0) {
    float* tail = reinterpret_cast<float*>(packed + (int64_t)FT_MFMA_WAVES * 4 * 2 * 64 * 8);
    tail[v] = inv;
    const int wcol = v - ldb;
    tail[FT_NV + v] = (bcat != nullptr && wcol >= 0 && wcol < W) ? bcat[wcol] : 0.f;
  }
}

size_t fused_tile_pack_bytes() { return (size_t)FT_MFMA_WAVES * 4 * 2 * 64 * 8 * sizeof(ft_u16) + 2 * FT_NV * sizeof(float); }

int fused_tile_pack(const float* wcat, const float* bcat, int f_in, int f_g, int w_cols, int ldb, void* packed, hipStream_t stream) {
  ft_pack_kernel<<<FT_NV, 64, 0, stream>>>(wcat, bcat, f_in, f_g, w_cols, ldb, (ft_u16*)packed);
  EGC_LAUNCH_CHECK("ft_pack_kernel");
  return EGC_OK;
}

// first index i in [0, n) with arr[i] >= key (n if none), by HALF a wavefront (lanes [32 h, 32 h + 32) share `key`), as
// egc_aggregate_tile.hip: the two halves of a wavefront run two searches side by side; deterministic on unsorted input.
__device__ inline int64_t ft_half_wave_lower_bound(const int64_t* __restrict__ arr, int64_t n, int64_t key, int lane) {
  const int l32 = lane & 31, sh = lane & 32;
  int64_t lo = 0, hi = n;
  while (__ballot(hi - lo > 32) != 0) {
    const bool live = hi - lo > 32;
    const int64_t step = live ? (hi - lo + 31) / 32 : 1;
    const int64_t i = lo + (int64_t)l32 * step;
    const bool ge = (live && i < hi) ? arr[i] >= key : true;
    const unsigned m = (unsigned)(__ballot(ge) >> sh);
    if (!live) continue;
    const int f = __ffs((int)m) - 1;
    if (f < 0) { lo = lo + 31 * step + 1; if (lo > hi) lo = hi; continue; }
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

__device__ inline void ft_error(const FusedTileArgs& t, int code) {
  atomicOr(t.status, code);
  if (t.host_flag != nullptr) *(volatile int32_t*)t.host_flag = 1;
}

// largest magnitude of a row = 32 consecutive lanes (bit pattern of a non-negative float), as egc_gemm_f16x2.hip
__device__ inline unsigned ft_row_amax(const f4 v) {
  float m;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|\n\tv_max_f32 %0, |%4|, %0" : "=&v"(m) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
  unsigned a = __float_as_uint(m);
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x141, 0xf, 0xf, true));  // row_half_mirror
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x140, 0xf, 0xf, true));  // row_mirror
  return max(a, (unsigned)__builtin_amdgcn_ds_swizzle((int)a, 0x401F));                 // lane ^ 16
}

template <int LPR_LOG2, int HPB, int NEED, class C>
__global__ void __launch_bounds__(FT_THREADS) fused_tile_kernel(AggArgs a, FusedTileArgs t) {
  constexpr int LPR = 1 << LPR_LOG2, G = 64 / LPR;
  extern __shared__ float smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int g = lane >> LPR_LOG2;
  const int q = lane & (LPR - 1);
  const int F_out = C::F_out(a);
  const bool is_helper = wave >= FT_FIRST_HELPER;
  const bool is_mfma = wave < t.n_ct;

  // ---- LDS image ----
  char* base = reinterpret_cast<char*>(smem);
  float* lds_bias = smem;                                   // [bias (x scale + shift)][scale]: one copy for the workgroup
  const bool post = a.post_scale != nullptr;
  float* lds_scale = lds_bias + a.bias_lds_floats;
  int* lds_rec = reinterpret_cast<int*>(base + t.off_rec);  // [2][8]: n0, n1, e0, e1 of the current / next tile
  char* lds_planes = base + t.off_planes;                   // [2 buffers][2 planes][16 rows][128 fp16], 16-byte pieces swizzled
  float* lds_rowinv = reinterpret_cast<float*>(base + t.off_rowinv);   // [2][16]
  f4* lds_bases4 = reinterpret_cast<f4*>(base + t.off_bases);
  float* lds_wt = reinterpret_cast<float*>(base + t.off_wt);
  unsigned short* lds_col = reinterpret_cast<unsigned short*>(base + t.off_col);
  int* lds_rowptr = reinterpret_cast<int*>(base + t.off_rowptr);
  int* lds_cnt = reinterpret_cast<int*>(base + t.off_cnt);
  int* lds_ns = reinterpret_cast<int*>(base + t.off_ns);
  float* lds_dis_raw = reinterpret_cast<float*>(base + t.off_dis_raw);
  float* lds_dis_looped = reinterpret_cast<float*>(base + t.off_dis_looped);
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  for (int o = tid; o < C::H(a) * C::Ls(a); o += FT_THREADS) {
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

  // ---- this wavefront's 16-column tile of the packed weights: its column's inverse scale and bias, and where its column
  //      goes in the LDS image.  The tile itself (both planes of 128 x 16 as B operands: lane -> column 16 wave + lane % 16,
  //      k = 32 s + 8 (lane / 16) ..+7) is fetched again for every tile of graphs, from L2: kept across the rows phase its 32
  //      registers push that phase's working set out of the register file (measured at compile time: 129 spilled VGPRs) ----
  // ONE register array serves both roles (the allocator cannot overlay two arrays whose live ranges only differ by the
  // wavefront's role): wavefronts 0-11 hold their weight tile in it during the GEMM phase -- u[2 s + p] = k-step s, plane p --
  // wavefronts 12-15 the x chunks in flight -- u[2 d + i] = piece i of the chunk in register set d
  static_assert(FT_DEPTH == 4, "the x prefetch shares the 8 x 16-byte registers of a weight tile");
  f4 u[8];
  float col_inv = 0.f, col_bias = 0.f;
  int dst_off = -1, dst_stride = 0;        // byte offset inside a row of the image area / bytes between its rows
  bool dst_act = false;
  if (is_mfma) {
    const float* tail = reinterpret_cast<const float*>(t.packed + (int64_t)FT_MFMA_WAVES * 4 * 2 * 64 * 8);
    const int v = 16 * wave + (lane & 15);
    col_inv = tail[v];
    col_bias = tail[FT_NV + v];
    if (v < a.ldb) {
      dst_off = t.off_bases + v * 4;
      dst_stride = a.ldb * 4;
    } else if (v - a.ldb < C::W(a)) {
      const int wc = v - a.ldb;
      const int hb = wc / C::A(a);
      dst_off = t.off_wt + (hb * 4 + (wc - hb * C::A(a))) * 4;
      dst_stride = t.wl_floats * 4;
      dst_act = true;
    }
  }

  // ---- the workgroup's graphs: [g_lo, g_hi) = those whose first node lies in its share of [0, N) ----
  const int64_t Gn = t.n_graphs;
  int64_t cur_g = 0, g_hi = 0;             // (meaningful in wavefront 15 only)
  const int ldb4 = a.ldb >> 2;
  FastRsrc R;
  R.bases = bases_rsrc(a);
  R.out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (unsigned)a.n_nodes * (unsigned)F_out * 4u, 0x00020000);
  R.res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual != nullptr ? a.residual : a.out), 0,
                                            (unsigned)a.n_nodes * (unsigned)F_out * 4u, 0x00020000);
  const bool looped_any = C::xl(a) || C::yl(a);
  const float* lds_dis = C::yl(a) ? lds_dis_looped : lds_dis_raw;
  const bool want_dis = a.dis != nullptr;
  const int max_index = (!C::loops_all(a) && t.max_index != nullptr) ? *t.max_index : 0x7fffffff;
  constexpr int RPP = FT_WAVES * G;
  const int grp_addr = (g << LPR_LOG2) << 2;

  // next tile of this workgroup -> lds_rec[slot] (wavefront 15, all lanes): graphs [cur_g, next_g) with at most tcap nodes
  auto plan_tile = [&](int slot) {
    int n0 = 0, n1 = 0, e0 = 0, e1 = 0;
    if (cur_g < g_hi) {
      const int64_t p0 = t.ptr[cur_g];
      const int64_t gi = cur_g + 1 + lane;
      const int64_t pv = t.ptr[gi <= g_hi ? gi : g_hi];
      const bool ok = gi <= g_hi && pv - p0 <= (int64_t)t.tcap && pv >= p0;
      const unsigned long long m = __ballot(ok);
      int n_ok = m == ~0ull ? 64 : __ffsll((long long)~m) - 1;     // graphs that fit (a prefix: ptr is non-decreasing)
      n_ok = n_ok < 1 ? 1 : n_ok;                                   // a single graph beyond the capacity: reported by the tile
      // offsets that decrease inside the run: reported (the node ranges of the tiles then no longer partition [0, N))
      const int64_t pprev = __shfl_up(pv, 1);
      const bool dec = gi <= g_hi && lane < n_ok && pv < (lane == 0 ? p0 : pprev);
      if (__ballot(dec) != 0 && lane == 0) ft_error(t, 1);
      const int64_t next_g = cur_g + n_ok;
      int64_t pn = __shfl(pv, n_ok - 1);
      int64_t a0 = p0, a1 = pn;
      a0 = a0 < 0 ? 0 : (a0 > a.n_nodes ? a.n_nodes : a0);
      a1 = a1 < a0 ? a0 : (a1 > a.n_nodes ? a.n_nodes : a1);
      int64_t b0, b1;
      if (t.edge_ptr != nullptr) {
        b0 = t.edge_ptr[cur_g];
        b1 = t.edge_ptr[next_g];
        if (b1 < b0 && lane == 0) ft_error(t, 1);
      } else {   // the two ends side by side in the two halves of the wavefront
        const int64_t r = ft_half_wave_lower_bound(t.dst, t.n_edges, lane < 32 ? a0 : a1, lane);
        b0 = __shfl(r, 0);
        b1 = __shfl(r, 32);
      }
      if (cur_g == 0) b0 = 0;
      if (next_g >= Gn) b1 = t.n_edges;
      b0 = b0 < 0 ? 0 : (b0 > t.n_edges ? t.n_edges : b0);
      b1 = b1 < b0 ? b0 : (b1 > t.n_edges ? t.n_edges : b1);
      n0 = (int)a0; n1 = (int)a1; e0 = (int)b0; e1 = (int)b1;
      cur_g = next_g;
      if (lane == 0) { lds_rec[slot * 8 + 4] = 1; }
    } else if (lane == 0) {
      lds_rec[slot * 8 + 4] = 0;           // no further tile
    }
    if (lane == 0) {
      lds_rec[slot * 8 + 0] = n0; lds_rec[slot * 8 + 1] = n1; lds_rec[slot * 8 + 2] = e0; lds_rec[slot * 8 + 3] = e1;
    }
  };

  if (wave == FT_WAVES - 1) {
    const int64_t N = a.n_nodes;
    const int64_t nb = gridDim.x, b = blockIdx.x;
    const int64_t k_lo = b * N / nb, k_hi = (b + 1) * N / nb;
    const int64_t r = ft_half_wave_lower_bound(t.ptr, Gn + 1, lane < 32 ? k_lo : k_hi, lane);
    int64_t lo = __shfl(r, 0), hi = __shfl(r, 32);
    if (b == 0) lo = 0;
    if (b == nb - 1) hi = Gn;
    lo = lo > Gn ? Gn : lo;
    hi = hi > Gn ? Gn : hi;
    cur_g = lo;
    g_hi = hi < lo ? lo : hi;
    if (b == 0 && lane == 0 && Gn > 0 && (t.ptr[0] != 0 || t.ptr[Gn] != N)) ft_error(t, 1);   // offsets that do not cover [0, N)
    plan_tile(0);
  }
  lds_barrier();

  // ---- edges of a tile: FT_EDGE_REGS per thread, local ids packed 16:16, -1 = absent / outside the tile (reported) ----
  int epk[FT_EDGE_REGS];
  bool bad = false;
  int edges_of = -1;
  auto request_edges = [&](int rn0, int rn1, int re0, int re1, int which) {
    const int tT = rn1 - rn0, tE = re1 - re0;
    bad = false;
    edges_of = which;
#pragma unroll
    for (int j = 0; j < FT_EDGE_REGS; ++j) epk[j] = -1;
    if (tT <= 0 || tT > t.tcap || tE > FT_THREADS * FT_EDGE_REGS) return;
#pragma unroll
    for (int j = 0; j < FT_EDGE_REGS; ++j) {
      const int i = tid + j * FT_THREADS;
      if (i < tE) {
        const int64_t s = t.src[(int64_t)re0 + i] - rn0, d = t.dst[(int64_t)re0 + i] - rn0;
        if (s < 0 || s >= tT || d < 0 || d >= tT) bad = true;
        else epk[j] = (int)((s << 16) | d);
      }
    }
  };

  // helper wavefronts: 16-byte pieces p = ht + 256 i (i = 0, 1) of a 16-row chunk <-> (row p / 32, k 4 (p % 32))
  const int ht = tid - FT_FIRST_HELPER * 64;

  for (int it = 0;; ++it) {
    const int slot = it & 1;
    if (__builtin_amdgcn_readfirstlane(lds_rec[slot * 8 + 4]) == 0) break;
    const int n0 = __builtin_amdgcn_readfirstlane(lds_rec[slot * 8 + 0]);
    const int T = __builtin_amdgcn_readfirstlane(lds_rec[slot * 8 + 1]) - n0;
    const int e0 = __builtin_amdgcn_readfirstlane(lds_rec[slot * 8 + 2]);
    const int Et = __builtin_amdgcn_readfirstlane(lds_rec[slot * 8 + 3]) - e0;
    const bool tile_ok = T > 0 && T <= t.tcap && Et <= t.emax;
    if (!tile_ok && (T > 0 || Et > 0) && tid == 0) ft_error(t, T <= 0 ? 1 : 2);
    const bool in_regs = Et <= FT_THREADS * FT_EDGE_REGS;
    const int nch = tile_ok ? (T + FT_CHUNK - 1) / FT_CHUNK : 0;
    // the tile's rows of x through a descriptor of their own (no 4 GiB limit on x; rows beyond the tile read as 0)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(t.x + (int64_t)n0 * t.F_in), 0, (unsigned)(tile_ok ? T : 0) * (unsigned)t.F_in * 4u, 0x00020000);
    // (`hv`: the helper thread's index through an opaque copy per use -- otherwise every unrolled body's address arithmetic
    // is hoisted out of the tile loop and lives in registers across the rows phase: 129 spilled VGPRs)
    auto x_load = [&](f4& d0, f4& d1, int c, int hv) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int p = hv + FT_HELPER_THREADS * i;
        const int row = FT_CHUNK * c + (p >> 5), k4 = (p & 31) * 4;
        const bool ok = row < T && k4 < t.F_in;
        (i == 0 ? d0 : d1) = load_slot(xrs, ok ? (unsigned)(row * t.F_in + k4) * 4u : OOB);
      }
    };
    auto split = [&](const f4 v0, const f4 v1, int buf, int hv) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f4 v = i == 0 ? v0 : v1;
        const int p = hv + FT_HELPER_THREADS * i;
        const int row = p >> 5, k4 = (p & 31) * 4;
        unsigned e = ft_row_amax(v) & 0x7f800000u;
        e = min(max(e, 13u << 23), 253u << 23);
        const float sc = __uint_as_float(0x7f000000u - e);                  // 2^-e
        const float sc2k = __uint_as_float(0x7f000000u + (11u << 23) - e);  // 2^(11-e)
        const ft_h2 h01 = __builtin_convertvector(ft_f2{v.x * sc, v.y * sc}, ft_h2);
        const ft_h2 h23 = __builtin_convertvector(ft_f2{v.z * sc, v.w * sc}, ft_h2);
        ft_h2 l01, l23;
        l01[0] = (_Float16)__builtin_fmaf((float)h01[0], -2048.f, v.x * sc2k);
        l01[1] = (_Float16)__builtin_fmaf((float)h01[1], -2048.f, v.y * sc2k);
        l23[0] = (_Float16)__builtin_fmaf((float)h23[0], -2048.f, v.z * sc2k);
        l23[1] = (_Float16)__builtin_fmaf((float)h23[1], -2048.f, v.w * sc2k);
        char* dstp = lds_planes + buf * (2 * FT_PLANE_BYTES) + row * (FT_KP * 2) + ((((k4 >> 3) ^ row) & 15) << 4) + (k4 & 7) * 2;
        *reinterpret_cast<ft_u2*>(dstp) = ft_u2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
        *reinterpret_cast<ft_u2*>(dstp + FT_PLANE_BYTES) = ft_u2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
        lds_rowinv[buf * FT_CHUNK + row] = __uint_as_float(e);              // 2^e (the 32 lanes of a row write the same word)
      }
    };

    // ---- (A) requests: edges (unless they came during the previous tile's rows), the first x chunks; counters zeroed;
    //      wavefront 15 plans the next tile ----
    if (in_regs && edges_of != it) request_edges(n0, n0 + T, e0, e0 + Et, it);
    if (is_mfma && tile_ok) {
      int lv = lane;
      asm volatile("" : "+v"(lv));
      const f4* wsrc = reinterpret_cast<const f4*>(t.packed) + (int64_t)wave * 8 * 64;   // wave-uniform base + lane
#pragma unroll
      for (int s = 0; s < 8; ++s) u[s] = wsrc[s * 64 + lv];
    } else if (!is_helper) {
#pragma unroll
      for (int s = 0; s < 8; ++s) u[s] = f4{0.f, 0.f, 0.f, 0.f};
    }
    if (is_helper && tile_ok) {
      int hv = ht;
      asm volatile("" : "+v"(hv));
#pragma unroll
      for (int d = 0; d < FT_DEPTH; ++d)
        if (d < nch) x_load(u[2 * d], u[2 * d + 1], d, hv);
    }
    if (tile_ok)
      for (int i = tid; i < T; i += FT_THREADS) { lds_cnt[i] = 0; lds_ns[i] = 0; }
    if (wave == FT_WAVES - 1) plan_tile(slot ^ 1);
    lds_barrier();
    const int nn0 = __builtin_amdgcn_readfirstlane(lds_rec[(slot ^ 1) * 8 + 0]);
    const int nn1 = __builtin_amdgcn_readfirstlane(lds_rec[(slot ^ 1) * 8 + 1]);
    const int ne0 = __builtin_amdgcn_readfirstlane(lds_rec[(slot ^ 1) * 8 + 2]);
    const int ne1 = __builtin_amdgcn_readfirstlane(lds_rec[(slot ^ 1) * 8 + 3]);
    const bool has_next = __builtin_amdgcn_readfirstlane(lds_rec[(slot ^ 1) * 8 + 4]) != 0;
    if (!tile_ok) { lds_barrier(); continue; }   // (the barrier: nobody rewrites this record slot before all have read it)

    // ---- (B) in-degrees ----
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < FT_EDGE_REGS; ++j)
        if (epk[j] >= 0) {
          const int s = epk[j] >> 16, d = epk[j] & 0xffff;
          atomicAdd(&lds_cnt[d], 1);
          if (s != d) atomicAdd(&lds_ns[d], 1);
        }
    } else {
      bad = false;
#pragma unroll 4
      for (int i = tid; i < Et; i += FT_THREADS) {
        const int64_t s = t.src[(int64_t)e0 + i] - n0, d = t.dst[(int64_t)e0 + i] - n0;
        if (s < 0 || s >= T || d < 0 || d >= T) { bad = true; continue; }
        atomicAdd(&lds_cnt[(int)d], 1);
        if (s != d) atomicAdd(&lds_ns[(int)d], 1);
      }
    }
    if (__ballot(bad) != 0 && lane == 0) ft_error(t, 1);
    lds_barrier();

    // ---- (C) exclusive scan -> rowptr, deg^-1/2 tables (wavefront 0) ----
    if (wave == 0) {
      const int per = (T + 63) >> 6;
      const int b0 = lane * per;
      int mine = 0;
      for (int j = 0; j < per; ++j) mine += (b0 + j < T) ? lds_cnt[b0 + j] : 0;
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
          const int c = lds_cnt[i];
          lds_rowptr[i] = run;
          lds_dis_raw[i] = c > 0 ? 1.0f / sqrtf((float)c) : 0.0f;
          lds_dis_looped[i] = 1.0f / sqrtf((float)(lds_ns[i] + 1));
          lds_cnt[i] = 0;
          run += c;
        }
      }
      if (lane == 63) lds_rowptr[T] = incl;
    }
    lds_barrier();

    // ---- (D) scatter; the helpers stage chunk 0 ----
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < FT_EDGE_REGS; ++j)
        if (epk[j] >= 0) {
          const int s = epk[j] >> 16, d = epk[j] & 0xffff;
          lds_col[lds_rowptr[d] + atomicAdd(&lds_cnt[d], 1)] = (unsigned short)s;
        }
    } else {
#pragma unroll 4
      for (int i = tid; i < Et; i += FT_THREADS) {
        const int64_t s = t.src[(int64_t)e0 + i] - n0, d = t.dst[(int64_t)e0 + i] - n0;
        if (s < 0 || s >= T || d < 0 || d >= T) continue;
        lds_col[lds_rowptr[(int)d] + atomicAdd(&lds_cnt[(int)d], 1)] = (unsigned short)s;
      }
    }
    if (is_helper) {
      int hv = ht;
      asm volatile("" : "+v"(hv));
      split(u[0], u[1], 0, hv);
      if (FT_DEPTH < nch) x_load(u[0], u[1], FT_DEPTH, hv);
    }
    lds_barrier();
    // the next tile's edges travel during this tile's GEMM and rows
    if (has_next) request_edges(nn0, nn1, ne0, ne1, it + 1);

    // ---- (G) [bases | weightings] of the tile, 16 rows per step ----
    for (int c0 = 0; c0 < nch; c0 += FT_DEPTH) {
#pragma unroll
      for (int d = 0; d < FT_DEPTH; ++d) {
        const int c = c0 + d;
        if (c < nch) {   // workgroup-uniform
          const int buf = c & 1;
          if (is_helper) {
            if (c + 1 < nch) {      // chunk c + 1 sits in register set (c + 1) % FT_DEPTH = (d + 1) % FT_DEPTH
              int hv = ht;
              asm volatile("" : "+v"(hv));
              split(u[2 * ((d + 1) % FT_DEPTH)], u[2 * ((d + 1) % FT_DEPTH) + 1], buf ^ 1, hv);
              if (c + 1 + FT_DEPTH < nch) x_load(u[2 * ((d + 1) % FT_DEPTH)], u[2 * ((d + 1) % FT_DEPTH) + 1], c + 1 + FT_DEPTH, hv);
            }
          } else if (is_mfma) {
            int lv = lane;
            asm volatile("" : "+v"(lv));
            const int m = lv & 15, qd = lv >> 4;
            const char* pa = lds_planes + buf * (2 * FT_PLANE_BYTES) + m * (FT_KP * 2);
            f4 acc0 = f4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const int piece = (((4 * s + qd) ^ m) & 15) << 4;
              const ft_h8 xh = *reinterpret_cast<const ft_h8*>(pa + piece);
              const ft_h8 xl = *reinterpret_cast<const ft_h8*>(pa + FT_PLANE_BYTES + piece);
              const ft_h8 wh = __builtin_bit_cast(ft_h8, u[2 * s]), wl = __builtin_bit_cast(ft_h8, u[2 * s + 1]);
              acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh, acc1, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl, acc2, 0, 0, 0);
            }
            // D: lane -> column lane % 16, rows 4 (lane / 16) + i.  2^ex 2^ew (acc0 + 2^-11 (acc1 + acc2)) + bias
            const f4 ri = *reinterpret_cast<const f4*>(lds_rowinv + buf * FT_CHUNK + 4 * qd);
            f4 o;
            o.x = __builtin_fmaf(__builtin_fmaf(acc1.x + acc2.x, 1.f / 2048.f, acc0.x), col_inv * ri.x, col_bias);
            o.y = __builtin_fmaf(__builtin_fmaf(acc1.y + acc2.y, 1.f / 2048.f, acc0.y), col_inv * ri.y, col_bias);
            o.z = __builtin_fmaf(__builtin_fmaf(acc1.z + acc2.z, 1.f / 2048.f, acc0.z), col_inv * ri.z, col_bias);
            o.w = __builtin_fmaf(__builtin_fmaf(acc1.w + acc2.w, 1.f / 2048.f, acc0.w), col_inv * ri.w, col_bias);
            if (dst_act) o = w_act<C>(a, o);
            if (dst_off >= 0) {
              char* po = base + dst_off + (FT_CHUNK * c + 4 * qd) * dst_stride;
              *reinterpret_cast<float*>(po) = o.x;
              *reinterpret_cast<float*>(po + dst_stride) = o.y;
              *reinterpret_cast<float*>(po + 2 * dst_stride) = o.z;
              *reinterpret_cast<float*>(po + 3 * dst_stride) = o.w;
            }
          }
          lds_barrier();
        }
      }
    }

    // ---- (E) rows: one lane group per row, G rows per wavefront and pass, everything from LDS ----
    for (int r0 = 0; r0 < T; r0 += RPP) {
      const int r = r0 + wave * G + g;
      const bool row_ok = r < T;
      const int row = n0 + (row_ok ? r : 0);
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
      if (want_self) vself = lds_bases4[r * ldb4 + q];

      FAcc<NEED> acc;
      acc.init();
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
            v[u] = in_x[u] ? lds_bases4[j * ldb4 + q] : f4{0.f, 0.f, 0.f, 0.f};
            w[u] = bperm(addr, dd) * dis_i;
            if (C::yl(a) && !C::xl(a)) w[u] = is_self ? 0.f : w[u];
          }
#pragma unroll
          for (int u = 0; u < FU; ++u) fold<NEED>(acc, v[u], w[u], in_x[u], start + ts + t0 + u);
        }
      }
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const f4 wdummy[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
      // the row's weightings sit in the LDS image as [h][b][4], nonlinearity applied (W_READY; a.w_lds_stride == 0)
      finish_group<LPR_LOG2, HPB, NEED, C, true>(a, R, ln, row, row_ok, acc, nd, nself, dis_i, vself, has_self, wdummy, true,
                                                 lds_wt + (row_ok ? r : 0) * t.wl_floats, lds_bias, lds_scale);
    }
    lds_barrier();   // every wavefront is done with the tile's LDS image
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
constexpr size_t FT_LDS_BUDGET = 160 * 1024 - 256;

struct FtLds {
  size_t total;
  int off_rec, off_planes, off_rowinv, off_bases, off_wt, off_col, off_rowptr, off_cnt, off_ns, off_dis_raw, off_dis_looped;
};

static FtLds ft_lds(const AggArgs& a, int wl_floats, int tcap, int emax, bool with_post) {
  FtLds L;
  auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const int bias_floats = (a.H * a.Ls + 3) & ~3;
  size_t at = up16((size_t)(with_post ? 2 : 1) * bias_floats * sizeof(float));
  L.off_rec = (int)at; at += 64;
  L.off_planes = (int)at; at += FT_PLANES_BYTES;
  L.off_rowinv = (int)at; at += up16(2 * FT_CHUNK * sizeof(float));
  L.off_bases = (int)at; at += up16((size_t)tcap * a.ldb * 4);
  L.off_wt = (int)at; at += up16((size_t)tcap * wl_floats * 4);
  L.off_col = (int)at; at += up16((size_t)emax * 2);
  L.off_rowptr = (int)at; at += up16((size_t)(tcap + 1) * 4);
  L.off_cnt = (int)at; at += up16((size_t)tcap * 4);
  L.off_ns = (int)at; at += up16((size_t)tcap * 4);
  L.off_dis_raw = (int)at; at += up16((size_t)tcap * 4);
  L.off_dis_looped = (int)at; at += up16((size_t)tcap * 4);
  L.total = at;
  return L;
}

bool fused_tile_shape(const AggArgs& a, int f_in) {
  return f_in >= 4 && f_in <= FT_KP && (f_in & 3) == 0 && a.ldb + a.W <= FT_NV && a.slots <= 64 && a.A <= AMAX;
}

// rows of a tile whose image (bases + weightings + CSR areas for max_tile_edges entries) fits the LDS of a CU; 0 = none
int fused_tile_capacity(const AggArgs& a, int f_in, int max_tile_edges, bool with_post) {
  if (!fused_tile_shape(a, f_in) || max_tile_edges < 0) return 0;
  const int wl = a.H * a.B * 4;
  int best = 0;
  for (int tcap = FT_CHUNK; tcap <= FT_MAX_NODES; tcap += FT_CHUNK) {
    if (ft_lds(a, wl, tcap, max_tile_edges, with_post).total <= FT_LDS_BUDGET) best = tcap; else break;
  }
  return best;
}

template <int LPR_LOG2, int HPB, int NEED, class C>
static int launch_ft_one(const AggArgs& a, const FusedTileArgs& t, unsigned grid, size_t lds, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_tile_kernel<LPR_LOG2, HPB, NEED, C>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_last_error("hipFuncSetAttribute(fused_tile_kernel)", e); return EGC_ERR_HIP; }
    attr_set = true;
  }
  fused_tile_kernel<LPR_LOG2, HPB, NEED, C><<<grid, FT_THREADS, lds, stream>>>(a, t);
  EGC_LAUNCH_CHECK("fused_tile_kernel");
  return EGC_OK;
}

template <int LPR_LOG2>
static int launch_ft_rt(const AggArgs& a, const FusedTileArgs& t, int need, unsigned grid, size_t lds, hipStream_t stream) {
  const int hpb = (a.H + a.B - 1) / a.B;
  if (need == 0) {
    if (hpb <= 1) return launch_ft_one<LPR_LOG2, 1, 0, RtCfg>(a, t, grid, lds, stream);
    if (hpb <= 2) return launch_ft_one<LPR_LOG2, 2, 0, RtCfg>(a, t, grid, lds, stream);
    return launch_ft_one<LPR_LOG2, 4, 0, RtCfg>(a, t, grid, lds, stream);
  }
  if (hpb <= 1) return launch_ft_one<LPR_LOG2, 1, NEED_SQ | NEED_MN, RtCfg>(a, t, grid, lds, stream);
  if (hpb <= 2) return launch_ft_one<LPR_LOG2, 2, NEED_SQ | NEED_MN, RtCfg>(a, t, grid, lds, stream);
  return launch_ft_one<LPR_LOG2, 4, NEED_SQ | NEED_MN, RtCfg>(a, t, grid, lds, stream);
}

int launch_fused_tile(AggArgs a, const int64_t* ptr, const int64_t* edge_ptr, int64_t n_graphs, const int64_t* src,
                      const int64_t* dst, int64_t n_edges, const int* max_index, const float* x, int f_in, const void* packed,
                      int tcap, int emax, int32_t* status, int32_t* host_flag, hipStream_t stream) {
  if (!fused_tile_shape(a, f_in)) return EGC_ERR_UNSUPPORTED;
  const int lpr = a.slots <= 16 ? 16 : a.slots <= 32 ? 32 : 64;
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
  a.w_lds_stride = 0;     // finish_group<W_READY>: the weight strip of a lane group IS the row of the LDS image it is handed
  a.bias_lds_floats = (a.H * a.Ls + 3) & ~3;
  a.lds_floats_per_wave = 0;
  FusedTileArgs t = {};
  t.ptr = ptr; t.edge_ptr = edge_ptr; t.n_graphs = n_graphs; t.src = src; t.dst = dst; t.n_edges = n_edges;
  t.max_index = max_index; t.status = status; t.host_flag = host_flag; t.x = x; t.packed = (const ft_u16*)packed;
  t.F_in = f_in;
  t.n_ct = (a.ldb + a.W + 15) / 16;
  t.tcap = tcap; t.emax = emax;
  t.wl_floats = a.H * a.B * 4;
  if (tcap < FT_CHUNK || tcap > FT_MAX_NODES || (tcap % FT_CHUNK) != 0 || emax < 0) return EGC_ERR_INVALID;
  const FtLds L = ft_lds(a, t.wl_floats, tcap, emax, a.post_scale != nullptr);
  if (L.total > FT_LDS_BUDGET) return EGC_ERR_UNSUPPORTED;
  t.off_rec = L.off_rec; t.off_planes = L.off_planes; t.off_rowinv = L.off_rowinv; t.off_bases = L.off_bases; t.off_wt = L.off_wt;
  t.off_col = L.off_col; t.off_rowptr = L.off_rowptr; t.off_cnt = L.off_cnt; t.off_ns = L.off_ns; t.off_dis_raw = L.off_dis_raw;
  t.off_dis_looped = L.off_dis_looped;
  // one workgroup per CU at most; fewer when the batch is small (a workgroup's share: at least ~16 nodes, at least one graph)
  int64_t grid = 256;
  if (const char* e = getenv("EGC_FT_GRID")) grid = std::max(1, atoi(e));
  grid = std::min<int64_t>(grid, std::max<int64_t>(1, n_graphs));
  grid = std::min<int64_t>(grid, std::max<int64_t>(1, (int64_t)a.n_nodes / 16));
  if (getenv("EGC_NO_STATIC_CFG") == nullptr && a.act == EGC_ACT_NONE && a.Ls == a.L) {
    constexpr int S = EGC_AGGR_SUM, M = EGC_AGGR_MEAN, X = EGC_AGGR_MAX, Y = EGC_AGGR_SYMNORM;
    unsigned pk = 0;
    for (int k = 0; k < a.A; ++k) pk |= (unsigned)a.aggr[k] << (3 * k);
    // EGConv EGC-M north star (configs 3 / 4 of BASELINE.json): d=128, H=8, B=4, sum+mean+max+symnorm, loops on every node
    if (a.H == 8 && a.B == 4 && a.L == 16 && a.A == 4 && pk == agg_pack(S, M, X, Y) && a.x_looped && a.y_looped && a.loops_all)
      return launch_ft_one<4, 2, 0, StCfg<8, 4, 16, 4, agg_pack(S, M, X, Y), EGC_ACT_NONE, true, true, true>>(a, t, (unsigned)grid,
                                                                                                              L.total, stream);
    // EfficientGraphConv EGC-M at d=128 (symadd looped, the others raw)
    if (a.H == 8 && a.B == 4 && a.L == 16 && a.A == 3 && pk == agg_pack(Y, X, M) && !a.x_looped && a.y_looped && a.loops_all)
      return launch_ft_one<4, 2, 0, StCfg<8, 4, 16, 3, agg_pack(Y, X, M), EGC_ACT_NONE, false, true, true>>(a, t, (unsigned)grid,
                                                                                                            L.total, stream);
  }
  switch (lpr) {
    case 16: return launch_ft_rt<4>(a, t, need, (unsigned)grid, L.total, stream);
    case 32: return launch_ft_rt<5>(a, t, need, (unsigned)grid, L.total, stream);
    default: return launch_ft_rt<6>(a, t, need, (unsigned)grid, L.total, stream);
  }
}

}  // namespace egc
