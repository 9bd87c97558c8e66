// The kernel of egc_fused_tile.hip (batches of whole graphs, the WHOLE layer in one launch) -- shared by its two translation
// units: egc_fused_tile.hip (the register-stationary form: F_in <= 128, ldb + H B A <= 192) and egc_fused_tile_wide.hip (the
// reference's wider batched shapes: weight slabs streamed from L2).  See egc_fused_tile.hip for the description.
#pragma once
#include <algorithm>
#include <cstdio>

#include "egc_aggregate_fast_dev.h"

namespace egc {

typedef _Float16 ft_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 ft_h2 __attribute__((ext_vector_type(2)));
typedef float ft_f2 __attribute__((ext_vector_type(2)));
typedef unsigned int ft_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int ft_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short ft_u16;

constexpr int FT_THREADS = 1024;
constexpr int FT_WAVES = FT_THREADS / 64;
constexpr int FT_MFMA_WAVES = 12;       // 16-column tiles of the virtual column space [bases (ldb) | weightings (W)]
constexpr int FT_FIRST_HELPER = 12;     // wavefronts 12-15: x rows -> planes
constexpr int FT_HELPER_THREADS = (FT_WAVES - FT_FIRST_HELPER) * 64;
constexpr int FT_KP = 128;              // k extent of the register-resident weight tiles (F_in <= 128, zero beyond)
constexpr int FT_CHUNK = 16;            // rows per GEMM step (one MFMA tile)
constexpr int FT_WORKER_THREADS = FT_FIRST_HELPER * 64;
constexpr int FT_EDGE_REGS = 4;         // edges per worker thread kept in registers (16:16 packed local ids)
constexpr int FT_CSR_WAVES = 3;         // helper wavefronts 12-14 build the tiles' CSR (15 plans the tiles)
constexpr int FT_PER = 3;               // rows per lane of the one-wavefront scan: 3 x 64 >= 16 FT_RING
constexpr int FT_RING = 10;             // 16-row chunks of x a tile may have: the helpers hold them all in registers (80 VGPRs)
static_assert(FT_PER * 64 >= FT_CHUNK * FT_RING, "the scan covers a whole tile");
constexpr int FT_MAX_NODES = 2048;      // local ids are 16-bit, the scan is one wavefront
constexpr int FT_NV = FT_MFMA_WAVES * 16;
constexpr int FT_PLANE_BYTES = FT_CHUNK * FT_KP * 2;     // one plane of one chunk
constexpr int FT_PBUF = 3;                                // chunk buffers: the helpers stage two chunks ahead of the workers' MFMAs
constexpr int FT_PLANES_BYTES = FT_PBUF * 2 * FT_PLANE_BYTES;  // [3 buffers][2 planes]
// The backward form's second GEMM: d rows of K2 = ldb + H B 4 <= 192 columns as two fp16 planes, rows padded by 16 bytes
constexpr int FTB_ROW_BYTES = 192 * 2 + 16;                // (25 sixteen-byte pieces: conflict-free A-operand reads)
constexpr int FTB_PLANE_BYTES = FT_CHUNK * FTB_ROW_BYTES;  // one plane of one 16-row chunk
constexpr int FTB_PBUF_BYTES = 2 * FTB_PLANE_BYTES;        // [2 planes]
constexpr int FTB_PLANES_BYTES = 2 * FTB_PBUF_BYTES;       // [2 buffers]: 25,600 bytes (the first GEMM's three buffers take 24,576)
// The WIDE form (egc_fused_tile_wide.hip: 128 < F_in <= 320 or more than 192 virtual columns -- the reference's 168 / 224 / 296 /
// 300 / 304-wide batched nets, run_pretrained.sh:7-48): 32-row GEMM chunks on v_mfma_f32_32x32x16_f16, one 32-column tile per
// worker (at most 12: 384 virtual columns), the weight fragments streamed from L2 per k-step (a 320 x 384 operand does not fit the
// register files), x staged in k-slabs of 128 through two plane buffers.
constexpr int FTW_CH = 32;                                 // rows per GEMM chunk
constexpr int FTW_PP = 10;                                 // 16-byte pieces of x per helper thread and chunk: 8 threads per row, F_in <= 320
constexpr int FTW_MAX_FIN = 8 * FTW_PP * 4;
constexpr int FTW_SLAB = 128;                              // k per staged slab
constexpr int FTW_LDX = FTW_SLAB + 8;                      // halves per plane row in LDS (+ 16 bytes: conflict-free A-operand reads)
constexpr int FTW_PLANE_BYTES = FTW_CH * FTW_LDX * 2;      // one plane of one slab
constexpr int FTW_PBUF_BYTES = 2 * FTW_PLANE_BYTES;        // [2 planes]
constexpr int FTW_PLANES_BYTES = 2 * FTW_PBUF_BYTES;       // [2 buffers]
constexpr int FTW_MAXCH = FT_CHUNK * FT_RING / FTW_CH;     // chunks of a tile (160 rows)
constexpr int FTW_MAX_CT = FT_MFMA_WAVES;                  // 32-column tiles
constexpr int FT_DBS_POISON = 0x7fffffff;                  // backward: the tile's fixed-point scale when its g or w' holds an Inf / NaN

#ifdef EGC_FT_STAMPS
__device__ unsigned long long* egc_ft_stamp_buf = nullptr;   // diagnostic build only: [grid][8] accumulated cycles per phase
#define FT_HSTAMP(k, cond) { if ((cond) && lane == 0 && blockIdx.x == 7 && it < 12 && egc_ft_stamp_buf != nullptr) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); egc_ft_stamp_buf[256 * 9 + it * 8 + k] = _t - ft_h0; } }
#define FT_STAMP(k) { if (tid == 0) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); ft_acc[k] += _t - ft_t0; if (blockIdx.x == 7 && it < 12 && egc_ft_stamp_buf != nullptr) egc_ft_stamp_buf[256 * 9 + it * 8 + k] = _t - ft_t0; ft_t0 = _t; } }
#else
#define FT_STAMP(k)
#endif

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
  int dbg;                   // diagnostic build (EGC_FT_STAMPS) only: bit 0 no split, bit 1 no MFMA, bit 2 no rows
  int off_rec, off_planes, off_rowinv, off_bases, off_wt;
  int off_col, off_rowptr, off_cnt, off_dis;   // the CSR areas of an even tile; csr_stride bytes further: those of an odd tile
  int csr_stride;
  // WIDE form only
  int n_slabs;               // ceil(F_in / 128)
  int k16;                   // k-steps of 16 of the streamed weight fragments: ceil(F_in / 16) rounded up to a multiple of 4 (zero steps)
  int ldbp;                  // ldb rounded up to 32: first virtual column of the weightings
  int w_aw;                  // floats per (h, b) block of a weightings row in LDS: 4 for A >= 3, else A
  int nsets, p0;             // rows of more than 64 slots: two passes, the first over p0 = ceil(P / 2) slots of every basis
  unsigned magic0, magic1;   // floor(2^32 / p0) + 1, floor(2^32 / (P - p0)) + 1
  // backward form only (MODE == 1: egc_layer_backward_batch_fused_f32, egc_fused_tile.hip)
  const float* grad_out;     // [n_nodes, F_out]
  float* d_x;                // [n_nodes, F_in]
  const float* d_x_add;      // [n_nodes, F_in] added to d_x in its store (the gradient reaching x past the layer), or nullptr
  float* d_cat;              // [n_nodes, ld_dcat]: the gradient of [bases | pre-activation weightings] (what x^T d needs), or nullptr
  int ld_dcat;
  const ft_u16* packed_t;    // [8][6][2][64][8] fp16 fragments of [bases_weight | comb_weight^T]^T, float col_inv[128]
  int off_db;                // LDS: d bases image [tcap][ldb]
  int off_rowinv2;           // LDS: row scales of the staged d chunks [2][16]
};

// first index i in [0, n) with arr[i] >= key (n if none), by HALF a wavefront (lanes [32 h, 32 h + 32) share `key`), as
// egc_aggregate_tile.hip: the two halves of a wavefront run two searches side by side; deterministic on unsorted input.
// 32-bit indices (n < 2^31): this runs in the wavefronts whose registers carry a tile of x.
__device__ inline int ft_half_wave_lower_bound(const int64_t* __restrict__ arr, int n, int64_t key, int lane) {
  const int l32 = lane & 31, sh = lane & 32;
  int lo = 0, hi = n;
  while (__ballot(hi - lo > 32) != 0) {
    const bool live = hi - lo > 32;
    const int step = live ? (hi - lo + 31) / 32 : 1;
    const int i = lo + l32 * step;
    const bool ge = (live && i < hi) ? arr[i] >= key : true;
    const unsigned m = (unsigned)(__ballot(ge) >> sh);
    if (!live) continue;
    const int f = __ffs((int)m) - 1;
    if (f < 0) { lo = lo + 31 * step + 1; if (lo > hi) lo = hi; continue; }
    const int nhi = lo + f * step;
    lo = f > 0 ? lo + (f - 1) * step + 1 : lo;
    hi = nhi < hi ? nhi : hi;
  }
  const int i = lo + l32;
  const bool ge = i < hi ? arr[i] >= key : true;
  const unsigned m = (unsigned)(__ballot(ge) >> sh);
  const int f = __ffs((int)m) - 1;
  return f < 0 ? hi : (lo + f < hi ? lo + f : hi);
}

// The same lower bound with ONE round of loads when the array is close to linear (graph offsets of a batch of similar
// graphs, the destination row of their edges): each half looks at the 64 entries around guess = key n / top first and
// falls back to the full search when the answer is not strictly inside that window (guess: the caller's, e.g. key n / top).  Same result as the full search on
// sorted input, a deterministic function of (arr, key) on any input (two workgroups that share a key get the same index).
__device__ inline int ft_guess_lower_bound(const int64_t* __restrict__ arr, int n, int64_t key, int64_t g64, int lane) {
  const int l32 = lane & 31, sh = lane & 32;
  g64 = g64 > n ? n : g64;
  int w0 = (int)(g64 < 31 ? 0 : g64 - 31);
  w0 = w0 > n - 64 ? n - 64 : w0;
  w0 = w0 < 0 ? 0 : w0;
  const int i0 = w0 + l32, i1 = w0 + 32 + l32;
  const int64_t v0 = i0 < n ? arr[i0] : key, v1 = i1 < n ? arr[i1] : key;     // beyond the array: counts as >= key
  const unsigned m0 = (unsigned)(__ballot(v0 >= key) >> sh), m1 = (unsigned)(__ballot(v1 >= key) >> sh);
  const int first = m0 != 0 ? __ffs((int)m0) - 1 : (m1 != 0 ? 32 + __ffs((int)m1) - 1 : 64);
  int r = w0 + first;
  r = r > n ? n : r;
  const bool sure = (first > 0 || w0 == 0) && (first < 64 || w0 + 64 >= n);
  if (__ballot(!sure) != 0) {       // (both halves take part in the full search; each keeps its window result if it was sure)
    const int full = ft_half_wave_lower_bound(arr, n, key, lane);
    r = sure ? r : full;
  }
  return r;
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

// largest magnitude of four floats as a bit pattern, Inf / NaN INCLUDED (integer maximum of the patterns without their sign: a NaN
// compares above everything; fmaxf would drop it, and the backward's fixed-point sums must know -- FT_DBS_POISON)
__device__ inline unsigned ft_amax_bits(const f4 v) {
  const unsigned a = __float_as_uint(v.x) & 0x7fffffffu, b = __float_as_uint(v.y) & 0x7fffffffu;
  const unsigned c = __float_as_uint(v.z) & 0x7fffffffu, d = __float_as_uint(v.w) & 0x7fffffffu;
  return max(max(a, b), max(c, d));
}

// largest of a non-negative bit pattern over the wavefront, uniform (four row maxima by DPP, then four lane reads)
__device__ inline unsigned ft_wave_umax(unsigned a) {
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x141, 0xf, 0xf, true));  // row_half_mirror
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x140, 0xf, 0xf, true));  // row_mirror
  return max(max((unsigned)__builtin_amdgcn_readlane((int)a, 0), (unsigned)__builtin_amdgcn_readlane((int)a, 16)),
             max((unsigned)__builtin_amdgcn_readlane((int)a, 32), (unsigned)__builtin_amdgcn_readlane((int)a, 48)));
}

// MODE 1 (WIDE == 0 only): the BACKWARD of the layer on the same tiles -- x, grad_out and the edge list in, d x and the
// gradient of [bases | weightings] out (described at egc_layer_backward_batch_fused_f32, egc_fused_tile.hip).
template <int LPR_LOG2, int HPB, int NEED, class C, int WIDE = 0, int MODE = 0>
__global__ void __launch_bounds__(FT_THREADS) fused_tile_kernel(AggArgs a, FusedTileArgs t) {
  static_assert(MODE == 0 || WIDE == 0, "the backward form is built on the register-stationary GEMM");
  constexpr int LPR = 1 << LPR_LOG2, G = 64 / LPR;
  extern __shared__ float smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const bool is_helper = wave >= FT_FIRST_HELPER;

  // ---- LDS image ----
  char* base = reinterpret_cast<char*>(smem);
  float* lds_bias = smem;                                   // [bias (x scale + shift)][scale]: one copy for the workgroup
  const bool post = a.post_scale != nullptr;
  float* lds_scale = lds_bias + a.bias_lds_floats;
  int* lds_rec = reinterpret_cast<int*>(base + t.off_rec);  // [3][8]: (n0, n1, e0, e1, valid) of tiles it, it + 1, it + 2; [24]: row counter
  int* lds_rowctr = lds_rec + 24;
  char* lds_planes = base + t.off_planes;                   // [2 buffers][2 planes][16 rows][128 fp16], 16-byte pieces swizzled
  float* lds_rowinv = reinterpret_cast<float*>(base + t.off_rowinv);   // [3][16]
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

#ifdef EGC_FT_STAMPS
  unsigned long long ft_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ft_t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ft_t0) :: "memory");
  const unsigned long long ft_start = ft_t0;
#endif

  // A tile's record -> (n0, T, e0, Et, rows of x to multiply); every wavefront derives the same values from the same record
  struct Tile { int n0, T, e0, Et, nch; bool valid, ok; };
  auto read_tile = [&](int slot) -> Tile {
    Tile r;
    r.valid = __builtin_amdgcn_readfirstlane(lds_rec[slot * 8 + 4]) != 0;
    r.n0 = __builtin_amdgcn_readfirstlane(lds_rec[slot * 8 + 0]);
    r.T = __builtin_amdgcn_readfirstlane(lds_rec[slot * 8 + 1]) - r.n0;
    r.e0 = __builtin_amdgcn_readfirstlane(lds_rec[slot * 8 + 2]);
    r.Et = __builtin_amdgcn_readfirstlane(lds_rec[slot * 8 + 3]) - r.e0;
    r.ok = r.valid && r.T > 0 && r.T <= t.tcap && r.Et <= t.emax;
    r.nch = r.ok ? (WIDE != 0 ? (r.T + FTW_CH - 1) / FTW_CH : (r.T + FT_CHUNK - 1) / FT_CHUNK) : 0;
    return r;
  };

  if (is_helper) {
    // =====================================================================================================================
    // wavefronts 12-15: the x stream.  A tile's rows sit in registers one whole tile ahead -- chunk c (16 rows) in xr[2 c],
    // xr[2 c + 1]; the NEXT tile's rows are requested into the same registers as soon as this tile's last chunk has been
    // split into the LDS planes (they travel during the rows phase of tile it and the CSR build of tile it + 1: a tile of
    // x per CU in flight, with no LDS staging) -- and wavefront 15 plans two tiles ahead.
    // The barrier sequence is the workers': 1 + nch + 1 per tile (nch = 0 for a tile that is skipped).
    // =====================================================================================================================
    const int ht = tid - FT_FIRST_HELPER * 64;
    if (ht == 0) lds_rec[25] = 0;            // the CSR builders' synchronisation counter (first used behind the barrier below)
    if constexpr (MODE == 1) { if (ht == 0) lds_rec[27] = 0; }     // (largest |w'| of a tile: zero again behind every B1)
    const int Gn = (int)t.n_graphs;          // (host: n_graphs, n_nodes, n_edges < 2^31)
    const int Nn = a.n_nodes, En = (int)t.n_edges;
    int cur_g = 0, g_hi = 0;                 // (meaningful in wavefront 15 only; wave-uniform)
    int last_n = 0, last_e = 0;              // where the last planned tile ended (node, edge): the next search starts its guess there
    auto clampi = [](int64_t v, int hi) -> int { return v < 0 ? 0 : (v > hi ? hi : (int)v); };
    // next tile of this workgroup -> lds_rec[slot] (wavefront 15, all lanes): graphs [cur_g, next_g) with at most tcap nodes.
    // Everything in 32 bits and wave-uniform values in scalar registers: this code runs with a tile of x in the vector registers.
    auto plan_tile = [&](int slot) {
      int n0 = 0, n1 = 0, e0 = 0, e1 = 0, valid = 0;
      if (cur_g < g_hi) {
        const int p0 = clampi(t.ptr[cur_g], Nn);
        const int gi = cur_g + 1 + lane;
        const int pv = clampi(t.ptr[gi <= g_hi ? gi : g_hi], Nn);
        const bool ok = gi <= g_hi && pv - p0 <= t.tcap && pv >= p0;
        const unsigned long long m = __ballot(ok);
        int n_ok = m == ~0ull ? 64 : __ffsll((long long)~m) - 1;     // graphs that fit (a prefix: ptr is non-decreasing)
        n_ok = n_ok < 1 ? 1 : n_ok;                                   // a single graph beyond the capacity: reported by the tile
        // offsets that decrease inside the run: reported (the node ranges of the tiles then no longer partition [0, N))
        const int pprev = __shfl_up(pv, 1);
        const bool dec = gi <= g_hi && lane < n_ok && pv < (lane == 0 ? p0 : pprev);
        if (__ballot(dec) != 0 && lane == 0) ft_error(t, 1);
        const int next_g = cur_g + n_ok;
        const int pn = __builtin_amdgcn_readfirstlane(__shfl(pv, n_ok - 1));
        n0 = p0;
        n1 = pn < p0 ? p0 : pn;
        if (t.edge_ptr != nullptr) {
          const int64_t b0 = t.edge_ptr[cur_g], b1 = t.edge_ptr[next_g];
          if (b1 < b0 && lane == 0) ft_error(t, 1);
          e0 = clampi(b0, En);
          e1 = clampi(b1, En);
        } else {   // the two ends side by side in the two halves of the wavefront, each around its own guess: the edge list is
                   // close to linear in the node id, locally (one tile further) even more so
          const int64_t den = Nn > 0 ? Nn : 1;
          const int64_t ga = (int64_t)last_e + (int64_t)(n0 - last_n) * En / den, gb = ga + (int64_t)(n1 - n0) * En / den;
          const int r = ft_guess_lower_bound(t.dst, En, lane < 32 ? n0 : n1, lane < 32 ? ga : gb, lane);
          e0 = __builtin_amdgcn_readfirstlane(__shfl(r, 0));
          e1 = __builtin_amdgcn_readfirstlane(__shfl(r, 32));
        }
        if (cur_g == 0) e0 = 0;
        if (next_g >= Gn) e1 = En;
        e1 = e1 < e0 ? e0 : e1;
        cur_g = next_g;
        valid = 1;
        last_n = n1;
        last_e = e1;
      }
      if (lane == 0) {
        lds_rec[slot * 8 + 0] = n0; lds_rec[slot * 8 + 1] = n1; lds_rec[slot * 8 + 2] = e0; lds_rec[slot * 8 + 3] = e1;
        lds_rec[slot * 8 + 4] = valid;
        if constexpr (MODE == 1) lds_rec[slot * 8 + 5] = 0;       // (largest |g| of the tile: the helpers' g_max below)
      }
    };
    if (wave == FT_WAVES - 1) {
      // the workgroup's graphs: [g_lo, g_hi) = those whose first node lies in its share of [0, N)
      const int64_t nb = gridDim.x, b = blockIdx.x;
      const int64_t k_lo = b * Nn / nb, k_hi = (b + 1) * Nn / nb;
      const int r = ft_guess_lower_bound(t.ptr, Gn + 1, lane < 32 ? k_lo : k_hi, (lane < 32 ? k_lo : k_hi) * (Gn + 1) / (Nn > 0 ? Nn : 1), lane);
      int lo = __builtin_amdgcn_readfirstlane(__shfl(r, 0)), hi = __builtin_amdgcn_readfirstlane(__shfl(r, 32));
      if (b == 0) lo = 0;
      if (b == nb - 1) hi = Gn;
      lo = lo > Gn ? Gn : lo;
      hi = hi > Gn ? Gn : hi;
      cur_g = lo;
      g_hi = hi < lo ? lo : hi;
      if (b == 0 && lane == 0 && Gn > 0 && (t.ptr[0] != 0 || t.ptr[Gn] != Nn)) ft_error(t, 1);   // offsets that do not cover [0, N)
      // The first TWO tiles from one window of 64 graph offsets and one round of edge offsets (plan_tile twice is four
      // dependent rounds in front of the first barrier: with the search above, 10,000 cycles of a 65,000-cycle launch).
      bool both = false;
      if (cur_g < g_hi) {
        const int p0 = clampi(t.ptr[cur_g], Nn);
        const int gi = cur_g + 1 + lane;
        const int pv = clampi(t.ptr[gi <= g_hi ? gi : g_hi], Nn);
        const bool ok0 = gi <= g_hi && pv - p0 <= t.tcap && pv >= p0;
        const unsigned long long m0 = __ballot(ok0);
        const int n0k = m0 == ~0ull ? 64 : __ffsll((long long)~m0) - 1;
        if (n0k >= 1 && n0k < 63) {
          const int p1 = __builtin_amdgcn_readfirstlane(__shfl(pv, n0k - 1));
          const bool ok1 = lane >= n0k && gi <= g_hi && pv - p1 <= t.tcap && pv >= p1;
          const unsigned long long m1 = __ballot(ok1) >> n0k;
          const int n1k = m1 == 0 ? 0 : (__ffsll((long long)~m1) - 1);
          const int g1 = cur_g + n0k, g2 = g1 + n1k;
          const int pprev = __shfl_up(pv, 1);       // (unconditionally, every lane active: inside the short-circuit lane 1 read an inactive lane 0)
          const bool mono = !(gi <= g_hi && lane < n0k + n1k && pv < (lane == 0 ? p0 : pprev));
          // (a second tile that would hold no graph although graphs remain, or a window that ends inside it: the general path)
          if ((n1k >= 1 || g1 >= g_hi) && n0k + n1k < 64 && __ballot(!mono) == 0) {
            const int p2 = n1k >= 1 ? __builtin_amdgcn_readfirstlane(__shfl(pv, n0k + n1k - 1)) : p1;
            int e0, e1, e2;
            if (t.edge_ptr != nullptr) {
              const int gsel = lane == 0 ? cur_g : (lane == 1 ? g1 : g2);
              const int64_t ev = lane < 3 ? t.edge_ptr[gsel] : 0;
              const int64_t b0 = __shfl(ev, 0), b1 = __shfl(ev, 1), b2 = __shfl(ev, 2);
              if ((b1 < b0 || b2 < b1) && lane == 0) ft_error(t, 1);
              e0 = clampi(b0, En); e1 = clampi(b1, En); e2 = clampi(b2, En);
            } else {
              const int ra = ft_guess_lower_bound(t.dst, En, lane < 32 ? p0 : p1, (int64_t)(lane < 32 ? p0 : p1) * En / (Nn > 0 ? Nn : 1), lane);
              e0 = __builtin_amdgcn_readfirstlane(__shfl(ra, 0));
              e1 = __builtin_amdgcn_readfirstlane(__shfl(ra, 32));
              const int rb = ft_guess_lower_bound(t.dst, En, p2, (int64_t)e1 + (int64_t)(p2 - p1) * En / (Nn > 0 ? Nn : 1), lane);
              e2 = __builtin_amdgcn_readfirstlane(__shfl(rb, 0));
            }
            if (cur_g == 0) e0 = 0;
            if (g1 >= Gn) e1 = En;
            if (g2 >= Gn) e2 = En;
            e1 = e1 < e0 ? e0 : e1;
            e2 = e2 < e1 ? e1 : e2;
            if (lane == 0) {
              lds_rec[0] = p0; lds_rec[1] = p1; lds_rec[2] = e0; lds_rec[3] = e1; lds_rec[4] = 1;
              lds_rec[8] = p1; lds_rec[9] = p2; lds_rec[10] = e1; lds_rec[11] = e2; lds_rec[12] = n1k >= 1 ? 1 : 0;
              if constexpr (MODE == 1) { lds_rec[5] = 0; lds_rec[13] = 0; }
            }
            cur_g = g2;
            last_n = p2;
            last_e = e2;
            both = true;
          }
        }
      }
      if (!both) {
        plan_tile(0);
        plan_tile(1);
      }
    }
    lds_barrier();

    // ---- the CSR of a tile, by wavefronts 12-14 (15 plans tiles meanwhile), synchronised among themselves through an LDS
    //      counter (the workgroup barrier belongs to the workers' schedule): in-degrees (in-degree | non-self in-degree << 16,
    //      one LDS atomic per edge) | wavefront scan -> rowptr and the layer's deg^-1/2 table (wavefront 12) | scatter through
    //      the counts counted back down -- which leaves them zero for the next tile that uses this set.  The edges are
    //      streamed from memory twice (second pass: L2), four per lane in flight; every edge is checked against its tile.
    //      Built into set (tile & 1) while the workers read the other set in their rows phase. ----
    int* lds_hsync = lds_rec + 25;
    int hs_target = 0;
    auto csr_sync = [&]() {
      hs_target += FT_CSR_WAVES;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(lds_hsync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      int spins = 0;
      while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(lds_hsync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < hs_target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { if (lane == 0) ft_error(t, 4); break; }     // (a bounded spin cannot hang the GPU)
      }
      asm volatile("" ::: "memory");
    };
    // Four stages -- S0 request the edges, S1 in-degrees, S2 scan (wavefront 12), S3 scatter -- with a synchronisation of the
    // three wavefronts in front of S2 and S3; the caller interleaves them with the requests for the tile's rows.  The first 8
    // edges of every lane stay in registers between S1 and S3 as packed local ids (tiles of up to 8 x 192 edges -- all of
    // configs 3 and 4 -- read their edges once).
    //
    // DETERMINISTIC, INPUT ORDER (round 6): a row's entries sit in the CSR in the order of the edge list, as the reference's
    // scatter sums them (SURVEY.md 8a note 9: "reference CPU order = edge order"), so two launches give the same bits and the
    // first entry of a row attaining a maximum is the first in input order (torch_scatter's arg rule) without a position array.
    //   * wavefront w of the three owns a CONTIGUOUS range of the tile's edges, [w S, (w + 1) S) with S = ceil(Et / 3) rounded
    //     up to 64, and walks it in rounds of 64: lane l of round j holds edge w S + 64 j + l;
    //   * per row one counter PER WAVEFRONT, 16 bits each, in the two words the finished CSR keeps for the row anyway:
    //     cnt[i] = c0 | c1 << 16 and rowptr[i] = c2 | self loops << 16 (no LDS beyond round 5's: the image's rows are what it
    //     costs); S2 turns them in place into the row's start and three cursors -- cnt[i] = start | (start + c0) << 16,
    //     rowptr[i] = start | (start + c0 + c1) << 16: the rows phase reads the LOW half of a rowptr word;
    //   * S3 takes an entry's position from a RETURNING add to its wavefront's cursor: the adds of one wavefront execute in
    //     program order (rounds), and the lanes of one instruction that hit the same word are served in ascending lane order
    //     (tools/src/lds_atomic_order.hip checks exactly this on the device; tests/test_fused_tile_gpu.py checks the result).
    // The two words of every row of a set are zeroed at the top of the tile loop's iteration that builds into it (csr_zero: the
    // set was last read in the rows phase of the tile before the current one, and the GEMM steps' barriers stand between the
    // zeroing and the first count).
    constexpr int KEEP = 8;                                  // rounds (edges per lane) kept in registers between the stages
    constexpr unsigned NO_EDGE = 0xffffffffu;
    const int cw = wave - FT_FIRST_HELPER;                   // 0 .. 2 in the CSR wavefronts
    auto edge_rsrc = [&](const Tile& r, const int64_t* p) {
      return __builtin_amdgcn_make_buffer_rsrc((void*)(p + r.e0), 0, (unsigned)(r.ok ? r.Et : 0) * 8u, 0x00020000);
    };
    // a wavefront's span of the tile's edges, and how many of them exist
    auto span_of = [&](const Tile& r, int& S, int& nw) {
      const int Et = r.ok ? r.Et : 0;
      S = (((Et + FT_CSR_WAVES - 1) / FT_CSR_WAVES) + 63) & ~63;
      const int left = Et - cw * S;
      nw = left < 0 ? 0 : (left > S ? S : left);
    };
    // S0: the first KEEP rounds of this wavefront's span, one edge per lane and 8-byte request, INTO THE REGISTERS OF ROW
    // CHUNKS that were split long ago (e: eight 16-byte values -- e[j] = sources of rounds 2 j, 2 j + 1, e[4 + j] = their
    // destinations; those chunks' requests for the next tile follow S1), so the edges in flight cost no register next to the
    // tile of x.  (Entries beyond the tile's range read as 0, without traffic, and are skipped below.)
    auto csr_s0 = [&](const Tile& r, f4* e) {
      const __amdgpu_buffer_rsrc_t es = edge_rsrc(r, t.src), ed = edge_rsrc(r, t.dst);
      int S, nw;
      span_of(r, S, nw);
#pragma unroll
      for (int j = 0; j < KEEP; ++j) {
        // (only the rounds the span has -- wave-uniform: a molecule tile has one or two, a superpixel tile five -- each behind
        // one scalar offset for the span with the round in the instruction's immediate; what follows the edges in the
        // vector-memory queue, the tile's rows, is unconditional, so the counted waits stay exact)
        // (the WIDE form requests every round: under the branches the compiler carried the chunk's registers the edges travel
        // in through copies -- up to 860 spilled registers in its 168 / 224-wide instances)
        if (WIDE != 0 || 64 * j < nw) {
          const ft_u2 sv = __builtin_bit_cast(ft_u2, __builtin_amdgcn_raw_buffer_load_b64(es, (unsigned)lane * 8u + 512u * j, cw * S * 8, 0));
          const ft_u2 dv = __builtin_bit_cast(ft_u2, __builtin_amdgcn_raw_buffer_load_b64(ed, (unsigned)lane * 8u + 512u * j, cw * S * 8, 0));
          if (j & 1) { e[j >> 1].z = __uint_as_float(sv.x); e[j >> 1].w = __uint_as_float(sv.y); e[KEEP / 2 + (j >> 1)].z = __uint_as_float(dv.x); e[KEEP / 2 + (j >> 1)].w = __uint_as_float(dv.y); }
          else { e[j >> 1].x = __uint_as_float(sv.x); e[j >> 1].y = __uint_as_float(sv.y); e[KEEP / 2 + (j >> 1)].x = __uint_as_float(dv.x); e[KEEP / 2 + (j >> 1)].y = __uint_as_float(dv.y); }
        }
      }
    };
    // an edge's ends as tile-local ids, or false: 64-bit ids whose upper halves are not zero lie outside every tile
    // (n_nodes < 2^31), the lower halves are compared without sign
    auto local_ids = [&](long long s64, long long d64, int n0, int T, unsigned& sl, unsigned& dl) -> bool {
      const unsigned hi = (unsigned)((unsigned long long)s64 >> 32) | (unsigned)((unsigned long long)d64 >> 32);
      sl = (unsigned)s64 - (unsigned)n0;
      dl = (unsigned)d64 - (unsigned)n0;
      return hi == 0u && sl < (unsigned)T && dl < (unsigned)T;
    };
    typedef long long ft_l2 __attribute__((ext_vector_type(2)));
    // one edge into its row's counters: this wavefront's 16 bits; a self loop also into the row's self-loop count (rare: one
    // wave-uniform test per edge, the second add inside it)
    auto count_edge = [&](int* cnt, int* rowptr, unsigned sl, unsigned dl) {
      if (cw == 2) atomicAdd(&rowptr[dl], sl != dl ? 1 : 0x10001);
      else {
        atomicAdd(&cnt[dl], cw == 0 ? 1 : 0x10000);
        if (__ballot(sl == dl) != 0) { if (sl == dl) atomicAdd(&rowptr[dl], 0x10000); }
      }
    };
    // the building set's two words per row back to zero (every CSR thread one row: tcap <= 160 < 192)
    auto csr_zero = [&](int set) {
      char* cb = base + set * t.csr_stride;
      if (ht < t.tcap) { reinterpret_cast<int*>(cb + t.off_cnt)[ht] = 0; reinterpret_cast<int*>(cb + t.off_rowptr)[ht] = 0; }
    };
    auto csr_s1 = [&](const Tile& r, int set, const f4* e, unsigned (&epk)[KEEP]) {
      int* cnt = reinterpret_cast<int*>(base + set * t.csr_stride + t.off_cnt);
      int* rowptr = reinterpret_cast<int*>(base + set * t.csr_stride + t.off_rowptr);
      const int T = r.ok ? r.T : 0;
      int S, nw;
      span_of(r, S, nw);
      bool bad = false;
#pragma unroll
      for (int j = 0; j < KEEP / 2; ++j) {
        const ft_l2 s2 = __builtin_bit_cast(ft_l2, e[j]), d2 = __builtin_bit_cast(ft_l2, e[KEEP / 2 + j]);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          unsigned pk = NO_EDGE;
          if (WIDE == 0 && 64 * (2 * j + k) >= nw) { epk[2 * j + k] = pk; continue; }     // (wave-uniform: the span has no such round)
          if (64 * (2 * j + k) + lane < nw) {
            unsigned sl, dl;
            if (!local_ids(s2[k], d2[k], r.n0, T, sl, dl)) bad = true;
            else {
              count_edge(cnt, rowptr, sl, dl);
              pk = sl | (dl << 16);
            }
          }
          epk[2 * j + k] = pk;
        }
      }
      if (__ballot(bad) != 0 && lane == 0) ft_error(t, 1);
    };
    // the rounds beyond the first KEEP of a wavefront's span (tiles of more than 8 x 192 edges): requested and counted in one go
    auto csr_s1_rest = [&](const Tile& r, int set) {
      int* cnt = reinterpret_cast<int*>(base + set * t.csr_stride + t.off_cnt);
      int* rowptr = reinterpret_cast<int*>(base + set * t.csr_stride + t.off_rowptr);
      const int T = r.ok ? r.T : 0;
      int S, nw;
      span_of(r, S, nw);
      if (__builtin_amdgcn_readfirstlane(nw) <= KEEP * 64) return;
      bool bad = false;
      const __amdgpu_buffer_rsrc_t es = edge_rsrc(r, t.src), ed = edge_rsrc(r, t.dst);
      for (int i0 = KEEP * 64; i0 < nw; i0 += 4 * 64) {
        long long s4[4], d4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s4[j] = __builtin_bit_cast(long long, __builtin_amdgcn_raw_buffer_load_b64(es, (unsigned)lane * 8u + 512u * j, (cw * S + i0) * 8, 0));
          d4[j] = __builtin_bit_cast(long long, __builtin_amdgcn_raw_buffer_load_b64(ed, (unsigned)lane * 8u + 512u * j, (cw * S + i0) * 8, 0));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (i0 + 64 * j + lane < nw) {
            unsigned sl, dl;
            if (!local_ids(s4[j], d4[j], r.n0, T, sl, dl)) bad = true;
            else count_edge(cnt, rowptr, sl, dl);
          }
        }
      }
      if (__ballot(bad) != 0 && lane == 0) ft_error(t, 1);
    };
    auto csr_s2 = [&](const Tile& r, int set) {
      if (wave != FT_FIRST_HELPER) return;
      char* cb = base + set * t.csr_stride;
      int* rowptr = reinterpret_cast<int*>(cb + t.off_rowptr);
      int* cnt = reinterpret_cast<int*>(cb + t.off_cnt);
      float* dis = reinterpret_cast<float*>(cb + t.off_dis);
      const int T = r.ok ? r.T : 0;
      // (while the workers' rows phase keeps the LDS pipeline full every dependent LDS round trip of this build costs
      // hundreds of cycles: all reads of a stage are issued before the first is used)
      const int per = (T + 63) >> 6;           // <= 3: T <= 160
      const int b0 = lane * per;
      int ca[FT_PER], cb2[FT_PER];
#pragma unroll
      for (int j = 0; j < FT_PER; ++j) { ca[j] = cnt[min(b0 + j, t.tcap - 1)]; cb2[j] = rowptr[min(b0 + j, t.tcap - 1)]; }
      int mine = 0;
#pragma unroll
      for (int j = 0; j < FT_PER; ++j) {
        const bool on = j < per && b0 + j < T;
        ca[j] = on ? ca[j] : 0;
        cb2[j] = on ? cb2[j] : 0;
        mine += (ca[j] & 0xffff) + (int)((unsigned)ca[j] >> 16) + (cb2[j] & 0xffff);
      }
      // inclusive scan over the wavefront on the DPP network (row shifts inside the rows of 16, then the two row broadcasts)
      int incl = mine;
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);   // row_shr:1
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);   // row_shr:2
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);   // row_shr:4
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);   // row_shr:8
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, true);   // row_bcast:15 -> rows 1 and 3
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, true);   // row_bcast:31 -> rows 2 and 3
      int run = incl - mine;
#pragma unroll
      for (int j = 0; j < FT_PER; ++j) {
        const int i = b0 + j;
        if (j < per && i < T) {
          const int c0 = ca[j] & 0xffff, c1 = (int)((unsigned)ca[j] >> 16), c2 = cb2[j] & 0xffff;
          const int c = c0 + c1 + c2, ns = c - (int)((unsigned)cb2[j] >> 16);
          cnt[i] = run | ((run + c0) << 16);                                  // the scatter's cursors, one per wavefront;
          rowptr[i] = run | ((run + c0 + c1) << 16);                          // the row's start (what the rows phase reads)
          run += c;
          // deg^-1/2 of the layer's symnorm edge set, as prepare_kernel / build_scan_kernel (egc_graph.hip)
          dis[i] = C::yl(a) ? 1.0f / sqrtf((float)(ns + 1)) : (c > 0 ? 1.0f / sqrtf((float)c) : 0.0f);
        }
      }
      if (lane == 63) rowptr[T] = incl;
    };
    // an entry's position: a returning add to its wavefront's cursor of the row (in-order per wavefront, ascending lanes)
    auto take_pos = [&](int* cnt, int* rowptr, unsigned dl) -> int {
      if (cw == 0) return __hip_atomic_fetch_add(&cnt[dl], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & 0xffff;
      return (int)((unsigned)__hip_atomic_fetch_add(cw == 1 ? &cnt[dl] : &rowptr[dl], 0x10000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 16);
    };
    auto csr_s3 = [&](const Tile& r, int set, const unsigned (&epk)[KEEP]) {
      char* cb = base + set * t.csr_stride;
      unsigned short* col = reinterpret_cast<unsigned short*>(cb + t.off_col);
      int* cnt = reinterpret_cast<int*>(cb + t.off_cnt);
      int* rowptr = reinterpret_cast<int*>(cb + t.off_rowptr);
      const int T = r.ok ? r.T : 0;
      int S, nw;
      span_of(r, S, nw);
      int pos[KEEP];
#pragma unroll
      for (int j = 0; j < KEEP; ++j) {         // (all the adds issued before the first position is used)
        pos[j] = 0;
        if ((WIDE != 0 || 64 * j < nw) && epk[j] != NO_EDGE) pos[j] = take_pos(cnt, rowptr, epk[j] >> 16);
      }
#pragma unroll
      for (int j = 0; j < KEEP; ++j)
        if ((WIDE != 0 || 64 * j < nw) && epk[j] != NO_EDGE) col[pos[j]] = (unsigned short)(epk[j] & 0xffffu);
      if (__builtin_amdgcn_readfirstlane(nw) > KEEP * 64) {      // (larger tiles: the rest of their edges a second time, from L2)
        const __amdgpu_buffer_rsrc_t es = edge_rsrc(r, t.src), ed = edge_rsrc(r, t.dst);
        for (int i0 = KEEP * 64; i0 < nw; i0 += 4 * 64) {
          long long s4[4], d4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            s4[j] = __builtin_bit_cast(long long, __builtin_amdgcn_raw_buffer_load_b64(es, (unsigned)lane * 8u + 512u * j, (cw * S + i0) * 8, 0));
            d4[j] = __builtin_bit_cast(long long, __builtin_amdgcn_raw_buffer_load_b64(ed, (unsigned)lane * 8u + 512u * j, (cw * S + i0) * 8, 0));
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (i0 + 64 * j + lane < nw) {
              unsigned sl, dl;
              if (local_ids(s4[j], d4[j], r.n0, T, sl, dl)) col[take_pos(cnt, rowptr, dl)] = (unsigned short)sl;
            }
          }
        }
      }
    };
    // both sets' counters start at zero (afterwards: csr_zero at the top of the tile loop)
    if (wave < FT_FIRST_HELPER + FT_CSR_WAVES) {
      csr_zero(0);
      csr_zero(1);
      csr_sync();
    }

    const bool csr_wave = wave < FT_FIRST_HELPER + FT_CSR_WAVES;
    if constexpr (WIDE == 0) {
    // (the backward form keeps d bases next to the images: tiles of at most 128 rows, 16 registers fewer for the rows in flight)
    // (H = 8 in a static configuration: tiles of at most 96 rows -- the host's capacity, fused_tile_bwd_capacity -- so six chunks; the
    // sixteen registers fewer keep the helpers' working set in the register file: with eight chunks the allocator spilled ~30
    // registers into the per-tile loops, 22 MB of scratch traffic per launch on the molhiv batch)
    constexpr int RINGN = MODE == 1 ? (C::kH == 8 ? 6 : 8) : FT_RING;
    constexpr int EREG = 2 * RINGN - 8 < RINGN ? 2 * RINGN - 8 : RINGN;     // first register of the eight the edges travel in ...
    constexpr int H1 = EREG / 2;                                            // ... = the chunks requested behind the CSR build's counts
    f4 xr[2 * RINGN];
    // 16-byte pieces p = ht + 256 i (i = 0, 1) of a 16-row chunk <-> (row p / 32, k 4 (p % 32)).  Per thread: the two byte
    // offsets inside a chunk (rows beyond the tile fall outside the tile's descriptor and read as 0; the chunk is a SCALAR
    // offset of the load) and the two destinations inside a plane buffer -- a few registers next to the 80 that carry x.
    // (Address arithmetic per use, in ten unrolled bodies, is hoisted out of the tile loop by the compiler and spills.)
    constexpr unsigned XOOB = 0x80000000u;     // out of range for any tile, also with the scalar chunk offset added
    unsigned xoff[2];
    int pdst[2], prow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = ht + FT_HELPER_THREADS * i;
      const int row = p >> 5, k4 = (p & 31) * 4;
      xoff[i] = k4 < t.F_in ? (unsigned)(row * t.F_in + k4) * 4u : XOOB;
      pdst[i] = row * (FT_KP * 2) + ((((k4 >> 3) ^ row) & 15) << 4) + (k4 & 7) * 2;
      prow[i] = row;
    }
    auto x_load = [&](f4& d0, f4& d1, const __amdgpu_buffer_rsrc_t rs, int c) {
      const int so = c * FT_CHUNK * t.F_in * 4;
      d0 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, xoff[0], so, 0));
      d1 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, xoff[1], so, 0));
    };
    auto split = [&](const f4 v0, const f4 v1, int buf) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f4 v = i == 0 ? v0 : v1;
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
        char* dstp = lds_planes + buf * (2 * FT_PLANE_BYTES) + pdst[i];
        *reinterpret_cast<ft_u2*>(dstp) = ft_u2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
        *reinterpret_cast<ft_u2*>(dstp + FT_PLANE_BYTES) = ft_u2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
        lds_rowinv[buf * FT_CHUNK + prow[i]] = __uint_as_float(e);          // 2^e (the 32 lanes of a row write the same word)
      }
    };
    // chunks 0 and 1 of a tile -> plane buffers 0 and 1: in front of the tile's first barrier, i.e. at the end of the tile
    // before (the buffers are free from the last GEMM step on, the rows arrived during the CSR build)
    auto stage01 = [&](const Tile& r) {
#ifdef EGC_FT_STAMPS
      if (t.dbg & 1) return;
#endif
      if (0 < r.nch) split(xr[0], xr[1], 0);           // (a tile that is skipped has no chunks: only the barriers remain)
      if (1 < r.nch) split(xr[2], xr[3], 1);
    };
    // a tile's rows of x through a descriptor of their own (no 4 GiB limit on x; rows beyond the tile read as 0)
    auto x_rsrc_of = [&](const Tile& r) {
      return __builtin_amdgcn_make_buffer_rsrc((void*)(t.x + (int64_t)r.n0 * t.F_in), 0,
                                               (unsigned)(r.ok ? r.T : 0) * (unsigned)t.F_in * 4u, 0x00020000);
    };
    // (backward) what the rows pass needs of a tile before it starts, done here a tile ahead: the largest |g| of its rows -- the
    // scale of the 64-bit fixed point d bases is summed in comes from it and from the largest |w'| -- into the tile's record
    // (four batches of four 16-byte pieces per thread cover 128 rows of 128 floats; pieces beyond the tile lie outside the
    // descriptor), and the d bases area zeroed.  (As a pass of the twelve workers behind the first GEMM, with g brought in by
    // LDS-DMA, this was 14 k cycles per tile: every instruction of straight-line code the workers all run costs twelve cycles.)
    auto g_max = [&](const Tile& r, int slot, auto wide_batches) {
      if constexpr (MODE == 1) {
        constexpr int PB = decltype(wide_batches)::value ? 8 : 4, NB = 16 / PB;      // 16 pieces per thread, PB in flight
        const int F_o = C::F_out(a);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(t.grad_out + (int64_t)r.n0 * F_o), 0,
                                                                           (unsigned)(r.ok ? r.T : 0) * (unsigned)F_o * 4u, 0x00020000);
        unsigned m = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          f4 v[PB];
#pragma unroll
          for (int i = 0; i < PB; ++i)       // (the piece is a SCALAR offset: as per-lane offsets the sixteen addresses were spilled, and
                                             //  every reload stood, with a vmcnt(0), in front of its request -- one round trip per piece)
            v[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rg, (unsigned)ht * 16u, (PB * b + i) * FT_HELPER_THREADS * 16, 0));
#pragma unroll
          for (int i = 0; i < PB; ++i)
            m = max(m, ft_amax_bits(v[i]));
        }
        m = ft_wave_umax(m);
        if (lane == 0) __hip_atomic_fetch_max(reinterpret_cast<unsigned*>(lds_rec) + slot * 8 + 5, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    };
    auto zero_db = [&](const Tile& r) {
      if constexpr (MODE == 1) {
        f4* db4 = reinterpret_cast<f4*>(base + t.off_db);       // (64-bit fixed point: two 16-byte pieces per slot)
        const int npd = r.ok ? r.T * (a.ldb >> 1) : 0;
        for (int i = ht; i < npd; i += FT_HELPER_THREADS) db4[i] = f4{0.f, 0.f, 0.f, 0.f};
      }
    };
    const Tile first = read_tile(0);
    // the first tile's CSR while its rows travel (every later one is built during the GEMM steps of the tile before)
    unsigned epk0[KEEP];
    {
      const __amdgpu_buffer_rsrc_t rs = x_rsrc_of(first);
      if (csr_wave) csr_s0(first, xr + EREG);
#pragma unroll
      for (int c = 0; c < H1; ++c) x_load(xr[2 * c], xr[2 * c + 1], rs, c);   // (chunks beyond the tile read as 0)
      if (csr_wave) {
        csr_s1(first, 0, xr + EREG, epk0);
        csr_s1_rest(first, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = H1; c < RINGN; ++c) x_load(xr[2 * c], xr[2 * c + 1], rs, c);
    }
    if (csr_wave) {
      csr_sync();
      csr_s2(first, 0);
      csr_sync();
      csr_s3(first, 0, epk0);
    }
    if constexpr (MODE == 1) g_max(first, 0, std::true_type{});        // (in front of the first barrier: two round trips instead of four)
    zero_db(first);
    stage01(first);
    for (int it = 0;; ++it) {
      const Tile cur = read_tile(it % 3);
      if (!cur.valid) break;
      const Tile nxt = read_tile((it + 1) % 3);
      const __amdgpu_buffer_rsrc_t rsn = x_rsrc_of(nxt);
      // (chunks 0 and 1 were staged at the end of the tile before / in front of the loop)
      lds_barrier();                                   // (chunks 0 and 1 staged; the CSR of this tile complete)
      unsigned epk[KEEP];                      // (declared per tile: nothing of the build is carried over)
      const int nset = (it + 1) & 1;
      // the counters of the set the next tile's CSR is built into (last read in the rows phase of tile it - 1); the GEMM steps'
      // barriers stand between this and the build's first count (a tile without steps: the builders' own synchronisation)
      if (csr_wave) { csr_zero(nset); if (cur.nch == 0) csr_sync(); }
#pragma unroll
      for (int c = 0; c < RINGN; ++c) {
        if (c < cur.nch) {   // workgroup-uniform
#ifdef EGC_FT_STAMPS
          if (!(t.dbg & 1))
#endif
          // two chunks ahead: the workers take the first half of chunk c + 1 while they are in step c, and chunk c - 1, whose
          // buffer this is, was read in step c - 1 at the latest
          if (c + 2 < RINGN && c + 2 < cur.nch) split(xr[2 * (c + 2)], xr[2 * (c + 2) + 1], (c + 2) % FT_PBUF);
          lds_barrier();
        }
#ifdef EGC_FT_EARLY_X
        // The NEXT tile's rows are requested as soon as the registers of a chunk are free -- chunks 0 and 1 at once (they were
        // staged at the end of the tile before), chunk c + 2 behind its split -- instead of all in the rows phase: the launch's
        // memory skeleton (no split, no matrix work, no rows: 110 k of a workgroup's 227 k cycles at config 4) is the x stream
        // running in the rows-phase window only, 40 % of the time.  Unconditional, straight-line (a chunk beyond the next tile
        // lies outside its descriptor); chunks 5 - 8 stay behind the CSR build's counts: its edges travel in their registers.
        if (c == 0) { x_load(xr[0], xr[1], rsn, 0); x_load(xr[2], xr[3], rsn, 1); }
        if (c + 2 < RINGN && (c + 2 < RINGN / 2 || c + 2 == RINGN - 1)) x_load(xr[2 * (c + 2)], xr[2 * (c + 2) + 1], rsn, c + 2);
#endif
      }
#ifdef EGC_FT_STAMPS
      unsigned long long ft_h0;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ft_h0) :: "memory");
#endif
      // Behind the workgroup's LAST tile there is nothing to plan, build or request: straight to the end-of-tile barrier.  (Same-box
      // A/B under rocprofv3, profiles/r05_last_tile_skip.log: ZINC b128 13.64 -> 13.24 us, config 3 31.5 -> 31.0, config 4 111.9 -> 111.0.)
#ifndef EGC_FT_NO_LAST_TILE_SKIP
      if constexpr (MODE == 0) {
        if (!nxt.valid) { lds_barrier(); break; }
      }
#endif

      // the tile after the next, while the workers are in their rows phase (its dependent loads -- graph offsets, then edge
      // offsets or the search -- take two to five memory round trips: in front of a barrier they were 15 % of the kernel).
      // The record's slot was last read during tile it - 1.
      // (the helpers' few instructions go first from here to the end of the tile: behind the twelve workers' streams of LDS
      // reads every dependent step of the CSR build waited its turn at the issue arbiter)
      __builtin_amdgcn_s_setprio(3);
      if (wave == FT_WAVES - 1) plan_tile((it + 2) % 3);
#ifdef EGC_FT_STAMPS
      FT_HSTAMP(7, wave == FT_WAVES - 1)
#endif
      // the next tile's CSR, into the other set of areas (last read during the rows of tile it - 1), interleaved with the
      // requests for its rows: edges | rows of chunks 0-4 | in-degrees | scan | scatter | rows of chunks 5-9.  The
      // vector-memory counter is in order: the edges are requested first, so the wait for them leaves the ten row requests
      // behind them in flight; they travel in the registers of the second half of the rows (csr_s0).  All ten chunks are
      // requested, unconditionally (the ones beyond the tile lie outside its descriptor and cost no traffic): the compiler can
      // then count the requests in flight.  (Requested chunk by chunk inside the loop above, its conservative vmcnt(0) in
      // front of every split made each step wait for the request it had just issued.)
      if (csr_wave) csr_s0(nxt, xr + EREG);
#ifndef EGC_FT_EARLY_X
#pragma unroll
      for (int c = 0; c < H1; ++c) x_load(xr[2 * c], xr[2 * c + 1], rsn, c);
#endif
      if (csr_wave) {
        csr_s1(nxt, nset, xr + EREG, epk);
        csr_s1_rest(nxt, nset);
      }
      __builtin_amdgcn_sched_barrier(0);       // (the second half of the rows into the registers the edges have left)
#ifdef EGC_FT_STAMPS
      FT_HSTAMP(1, wave == FT_FIRST_HELPER)
#endif
      if (csr_wave) {
        csr_sync();
        csr_s2(nxt, nset);
        csr_sync();
        csr_s3(nxt, nset, epk);
      }
#ifdef EGC_FT_STAMPS
      FT_HSTAMP(3, wave == FT_FIRST_HELPER)
#endif
      // (chunks 5-9 are not split before the next tile's fifth step: their requests -- 1,300 cycles of the CU's one
      // vector-memory pipeline -- need not stand between the in-degrees and the scan)
#ifdef EGC_FT_EARLY_X
#pragma unroll
      for (int c = H1; c < RINGN - 1; ++c) x_load(xr[2 * c], xr[2 * c + 1], rsn, c);
#else
#pragma unroll
      for (int c = H1; c < RINGN; ++c) x_load(xr[2 * c], xr[2 * c + 1], rsn, c);
#endif
#ifdef EGC_FT_STAMPS
      FT_HSTAMP(2, wave == FT_FIRST_HELPER)
#endif
      if constexpr (MODE == 1) { if (nxt.valid) g_max(nxt, (it + 1) % 3, std::false_type{}); }
      if constexpr (MODE == 0) {
        stage01(nxt);
        __builtin_amdgcn_s_setprio(0);
        lds_barrier();                                   // (end of tile)
      } else {
        // ---- backward: the rows pass has left d bases [T][ldb] and d weightings [T][H B 4] in the LDS images.  Their rows
        //      are staged for the second GEMM (d x = [d bases | d weightings] [bases_weight | comb_weight^T]^T) exactly as x
        //      was for the first -- row scale, two fp16 planes, one 16-row chunk ahead of the MFMAs -- by 16 threads per row
        //      (thread j: the row's 16-byte pieces j, j + 16, j + 32), and leave for memory as d_cat on the way (the weight
        //      gradient x^T d is a launch of its own).
        __builtin_amdgcn_s_setprio(0);
        lds_barrier();                                   // (B1: the rows pass is done)
        const int dbs_h = __builtin_amdgcn_readfirstlane(lds_rec[28]);
        const bool poison = dbs_h == FT_DBS_POISON;      // (a non-finite g or w' in the tile: its d bases rows leave as NaN)
        const double sinv = __builtin_ldexp(1.0, -(poison ? 0 : dbs_h));
        const int drow = ht >> 4, dj = ht & 15;
        const int np2 = (a.ldb + t.wl_floats) >> 6;      // pieces per thread: 2 (H = 4) or 3 (H = 8)
        auto stage_d = [&](int c, int buf) {
          const int r = FT_CHUNK * c + drow;
          f4 pc[3];
          {   // d bases: 64-bit fixed point (the rows pass adds with integer LDS atomics), scale 2^-dbs in lds_rec[28]
            const long long* dq = reinterpret_cast<const long long*>(base + t.off_db) + (r * a.ldb + 4 * dj);
            pc[0] = f4{(float)((double)dq[0] * sinv), (float)((double)dq[1] * sinv), (float)((double)dq[2] * sinv), (float)((double)dq[3] * sinv)};
            if (poison) pc[0] = f4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
          }
          pc[1] = *reinterpret_cast<const f4*>(base + t.off_wt + (r * t.wl_floats + 4 * dj) * 4);
          pc[2] = np2 > 2 ? *reinterpret_cast<const f4*>(base + t.off_wt + (r * t.wl_floats + 64 + 4 * dj) * 4) : f4{0.f, 0.f, 0.f, 0.f};
          // ... and leave for memory as d_cat [n_nodes][ld_dcat] = [d bases (ldb) | d weightings in the layer's column order
          // (h B + b) A + a] on the way (block (h, b) = piece dj / 16 + dj of the w' image)
#ifdef EGC_FT_STAMPS
          if (!(t.dbg & 2048))
#endif
          if (t.d_cat != nullptr && r < cur.T) {
            float* dc = t.d_cat + (int64_t)(cur.n0 + r) * t.ld_dcat;
            __builtin_nontemporal_store(pc[0], reinterpret_cast<f4*>(dc + 4 * dj));
            const int Ad = C::A(a);
#pragma unroll
            for (int i = 1; i < 3; ++i) {
              if (i < np2) {
                const int hb = 16 * (i - 1) + dj;
                if (Ad == 4) {
                  __builtin_nontemporal_store(pc[i], reinterpret_cast<f4*>(dc + a.ldb + 4 * hb));
                } else {
                  float* dw = dc + a.ldb + hb * Ad;
                  dw[0] = pc[i].x;
                  if (Ad > 1) dw[1] = pc[i].y;
                  if (Ad > 2) dw[2] = pc[i].z;
                }
              }
            }
          }
          float m = 0.f;
#pragma unroll
          for (int i = 0; i < 3; ++i) m = fmaxf(fmaxf(fmaxf(m, fabsf(pc[i].x)), fmaxf(fabsf(pc[i].y), fabsf(pc[i].z))), fabsf(pc[i].w));
          unsigned am = __float_as_uint(m);
          am = max(am, (unsigned)__builtin_amdgcn_update_dpp(0, (int)am, 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
          am = max(am, (unsigned)__builtin_amdgcn_update_dpp(0, (int)am, 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
          am = max(am, (unsigned)__builtin_amdgcn_update_dpp(0, (int)am, 0x141, 0xf, 0xf, true));  // row_half_mirror
          am = max(am, (unsigned)__builtin_amdgcn_update_dpp(0, (int)am, 0x140, 0xf, 0xf, true));  // row_mirror
          unsigned e = am & 0x7f800000u;
          e = min(max(e, 13u << 23), 253u << 23);
          const float sc = __uint_as_float(0x7f000000u - e), sc2k = __uint_as_float(0x7f000000u + (11u << 23) - e);
          char* dst0 = lds_planes + buf * FTB_PBUF_BYTES + drow * FTB_ROW_BYTES + 8 * dj;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            if (i < np2) {
              const f4 v = pc[i];
              const ft_h2 h01 = __builtin_convertvector(ft_f2{v.x * sc, v.y * sc}, ft_h2);
              const ft_h2 h23 = __builtin_convertvector(ft_f2{v.z * sc, v.w * sc}, ft_h2);
              ft_h2 l01, l23;
              l01[0] = (_Float16)__builtin_fmaf((float)h01[0], -2048.f, v.x * sc2k);
              l01[1] = (_Float16)__builtin_fmaf((float)h01[1], -2048.f, v.y * sc2k);
              l23[0] = (_Float16)__builtin_fmaf((float)h23[0], -2048.f, v.z * sc2k);
              l23[1] = (_Float16)__builtin_fmaf((float)h23[1], -2048.f, v.w * sc2k);
              char* dstp = dst0 + 128 * i;
              *reinterpret_cast<ft_u2*>(dstp) = ft_u2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
              *reinterpret_cast<ft_u2*>(dstp + FTB_PLANE_BYTES) = ft_u2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
            }
          }
          reinterpret_cast<float*>(base + t.off_rowinv2)[buf * FT_CHUNK + drow] = __uint_as_float(e);
        };
#ifdef EGC_FT_STAMPS
        const bool do_stage = !(t.dbg & 8192);
#else
        constexpr bool do_stage = true;
#endif
        if (cur.nch > 0 && do_stage) stage_d(0, 0);
        lds_barrier();                                   // (A: chunk 0 of d staged)
        for (int c = 0; c < cur.nch; ++c) {
          if (c + 1 < cur.nch && do_stage) stage_d(c + 1, (c + 1) & 1);
          lds_barrier();
        }
        zero_db(nxt);
        stage01(nxt);
      }
    }
    return;
    } else {
    // =====================================================================================================================
    // WIDE: the x stream in 32-row chunks.  Eight threads per row (thread j of a row: its 16-byte pieces j, j + 8, ... --
    // FTW_PP per chunk, 128 contiguous bytes per row and request), TWO chunks in registers (slot = chunk & 1: xa, xb); a
    // chunk leaves for the LDS planes in k-slabs of 128 (a whole row's planes of 16 x 320 x 2 would take the LDS the tile's
    // image needs), one slab per GEMM step, staged one step ahead into the other of two plane buffers; when a chunk's last
    // slab has left, its registers are re-requested with the chunk two further on (rows beyond the tile lie outside its
    // descriptor: no traffic, but always issued -- the counted waits below rely on the order of the requests).
    // The loads are inline assembly with counted waits: inside the uniform branches of the unrolled steps the compiler's own
    // bookkeeping falls back to vmcnt(0) in front of every use.  Order of a tile's requests: [chunk 0][chunk 1] (during the
    // rows phase of the tile before, drained before stage0), then chunk c + 2 right after the last slab of chunk c is split --
    // always BEFORE slab 0 of chunk c + 1 is needed, so that wait leaves exactly the FTW_PP requests of chunk c + 2 in flight.
    // Barrier sequence = the workers': 1 + nch n_slabs + 1 per tile.
    // =====================================================================================================================
    // The requests stand in straight-line code, every one of them unconditional (as in the narrow form): the steps of a tile are
    // unrolled with the number of k-slabs a template parameter, so that each chunk has ONE place where its registers are
    // re-requested; a request for a chunk the tile does not have lies outside the tile's descriptor (zeros, no traffic) and
    // lands in registers nobody reads again.  (Requests inside the uniform branches of the steps were merged by the compiler
    // through copies of all 40 registers of a chunk -- and as inline assembly with counted waits some of those copies ran while
    // the load into their source was in flight.)
    constexpr int NS = WIDE;             // k-slabs of 128 per chunk: ceil(F_in / 128) -- the host picks the instance
    // 16-byte pieces per thread and chunk: a row's groups of eight pieces -- F_in <= 256 needs 8 of them, 320 needs 10 (the two
    // fewer are 16 registers the split's temporaries otherwise spilled for, with a vmcnt(0) behind every reload)
    constexpr int PP = NS == 3 ? FTW_PP : 8;
    f4 xa[PP], xb[PP];
    const int hrow = ht >> 3, hj = ht & 7;
    constexpr unsigned XOOB = 0x80000000u;
    // piece i of a row = columns 4 (hj + 8 i) ..+3.  A row's pieces come in groups of eight: group i is whole for i < F_in / 32,
    // partial (lanes hj < (F_in / 4) % 8) for i == F_in / 32, absent beyond.  Absent groups are switched off through the SCALAR
    // offset (no traffic); the lanes beyond a partial group's end read the first floats of the NEXT row and are masked where the
    // piece is read (piece()).
    const unsigned xoff0 = (unsigned)(hrow * t.F_in + 4 * hj) * 4u;
    const int npf = (t.F_in >> 2) >> 3;                                       // whole groups of a row
    const int npg = ((t.F_in >> 2) + 7) >> 3;                                 // groups of a row
    const bool lane_cut = npg != npf && hj >= ((t.F_in >> 2) & 7);            // this lane has no piece in the partial group
    float* lds_rowinv_w = lds_rowinv;                        // [2][32]: chunk parity, row
    auto x_rsrc_of = [&](const Tile& r) {
      return __builtin_amdgcn_make_buffer_rsrc((void*)(t.x + (int64_t)r.n0 * t.F_in), 0,
                                               (unsigned)(r.ok ? r.T : 0) * (unsigned)t.F_in * 4u, 0x00020000);
    };
    auto request = [&](f4 (&xs)[PP], const __amdgpu_buffer_rsrc_t rs, int chunk) {
      const int so = chunk * FTW_CH * t.F_in * 4;
#pragma unroll
      for (int i = 0; i < PP; ++i)
        xs[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, xoff0 + 128u * i, i < npg ? so : (int)XOOB, 0));
    };
    // piece i of a landed chunk (the partial group's missing lanes read as zero)
    auto piece = [&](const f4 (&xs)[PP], int i) -> f4 {
      const bool cut = lane_cut && i == npf;
      return f4{cut ? 0.f : xs[i].x, cut ? 0.f : xs[i].y, cut ? 0.f : xs[i].z, cut ? 0.f : xs[i].w};
    };
    // exponent field of a row's largest magnitude (8 lanes hold a row), clamped as in the narrow form
    auto row_exp = [&](const f4 (&xs)[PP]) -> unsigned {
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < PP; ++i) {
        const f4 v = piece(xs, i);
        m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fabsf(v.z))), fabsf(v.w));
      }
      unsigned am = __float_as_uint(m);
      am = max(am, (unsigned)__builtin_amdgcn_update_dpp(0, (int)am, 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
      am = max(am, (unsigned)__builtin_amdgcn_update_dpp(0, (int)am, 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
      am = max(am, (unsigned)__builtin_amdgcn_update_dpp(0, (int)am, 0x141, 0xf, 0xf, true));  // row_half_mirror
      unsigned e = am & 0x7f800000u;
      return min(max(e, 13u << 23), 253u << 23);
    };
    // slab sl of the chunk in xs -> plane buffer buf: pieces 4 sl .. 4 sl + 3 (k = 128 sl + 4 (hj + 8 (i - 4 sl)) ..+3)
    auto split_slab = [&](const f4 (&xs)[PP], int sl, unsigned e, int buf) {
      const float sc = __uint_as_float(0x7f000000u - e);                  // 2^-e
      const float sc2k = __uint_as_float(0x7f000000u + (11u << 23) - e);  // 2^(11-e)
      char* dst0 = lds_planes + buf * FTW_PBUF_BYTES + hrow * (FTW_LDX * 2) + 8 * hj;
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = 4 * sl + ii;
        if (i < PP) {
          const f4 v = piece(xs, i);
          const ft_h2 h01 = __builtin_convertvector(ft_f2{v.x * sc, v.y * sc}, ft_h2);
          const ft_h2 h23 = __builtin_convertvector(ft_f2{v.z * sc, v.w * sc}, ft_h2);
          ft_h2 l01, l23;
          l01[0] = (_Float16)__builtin_fmaf((float)h01[0], -2048.f, v.x * sc2k);
          l01[1] = (_Float16)__builtin_fmaf((float)h01[1], -2048.f, v.y * sc2k);
          l23[0] = (_Float16)__builtin_fmaf((float)h23[0], -2048.f, v.z * sc2k);
          l23[1] = (_Float16)__builtin_fmaf((float)h23[1], -2048.f, v.w * sc2k);
          char* dstp = dst0 + 64 * ii;
          *reinterpret_cast<ft_u2*>(dstp) = ft_u2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
          *reinterpret_cast<ft_u2*>(dstp + FTW_PLANE_BYTES) = ft_u2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
          // (one piece at a time: interleaved, the four pieces' temporaries were a register too many next to the chunks in
          // flight -- one spilled dword, reloaded inside the GEMM steps with a vmcnt(0) that also waited for the rows just requested)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    unsigned e_cur = 13u << 23;          // the exponent of the chunk whose slabs are being split
    // chunk `chunk` (in xs, landed) starts: its rows' scales, its slab 0 -> buf   (reads of xs only)
    auto start_chunk = [&](const f4 (&xs)[PP], int chunk, int buf) {
      e_cur = row_exp(xs);
      lds_rowinv_w[(chunk & 1) * FTW_CH + hrow] = __uint_as_float(e_cur);      // 2^e (the 8 lanes of a row write the same word)
      split_slab(xs, 0, e_cur, buf);
    };
    // chunk 0 of a tile (requested, with chunk 1, during the rows phase before): slab 0 -> buffer 0; with one slab per chunk its
    // registers are free for chunk 2 at once
    auto stage0 = [&](const Tile& r, const __amdgpu_buffer_rsrc_t rs) {
      if (r.nch > 0) start_chunk(xa, 0, 0);
      if constexpr (NS == 1) request(xa, rs, 2);
    };

    const Tile first = read_tile(0);
    unsigned epk0[KEEP];
    {
      const __amdgpu_buffer_rsrc_t rs = x_rsrc_of(first);
      if (csr_wave) csr_s0(first, xb + (PP - 8));
      request(xa, rs, 0);
      if (csr_wave) {
        csr_s1(first, 0, xb + (PP - 8), epk0);
        csr_s1_rest(first, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      request(xb, rs, 1);
      if (csr_wave) {
        csr_sync();
        csr_s2(first, 0);
        csr_sync();
        csr_s3(first, 0, epk0);
      }
      stage0(first, rs);
    }
    for (int it = 0;; ++it) {
      const Tile cur = read_tile(it % 3);
      if (!cur.valid) break;
      const Tile nxt = read_tile((it + 1) % 3);
      const __amdgpu_buffer_rsrc_t rsc = x_rsrc_of(cur), rsn = x_rsrc_of(nxt);
      lds_barrier();                                   // (slab 0 of chunk 0 staged; the CSR of this tile complete)
      unsigned epk[KEEP];
      const int nset = (it + 1) & 1;
      if (csr_wave) { csr_zero(nset); if (cur.nch == 0) csr_sync(); }     // (as in the narrow form)
      int qn = 1;                                      // plane buffer of the step being staged (the step after the one running)
#pragma unroll
      for (int c = 0; c < FTW_MAXCH; ++c) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
#ifdef EGC_FT_STAMPS
          const bool on = c < cur.nch, work = on && !(t.dbg & 1);       // (diagnostic build: EGC_FT_DBG bit 0 = no split)
#else
          const bool on = c < cur.nch;                                  // the step exists (workgroup-uniform)
          const bool work = on;
#endif
          if (c & 1) {      // chunk c lives in xb, chunk c + 1 in xa
            if (work && sl + 1 < NS) split_slab(xb, sl + 1, e_cur, qn);                     // the chunk's next slab ...
            if (work && sl + 1 == NS && c + 1 < cur.nch) start_chunk(xa, c + 1, qn);        // ... or the next chunk's first
#ifdef EGC_FT_STAMPS
            if (!(t.dbg & 8))        // (diagnostic build: EGC_FT_DBG bit 3 = no requests inside the GEMM steps)
#endif
            {
            if (NS >= 2 && sl + 2 == NS) request(xb, rsc, c + 2);                           // the chunk's last slab has left its registers
            if (NS == 1) request(xa, rsc, c + 3);
            }
          } else {
            if (work && sl + 1 < NS) split_slab(xa, sl + 1, e_cur, qn);
            if (work && sl + 1 == NS && c + 1 < cur.nch) start_chunk(xb, c + 1, qn);
#ifdef EGC_FT_STAMPS
            if (!(t.dbg & 8))
#endif
            {
            if (NS >= 2 && sl + 2 == NS) request(xa, rsc, c + 2);
            if (NS == 1) request(xb, rsc, c + 3);
            }
          }
          if (on) {
            lds_barrier();
            qn ^= 1;
          }
        }
      }
      __builtin_amdgcn_s_setprio(3);
      if (wave == FT_WAVES - 1) plan_tile((it + 2) % 3);
      if (csr_wave) csr_s0(nxt, xb + (PP - 8));
      request(xa, rsn, 0);
      if (csr_wave) {
        csr_s1(nxt, nset, xb + (PP - 8), epk);
        csr_s1_rest(nxt, nset);
      }
      __builtin_amdgcn_sched_barrier(0);
      request(xb, rsn, 1);
      if (csr_wave) {
        csr_sync();
        csr_s2(nxt, nset);
        csr_sync();
        csr_s3(nxt, nset, epk);
      }
      stage0(nxt, rsn);
      __builtin_amdgcn_s_setprio(0);
      lds_barrier();                                   // (end of tile)
    }
    return;
    }
  }

  // =======================================================================================================================
  // wavefronts 0-11: CSR of the tile in LDS, the matrix-core step of every chunk the helpers stage, the rows
  // =======================================================================================================================
  const int g = lane >> LPR_LOG2;
  const int q = lane & (LPR - 1);
  const int F_out = C::F_out(a);
  const bool is_mfma = wave < t.n_ct;
  f4* lds_bases4 = reinterpret_cast<f4*>(base + t.off_bases);
  float* lds_wt = reinterpret_cast<float*>(base + t.off_wt);

  // this wavefront's 16-column tile of the packed weights: its column's inverse scale and bias, and where its column goes in
  // the LDS image.  The tile itself (both planes of 128 x 16 as B operands: lane -> column 16 wave + lane % 16, k = 32 s +
  // 8 (lane / 16) ..+7) is fetched again for every tile of graphs, from L2: kept across the rows phase its 32 registers
  // push that phase's working set out of the register file.
  float col_inv = 0.f, col_bias = 0.f;
  int dst_off = -1, dst_stride = 0;        // byte offset inside a row of the image area / bytes between its rows
  bool dst_act = false;
  if constexpr (WIDE == 0) {
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
  } else if (is_mfma) {
    // WIDE: this wavefront's 32 virtual columns [bases: 0 .. ldb) | padding to a multiple of 32 | weightings: ldbp .. ldbp + W)
    const float* tail = reinterpret_cast<const float*>(t.packed + (int64_t)t.n_ct * t.k16 * 2 * 64 * 8);
    const int v = 32 * wave + (lane & 31);
    col_inv = tail[v];
    col_bias = tail[FTW_MAX_CT * 32 + v];
    if (v < a.ldb) {
      dst_off = t.off_bases + v * 4;
      dst_stride = a.ldb * 4;
    } else if (v >= t.ldbp && v - t.ldbp < C::W(a)) {
      const int wc = v - t.ldbp;
      const int hb = wc / C::A(a);
      dst_off = t.off_wt + (hb * t.w_aw + (wc - hb * C::A(a))) * 4;
      dst_stride = t.wl_floats * 4;
      dst_act = true;
    }
  }
  const int ldb4 = a.ldb >> 2;
  const unsigned ldb_bytes = (unsigned)a.ldb * 4u;
  const int zrow = t.tcap;                     // the row behind the image's last one: zeros (written once, below)
  const char* bases_q = base + t.off_bases + (q < C::slots(a) ? q : 0) * 16;
  FastRsrc R;
  R.bases = bases_rsrc(a);
  R.out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (unsigned)a.n_nodes * (unsigned)F_out * 4u, 0x00020000);
  R.res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual != nullptr ? a.residual : a.out), 0,
                                            (unsigned)a.n_nodes * (unsigned)F_out * 4u, 0x00020000);
  const bool looped_any = C::xl(a) || C::yl(a);
  const bool want_dis = a.dis != nullptr;
  const int max_index = (!C::loops_all(a) && t.max_index != nullptr) ? *t.max_index : 0x7fffffff;
  const int grp_addr = (g << LPR_LOG2) << 2;
  for (int i = tid; i < ldb4; i += FT_WORKER_THREADS) lds_bases4[zrow * ldb4 + i] = f4{0.f, 0.f, 0.f, 0.f};
  lds_barrier();       // bias strips, the first two tile records
#ifdef EGC_FT_STAMPS
  unsigned long long ft_pro = 0;
  if (tid == 0) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); ft_pro = _t - ft_t0; ft_t0 = _t; }
#endif

  // this wavefront's 16-column tile of the packed weights, from L2, once per tile of graphs (kept across the rows phase its 32
  // registers push that phase's working set out of the register file): requested when the wavefront has left the rows of the
  // tile before, so that it travels while the others finish theirs and the helpers stage the first chunks
  f4 u[8];     // u[2 s + p] = k-step s, plane p of the weight tile
  auto request_weights = [&]() {
    if constexpr (WIDE != 0) return;      // (the wide form streams its weight fragments: nothing is resident)
    if (is_mfma) {
      int lv = lane;
      asm volatile("" : "+v"(lv));
      const f4* wsrc = reinterpret_cast<const f4*>(t.packed) + (int64_t)wave * 8 * 64;   // wave-uniform base + lane
#pragma unroll
      for (int s = 0; s < 8; ++s) u[s] = wsrc[s * 64 + lv];
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) u[s] = f4{0.f, 0.f, 0.f, 0.f};
    }
  };
  request_weights();
  for (int it = 0;; ++it) {
    const Tile cur = read_tile(it % 3);
    if (!cur.valid) break;
    const int n0 = cur.n0, T = cur.T, nch = cur.nch;
    if (!cur.ok && (T > 0 || cur.Et > 0)) {
      if (tid == 0) ft_error(t, T <= 0 ? 1 : 2);
      // a tile beyond the LDS image is reported and its rows become zeros, not whatever the allocation held (the caller may
      // read `out` before it checks the status)
      if (T > 0) {
        const int64_t lo = (int64_t)n0 * F_out, hi = (int64_t)(n0 + T > a.n_nodes ? a.n_nodes : n0 + T) * F_out;
        for (int64_t i = lo + tid; i < hi; i += FT_WORKER_THREADS) a.out[i] = 0.f;
      }
    }
    // this tile's CSR (built by wavefront 14 during the previous tile's rows phase)
    char* cb = base + (it & 1) * t.csr_stride;
    const unsigned short* lds_col = reinterpret_cast<const unsigned short*>(cb + t.off_col);
    // (a row's start is the LOW half of its rowptr word; the high half is a cursor of the CSR build: csr_s2)
    const int* lds_rowptr = reinterpret_cast<const int*>(cb + t.off_rowptr);
    const float* lds_dis = reinterpret_cast<const float*>(cb + t.off_dis);

    // ---- this wavefront's weight tile: requested when the wavefront left the rows of the tile before (below); the row counter ----
    if (tid == 0) *lds_rowctr = 0;
    // the weight tile has landed HERE as far as the compiler is concerned (else it waits for it inside the GEMM loop)
    if constexpr (WIDE == 0) asm volatile("" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]));
    lds_barrier();       // chunk 0 is staged, the tile's CSR complete, the row counter zero
    FT_STAMP(0)
    unsigned wmx = 0;                      // (backward: largest |w'| of the tile, kept by the GEMM's epilogue below)

    // ---- (G) [bases | weightings] of the tile, 16 rows per step.  The A fragments of a chunk are read in two halves: k-steps
    //      2, 3 at the start of its step (the MFMAs of k-steps 0, 1 run meanwhile), k-steps 0, 1 at the END OF THE STEP BEFORE
    //      (behind its last MFMA: their LDS latency passes during the epilogue and the barrier) -- the helpers stage two
    //      chunks ahead for that.  (In lock step -- barrier, eight reads, twelve MFMAs, epilogue -- the matrix pipe of a SIMD
    //      was busy 576 of a step's 1,100 cycles.) ----
    if constexpr (WIDE == 0) {
    int lvm = lane;
    asm volatile("" : "+v"(lvm));
    const int m = lvm & 15, qd = lvm >> 4;
    auto a_read = [&](int buf, int s, ft_h8& xh, ft_h8& xl) {
      const char* pa = lds_planes + buf * (2 * FT_PLANE_BYTES) + m * (FT_KP * 2) + ((((4 * s + qd) ^ m) & 15) << 4);
      xh = *reinterpret_cast<const ft_h8*>(pa);
      xl = *reinterpret_cast<const ft_h8*>(pa + FT_PLANE_BYTES);
    };
    // (the same two reads, written out: the compiler then does not know of them and puts no wait for them in front of the
    // step's first MFMA -- which would also wait for the step's own four reads, LDS returning in order.  The barrier that
    // always stands between such a request and its use waits for the LDS counter itself.)
    static_assert(FT_PLANE_BYTES == 4096, "offset of the low plane in a_prefetch");
    auto a_prefetch = [&](int buf, int s, ft_h8& xh, ft_h8& xl) {
      const char* pa = lds_planes + buf * (2 * FT_PLANE_BYTES) + m * (FT_KP * 2) + ((((4 * s + qd) ^ m) & 15) << 4);
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096"
                   : "=&v"(xh), "=&v"(xl) : "v"((unsigned)(uintptr_t)pa) : "memory");
    };
    ft_h8 ah[2], al[2];        // k-steps 0, 1 of the chunk of the coming step
    ah[0] = al[0] = ah[1] = al[1] = ft_h8{0, 0, 0, 0, 0, 0, 0, 0};
#ifdef EGC_FT_STAMPS
    if (!(t.dbg & 2))
#endif
    if (is_mfma && nch > 0) {
      a_prefetch(0, 0, ah[0], al[0]);
      a_prefetch(0, 1, ah[1], al[1]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the one time a tile that no barrier follows the request)
    }
    int buf = 0;
    for (int c = 0; c < nch; ++c) {
#ifdef EGC_FT_STAMPS
      if (!(t.dbg & 2))
#endif
      if (is_mfma) {
        ft_h8 bh[2], bl[2];
        a_read(buf, 2, bh[0], bl[0]);
        a_read(buf, 3, bh[1], bl[1]);
        const f4 ri = *reinterpret_cast<const f4*>(lds_rowinv + buf * FT_CHUNK + 4 * qd);   // (the rows' scales, for the epilogue)
        __builtin_amdgcn_sched_barrier(0);
        f4 acc0 = f4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const ft_h8 wh = __builtin_bit_cast(ft_h8, u[2 * s]), wl = __builtin_bit_cast(ft_h8, u[2 * s + 1]);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s], wh, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[s], wh, acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s], wl, acc2, 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const ft_h8 wh = __builtin_bit_cast(ft_h8, u[4 + 2 * s]), wl = __builtin_bit_cast(ft_h8, u[4 + 2 * s + 1]);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[s], wh, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[s], wh, acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[s], wl, acc2, 0, 0, 0);
        }
        const int nbuf = buf + 1 == FT_PBUF ? 0 : buf + 1;
        if (c + 1 < nch) {
          a_prefetch(nbuf, 0, ah[0], al[0]);
          a_prefetch(nbuf, 1, ah[1], al[1]);
        }
        // D: lane -> column lane % 16, rows 4 (lane / 16) + i.  2^ex 2^ew (acc0 + 2^-11 (acc1 + acc2)) + bias
        // (on the packed fp32 pipe: the GEMM phase is bound by the SIMD's vector ISSUE -- 36 MFMAs hold it 288 of a step's
        // cycles, the helpers' split 340, this epilogue and the LDS instructions the rest -- so eight instructions instead of
        // sixteen are time; the same operations per component, the same bits)
        const ft_f2 k2 = ft_f2{1.f / 2048.f, 1.f / 2048.f}, ci2 = ft_f2{col_inv, col_inv}, cb2 = ft_f2{col_bias, col_bias};
        const ft_f2 a0l = __builtin_shufflevector(acc0, acc0, 0, 1), a0h = __builtin_shufflevector(acc0, acc0, 2, 3);
        const ft_f2 tl = __builtin_shufflevector(acc1, acc1, 0, 1) + __builtin_shufflevector(acc2, acc2, 0, 1);
        const ft_f2 th = __builtin_shufflevector(acc1, acc1, 2, 3) + __builtin_shufflevector(acc2, acc2, 2, 3);
        const ft_f2 ul = __builtin_elementwise_fma(tl, k2, a0l), uh = __builtin_elementwise_fma(th, k2, a0h);
        const ft_f2 sl = ci2 * __builtin_shufflevector(ri, ri, 0, 1), sh2 = ci2 * __builtin_shufflevector(ri, ri, 2, 3);
        const ft_f2 ol = __builtin_elementwise_fma(ul, sl, cb2), oh = __builtin_elementwise_fma(uh, sh2, cb2);
        f4 o = __builtin_shufflevector(ol, oh, 0, 1, 2, 3);
        if (dst_act) o = w_act<C>(a, o);
        if constexpr (MODE == 1) {
          if (dst_act) wmx = max(wmx, ft_amax_bits(o));
        }
        if (dst_off >= 0) {
          char* po = base + dst_off + (FT_CHUNK * c + 4 * qd) * dst_stride;
          *reinterpret_cast<float*>(po) = o.x;
          *reinterpret_cast<float*>(po + dst_stride) = o.y;
          *reinterpret_cast<float*>(po + 2 * dst_stride) = o.z;
          *reinterpret_cast<float*>(po + 3 * dst_stride) = o.w;
        }
      }
      buf = buf + 1 == FT_PBUF ? 0 : buf + 1;
      if constexpr (MODE == 1) {
        // (backward) the largest |w'| of the tile, for the scale of d bases: in front of the LAST step's barrier
        if (c + 1 == nch && dst_act) {
          const unsigned m = ft_wave_umax(wmx);
          if (lane == 0) __hip_atomic_fetch_max(reinterpret_cast<unsigned*>(lds_rec) + 27, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      lds_barrier();
    }
    } else {
    // ---- (G, WIDE) 32 rows per chunk, one 32-column tile per wavefront on v_mfma_f32_32x32x16_f16 (x as the A operand: lane ->
    //      row lane % 32, k = 16 s + 8 (lane / 32) ..+7; the weights as B: lane -> column 32 wave + lane % 32, same k); per
    //      k-step of 16 both planes of the weight fragment come from L2 (packed[tile][k-step][plane][lane][8]: one KiB per
    //      request), requested FOUR k-steps ahead into a ring of four (with one step ahead and a register copy at the end of
    //      the step every k-step paid the whole L2 latency: 22,000 cycles per chunk against 1,300 of matrix work); three
    //      products, acc0 = xh wh, acc1 = xl wh, acc2 = xh wl.  One barrier per k-slab of 128 the helpers stage; the D tile leaves
    //      behind the chunk's last k-step. ----
    typedef float ft_f16v __attribute__((ext_vector_type(16)));
    int lvm = lane;
    asm volatile("" : "+v"(lvm));
    const int l31 = lvm & 31, hh = lvm >> 5;
    constexpr int NS = WIDE;
    constexpr int BD = 4;                 // weight fragments in flight per wavefront (k-steps); t.k16 is a multiple of it (zero k-steps)
    const ft_h8* bsrc = reinterpret_cast<const ft_h8*>(t.packed) + (int64_t)wave * t.k16 * 128 + lvm;   // [tile][k16][plane][64 lanes]
    const int NG = t.k16 / BD;            // groups of BD k-steps per chunk
    // ring of BD fragments, slot = k-step % BD; a k-step's fragment is requested BD steps ahead (the chunk's last BD steps request
    // the next chunk's first), always, so that the compiler counts the requests exactly; every request is 1 KiB per plane
    ft_h8 wh[BD], wl[BD];
#pragma unroll
    for (int j = 0; j < BD; ++j) { wh[j] = ft_h8{0, 0, 0, 0, 0, 0, 0, 0}; wl[j] = wh[j]; }
    if (is_mfma && nch > 0) {
#pragma unroll
      for (int j = 0; j < BD; ++j) { wh[j] = bsrc[j * 128]; wl[j] = bsrc[j * 128 + 64]; }
    }
    int qb = 0;                         // plane buffer of the running slab step
    for (int c = 0; c < nch; ++c) {
      // three accumulators, one per product: a second product on the same accumulator waits for the first (64 cycles of a
      // 32 x 32 x 16 MFMA's latency against 32 of issue) -- 2,300 cycles of matrix work per chunk where 1,150 do
      ft_f16v acc0, acc1, acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }
      if (is_mfma) {
        const char* pa = lds_planes + qb * FTW_PBUF_BYTES + l31 * (FTW_LDX * 2) + hh * 16;
        for (int g = 0; g < NG; ++g) {
#ifdef EGC_FT_STAMPS
          if (t.dbg & 2) {                 // (diagnostic build: EGC_FT_DBG bit 1 = no matrix work, the barriers stay)
            if ((g & 1) && g + 1 < NG) { qb ^= 1; lds_barrier(); }
            continue;
          }
#endif
          const ft_h8* bnext = g + 1 < NG ? bsrc + (int64_t)(g + 1) * (BD * 128) : bsrc;
          // the A fragments of the group's four k-steps: requested together (their LDS latency passes once per group)
          ft_h8 xh[BD], xl[BD];
          const char* pg = pa + (g & 1) * (BD * 32);
#pragma unroll
          for (int j = 0; j < BD; ++j) {
            xh[j] = *reinterpret_cast<const ft_h8*>(pg + j * 32);
            xl[j] = *reinterpret_cast<const ft_h8*>(pg + j * 32 + FTW_PLANE_BYTES);
          }
#pragma unroll
          for (int j = 0; j < BD; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[j], wh[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[j], wh[j], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[j], wl[j], acc2, 0, 0, 0);
            wh[j] = bnext[j * 128];
            wl[j] = bnext[j * 128 + 64];
          }
          if ((g & 1) && g + 1 < NG) {     // a k-slab of 128 (two groups) is done and another follows: the helpers have staged it
            qb ^= 1;
            lds_barrier();
            pa = lds_planes + qb * FTW_PBUF_BYTES + l31 * (FTW_LDX * 2) + hh * 16;
          }
        }
#ifdef EGC_FT_STAMPS
        if (!(t.dbg & 16))           // (diagnostic build: EGC_FT_DBG bit 4 = no D-tile epilogue)
#endif
        {
          // D: lane -> column 32 wave + lane % 32, rows 8 j + 4 (lane / 32) + i.  2^ex 2^ew (acc0 + 2^-11 acc1) + bias
          const float* rinv = lds_rowinv + (c & 1) * FTW_CH + 4 * hh;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f4 ri = *reinterpret_cast<const f4*>(rinv + 8 * j);
            f4 o;
            o.x = __builtin_fmaf(__builtin_fmaf(acc1[4 * j] + acc2[4 * j], 1.f / 2048.f, acc0[4 * j]), col_inv * ri.x, col_bias);
            o.y = __builtin_fmaf(__builtin_fmaf(acc1[4 * j + 1] + acc2[4 * j + 1], 1.f / 2048.f, acc0[4 * j + 1]), col_inv * ri.y, col_bias);
            o.z = __builtin_fmaf(__builtin_fmaf(acc1[4 * j + 2] + acc2[4 * j + 2], 1.f / 2048.f, acc0[4 * j + 2]), col_inv * ri.z, col_bias);
            o.w = __builtin_fmaf(__builtin_fmaf(acc1[4 * j + 3] + acc2[4 * j + 3], 1.f / 2048.f, acc0[4 * j + 3]), col_inv * ri.w, col_bias);
            if (dst_act) o = w_act<C>(a, o);
            if (dst_off >= 0) {
              char* po = base + dst_off + (FTW_CH * c + 8 * j + 4 * hh) * dst_stride;
              *reinterpret_cast<float*>(po) = o.x;
              *reinterpret_cast<float*>(po + dst_stride) = o.y;
              *reinterpret_cast<float*>(po + 2 * dst_stride) = o.z;
              *reinterpret_cast<float*>(po + 3 * dst_stride) = o.w;
            }
          }
        }
        qb ^= 1;
        lds_barrier();
      } else {
        // a wavefront without a column tile keeps the barrier sequence: one per k-slab
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) lds_barrier();
        qb ^= NS & 1;
      }
    }
    }
    FT_STAMP(4)

    // ---- (E) rows: one lane group per row, G rows per wavefront and turn (turns handed out by an LDS counter: a wavefront
    //      whose rows are short takes the next ones), everything from LDS ----
    if constexpr (MODE == 0) {
#ifdef EGC_FT_STAMPS
    if (!(t.dbg & 4))
#endif
    for (; cur.ok;) {
      int r0 = 0;
      // (workgroup scope, relaxed: the plain atomicAdd drains the vector-memory counter first, i.e. waits for the `out`
      // stores of the wavefront's previous turn)
      if (lane == 0) r0 = __hip_atomic_fetch_add(lds_rowctr, G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      r0 = __builtin_amdgcn_readfirstlane(r0);
      if (r0 >= T) break;
      const int r = r0 + g;
      const bool row_ok = r < T;
      const int row = n0 + (row_ok ? r : 0);
      const int start = row_ok ? (lds_rowptr[r] & 0xffff) : 0;             // (two adjacent words: one ds_read2_b32)
      const int nd = row_ok ? (lds_rowptr[r + 1] & 0xffff) - start : 0;
      int maxd = nd;
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1) maxd = max(maxd, bperm((lane ^ off) << 2, maxd));
      maxd = __builtin_amdgcn_readfirstlane(maxd);
      const float dis_i = (want_dis && row_ok) ? lds_dis[r] : 0.f;
      const bool has_self = row_ok && (C::loops_all(a) || row <= max_index);
      // Rows of 65 .. 128 slots (the reference's ogbg-code nets: 300 / H4 / B4 -> 76, 304 / H8 / B8 -> 80; run_pretrained.sh:47-48)
      // are finished in TWO passes of at most 64 lanes, as in agg_wide_kernel (egc_aggregate_fast.hip): the P = Ls / 4 slots of
      // every basis are cut into a first set of P0 and a second of P - P0, each set a complete sub-layer over its own channels
      // (lane q <-> basis q / Ps, slot q % Ps of the set), the epilogue unchanged but for the channel offset.
      constexpr bool TWO = WIDE != 0 && LPR_LOG2 == 6 && __is_same(C, RtCfg);
      const int nsets = TWO ? t.nsets : 1;
      for (int set = 0; set < nsets; ++set) {
        AggArgs as_store;
        int qslot = q;
        const char* bq = bases_q;
        if constexpr (TWO) {
          as_store = a;
          if (nsets == 2) {
            const int P = a.Ls >> 2, Ps = set == 0 ? t.p0 : P - t.p0;
            as_store.lanes_pb = Ps;
            as_store.slots = a.B * Ps;
            as_store.magic_P = set == 0 ? t.magic0 : t.magic1;
            as_store.lpb_log2 = -1;
            as_store.l4_off = set == 0 ? 0 : t.p0;
            const int bb = min((int)__umulhi((unsigned)q, as_store.magic_P), a.B - 1);
            qslot = bb * P + as_store.l4_off + (q - bb * Ps);
            bq = base + t.off_bases + (q < as_store.slots ? qslot : 0) * 16;
          }
        }
        const AggArgs& as = TWO ? as_store : a;
        const bool want_self = looped_any && has_self && q < C::slots(as);
        f4 vself = f4{0.f, 0.f, 0.f, 0.f};
        if (want_self) vself = lds_bases4[r * ldb4 + qslot];

        FAcc<NEED> acc;
        acc.init();
        if constexpr (NEED & NEED_SQ)     // the variance's shift: the row's first entry in the tile's CSR (FAcc::sh)
          acc.sh = *reinterpret_cast<const f4*>(bq + __umul24((unsigned)((row_ok && nd > 0) ? (int)lds_col[start] : zrow), ldb_bytes));
        int nself = 0;
        for (int ts = 0; ts < maxd; ts += LPR) {
          // lane q of the group stages entry ts + q of the row: its source row (the image's all-zero row when the entry is
          // absent, or a self-entry the layer's x-part excludes: one 24-bit multiply-add then addresses every entry, and an
          // entry takes part in the extrema iff its row is not that one) and its symnorm weight, whole
          const bool pv = ts + q < nd;
          const int jj = pv ? (int)lds_col[start + ts + q] : 0;
          const bool self_e = pv && jj == r;
          float dd = (pv && want_dis) ? lds_dis[jj] * dis_i : 0.f;
          if (C::yl(a) && !C::xl(a)) dd = self_e ? 0.f : dd;     // mixed sets: the self-entry counts for sum / max only
          if (looped_any) {
            const unsigned long long sb = __ballot(self_e);
            nself += __popcll((sb >> (g << LPR_LOG2)) & ((LPR == 64) ? ~0ull : ((1ull << LPR) - 1ull)));
          }
          const int jx = (pv && !(C::xl(a) && self_e)) ? jj : zrow;
          const int cnt = min(LPR, maxd - ts);
          for (int t0 = 0; t0 < cnt; t0 += FU) {
            f4 v[FU];
            float w[FU];
            bool in_x[FU];
  #pragma unroll
            for (int uu = 0; uu < FU; ++uu) {
              const int addr = grp_addr + ((t0 + uu) << 2);
              const int j = bperm(addr, jx);
              in_x[uu] = j != zrow;
              v[uu] = *reinterpret_cast<const f4*>(bq + __umul24((unsigned)j, ldb_bytes));
              w[uu] = bperm(addr, dd);
            }
            if constexpr (!(NEED & NEED_ARG) && FU == 4) {
              // every lane's four entries present (wavefront-uniform; the common case of a regular graph): the sums as in fold,
              // in the same order, and the extrema two entries at a time -- no lane masks
              if (__ballot(!(in_x[0] && in_x[1] && in_x[2] && in_x[3])) == 0) {
  #pragma unroll
                for (int uu = 0; uu < FU; ++uu) {
                  acc.sum += v[uu];
                  acc.ws = f4_fma(splat(w[uu]), v[uu], acc.ws);
                  if constexpr (NEED & NEED_SQ) acc.sq += f4_sqr_rn(v[uu] - acc.sh);
                }
                acc.mx = f4_vmax3(f4_vmax3(acc.mx, v[0], v[1]), v[2], v[3]);
                if constexpr (NEED & NEED_MN) acc.mn = f4_vmin3(f4_vmin3(acc.mn, v[0], v[1]), v[2], v[3]);
                continue;
              }
            }
  #pragma unroll
            for (int uu = 0; uu < FU; ++uu) fold<NEED>(acc, v[uu], w[uu], in_x[uu], start + ts + t0 + uu);
          }
        }
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const f4 wdummy[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        // the row's weightings sit in the LDS image as [h][b][4], nonlinearity applied (W_READY; a.w_lds_stride == 0)
        finish_group<LPR_LOG2, HPB, NEED, C, true, WIDE != 0 ? 0 : 4>(as, R, ln, row, row_ok, acc, nd, nself, dis_i, vself, has_self, wdummy,
                                                                 true, lds_wt + (row_ok ? r : 0) * t.wl_floats, lds_bias, lds_scale);
      }
    }
    } else {
    // ---- (E, backward) rows: one lane group per DESTINATION row, as the forward -- the row's aggregates are formed again from
    //      the LDS image (with the source and the input position of the entry attaining each maximum), then, head by head,
    //      d agg[t] += w'[h][b][t] g[h] and d w'[h][b][t] = <g[h], agg[t]> (reduced over the four lanes of a basis and written
    //      over w' in the image), and the row's gradients travel to its SOURCES' rows of the d bases image as 64-bit fixed
    //      point on integer LDS atomics: sum / mean / symnorm along every entry, max to the one entry that attained it.  (B = 4 bases of 16
    //      channels: 16 slots, four lanes per basis; aggregators sum / mean / max / symnorm; no weight nonlinearity.)
    long long* lds_db = reinterpret_cast<long long*>(base + t.off_db);
    // The sources' gradients are summed by LDS atomics, and float LDS atomics run at ONE LANE PER CLOCK for the whole CU on
    // gfx950 (tools/src/lds_atomic_bench.hip: 768 cycles per wave instruction with twelve wavefronts adding, 62 for ds_add_u64):
    // d bases is kept as 64-bit fixed point.  Its scale comes from a pre-pass over the tile: every contribution is at most
    // A H max|w'| max|g| in magnitude, so with 2^dbs = 2^(34 - e(max|g|) - e(H max|w'|)) a contribution stays below 2^36 and
    // the sum of a tile's at most 2^14 entries below 2^50 -- 34 bits under the bound where fp32 carries 24.
    int dbs_tile = 0;
    {
      // (largest |g|: by the helpers a tile ahead, in the tile's record; largest |w'|: the GEMM's epilogue, above)
      const int eg = (int)((unsigned)__builtin_amdgcn_readfirstlane(lds_rec[(it % 3) * 8 + 5]) >> 23) - 127;
      const int ew = (int)((unsigned)__builtin_amdgcn_readfirstlane(lds_rec[27]) >> 23) - 127;
      int dbs = 34 - (eg + 1) - (ew + 1) - 5;      // (H <= 8: 3 bits; A <= 4: 2 bits)
      dbs = dbs < -1000 ? -1000 : (dbs > 1000 ? 1000 : dbs);
      // An Inf or NaN among the tile's g or w' (exponent field 255 in either maximum) has no fixed-point image: the integer sums
      // would come out as finite garbage where autograd -- and the CSR path -- give Inf / NaN, and a GradScaler or
      // clip_grad_norm_(error_if_nonfinite=True) would not see it.  The tile's d bases rows are then staged as NaN (FT_DBS_POISON).
      const bool nonfinite = eg >= 128 || ew >= 128;
      if (tid == 0) lds_rec[28] = nonfinite ? FT_DBS_POISON : dbs;     // (for the helpers, who stage d bases behind barrier B1;
      dbs_tile = nonfinite ? 0 : dbs;                                   //  every wavefront here forms the same number itself)
    }
    FT_STAMP(1)
    const double db_scale = __builtin_ldexp(1.0, dbs_tile);
    const __amdgpu_buffer_rsrc_t rgo =
        __builtin_amdgcn_make_buffer_rsrc((void*)t.grad_out, 0, (unsigned)a.n_nodes * (unsigned)F_out * 4u, 0x00020000);
    const int bq = q >> 2, l4 = q & 3;
    const int A = C::A(a), H = C::H(a);
    constexpr int SELF_POS = 0x10000;         // the appended self loop: behind every edge of the tile (16-bit positions)
    for (; cur.ok;) {
      int r0 = 0;
      if (lane == 0) r0 = __hip_atomic_fetch_add(lds_rowctr, G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      r0 = __builtin_amdgcn_readfirstlane(r0);
      if (r0 >= T) break;
      const int r = r0 + g;
      const bool row_ok = r < T;
      const int row = n0 + (row_ok ? r : 0);
      // the row's g, eight heads x 16 bytes per lane (lane (b, l4): channels 4 l4 ..+3 of every head): requested first
      f4 gv[8];
#pragma unroll
      for (int h = 0; h < 8; ++h)
#ifdef EGC_FT_STAMPS
        gv[h] = load_slot(rgo, (row_ok && h < H && !(t.dbg & 256)) ? ((unsigned)row * (unsigned)F_out + (unsigned)(h * 16 + 4 * l4)) * 4u : OOB);
#else
        gv[h] = load_slot(rgo, (row_ok && h < H) ? ((unsigned)row * (unsigned)F_out + (unsigned)(h * 16 + 4 * l4)) * 4u : OOB);
#endif
      const int start = row_ok ? (lds_rowptr[r] & 0xffff) : 0;             // (two adjacent words: one ds_read2_b32)
      const int nd = row_ok ? (lds_rowptr[r + 1] & 0xffff) - start : 0;
      int maxd = nd;
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1) maxd = max(maxd, bperm((lane ^ off) << 2, maxd));
      maxd = __builtin_amdgcn_readfirstlane(maxd);
      const float dis_i = (want_dis && row_ok) ? lds_dis[r] : 0.f;
      const bool has_self = row_ok && (C::loops_all(a) || row <= max_index);
      f4 vself = f4{0.f, 0.f, 0.f, 0.f};
      if (looped_any && has_self) vself = lds_bases4[r * ldb4 + q];
      f4 sum = f4{0.f, 0.f, 0.f, 0.f}, ws = sum, mx = f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      i4 ax = i4{ARG_NONE, ARG_NONE, ARG_NONE, ARG_NONE}, aj = i4{-1, -1, -1, -1};
      auto take = [&](f4 v, int pos, int j) {     // (value, position in the row = input order) lexicographic: the FIRST entry attaining the maximum
        const bool cx = v.x > mx.x || (v.x == mx.x && pos < ax.x), cy = v.y > mx.y || (v.y == mx.y && pos < ax.y);
        const bool cz = v.z > mx.z || (v.z == mx.z && pos < ax.z), cw = v.w > mx.w || (v.w == mx.w && pos < ax.w);
        mx = f4{cx ? v.x : mx.x, cy ? v.y : mx.y, cz ? v.z : mx.z, cw ? v.w : mx.w};
        ax = i4{cx ? pos : ax.x, cy ? pos : ax.y, cz ? pos : ax.z, cw ? pos : ax.w};
        aj = i4{cx ? j : aj.x, cy ? j : aj.y, cz ? j : aj.z, cw ? j : aj.w};
      };
      int nself = 0;
      for (int ts = 0; ts < maxd; ts += LPR) {
#ifdef EGC_FT_STAMPS
        if (t.dbg & 512) break;
#endif
        const bool pv = ts + q < nd;
        const int jj = pv ? (int)lds_col[start + ts + q] : 0;
        const bool self_e = pv && jj == r;
        float dd = (pv && want_dis) ? lds_dis[jj] * dis_i : 0.f;
        if (C::yl(a) && !C::xl(a)) dd = self_e ? 0.f : dd;
        if (looped_any) {
          const unsigned long long sb = __ballot(self_e);
          nself += __popcll((sb >> (g << LPR_LOG2)) & ((LPR == 64) ? ~0ull : ((1ull << LPR) - 1ull)));
        }
        const int jx = (pv && !(C::xl(a) && self_e)) ? jj : zrow;
        const int cnt_e = min(LPR, maxd - ts);
        for (int t0 = 0; t0 < cnt_e; t0 += FU) {
          f4 v[FU];
          float w[FU];
          int jn[FU];
#pragma unroll
          for (int uu = 0; uu < FU; ++uu) {
            const int addr = grp_addr + ((t0 + uu) << 2);
            jn[uu] = bperm(addr, jx);
            v[uu] = *reinterpret_cast<const f4*>(bases_q + __umul24((unsigned)jn[uu], ldb_bytes));
            w[uu] = bperm(addr, dd);
          }
#pragma unroll
          for (int uu = 0; uu < FU; ++uu) {
            sum += v[uu];
            ws = f4_fma(splat(w[uu]), v[uu], ws);
#ifdef EGC_FT_STAMPS
            if (!(t.dbg & 32))
#endif
            if (jn[uu] != zrow) take(v[uu], ts + t0 + uu, jn[uu]);     // (the row's entries are in input order: csr_s3)
          }
        }
      }
      // the self-loop term, as finish_group
      int cnt = nd;
      if (C::xl(a)) {
        cnt = nd - nself + (has_self ? 1 : 0);
        sum += vself;                                          // (0 where the row has no self loop)
        ws = f4_fma(splat(dis_i * dis_i), vself, ws);
        if (has_self) take(vself, SELF_POS, r);
      } else if (C::yl(a)) {
        ws = f4_fma(splat(dis_i * dis_i), vself, ws);
      }
      const float rcnt = __builtin_amdgcn_rcpf((float)max(cnt, 1));
      const bool nonempty = cnt > 0;
      const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
      f4 val[4], dagg[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        val[tt] = dagg[tt] = zero4;
        if (tt < A) {
          switch (C::aggr(a, tt)) {
            case EGC_AGGR_SUM: val[tt] = sum; break;
            case EGC_AGGR_MEAN: val[tt] = sum * splat(rcnt); break;
            case EGC_AGGR_MAX: val[tt] = nonempty ? mx : zero4; break;
            default: val[tt] = ws; break;      // EGC_AGGR_SYMNORM
          }
        }
      }
      // head by head: d agg, d w'
      float* wrow = lds_wt + (row_ok ? r : 0) * t.wl_floats;
#pragma unroll
      for (int h = 0; h < 8; ++h) {
#ifdef EGC_FT_STAMPS
        if (t.dbg & 64) break;
#endif
        if (h < H) {
          const f4 wv = *reinterpret_cast<const f4*>(wrow + (h * 4 + bq) * 4);
          dagg[0] = f4_fma(splat(wv.x), gv[h], dagg[0]);
          if (A > 1) dagg[1] = f4_fma(splat(wv.y), gv[h], dagg[1]);
          if (A > 2) dagg[2] = f4_fma(splat(wv.z), gv[h], dagg[2]);
          if (A > 3) dagg[3] = f4_fma(splat(wv.w), gv[h], dagg[3]);
          float dw[4];
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) {
            float d = tt < A ? fmaf(gv[h].w, val[tt].w, fmaf(gv[h].z, val[tt].z, fmaf(gv[h].y, val[tt].y, gv[h].x * val[tt].x))) : 0.f;
            d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
            d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
            dw[tt] = d;
          }
          // (the four lanes of the basis have read w'[h][b][.]: in-order LDS, the block can be overwritten)
          if (row_ok && l4 == (h & 3)) *reinterpret_cast<f4*>(wrow + (h * 4 + bq) * 4) = f4{dw[0], dw[1], dw[2], dw[3]};
        }
      }
      // the source side: d bases[j] += (sum / mean part) + weight x (symnorm part) along every entry; the maximum's to its entry
      f4 d_t = zero4, d_s = zero4, d_x = zero4;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        if (tt < A) {
          switch (C::aggr(a, tt)) {
            case EGC_AGGR_SUM: d_t += dagg[tt]; break;
            case EGC_AGGR_MEAN: d_t += dagg[tt] * splat(rcnt); break;
            case EGC_AGGR_MAX: d_x += nonempty ? dagg[tt] : zero4; break;
            default: d_s += dagg[tt]; break;
          }
        }
      }
      // float -> 64-bit fixed point: x 2^dbs in double, + 1.5 2^52 (the integer then sits in the low bits of the mantissa), - its bits
      auto fix = [&](float v) -> unsigned long long {
        const double tq = (double)v * db_scale + 6755399441055744.0;
        return (unsigned long long)(__builtin_bit_cast(long long, tq) - 0x4338000000000000ll);
      };
      auto add_at = [&](long long* p, float v) { atomicAdd(reinterpret_cast<unsigned long long*>(p), fix(v)); };
      auto add_row = [&](int j, f4 c) {
        long long* p = lds_db + j * a.ldb + 4 * q;
        add_at(p, c.x); add_at(p + 1, c.y); add_at(p + 2, c.z); add_at(p + 3, c.w);
      };
      for (int ts = 0; ts < maxd; ts += LPR) {
#ifdef EGC_FT_STAMPS
        if (t.dbg & 128) break;
#endif
        const bool pv = ts + q < nd;
        const int jj = pv ? (int)lds_col[start + ts + q] : 0;
        const bool self_e = pv && jj == r;
        float dd = (pv && want_dis) ? lds_dis[jj] * dis_i : 0.f;
        if (C::yl(a) && !C::xl(a)) dd = self_e ? 0.f : dd;
        const int jx = (pv && !(C::xl(a) && self_e)) ? jj : zrow;
        const int cnt_e = min(LPR, maxd - ts);
        for (int e0 = 0; e0 < cnt_e; ++e0) {
          const int addr = grp_addr + (e0 << 2);
          const int j = bperm(addr, jx);
          const float w = bperm(addr, dd);
          if (j != zrow) add_row(j, f4_fma(splat(w), d_s, d_t));
        }
      }
#ifdef EGC_FT_STAMPS
      if (!(t.dbg & 1024))
#endif
      if (row_ok) {
        if (C::xl(a)) { if (has_self) add_row(r, f4_fma(splat(dis_i * dis_i), d_s, d_t)); }
        else if (C::yl(a) && has_self) add_row(r, d_s * splat(dis_i * dis_i));
        if (nonempty) {
          if (aj.x >= 0) add_at(lds_db + aj.x * a.ldb + 4 * q, d_x.x);
          if (aj.y >= 0) add_at(lds_db + aj.y * a.ldb + 4 * q + 1, d_x.y);
          if (aj.z >= 0) add_at(lds_db + aj.z * a.ldb + 4 * q + 2, d_x.z);
          if (aj.w >= 0) add_at(lds_db + aj.w * a.ldb + 4 * q + 3, d_x.w);
        }
      }
    }
    }
    FT_STAMP(5)
    if constexpr (MODE == 0) {
      request_weights();
      lds_barrier();   // every wavefront is done with the tile's LDS image
    } else {
      // ---- (G2, backward) d x [T][F_in] = [d bases | d w'] [T][K2] x [bases_weight | comb_weight^T]^T [K2][F_in]: the helpers stage
      //      the rows of the two images as fp16 planes (above), wavefront w < F_in / 16 keeps the 16 output features 16 w ..+15
      //      of the transposed operand (K2 / 32 k-steps x two planes: 48 registers at K2 = 192); three products, the D tile
      //      scaled and stored straight to d_x (lane -> feature lane % 16, rows 4 (lane / 16) + i).
      const int n_ft = (t.F_in + 15) >> 4;
      const int K2S = (a.ldb + t.wl_floats) >> 5;          // k-steps of 32: 4 (H = 4) or 6 (H = 8)
      const bool is_mfma2 = wave < n_ft;
      f4 u2[12];
      float col_inv2 = 0.f;
      {
        int lv = lane;
        asm volatile("" : "+v"(lv));
        const f4* wsrc = reinterpret_cast<const f4*>(t.packed_t) + (int64_t)(is_mfma2 ? wave : 0) * 12 * 64;
#pragma unroll
        for (int k2 = 0; k2 < 12; ++k2) u2[k2] = wsrc[k2 * 64 + lv];
        col_inv2 = reinterpret_cast<const float*>(t.packed_t + (int64_t)8 * 6 * 2 * 64 * 8)[16 * (is_mfma2 ? wave : 0) + (lv & 15)];
      }
      lds_barrier();   // (B1: every row's d bases / d w' are in the images)
      if (tid == 0) lds_rec[27] = 0;
      lds_barrier();   // (A: the helpers have staged chunk 0)
      const __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(t.d_x + (int64_t)n0 * t.F_in), 0, (unsigned)(cur.ok ? T : 0) * (unsigned)t.F_in * 4u, 0x00020000);
      // (what reaches x past the layer: read at the start of a step, added in the store; without it a descriptor of no bytes)
      const __amdgpu_buffer_rsrc_t rda = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(t.d_x_add != nullptr ? t.d_x_add + (int64_t)n0 * t.F_in : t.x), 0,
          (unsigned)((cur.ok && t.d_x_add != nullptr) ? T : 0) * (unsigned)t.F_in * 4u, 0x00020000);
      int lvm = lane;
      asm volatile("" : "+v"(lvm));
      const int m2 = lvm & 15, qd2 = lvm >> 4;
      int buf2 = 0;
      for (int c = 0; c < nch; ++c) {
#ifdef EGC_FT_STAMPS
        if (!(t.dbg & 4096))
#endif
        if (is_mfma2) {
          const int f = 16 * wave + m2;
          const unsigned off0 = f < t.F_in ? ((unsigned)(FT_CHUNK * c + 4 * qd2) * (unsigned)t.F_in + (unsigned)f) * 4u : OOB;
          const unsigned rs4 = (unsigned)t.F_in * 4u;
          const float ad0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rda, off0, 0, 0));
          const float ad1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rda, off0 == OOB ? OOB : off0 + rs4, 0, 0));
          const float ad2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rda, off0 == OOB ? OOB : off0 + 2 * rs4, 0, 0));
          const float ad3 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rda, off0 == OOB ? OOB : off0 + 3 * rs4, 0, 0));
          const char* pa = lds_planes + buf2 * FTB_PBUF_BYTES + m2 * FTB_ROW_BYTES + qd2 * 16;
          f4 acc0 = f4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
#pragma unroll
          for (int k2 = 0; k2 < 6; ++k2) {
            if (k2 < K2S) {
              const ft_h8 xh = *reinterpret_cast<const ft_h8*>(pa + k2 * 64);
              const ft_h8 xl = *reinterpret_cast<const ft_h8*>(pa + k2 * 64 + FTB_PLANE_BYTES);
              const ft_h8 wh = __builtin_bit_cast(ft_h8, u2[2 * k2]), wl = __builtin_bit_cast(ft_h8, u2[2 * k2 + 1]);
              acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh, acc1, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl, acc2, 0, 0, 0);
            }
          }
          const f4 ri = *reinterpret_cast<const f4*>(reinterpret_cast<const float*>(base + t.off_rowinv2) + buf2 * FT_CHUNK + 4 * qd2);
          const f4 tt4 = acc1 + acc2;
          const f4 o = f4{__builtin_fmaf(tt4.x, 1.f / 2048.f, acc0.x) * (col_inv2 * ri.x) + ad0, __builtin_fmaf(tt4.y, 1.f / 2048.f, acc0.y) * (col_inv2 * ri.y) + ad1,
                          __builtin_fmaf(tt4.z, 1.f / 2048.f, acc0.z) * (col_inv2 * ri.z) + ad2, __builtin_fmaf(tt4.w, 1.f / 2048.f, acc0.w) * (col_inv2 * ri.w) + ad3};
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o.x), rdx, off0, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o.y), rdx, off0 == OOB ? OOB : off0 + rs4, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o.z), rdx, off0 == OOB ? OOB : off0 + 2 * rs4, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o.w), rdx, off0 == OOB ? OOB : off0 + 3 * rs4, 0, 0);
        }
        buf2 ^= 1;
        lds_barrier();
      }
      request_weights();
    }
    FT_STAMP(6)
  }
#ifdef EGC_FT_STAMPS
  if (tid == 0 && egc_ft_stamp_buf != nullptr) {
    unsigned long long tend;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tend) :: "memory");
    ft_acc[7] = tend - ft_start;
    for (int k = 0; k < 8; ++k) egc_ft_stamp_buf[blockIdx.x * 8 + k] = ft_acc[k];
    egc_ft_stamp_buf[256 * 8 + blockIdx.x] = ft_pro;
  }
#endif
}

// egc_fused_tile_wide.hip: the WIDE instances (lpr = lanes per row group: 16 / 32 / 64; need = NEED_* mask of the layer)
int launch_fused_tile_wide1(const AggArgs& a, const FusedTileArgs& t, int lpr, int need, unsigned grid, size_t lds, hipStream_t stream);
int launch_fused_tile_wide2(const AggArgs& a, const FusedTileArgs& t, int lpr, int need, unsigned grid, size_t lds, hipStream_t stream);
int launch_fused_tile_wide3(const AggArgs& a, const FusedTileArgs& t, int lpr, int need, unsigned grid, size_t lds, hipStream_t stream);

}  // namespace egc
