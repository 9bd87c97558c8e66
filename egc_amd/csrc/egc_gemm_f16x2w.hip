// Basis transform + weightings Linear for 128 < F_in <= 352 on the fp16 matrix cores with fp32-level accuracy
// (gfx950), round 3: the register-stationary fp16x2 scheme with THIRTY-TWO columns per wavefront.
//
//     [bases | weightings] = x[N,F_in] @ [bases_weight | comb.weight^T]  (+ comb.bias)
//
// Reference behaviour replaced: torch.matmul(x, bases_weight) (experiments/layers.py:97-101,
// optimized_layers.py:180) and comb_weights(x) (layers.py:110, optimized_layers.py:182) at the reference's wide layers
// (ogbn-mag 352, mag/models.py:23-53; the 136 ... 304 wide trained nets, hyperparameters.md).
//
// Numerics: exactly those of egc_gemm_f16x2.hip / egc_gemm_f16x2k.hip -- every x row and weight column scaled by a
// power of two into [1, 2), two fp16 planes each (22 bits), three products accumulated in fp32.
//
// Why another kernel.  egc_gemm_f16x2k.hip gives a wavefront 16 columns, so 13 wavefronts at the ogbn-mag shape each
// read the whole 16 x 352 x-tile (two planes) from LDS: 293 KB of LDS reads per 16 rows, ~2.3 k cycles of the LDS
// port per tile, on top of one barrier and one round of DMA bookkeeping per 16 rows -- it ran at 2.9-3.7 TB/s.  Here:
//   * a wavefront owns 32 columns for the whole k range (v_mfma_f32_32x32x16_f16, both weight planes in 8 registers per
//     k-step of 16: 176 at F_in = 352) and a workgroup is at most 8 wavefronts -- two per SIMD, 256 registers each;
//   * tiles are 32 rows: half the barriers per row, and per row each x fragment is read by 7 wavefronts instead of 13;
//   * the raw fp32 tile is split IN PLACE: a 16-byte piece (4 floats) becomes its 4 high halves + 4 low halves, so
//     the ring of three DMA slots is all the LDS the kernel needs (137 KB at F_in = 352; separate plane buffers would
//     not fit beside a ring deep enough to cover the HBM latency);
//   * rows 16-31 store their halves swapped and a row is padded to an odd number of pieces: the fragment reads
//     (two 8-byte halves of two adjacent pieces per lane) are bank-conflict free;
//   * x as the A operand: a lane holds ONE output column for 16 rows, stores are dwords that fill 128-byte lines
//     (as in egc_gemm_f16x2.hip), the column's inverse scale and bias stay in two registers.
// One barrier per tile: DMA of tile t + 2, in-place split of tile t + 1 and the MFMAs of tile t sit between the same two
// barriers; the two wavefronts of a SIMD take split and MFMAs in opposite order.
//
// OUTCOME (MI355X, ogbn-mag shape N = 736,389, 352 -> 176 + 32): 552-564 us against 543 us for egc_gemm_f16x2k.hip --
// not faster, so this kernel is OPT-IN (EGC_GEMM_F16X2W=1) and the 16-column kernel stays the default.  Per tile and
// wavefront (cycles): DMA issue 1.5 k (an LDS-DMA wave-instruction costs its issuer 100-200 cycles), split 3.0-3.9 k
// (3.0 k even with the DMA and the MFMAs removed: 27 LDS wave-instructions of 1 KiB per wavefront, seven wavefronts on
// one LDS), MFMAs + stores 3.7 k (2.1 k of pipe time), barrier 1 k -- with 176 registers of weights a SIMD holds two
// wavefronts, and each runs these phases one after the other: 10 k cycles per 32 rows where the 16-column kernel takes
// 2 x 4.4 k.  Ablations: MFMAs only 405 us, split only 233 us, DMA only 280 us (one tile in flight in that variant).
#include <stdlib.h>

#include <algorithm>
#include <cstdio>

#include "egc_common.h"
#include "egc_gemm_split.h"

namespace egc {

typedef float f32x16w __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8w __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2w __attribute__((ext_vector_type(2)));
typedef float f32x2w __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2w __attribute__((ext_vector_type(2)));

constexpr int WROWS = 32;        // rows of an x tile = one MFMA row block
constexpr int W_MAX_WAVES = 8;   // two wavefronts per SIMD: 256 registers each
constexpr int W_STORES = 16;     // store instructions per tile and wavefront (vmcnt arithmetic)
constexpr int W_KMAX = 352;

__device__ inline void lds_barrier_w() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Virtual column space: column tiles of 32; tiles [0, TB) hold the bases columns (ldb of them; what is left of the last
// tile is never stored), tiles [TB, NT) the weightings columns.
struct WCols {
  int F_g, ldb, W, TB, NT;
  __host__ __device__ inline int source(int v) const {  // column of wcat behind virtual column v, or -1
    const int t = v >> 5;
    if (t < TB) return v < F_g ? v : -1;
    const int w = v - 32 * TB;
    return w < W ? F_g + w : -1;
  }
};

static WCols wcols(int f_g, int ldb, int w_cols) {
  WCols c;
  c.F_g = f_g; c.ldb = ldb; c.W = w_cols;
  c.TB = (ldb + 31) / 32;
  c.NT = c.TB + (w_cols + 31) / 32;
  return c;
}

// packed: [NT][KS16][2 planes][64 lanes][8] fp16 -- the B fragments of v_mfma_f32_32x32x16_f16 in register order
// (lane 32 hh + i holds k = 16 s + 8 hh .. + 7 of column 32 t + i) -- followed by float inv_scale[32 NT].
// One wavefront per virtual column: lanes stride over k, the column maximum is a wavefront all-reduce.
__global__ void __launch_bounds__(64) pack_f16x2w_kernel(const float* __restrict__ wcat, int64_t rs, int64_t cs, int K,
                                                         WCols c, int KS16, u16* __restrict__ packed) {
  const int v = blockIdx.x;
  const int lane = threadIdx.x;
  const int src = c.source(v);
  unsigned amax = 0;
  if (src >= 0)
    for (int k = lane; k < K; k += 64) amax = max(amax, __float_as_uint(wcat[k * rs + src * cs]) & 0x7fffffffu);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, d));
  unsigned be = amax >> 23;
  be = be > 253u ? 253u : be;
  const float scale = __uint_as_float((254u - be) << 23), inv = __uint_as_float(be << 23);
  const int t = v >> 5, i = v & 31;
  for (int k = lane; k < KS16 * 16; k += 64) {
    const float w = (src >= 0 && k < K) ? wcat[k * rs + src * cs] * scale : 0.f;
    const _Float16 h = (_Float16)w;
    const _Float16 l = (_Float16)((w - (float)h) * 2048.f);
    const int s = k >> 4, hh = (k >> 3) & 1, e = k & 7;
    const int64_t base = ((((int64_t)t * KS16 + s) * 2) * 64 + (hh * 32 + i)) * 8 + e;
    packed[base] = __builtin_bit_cast(u16, h);
    packed[base + 64 * 8] = __builtin_bit_cast(u16, l);
  }
  if (lane == 0) reinterpret_cast<float*>(packed + (int64_t)c.NT * KS16 * 2 * 64 * 8)[v] = inv;
}

template <int N>
__device__ inline void vmwait_w() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#ifdef EGC_GEMMW_STAMPS
__device__ unsigned long long egc_stampw[8];  // diagnostic build only: cycles per phase, summed over wavefronts
#define WST(k) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); wsum[k] += _t - wt0; wt0 = _t; }
#else
#define WST(k)
#endif

// RP: 16-byte pieces per LDS row (K / 4 real ones, the rest read as zero: k beyond F_in up to 16 KS16, and the padding
// that makes the row stride an odd number of pieces); R: DMA wave-instructions per wavefront and tile.
template <int KS16>
__global__ void __launch_bounds__(W_MAX_WAVES * 64) basis_gemm_f16x2w_kernel(const float* __restrict__ x, const u16* __restrict__ packed,
                                                                            const float* __restrict__ bcat, int64_t M, int K, WCols c,
                                                                            float* __restrict__ bases, float* __restrict__ weightings,
                                                                            int n_tiles, int RP, int R, int slot_bytes, int tile0) {
  extern __shared__ __attribute__((aligned(16))) char smem_w[];
  char* ring = smem_w;                                                    // [3][slot_bytes]: raw fp32 tile, then its two fp16 planes in place
  float* row_inv = reinterpret_cast<float*>(smem_w + 3 * slot_bytes);     // [3][WROWS]
  char* sink = smem_w + 3 * slot_bytes + 3 * WROWS * sizeof(float);       // [8 wavefronts][64 lanes] x 16 B: stores of rows beyond the tile
  const int tid = threadIdx.x;
  const int nthreads = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = tile0 + wave;                                            // this wavefront's column tile
  const int l31 = lane & 31, hh = lane >> 5;
  const int K4 = K >> 2;
  const int RS = RP * 16;                                                 // LDS row stride, bytes
  constexpr unsigned GOOB = 0xFFFFFFF0u;
  constexpr unsigned SOOB = 0x80000000u;
  const u32x4w rx = {(unsigned)(uintptr_t)x, (unsigned)((uintptr_t)x >> 32) & 0xffffu, (unsigned)(M * K * 4), 0x00020000u};
  const unsigned ring_lds = (unsigned)(uintptr_t)ring;
  const unsigned magic_RP = (unsigned)(((uint64_t)1 << 32) / (uint64_t)RP) + 1u;

  // one LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global offsets to 1 KiB of contiguous LDS.  Piece
  // tid + nthreads i of the LDS image is (row, k4) = that index divided by RP: stepped, not divided (the division and the
  // 64-bit offset arithmetic per piece were ~1.5 k cycles per tile and wavefront)
  const int row_first = (int)__umulhi((unsigned)tid, magic_RP);   // tid / RP
  const int k4_first = tid - row_first * RP;
  const int step_row = nthreads / RP, step_k4 = nthreads - step_row * RP;   // wave-uniform
  const unsigned Kb = (unsigned)K * 4u;
  auto dma_tile = [&](int tile, int slot) {
    // rows of this tile that exist (0 beyond the last tile): the host keeps row offsets below 2^31
    const int rows_here = tile < n_tiles ? (int)min((int64_t)WROWS, M - (int64_t)tile * WROWS) : 0;
    const unsigned tile_off = (unsigned)tile * (unsigned)WROWS * Kb;
    int row = row_first, k4 = k4_first;
    unsigned dst = ring_lds + slot * slot_bytes + wave * 1024;
    for (int i = 0; i < R; ++i) {
      const bool ok = (row < rows_here) & (k4 < K4);
      const unsigned voff = ok ? tile_off + (unsigned)row * Kb + (unsigned)k4 * 16u : GOOB;   // out of range: zeros land in LDS
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(voff), "s"(dst), "s"(rx)
                   : "memory");
      k4 += step_k4; row += step_row;
      const bool wrap = k4 >= RP;
      k4 = wrap ? k4 - RP : k4;
      row = wrap ? row + 1 : row;
      dst += nthreads * 16;
    }
  };

  const int stride = gridDim.x;
  int tile = blockIdx.x;
  if (tile >= n_tiles) return;
  dma_tile(tile, 0);
  dma_tile(tile + stride, 1);

  // both planes of this wavefront's F_in x 32 weight block, as B operands (one contiguous KiB per load)
  f16x8w wf[KS16][2];
  {
    const u16* src = packed + ((int64_t)ct * KS16 * 2 * 64 + lane) * 8;
#pragma unroll
    for (int s = 0; s < KS16; ++s) {
      wf[s][0] = *reinterpret_cast<const f16x8w*>(src + (s * 2) * 64 * 8);
      wf[s][1] = *reinterpret_cast<const f16x8w*>(src + (s * 2 + 1) * 64 * 8);
    }
  }
  // output addressing: every element of a lane belongs to ONE virtual column 32 ct + l31
  const bool to_bases = ct < c.TB;
  const int out_ld = to_bases ? c.ldb : c.W;
  const int out_col = (to_bases ? 32 * ct : 32 * (ct - c.TB)) + l31;
  const bool col_ok = out_col < out_ld;
  const __amdgpu_buffer_rsrc_t ro =
      to_bases ? __builtin_amdgcn_make_buffer_rsrc((void*)bases, 0, (unsigned)(M * c.ldb * 4), 0x00020000)
               : __builtin_amdgcn_make_buffer_rsrc((void*)weightings, 0, (unsigned)(M * (int64_t)c.W * 4), 0x00020000);
  float col_inv = reinterpret_cast<const float*>(packed + (int64_t)c.NT * KS16 * 2 * 64 * 8)[32 * ct + l31];
  float col_bias = (!to_bases && col_ok && bcat != nullptr) ? bcat[out_col] : 0.f;
  vmwait_w<0>();
  // the compiler counts only its own loads: let it retire the weight loads HERE
#pragma unroll
  for (int s = 0; s < KS16; ++s) asm volatile("" : "+v"(wf[s][0]), "+v"(wf[s][1]));
  asm volatile("" : "+v"(col_inv), "+v"(col_bias));
  // the first two tiles were requested before the weight loads: every wavefront's pieces have landed (vmcnt(0) above)
  lds_barrier_w();

  const bool late_half = wave >= 4;           // shares its SIMD with wavefront wave - 4
  const int hw = tid >> 5;                    // half-wavefront index: one row of the tile per half-wavefront and pass
  const int hl = tid & 31;
  const int n_hw = nthreads >> 5;
  // raw fp32 rows of ring slot `slot` -> two fp16 planes IN PLACE: piece (4 floats) -> [h0 h1 h2 h3 | l0 l1 l2 l3], the
  // halves swapped in rows 16-31; + the row scales
  // Rows in batches of SB per half-wavefront: the loads of a batch are issued together, so the two LDS round trips of
  // the split (maximum, then conversion) are paid once per batch, not once per row.
  constexpr int SB = 3;
  auto split = [&](int slot) {
    char* rsl = ring + slot * slot_bytes;
    for (int row0 = hw; row0 < WROWS; row0 += SB * n_hw) {
      // Straight-line code: every load is issued whatever the lane's piece -- pieces beyond F_in read (and rewrite) the
      // row's first padding piece, which is and stays zero; rows beyond the tile read row0 and store nothing.  (With the
      // loads under `if (row < ... && k4 < ...)` each of the 18 became a branch with its own wait: 3.6 k cycles per tile.)
      char* pp[SB][3];
      bool rv[SB];
      float m[SB];
#pragma unroll
      for (int r = 0; r < SB; ++r) {
        const int row = row0 + r * n_hw;
        rv[r] = row < WROWS;
        char* rp = rsl + (rv[r] ? row : row0) * RS;
#pragma unroll
        for (int i = 0; i < 3; ++i) pp[r][i] = rp + min(hl + 32 * i, K4) * 16;
      }
#pragma unroll
      for (int r = 0; r < SB; ++r) {
        m[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(pp[r][i]);
          float mi;
          asm("v_max3_f32 %0, |%1|, |%2|, |%3|\n\tv_max_f32 %0, |%4|, %0" : "=&v"(mi) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
          m[r] = fmaxf(m[r], mi);
        }
      }
      float sc[SB], sc2k[SB];
#pragma unroll
      for (int r = 0; r < SB; ++r) {
        unsigned a = __float_as_uint(m[r]);
        a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
        a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
        a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x141, 0xf, 0xf, true));  // row_half_mirror
        a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x140, 0xf, 0xf, true));  // row_mirror
        a = max(a, (unsigned)__builtin_amdgcn_ds_swizzle((int)a, 0x401F));                    // lane ^ 16
        unsigned e = a & 0x7f800000u;
        e = min(max(e, 13u << 23), 253u << 23);
        sc[r] = __uint_as_float(0x7f000000u - e);                  // 2^-e
        sc2k[r] = __uint_as_float(0x7f000000u + (11u << 23) - e);  // 2^(11-e)
        if (hl == 0 && rv[r]) row_inv[slot * WROWS + row0 + r * n_hw] = __uint_as_float(e);  // 2^e
      }
#pragma unroll
      for (int r = 0; r < SB; ++r) {
        const bool swapped = ((row0 + r * n_hw) & 16) != 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(pp[r][i]);   // second read: the registers hold the weights
          const f16x2w h01 = __builtin_convertvector(f32x2w{v.x * sc[r], v.y * sc[r]}, f16x2w);
          const f16x2w h23 = __builtin_convertvector(f32x2w{v.z * sc[r], v.w * sc[r]}, f16x2w);
          f16x2w l01, l23;
          l01[0] = (_Float16)__builtin_fmaf((float)h01[0], -2048.f, v.x * sc2k[r]);
          l01[1] = (_Float16)__builtin_fmaf((float)h01[1], -2048.f, v.y * sc2k[r]);
          l23[0] = (_Float16)__builtin_fmaf((float)h23[0], -2048.f, v.z * sc2k[r]);
          l23[1] = (_Float16)__builtin_fmaf((float)h23[1], -2048.f, v.w * sc2k[r]);
          const unsigned uh0 = __builtin_bit_cast(unsigned, h01), uh1 = __builtin_bit_cast(unsigned, h23);
          const unsigned ul0 = __builtin_bit_cast(unsigned, l01), ul1 = __builtin_bit_cast(unsigned, l23);
          // rows beyond the tile write into a sink behind the ring: no branch anywhere in the split
          char* dst = rv[r] ? pp[r][i] : sink + (wave * 64 + lane) * 16;
          *reinterpret_cast<u32x4w*>(dst) = u32x4w{swapped ? ul0 : uh0, swapped ? ul1 : uh1, swapped ? uh0 : ul0, swapped ? uh1 : ul1};
        }
        if (KS16 >= 20 && r == 1) __builtin_amdgcn_sched_barrier(0);   // 176 weight registers: at most six pieces in flight here
      }
    }
  };
  split(0);
#ifdef EGC_GEMMW_STAMPS
  unsigned long long wt0, wsum[6] = {0, 0, 0, 0, 0, 0};
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wt0) :: "memory");
#endif
  // One barrier per tile.  At the top of iteration t: slot (t + 1) % 3 holds the raw tile t + 1 (this wavefront's own
  // pieces are waited for by its counter, the barrier covers everybody else's), slot t % 3 the split tile t.  After the
  // barrier slot (t + 2) % 3 = (t - 1) % 3 is free (everybody has multiplied tile t - 1) and is re-armed with tile
  // t + 2 at once; then the wavefronts split tile t + 1 and multiply tile t, in whatever order they get there.
  int st = 0;          // slot of tile t
  for (bool first = true; tile < n_tiles; tile += stride, first = false) {
    const int sn = st == 2 ? 0 : st + 1, snn = sn == 2 ? 0 : sn + 1;
    if (first) vmwait_w<0>(); else vmwait_w<W_STORES>();   // own pieces of tile t + 1: only tile t - 1's stores came after them
    WST(0)
    lds_barrier_w();
    WST(1)
#ifndef EGC_W_NODMA
    dma_tile(tile + 2 * stride, snn);
#endif
    WST(2)
    auto multiply = [&]() {
      f32x16w acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
      // A operands (x): lane -> row l31, k = 16 s + 8 hh .. + 7 = pieces 4 s + 2 hh, + 1; 8 bytes of each per plane
      const char* xb = ring + st * slot_bytes + l31 * RS + hh * 32;
      const int offh = (l31 & 16) ? 8 : 0, offl = 8 - offh;
      // three k-steps of fragments in flight (the compiler's own schedule keeps one: every MFMA group then waits a
      // whole LDS round trip -- 3.8 k cycles per tile for 2.1 k of matrix work)
      auto frag = [&](int s, f16x8w& xh, f16x8w& xl) {
        const char* p = xb + 64 * s;
        const u32x2w h0 = *reinterpret_cast<const u32x2w*>(p + offh), h1 = *reinterpret_cast<const u32x2w*>(p + 16 + offh);
        const u32x2w l0 = *reinterpret_cast<const u32x2w*>(p + offl), l1 = *reinterpret_cast<const u32x2w*>(p + 16 + offl);
        xh = __builtin_bit_cast(f16x8w, u32x4w{h0[0], h0[1], h1[0], h1[1]});
        xl = __builtin_bit_cast(f16x8w, u32x4w{l0[0], l0[1], l1[0], l1[1]});
      };
      constexpr int DEPTH = KS16 >= 22 ? 2 : 3;   // (176 weight registers at F_in = 352 leave room for two sets)
      f16x8w fh[DEPTH], fl[DEPTH];
#pragma unroll
      for (int s = 0; s < DEPTH; ++s) frag(s, fh[s], fl[s]);
#pragma unroll
      for (int s = 0; s < KS16; ++s) {
        const f16x8w xh = fh[s % DEPTH], xl = fl[s % DEPTH];
        if (s + DEPTH < KS16) frag(s + DEPTH, fh[s % DEPTH], fl[s % DEPTH]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wf[s][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wf[s][0], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wf[s][1], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // D layout (x as the A operand): lane -> column l31, rows 8 j + 4 hh + i.  Every store instruction is always issued
      // (masked lanes and rows past M fall outside the buffer's range): the counted wait at the top of the loop relies
      // on a fixed number of vector-memory operations per tile.
      const unsigned voff = col_ok ? (unsigned)((((int64_t)tile * WROWS + 4 * hh) * out_ld + out_col) * 4) : SOOB;
      const float* rinv = row_inv + st * WROWS;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 ri = *reinterpret_cast<const float4*>(rinv + 8 * j + 4 * hh);
        const float rr[4] = {ri.x, ri.y, ri.z, ri.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float tsum = __builtin_fmaf(acc1[4 * j + i], 1.f / 2048.f, acc0[4 * j + i]);
          const float v = __builtin_fmaf(tsum, col_inv * rr[i], col_bias);   // the scale product is a power of two: one rounding
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro, voff, (8 * j + i) * out_ld * 4, 0);
        }
      }
    };
    // The wavefronts of a SIMD (w and w + 4) take the two halves of the iteration in opposite order: while one splits
    // the next tile on the vector unit the other multiplies on the matrix pipe (all in the same order, every SIMD ran
    // vector phase and matrix phase back to back: 11.2 k cycles per tile instead of ~5 k).
#if defined(EGC_W_NOTHING)
    WST(4)
#elif defined(EGC_W_NOSPLIT)
    multiply(); WST(4)
#elif defined(EGC_W_NOMFMA)
    split(sn); WST(3)
#elif defined(EGC_W_SAMEORDER)
    split(sn); WST(3) multiply(); WST(4)
#else
    if (late_half) { multiply(); WST(4) split(sn); WST(3) } else { split(sn); WST(3) multiply(); WST(4) }
#endif
    WST(5)
    st = sn;
  }
  vmwait_w<0>();  // no DMA may still be writing this block's LDS when it is handed to the next block
#ifdef EGC_GEMMW_STAMPS
  if (lane == 0)
    for (int k = 0; k < 6; ++k) atomicAdd(&egc_stampw[k], wsum[k]);
#endif
}

static int ks16_of(int f_in) { return (f_in + 15) / 16; }
static int row_pieces(int f_in) {   // >= 4 KS16 (k up to 16 KS16 reads zeros) and odd (conflict-free fragment reads)
  int rp = std::max(f_in / 4 + 1, 4 * ks16_of(f_in));
  return (rp & 1) ? rp : rp + 1;
}

struct WLaunch { int launches, waves, R, slot_bytes; size_t lds; };
static WLaunch wlaunch(int f_in, int NT) {
  WLaunch l;
  l.launches = (NT + W_MAX_WAVES - 1) / W_MAX_WAVES;
  l.waves = (NT + l.launches - 1) / l.launches;           // the widest launch
  // the narrowest launch stages the same tile with fewer threads: size the slot for it
  const int narrow = NT / l.launches;
  const int pieces = WROWS * row_pieces(f_in);
  const int rn = (pieces + narrow * 64 - 1) / (narrow * 64), rw = (pieces + l.waves * 64 - 1) / (l.waves * 64);
  l.R = rn;
  l.slot_bytes = std::max(rn * narrow, rw * l.waves) * 64 * 16;
  l.lds = (size_t)3 * l.slot_bytes + (size_t)3 * WROWS * sizeof(float) + (size_t)W_MAX_WAVES * 64 * 16;   // ring, row scales, sink
  return l;
}

bool f16x2w_shape(int f_in, int f_g, int ldb, int w_cols) {
  // OPT-IN (EGC_GEMM_F16X2W=1): parity-green but not faster than the 16-column kernel where it was meant to be (ogbn-mag
  // shape 552-564 us against 543; DESIGN.md 3.2b has the stamps and ablations) -- the 16-column kernel stays the default
  const char* on = getenv("EGC_GEMM_F16X2W");
  if (on == nullptr || on[0] == '\0' || on[0] == '0') return false;
  if (f_in <= 128 || f_in > W_KMAX || (f_in & 3) != 0) return false;
  const WCols c = wcols(f_g, ldb, w_cols);
  if (c.NT < 1 || c.NT > 2 * W_MAX_WAVES) return false;
  const WLaunch l = wlaunch(f_in, c.NT);
  return l.lds <= 160 * 1024 && WROWS * row_pieces(f_in) < 65536;
}

size_t f16x2w_pack_bytes(int f_in, int f_g, int ldb, int w_cols) {
  if (f_in <= 128 || f_in > W_KMAX) return 0;
  const WCols c = wcols(f_g, ldb, w_cols);
  return (size_t)c.NT * ks16_of(f_in) * 2 * 64 * 8 * sizeof(u16) + (size_t)32 * c.NT * sizeof(float);
}

int f16x2w_pack(const float* wcat, int64_t rs, int64_t cs, int f_in, int f_g, int ldb, int w_cols, void* packed,
                hipStream_t stream) {
  const WCols c = wcols(f_g, ldb, w_cols);
  pack_f16x2w_kernel<<<32 * c.NT, 64, 0, stream>>>(wcat, rs, cs, f_in, c, ks16_of(f_in), (u16*)packed);
  EGC_LAUNCH_CHECK("pack_f16x2w_kernel");
  return EGC_OK;
}

template <int KS16>
static int launch_w(const float* x, const u16* packed, const float* bcat, int64_t M, int K, const WCols& c, float* bases,
                    float* weightings, hipStream_t stream) {
  const WLaunch l = wlaunch(K, c.NT);
  const int64_t n_tiles64 = ceil_div(M, WROWS);
  if (n_tiles64 >= ((int64_t)1 << 31)) return EGC_ERR_INVALID;
  const int n_tiles = (int)n_tiles64;
  if (l.lds > 160 * 1024) return EGC_ERR_UNSUPPORTED;
  auto kern = &basis_gemm_f16x2w_kernel<KS16>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_last_error("hipFuncSetAttribute(f16x2w)", e); return EGC_ERR_HIP; }
    attr_set = true;
  }
  const int RP = row_pieces(K);
  for (int q = 0, t0 = 0; q < l.launches; ++q) {
    const int ntl = (c.NT - t0 + (l.launches - q) - 1) / (l.launches - q);
    const int threads = ntl * 64;
    const int R = (WROWS * RP + threads - 1) / threads;
    // workgroups per CU: as many as the LDS and 8 wavefronts allow (short k, few column tiles: the d x GEMM of training)
    int per_cu = (int)std::min<size_t>((size_t)160 * 1024 / l.lds, (size_t)(W_MAX_WAVES / ntl));
    if (per_cu < 1) per_cu = 1;
    int grid = 256 * per_cu;
    if (const char* e = getenv("EGC_GEMM_GRID")) grid = atoi(e);
    if (grid > n_tiles) grid = n_tiles;
    kern<<<grid, threads, l.lds, stream>>>(x, packed, bcat, M, K, c, bases, weightings, n_tiles, RP, R, l.slot_bytes, t0);
    EGC_LAUNCH_CHECK("basis_gemm_f16x2w_kernel");
    t0 += ntl;
  }
#ifdef EGC_GEMMW_STAMPS
  {
    static int calls = 0;
    if (++calls == 10) {
      hipDeviceSynchronize();
      unsigned long long h[8];
      hipMemcpyFromSymbol(h, HIP_SYMBOL(egc_stampw), sizeof(h));
      const double per = (double)calls * n_tiles * c.NT;
      fprintf(stderr, "[gemmw stamps] K=%d NT=%d per tile and wavefront (s_memtime ticks): dma-wait %.0f barrier %.0f dma-issue %.0f split %.0f mfma %.0f store %.0f\n",
              K, c.NT, h[0] / per, h[1] / per, h[2] / per, h[3] / per, h[4] / per, h[5] / per);
    }
  }
#endif
  return EGC_OK;
}

int f16x2w_launch(const float* x, const void* packed, const float* bcat, int64_t M, int K, int f_g, int ldb, int W,
                  float* bases, float* weightings, hipStream_t stream) {
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return EGC_ERR_UNSUPPORTED;
  const WCols c = wcols(f_g, ldb, W);
  const u16* pk = (const u16*)packed;
  // 32-bit buffer offsets (masked stores sit at 2^31 + a scalar row offset): row ranges of less than 2 GiB per array
  const int64_t widest = std::max(std::max(K, ldb), W);
  int64_t max_rows = ((int64_t)0x7FFFFFF0 / (4 * widest)) & ~(int64_t)(WROWS - 1);
  if (const char* e = getenv("EGC_GEMM_MAX_ROWS")) max_rows = std::max<int64_t>(WROWS, atoll(e) & ~(int64_t)(WROWS - 1));  // tests
  for (int64_t r0 = 0; r0 < M; r0 += max_rows) {
    const int64_t rows = std::min(max_rows, M - r0);
    const float* xr = x + r0 * K;
    float* br = bases + r0 * ldb;
    float* wr = weightings != nullptr ? weightings + r0 * W : nullptr;
    int st;
    switch (ks16_of(K)) {
#define EGC_W_CASE(n) case n: st = launch_w<n>(xr, pk, bcat, rows, K, c, br, wr, stream); break;
      EGC_W_CASE(9) EGC_W_CASE(10) EGC_W_CASE(11) EGC_W_CASE(12) EGC_W_CASE(13) EGC_W_CASE(14) EGC_W_CASE(15)
      EGC_W_CASE(16) EGC_W_CASE(17) EGC_W_CASE(18) EGC_W_CASE(19) EGC_W_CASE(20) EGC_W_CASE(21) EGC_W_CASE(22)
#undef EGC_W_CASE
      default: return EGC_ERR_UNSUPPORTED;
    }
    if (st != EGC_OK) return st;
  }
  return EGC_OK;
}

}  // namespace egc
