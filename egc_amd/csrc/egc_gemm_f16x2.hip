// Basis transform + weightings Linear on the fp16 matrix cores with fp32-level accuracy (gfx950), for the
// shapes whose whole weight matrix fits the register files of one block (F_in <= 128, 192 virtual columns:
// the north-star layer).
//
//     [bases | weightings] = x[N,F_in] @ [bases_weight | comb.weight^T]  (+ comb.bias)
//
// Reference behaviour replaced: torch.matmul(x, bases_weight) (experiments/layers.py:97-101,
// optimized_layers.py:180) and comb_weights(x) (layers.py:110, optimized_layers.py:182).
//
// Split: every x row and every weight column is first scaled by a power of two so that its largest
// magnitude lies in [1, 2) (exact), then written as
//     xs = xh + 2^-11 xl,     ws = wh + 2^-11 wl        (xh = fp16(xs), xl = fp16(2^11 (xs - xh)))
// fp16 carries an 11-bit significand, so the two planes hold ~22 bits and the remainder is < 2^-22 of the
// row / column maximum; xl and wl are stored pre-multiplied by 2^11 so that they stay normal numbers.
// Three products are accumulated in fp32 on v_mfma_f32_32x32x16_f16,
//     acc0 = xh wh,     acc1 = xh wl + xl wh,     result = 2^ex 2^ew (acc0 + 2^-11 acc1),
// the dropped xl*wl term and the plane remainders are ~2^-22 relative: the result is within a few fp32
// roundings of the reference's fp32 GEMM (parity tests: <= 1e-5).  Three MFMAs per k-step instead of the
// six of the bf16x3 split (egc_gemm_bf16x3.hip) put this GEMM back under its 217 MB of HBM traffic.
//
// Structure: persistent blocks of 12 wavefronts (one per CU), 64-row x tiles double-buffered in LDS as two
// fp16 planes, wavefront (ct, rt) owns column tile ct of row half rt and keeps BOTH planes of its weight tile in registers for the
// whole kernel; all global traffic through buffer instructions; LDS-only barriers (details at the kernel).
#include <stdlib.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#include "egc_common.h"
#include "egc_gemm_split.h"

namespace egc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ inline void lds_barrier2() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int F16X2_KP = 128;          // k extent of the register-resident weight block (F_in <= 128, zero beyond F_in)

// biased exponent of the scale group's largest magnitude -> (scale, inverse scale), both exact powers of two
__device__ inline void scales_of(unsigned amax_bits, float& scale, float& inv) {
  unsigned be = amax_bits >> 23;        // sign already cleared
  be = be > 253u ? 253u : be;           // huge / inf / nan: keep the scale a normal number (values propagate)
  scale = __uint_as_float((254u - be) << 23);
  inv = __uint_as_float(be << 23);      // be == 0 (all-zero or denormal group): result flushes to 0
}

// two scaled floats -> packed fp16 pairs (h plane, 2^11-scaled l plane); element 0 in the low half
__device__ inline void split2_pk(f32x2 v, unsigned& h, unsigned& l) {
  const f16x2 hh = __builtin_convertvector(v, f16x2);
  h = __builtin_bit_cast(unsigned, hh);
  const f32x2 r = (v - __builtin_convertvector(hh, f32x2)) * 2048.f;
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

// packed[column tile][k-step][plane][lane][8] (fp16 bits, the MFMA B fragments as they are loaded) followed by float inv_scale[NV].  One wavefront per virtual column
// (runs once per parameter update -- every step when training): lanes stride over k, the column maximum is a
// wavefront all-reduce.
__global__ void __launch_bounds__(64) pack_f16x2_kernel(const float* __restrict__ wcat, int64_t rs, int64_t cs, int K, int F_g,
                                                        int W, int ldb, int NV, int KS, u16* __restrict__ packed) {
  const int v = blockIdx.x;
  const int lane = threadIdx.x;
  const int src = (v < F_g) ? v : ((v < ldb || v >= ldb + W) ? -1 : v - ldb + F_g);
  unsigned amax = 0;
  if (src >= 0)
    for (int k = lane; k < K; k += 64) amax = max(amax, __float_as_uint(wcat[k * rs + src * cs]) & 0x7fffffffu);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, d));
  float scale, inv;
  scales_of(amax, scale, inv);
  for (int k = lane; k < KS * GEMM_KT; k += 64) {
    const float w = (src >= 0 && k < K) ? wcat[k * rs + src * cs] * scale : 0.f;
    const _Float16 h = (_Float16)w;
    const _Float16 l = (_Float16)((w - (float)h) * 2048.f);
    // fragment order: the B operand of k-step s (16 k), plane p, column tile ct is 64 lanes x 8 halves, lane
    // 32 (k % 16 / 8) + column % 32 -- a wavefront of the GEMM fetches it as ONE contiguous KiB (with the planes laid
    // out [k-slab][plane][column][32 k] every lane's 16 bytes sat in a line of their own, and the 196 KB of weight
    // loads of a block took ~3 us of its prologue)
    const int64_t base = ((((int64_t)(v >> 5) * (F16X2_KP / 16) + (k >> 4)) * 2) * 64 + 32 * ((k & 15) >> 3) + (v & 31)) * 8 + (k & 7);
    packed[base] = __builtin_bit_cast(u16, h);
    packed[base + 64 * 8] = __builtin_bit_cast(u16, l);
  }
  if (lane == 0) reinterpret_cast<float*>(packed + (int64_t)KS * 2 * NV * GEMM_KT)[v] = inv;
}

#ifdef EGC_GEMM_STAMPS
__device__ unsigned long long* egc_stamp_buf2 = nullptr;  // diagnostic build only
#endif

// What bounds this kernel is each SIMD's vector issue port and its matrix pipe TOGETHER: a VALU instruction
// holds the port for 4 cycles, a v_mfma_f32_32x32x16_f16 for 8 and the pipe for 32 (MI355X_MICROARCH.md,
// constants table), so a tile runs at the pipe's pace only if about six vector instructions sit behind every
// MFMA, everywhere in the loop.  The vector work of a tile is about 150 instructions per wavefront (split of
// the next tile ~25 per 16-byte piece, scale/bias/store of the outputs ~68) against 24 MFMAs, so:
//   * a block is 12 wavefronts = 6 column tiles x 2 row halves of a 64-row x tile, one block per CU, 3
//     wavefronts per SIMD (two 6-wavefront blocks per CU do not become co-resident: measured);
//   * the epilogue of tile i is deferred into the MFMA loop of tile i+1: only t = acc0 + 2^-11 acc1 is formed
//     right after the loop (16 registers carried across the barrier), scaling, bias and stores follow beside
//     k-steps 0-3 of the next tile; the split of tile i+2 sits beside k-steps 4-7;
//   * no packed-f32 arithmetic (slower than two scalar operations next to MFMAs), |x| maxima through source
//     modifiers, the row exponent by integer operations, the low plane by one mixed-precision fma per element.
//
// x reaches the CU by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR destination) into a ring of two raw fp32
// tiles per block: registers cannot hold a prefetch deep enough to cover HBM latency (measured with one tile
// of register prefetch: 19 GB/s per CU).  Every thread later reads back exactly the 16-byte pieces its own
// wavefront requested -- ordered by that wavefront's counted vmcnt alone -- splits them into the fp16 planes
// and the wavefront immediately re-arms the slot with the tile three ahead: 64 KB per CU always in flight.
constexpr int F16X2_THREADS = 768;      // 6 column tiles x 2 row halves
constexpr int F16X2_ROWS = 64;
constexpr int F16X2_LDX = F16X2_KP + 8;
constexpr int F16X2_RAW_BYTES = F16X2_ROWS * F16X2_KP * 4;                  // one raw fp32 tile
constexpr int F16X2_PLANE_BYTES = 2 * 2 * F16X2_ROWS * F16X2_LDX * 2;       // two buffers x two planes
constexpr int F16X2_STORES_PER_TILE = 16;  // per wavefront (vmcnt arithmetic below)

#define EGC_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

__global__ void __launch_bounds__(F16X2_THREADS) basis_gemm_f16x2_kernel(const float* __restrict__ x,
                                                                          const u16* __restrict__ packed,
                                                                          const float* __restrict__ bcat, int64_t M, int K,
                                                                          int W, float* __restrict__ bases, int ldb,
                                                                          float* __restrict__ weightings, int NV,
                                                                          int rows_per_block) {
  constexpr int KP = F16X2_KP;
  constexpr int KSUB = KP / 16;         // 16-k MFMA steps
  constexpr int LDX = F16X2_LDX;
  constexpr int ROWS = F16X2_ROWS;
  constexpr int XBUF = 2 * ROWS * LDX;  // fp16 elements of one x buffer (2 planes)
  constexpr int nthreads = F16X2_THREADS;
  extern __shared__ __attribute__((aligned(16))) u16 smem_h2[];
  u16* xs = smem_h2;                                                             // [2][2][ROWS][LDX] fp16 planes
  char* raw = reinterpret_cast<char*>(smem_h2) + F16X2_PLANE_BYTES;              // [2][ROWS][KP] fp32 ring
  // inverse row scales of tile t live in row_inv[t % 3]: tile t-1's are still being read by its deferred
  // epilogue while the split of tile t+1 writes its own
  float* row_inv = reinterpret_cast<float*>(raw + 2 * F16X2_RAW_BYTES);          // [3][ROWS]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave % 6, rt = wave / 6;                       // column tile, 32-row half of the x tile
#ifdef EGC_GEMM_STAMPS
  unsigned long long t_entry;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry) :: "memory");
#endif
  const int l31 = lane & 31, hh = lane >> 5;
  const int cb = 32 * ct;
  // A block owns a CONTIGUOUS row range of M / gridDim rows (its last tile is partial: the DMA and the stores of the
  // rows beyond the range are dropped by range checks, and the loop is bound by those bytes): every block moves the
  // same bytes.  (Whole 64-row tiles dealt round-robin left 86 of 256 blocks with an 11th tile at config 2.)
  const int64_t row_lo = (int64_t)blockIdx.x * rows_per_block;
  const int64_t row_hi = row_lo + rows_per_block < M ? row_lo + rows_per_block : M;
  const int n_tiles = (int)((row_hi - row_lo + F16X2_ROWS - 1) / F16X2_ROWS);

  f16x8 wf[KSUB][2];
  float col_inv, col_bias;
  // 16-byte pieces of a tile: piece pc = tid + 768 i  <->  (row pc / 32, k 4 (pc % 32)); i = 2 exists for
  // wavefronts 0-7 only (2048 pieces)
  constexpr unsigned GOOB = 0xFFFFFFF0u;  // out-of-range offset: loads return 0 -- no branches
  constexpr unsigned SOOB = 0x80000000u;  // same for the stores, which add a scalar offset (host: buffers < 2 GiB)
  const u32x4 rx = {(unsigned)(uintptr_t)x, (unsigned)((uintptr_t)x >> 32) & 0xffffu, (unsigned)(M * K * 4), 0x00020000u};
  // this wavefront's 32 columns lie either in `bases` or in `weightings` (host: ldb % 32 == 0, or W == 0: every tile in `bases`)
  const bool to_bases = cb < ldb;
  const __amdgpu_buffer_rsrc_t ro =
      to_bases ? __builtin_amdgcn_make_buffer_rsrc((void*)bases, 0, (unsigned)(row_hi * ldb * 4), 0x00020000)
               : __builtin_amdgcn_make_buffer_rsrc((void*)weightings, 0, (unsigned)(row_hi * (int64_t)W * 4), 0x00020000);
  const int out_ld = to_bases ? ldb : W;
  const int out_col = (to_bases ? cb : cb - ldb) + l31;
#if defined(EGC_DIAG_GEMM_NO_W_STORE) || defined(EGC_DIAG_GEMM_NO_W)   // diagnostic builds (tools/f3_roundtrip_cost.py): the weightings are
  const bool col_ok = to_bases && out_col < ldb;                        // not written / not computed at all (wrong results, honest times)
#else
  const bool col_ok = out_col < (to_bases ? ldb : W);
#endif
  const unsigned raw_lds = (unsigned)(uintptr_t)raw;  // LDS byte address of the ring
  const bool third = wave < 8;

  // one LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global offsets to 1 KiB of contiguous LDS
  auto dma_piece = [&](int tile, int slot, int i) {
    const int pc = tid + nthreads * i;
    const int row = pc >> 5;
    const int k4 = (pc & 31) * 4;
    const int64_t gm = row_lo + (int64_t)tile * ROWS + row;
    const bool ok = (tile < n_tiles) & (gm < row_hi) & (k4 < K);
    const unsigned voff = ok ? (unsigned)((gm * K + k4) * 4) : GOOB;
    const unsigned dst =
        __builtin_amdgcn_readfirstlane(raw_lds + slot * F16X2_RAW_BYTES + (wave * 64 + nthreads * i) * 16);  // wave-uniform
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(dst), "s"(rx)
                 : "memory");
  };
  auto dma_tile = [&](int tile, int slot) {
    dma_piece(tile, slot, 0);
    dma_piece(tile, slot, 1);
    if (third) dma_piece(tile, slot, 2);
  };
  // A row is 32 consecutive pieces = one half wavefront: its largest magnitude is an all-reduce over 32 lanes
  // (4 DPP steps inside the rows of 16, one cross-row exchange).  NaNs drop out of the maxima and propagate
  // through the products instead.
  auto row_amax = [&](const float4 v) -> unsigned {
    // |.| through source modifiers (two instructions for four elements; written as asm because the compiler
    // canonicalises every fmax operand), then unsigned maxima on the bit patterns of these non-negative floats
    float m;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|\n\tv_max_f32 %0, |%4|, %0" : "=&v"(m) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
    unsigned a = __float_as_uint(m);
    a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x141, 0xf, 0xf, true));  // row_half_mirror
    a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x140, 0xf, 0xf, true));  // row_mirror
    return max(a, (unsigned)__builtin_amdgcn_ds_swizzle((int)a, 0x401F));                 // lane ^ 16
  };
  auto split_store = [&](int buf, int ri, int i, const float4 v, unsigned amax) {
    const int pc = tid + nthreads * i;
    const int row = pc >> 5;
    const int k4 = (pc & 31) * 4;
    // exponent field of the row maximum, kept where both 2^-e and 2^(11-e) are normal numbers (rows below
    // 2^-113 are scaled by 2^114 only and keep fewer bits; rows above 2^126 overflow as they would in fp32)
    unsigned e = amax & 0x7f800000u;
    e = min(max(e, 13u << 23), 253u << 23);
    const float sc = __uint_as_float(0x7f000000u - e);                  // 2^-e
    const float sc2k = __uint_as_float(0x7f000000u + (11u << 23) - e);  // 2^(11-e)
    const f16x2 h01 = __builtin_convertvector(f32x2{v.x * sc, v.y * sc}, f16x2);
    const f16x2 h23 = __builtin_convertvector(f32x2{v.z * sc, v.w * sc}, f16x2);
    // (xs - h) * 2^11 = fma(h, -2^11, x * 2^(11-e)): one mixed-precision fma, rounded once to fp16
    f16x2 l01, l23;
    l01[0] = (_Float16)__builtin_fmaf((float)h01[0], -2048.f, v.x * sc2k);
    l01[1] = (_Float16)__builtin_fmaf((float)h01[1], -2048.f, v.y * sc2k);
    l23[0] = (_Float16)__builtin_fmaf((float)h23[0], -2048.f, v.z * sc2k);
    l23[1] = (_Float16)__builtin_fmaf((float)h23[1], -2048.f, v.w * sc2k);
    u16* dst = xs + buf * XBUF + row * LDX + k4;
    *reinterpret_cast<u32x2*>(dst) = u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
    *reinterpret_cast<u32x2*>(dst + ROWS * LDX) = u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
    row_inv[ri * ROWS + row] = __uint_as_float(e);  // 2^e; all 32 lanes of the row write the same word: no branch
  };
  auto raw_piece = [&](int slot, int i) -> float4 {
    return *reinterpret_cast<const float4*>(raw + slot * F16X2_RAW_BYTES + (tid + nthreads * i) * 16);
  };
  auto stage = [&](int buf, int ri, int slot, int i) {
    const float4 v = raw_piece(slot, i);
    split_store(buf, ri, i, v, row_amax(v));
  };
  // A operands (x): lane -> row 32 rt + l31, k = 16 s + 8 hh .. + 7
  auto frag = [&](int buf, int s, f16x8& xh, f16x8& xl) {
    const u16* xb = xs + buf * XBUF + (32 * rt + l31) * LDX + 16 * s + 8 * hh;
    xh = *reinterpret_cast<const f16x8*>(xb);
    xl = *reinterpret_cast<const f16x8*>(xb + ROWS * LDX);
  };
  f32x16 acc0, acc1, t;   // t: previous tile's acc0 + 2^-11 acc1, waiting for its scales
  auto mfma_step = [&](int s, const f16x8 xh, const f16x8 xl) {
#ifdef EGC_DIAG_GEMM_NO_W
    if (!to_bases) return;
#endif
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wf[s][0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wf[s][0], acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wf[s][1], acc1, 0, 0, 0);
  };
  // D layout (x as the A operand): lane -> column cb + l31, rows 8 j + 4 hh + i: one dword store writes two
  // full 128-byte lines.  (With x as B a lane would hold 4 consecutive columns of one row and a dwordx4 store
  // would touch 32 lines: the stores, not the arithmetic, then set the tile time.)  Rows past M fall outside
  // the buffer and are dropped by its range check; the row part of the address is a scalar offset.
  auto epilogue = [&](unsigned voff, int j, const float* rinv_t) {
    const float4 ri = *reinterpret_cast<const float4*>(rinv_t + 32 * rt + 8 * j + 4 * hh);
    // 2^ex 2^ew (acc0 + 2^-11 acc1) + bias; the scale product is a power of two, so the fma rounds once
    const float v0 = __builtin_fmaf(t[4 * j], col_inv * ri.x, col_bias);
    const float v1 = __builtin_fmaf(t[4 * j + 1], col_inv * ri.y, col_bias);
    const float v2 = __builtin_fmaf(t[4 * j + 2], col_inv * ri.z, col_bias);
    const float v3 = __builtin_fmaf(t[4 * j + 3], col_inv * ri.w, col_bias);
    const int so = (8 * j) * out_ld * 4;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), ro, voff, so, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), ro, voff, so + out_ld * 4, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v2), ro, voff, so + 2 * out_ld * 4, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v3), ro, voff, so + 3 * out_ld * 4, 0);
  };
  auto out_offset = [&](int tile, bool valid) -> unsigned {
    return (valid & col_ok) ? (unsigned)(((row_lo + (int64_t)tile * ROWS + 32 * rt + 4 * hh) * out_ld + out_col) * 4) : SOOB;
  };
#define EGC_PIN __builtin_amdgcn_sched_barrier(0)

  constexpr int stride = 1;   // (tiles of this block's own range, in order)
  int tile = 0;
  if (n_tiles <= 0) return;
  if (K < KP) {  // columns k >= K of a tile are out of range for the DMA and must read as 0
    for (int i = tid; i < 2 * F16X2_RAW_BYTES / 16; i += nthreads) reinterpret_cast<u32x4*>(raw)[i] = u32x4{0, 0, 0, 0};
    lds_barrier2();
  }
  // first two tiles on their way before anything else: the weight loads below overlap their latency
  dma_tile(tile, 0);
  dma_tile(tile + stride, 1);
#ifdef EGC_GEMM_STAMPS
  unsigned long long t_a, t_b;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_a) :: "memory");
#endif
  // both planes of this wavefront's 128 x 32 weight block, as B operands of v_mfma_f32_32x32x16_f16:
  // lane -> column cb + l31, k = 16 s + 8 hh .. + 7
#pragma unroll
  for (int s = 0; s < KSUB; ++s)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      wf[s][p] = *reinterpret_cast<const f16x8*>(packed + ((((int64_t)ct * KSUB + s) * 2 + p) * 64 + lane) * 8);   // one KiB per wavefront
    }
  // every output element of a lane belongs to ONE column (cb + l31): its inverse scale and bias stay in registers
  col_inv = reinterpret_cast<const float*>(packed + (int64_t)(KP / GEMM_KT) * 2 * NV * GEMM_KT)[cb + l31];
  const int wcol = cb + l31 - ldb;
  col_bias = (bcat != nullptr && wcol >= 0 && wcol < W) ? bcat[wcol] : 0.f;
  EGC_VMCNT(0);
  // the compiler counts only its own loads: let it retire the weight loads HERE (the counter is already zero),
  // or its waits in the first tile would also drain the DMAs issued below
  asm volatile("" : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[1][0]), "+v"(wf[1][1]), "+v"(wf[2][0]), "+v"(wf[2][1]),
               "+v"(wf[3][0]), "+v"(wf[3][1]), "+v"(wf[4][0]), "+v"(wf[4][1]), "+v"(wf[5][0]), "+v"(wf[5][1]),
               "+v"(wf[6][0]), "+v"(wf[6][1]), "+v"(wf[7][0]), "+v"(wf[7][1]), "+v"(col_inv), "+v"(col_bias));
#ifdef EGC_GEMM_STAMPS
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_b) :: "memory");
#endif
  stage(0, 0, 0, 0);
  stage(0, 0, 0, 1);
  if (third) stage(0, 0, 0, 2);
  dma_tile(tile + 2 * stride, 0);
  lds_barrier2();
  int buf = 0;
  int ri_cur = 0;                 // row_inv slot of the tile being multiplied (tile index mod 3)
  unsigned prev_off = SOOB;       // no previous tile yet: its stores are dropped
  const float* prev_ri = row_inv;
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] = 0.f;
#ifdef EGC_GEMM_STAMPS
  unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, t0, t1, r0, r1;
#define EGC_STAMP(k) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); tsum[k] += t1 - t0; t0 = t1; }
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0) :: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#else
#define EGC_STAMP(k)
#endif
  // In the loop every wavefront issues, per tile, F16X2_STORES_PER_TILE stores (k-steps 0-3) and then 3
  // (wavefronts 0-7) or 2 (8-11) DMA pieces (k-step 7), always in this order, and the counter retires them in
  // order.  When the pieces of the next tile are read (k-step 4), the operations issued after the DMA of piece
  // i are the rest of that DMA group, one whole tile of stores + DMA, and this tile's stores:
  //     piece 0: (2|1) + 16 + (3|2) + 16 = 37 | 35     piece 1: (1|0) + 16 + (3|2) + 16 = 36 | 34
  //     piece 2 (wavefronts 0-7 only):   0 + 16 +  3    + 16 = 35
  // A smaller count is always safe: one wait for 34 covers every piece of every wavefront.
  static_assert(F16X2_STORES_PER_TILE == 16, "vmcnt count below");
  for (; tile < n_tiles; tile += stride) {
    const int slot = buf ^ 1;  // ring slot of the next tile (tile index parity == plane buffer parity)
    const int ri_next = ri_cur == 2 ? 0 : ri_cur + 1;
    f16x8 xh, xl, yh, yl;
    frag(buf, 0, xh, xl);
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    // k-steps 0-3: the previous tile leaves
    frag(buf, 1, yh, yl);
    mfma_step(0, xh, xl);
    epilogue(prev_off, 0, prev_ri);
    EGC_PIN;
    frag(buf, 2, xh, xl);
    mfma_step(1, yh, yl);
    epilogue(prev_off, 1, prev_ri);
    EGC_PIN;
    frag(buf, 3, yh, yl);
    mfma_step(2, xh, xl);
    epilogue(prev_off, 2, prev_ri);
    EGC_PIN;
    frag(buf, 4, xh, xl);
    mfma_step(3, yh, yl);
    epilogue(prev_off, 3, prev_ri);
    EGC_PIN;
    EGC_STAMP(0)
    // k-steps 4-7: the next tile is split and staged, its ring slot re-armed
    EGC_VMCNT(34);
    frag(buf, 5, yh, yl);
    mfma_step(4, xh, xl);
    stage(buf ^ 1, ri_next, slot, 0);
    EGC_PIN;
    frag(buf, 6, xh, xl);
    mfma_step(5, yh, yl);
    stage(buf ^ 1, ri_next, slot, 1);
    EGC_PIN;
    frag(buf, 7, yh, yl);
    mfma_step(6, xh, xl);
    if (third) stage(buf ^ 1, ri_next, slot, 2);
    EGC_PIN;
    mfma_step(7, yh, yl);
    dma_tile(tile + 3 * stride, slot);
    EGC_PIN;
    EGC_STAMP(1)
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = __builtin_fmaf(acc1[r], 1.f / 2048.f, acc0[r]);
    prev_off = out_offset(tile, true);
    prev_ri = row_inv + ri_cur * ROWS;
    ri_cur = ri_cur == 2 ? 0 : ri_cur + 1;
    EGC_STAMP(2)
    lds_barrier2();
    EGC_STAMP(3)
    buf ^= 1;
  }
  epilogue(prev_off, 0, prev_ri);
  epilogue(prev_off, 1, prev_ri);
  epilogue(prev_off, 2, prev_ri);
  epilogue(prev_off, 3, prev_ri);
  EGC_VMCNT(0);  // no DMA may still be writing this block's LDS when it is handed to the next block
#ifdef EGC_GEMM_STAMPS
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
  tsum[5] = r1 - r0;
  if (lane == 0 && egc_stamp_buf2 != nullptr) {
    for (int k = 0; k < 6; ++k) egc_stamp_buf2[(blockIdx.x * 16 + wave) * 6 + k] = tsum[k];
    egc_stamp_buf2[1024 * 16 * 6 + (blockIdx.x * 16 + wave) * 2] = t_entry;
    egc_stamp_buf2[1024 * 16 * 6 + (blockIdx.x * 16 + wave) * 2 + 1] = r1;
    egc_stamp_buf2[(blockIdx.x * 16 + wave) * 6 + 4] = r0 - t_entry;
    egc_stamp_buf2[(blockIdx.x * 16 + wave) * 6 + 3] = ((t_a - t_entry) << 32) | (t_b - t_entry);
  }
#endif
}

size_t f16x2_pack_bytes(int KS, int NV) { return (size_t)KS * 2 * NV * GEMM_KT * sizeof(u16) + (size_t)NV * sizeof(float); }

int f16x2_pack(const float* wcat, int64_t rs, int64_t cs, int f_in, int f_g, int w_cols, int ldb, int NV, int KS, void* packed,
               hipStream_t stream) {
  pack_f16x2_kernel<<<NV, 64, 0, stream>>>(wcat, rs, cs, f_in, f_g, w_cols, ldb, NV, KS, (u16*)packed);
  EGC_LAUNCH_CHECK("pack_f16x2_kernel");
  return EGC_OK;
}

static int f16x2_launch_rows(const float* x, const void* packed, const float* bcat, int64_t M, int K, int W, float* bases,
                             int ldb, float* weightings, int NV, hipStream_t stream) {
  constexpr int ROWS = F16X2_ROWS;
  const int threads = F16X2_THREADS;
  const int64_t n_tiles64 = (M + ROWS - 1) / ROWS;
  if (n_tiles64 >= ((int64_t)1 << 31)) return EGC_ERR_INVALID;
  const int n_tiles = (int)n_tiles64;
  const size_t lds = (size_t)F16X2_PLANE_BYTES + 2 * (size_t)F16X2_RAW_BYTES + (size_t)(3 * ROWS) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&basis_gemm_f16x2_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_last_error("hipFuncSetAttribute(f16x2)", e); return EGC_ERR_HIP; }
    attr_set = true;
  }
  int grid = 256;  // one 12-wavefront block per CU (registers: 3 wavefronts per SIMD)
  if (grid > n_tiles) grid = n_tiles;
  const int rows_per_block = (int)((M + grid - 1) / grid);   // contiguous, equal row ranges
#ifdef EGC_GEMM_STAMPS
  static unsigned long long* dbuf = nullptr;
  if (dbuf == nullptr) {
    hipMalloc(&dbuf, 1024 * 16 * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(egc_stamp_buf2), &dbuf, sizeof(dbuf));
  }
  hipMemset(dbuf, 0, 1024 * 16 * 8 * 8);
#endif
  basis_gemm_f16x2_kernel<<<grid, threads, lds, stream>>>(x, (const u16*)packed, bcat, M, K, W, bases, ldb, weightings,
                                                               NV, rows_per_block);
  EGC_LAUNCH_CHECK("basis_gemm_f16x2_kernel");
#ifdef EGC_GEMM_STAMPS
  {
    hipDeviceSynchronize();
    static int calls = 0;
    if (++calls == 20) {
      std::vector<unsigned long long> h(1024 * 16 * 8);
      hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
      double sum[6] = {0, 0, 0, 0, 0, 0}; int nw = 0;
      for (int b = 0; b < grid; ++b)
        for (int w = 0; w < threads / 64; ++w) {
          for (int k = 0; k < 6; ++k) sum[k] += (double)h[(b * 16 + w) * 6 + k];
          ++nw;
        }
      {
        unsigned long long lo = ~0ull, hi = 0; int early = 0;
        for (int b = 0; b < grid; ++b) { unsigned long long e = h[1024 * 16 * 6 + (b * 16) * 2]; if (e && e < lo) lo = e; }
        for (int b = 0; b < grid; ++b) {
          unsigned long long e = h[1024 * 16 * 6 + (b * 16) * 2], x = h[1024 * 16 * 6 + (b * 16) * 2 + 1];
          if (x > hi) hi = x;
          if (e - lo < 300) ++early;  // entered within 3 us of the first block
        }
        double pro = 0, lmin = 1e30, lmax = 0, pa = 0, pb = 0;
        for (int b = 0; b < grid; ++b) {
          pro += (double)h[(b * 16) * 6 + 4];
          pa += (double)(h[(b * 16) * 6 + 3] >> 32); pb += (double)(h[(b * 16) * 6 + 3] & 0xffffffffull);
          lmin = std::min(lmin, (double)h[(b * 16) * 6 + 5]);
          lmax = std::max(lmax, (double)h[(b * 16) * 6 + 5]);
        }
        fprintf(stderr, "[stamps f16x2] kernel span %.1f us, %d of %d blocks entered within 3 us; prologue avg %.1f us (dma issued %.1f, all landed %.1f); loop min %.1f max %.1f us\n",
                (hi - lo) * 0.01, early, grid, pro / grid * 0.01, pa / grid * 0.01, pb / grid * 0.01, lmin * 0.01, lmax * 0.01);
      }
      const double tpb = (double)n_tiles / grid;
      const double cyc = sum[0] + sum[1] + sum[2] + sum[3];
      fprintf(stderr, "[stamps f16x2] per tile per wave (cycles): k0-3+epi %.0f  k4-7+stage %.0f  t-fma %.0f  barrier %.0f  (-) %.0f "
              "(tiles/block %.1f)  clock %.2f GHz  loop %.1f us\n", sum[0] / nw / tpb, sum[1] / nw / tpb, sum[2] / nw / tpb,
              sum[3] / nw / tpb, sum[4] / nw / tpb, tpb, cyc / sum[5] * 0.1, sum[5] / nw * 0.01);
    }
  }
#endif
  return EGC_OK;
}

// The kernel addresses x, bases and weightings through 32-bit buffer offsets (and drops masked stores at
// offset 2^31 + scalar row offset): row ranges of less than 2 GiB per array are launched one after another.
int f16x2_launch(const float* x, const void* packed, const float* bcat, int64_t M, int K, int W, float* bases, int ldb,
                 float* weightings, int NV, hipStream_t stream) {
  if (NV != 192 || K > F16X2_KP || K % 4 != 0 || (ldb % 32 != 0 && W != 0) || (reinterpret_cast<uintptr_t>(x) & 15) != 0)
    return EGC_ERR_UNSUPPORTED;
  const int64_t widest = std::max(std::max(K, ldb), W);
  int64_t max_rows = ((int64_t)0x7FFFFFF0 / (4 * widest)) & ~(int64_t)(F16X2_ROWS - 1);
  if (const char* e = getenv("EGC_GEMM_MAX_ROWS")) max_rows = std::max<int64_t>(F16X2_ROWS, atoll(e) & ~(int64_t)(F16X2_ROWS - 1));  // tests
  for (int64_t r0 = 0; r0 < M; r0 += max_rows) {
    const int64_t rows = std::min(max_rows, M - r0);
    const int st = f16x2_launch_rows(x + r0 * K, packed, bcat, rows, K, W, bases + r0 * ldb, ldb,
                                     weightings != nullptr ? weightings + r0 * W : nullptr, NV, stream);
    if (st != EGC_OK) return st;
  }
  return EGC_OK;
}

}  // namespace egc
