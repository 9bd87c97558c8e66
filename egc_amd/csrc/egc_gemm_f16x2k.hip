// Basis transform + weightings Linear for F_in > 128 on the fp16 matrix cores with fp32-level accuracy (gfx950):
// the register-stationary fp16x2 scheme of egc_gemm_f16x2.hip carried to long k (128 < F_in <= 384) -- the
// reference's ogbn-mag layers (352 wide, mag/models.py:23-53) and its other trained nets (136 ... 304 wide).
//
//     [bases | weightings] = x[N,F_in] @ [bases_weight | comb.weight^T]  (+ comb.bias)
//
// Reference behaviour replaced: torch.matmul(x, bases_weight) (experiments/layers.py:97-101,
// optimized_layers.py:180) and comb_weights(x) (layers.py:110, optimized_layers.py:182).
//
// Numerics: exactly those of egc_gemm_f16x2.hip -- every x row and weight column scaled by a power of two so that
// its largest magnitude lies in [1, 2), xs = xh + 2^-11 xl in two fp16 planes (same for w), three products
// (xh wh, xh wl + xl wh) accumulated in fp32, result 2^ex 2^ew (acc0 + 2^-11 acc1): ~2^-22 relative.
//
// What changes with long k is WHERE the weights can live.  [F_in x 208 columns] in two fp16 planes is 293 KB at the
// ogbn-mag shape: no LDS holds it and a 32-column register block (176 registers) leaves no room for anything else,
// which is why the bf16x3 kernel re-stages the weights from L2 every k-step and ends up LDS-bandwidth bound (1.0 ms
// against 0.26 ms of HBM or MFMA time).  Here a wavefront owns SIXTEEN columns for the whole k range
// (v_mfma_f32_16x16x32_f16: 4 registers per plane and k-step of 32 -> 88 registers at F_in = 352) and one workgroup
// of up to 16 wavefronts covers up to 256 virtual columns: the weights are read from memory ONCE per workgroup.
// x arrives by LDS-DMA into a ring of two raw fp32 tiles of 16 rows (the row maximum needs the whole row before
// the first MFMA, so a tile is complete rows), is split into two fp16 planes (double-buffered) half a wavefront
// per row, and every wavefront multiplies the same 16 x F_in planes by its own weights: per tile and wavefront
// 3 KS MFMAs of 16 cycles against 2 KS LDS reads of 1 KB.  One barrier per tile: the split of tile t + 1 and
// the MFMAs of tile t sit between the same two barriers, so the vector work of one wavefront runs beside the matrix
// work of another.
#include <stdlib.h>

#include <algorithm>
#include <cstdio>

#include "egc_common.h"
#include "egc_gemm_split.h"

namespace egc {

typedef float f32x4k __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8k __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2k __attribute__((ext_vector_type(2)));
typedef float f32x2k __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef unsigned int u32x4k __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2k __attribute__((ext_vector_type(2)));

constexpr int KROWS = 16;  // rows of an x tile = one MFMA row block

__device__ inline void lds_barrier_k() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Virtual column space: column tiles of 16; tiles [0, TB) hold the bases columns (ldb of them, the rest of the last
// tile is padding that is never stored), tiles [TB, NT) the weightings columns.
struct KCols {
  int F_g, ldb, W, TB, NT;
  __host__ __device__ inline int source(int v) const {  // column of wcat behind virtual column v, or -1
    const int t = v >> 4;
    if (t < TB) return v < F_g ? v : -1;
    const int w = v - 16 * TB;
    return w < W ? F_g + w : -1;
  }
};

// packed: [NT][KS][2 planes][64 lanes][8] fp16 -- the A fragments of v_mfma_f32_16x16x32_f16 in register order
// (lane (i, kq) holds k = 32 s + 8 kq .. + 7 of column 16 t + i) -- followed by float inv_scale[16 NT].
// One wavefront per virtual column: lanes stride over k, the column maximum is a wavefront all-reduce.
__global__ void __launch_bounds__(64) pack_f16x2k_kernel(const float* __restrict__ wcat, int64_t rs, int64_t cs, int K,
                                                         KCols c, int KS, u16* __restrict__ packed) {
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
  const int t = v >> 4, i = v & 15;
  for (int k = lane; k < KS * 32; k += 64) {
    const float w = (src >= 0 && k < K) ? wcat[k * rs + src * cs] * scale : 0.f;
    const _Float16 h = (_Float16)w;
    const _Float16 l = (_Float16)((w - (float)h) * 2048.f);
    const int s = k >> 5, kq = (k >> 3) & 3, e = k & 7;
    const int64_t base = ((((int64_t)t * KS + s) * 2) * 64 + (kq * 16 + i)) * 8 + e;
    packed[base] = __builtin_bit_cast(u16, h);
    packed[base + 64 * 8] = __builtin_bit_cast(u16, l);
  }
  if (lane == 0) reinterpret_cast<float*>(packed + (int64_t)c.NT * KS * 2 * 64 * 8)[v] = inv;
}

template <int N>
__device__ inline void vmwait_k() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wave-uniform runtime count (DMA wave-instructions per tile depend on F_in and the workgroup size)
__device__ inline void vmwait_rt(int n) {
  switch (n) {
    case 0: vmwait_k<0>(); break;   case 1: vmwait_k<1>(); break;   case 2: vmwait_k<2>(); break;
    case 3: vmwait_k<3>(); break;   case 4: vmwait_k<4>(); break;   case 5: vmwait_k<5>(); break;
    case 6: vmwait_k<6>(); break;   case 7: vmwait_k<7>(); break;   case 8: vmwait_k<8>(); break;
    case 9: vmwait_k<9>(); break;   case 10: vmwait_k<10>(); break; case 11: vmwait_k<11>(); break;
    case 12: vmwait_k<12>(); break; case 13: vmwait_k<13>(); break; case 14: vmwait_k<14>(); break;
    case 15: vmwait_k<15>(); break; case 16: vmwait_k<16>(); break; case 17: vmwait_k<17>(); break;
    case 18: vmwait_k<18>(); break; case 19: vmwait_k<19>(); break; case 20: vmwait_k<20>(); break;
    case 21: vmwait_k<21>(); break; case 22: vmwait_k<22>(); break; case 23: vmwait_k<23>(); break;
    case 24: vmwait_k<24>(); break; case 25: vmwait_k<25>(); break; case 26: vmwait_k<26>(); break;
    case 27: vmwait_k<27>(); break; case 28: vmwait_k<28>(); break; case 29: vmwait_k<29>(); break;
    case 30: vmwait_k<30>(); break; case 31: vmwait_k<31>(); break; case 32: vmwait_k<32>(); break;
    default: vmwait_k<0>(); break;
  }
}

#ifdef EGC_GEMMK_STAMPS
__device__ unsigned long long egc_stampk[8];  // diagnostic build only: cycles per phase, summed over wavefronts
#define KST(k) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); ksum[k] += _t - kt0; kt0 = _t; }
#else
#define KST(k)
#endif

template <int KS, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) basis_gemm_f16x2k_kernel(const float* __restrict__ x, const u16* __restrict__ packed,
                                                                      const float* __restrict__ bcat, int64_t M, int K,
                                                                      KCols c, float* __restrict__ bases,
                                                                      float* __restrict__ weightings, int n_tiles, int LDX,
                                                                      int R, int slot_bytes, int tile0, int ring,
                                                                      const float* __restrict__ addend) {
  extern __shared__ __attribute__((aligned(16))) char smem_k[];
  char* raw = smem_k;                                                    // [ring][slot_bytes] raw fp32 tiles (DMA ring)
  u16* xs = reinterpret_cast<u16*>(smem_k + ring * slot_bytes);          // [2 buffers][2 planes][KROWS][LDX] fp16
  float* row_inv = reinterpret_cast<float*>(xs + 4 * KROWS * LDX);       // [2 buffers][KROWS]
  float* colinfo = row_inv + 2 * KROWS;                                  // [16 NT][2]: inverse column scale, bias
  const int tid = threadIdx.x;
  const int nthreads = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = tile0 + wave;                                           // this wavefront's column tile (a launch covers tiles tile0 .. tile0 + wavefronts - 1)
  const int j = lane & 15, quad = lane >> 4;
  const int K4 = K >> 2;                                                 // 16-byte pieces per row
  constexpr unsigned GOOB = 0xFFFFFFF0u;
  const u32x4k rx = {(unsigned)(uintptr_t)x, (unsigned)((uintptr_t)x >> 32) & 0xffffu, (unsigned)(M * K * 4), 0x00020000u};
  const unsigned raw_lds = (unsigned)(uintptr_t)raw;
  const unsigned magic_K4 = (unsigned)(((uint64_t)1 << 32) / (uint64_t)K4) + 1u;

  // one LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global offsets to 1 KiB of contiguous LDS
  auto dma_tile = [&](int tile, int slot) {
    for (int i = 0; i < R; ++i) {
      const int pc = tid + nthreads * i;
      const int row = (int)__umulhi((unsigned)pc, magic_K4);   // pc / K4 for pc < 2^16
      const int k4 = pc - row * K4;
      const int64_t gm = (int64_t)tile * KROWS + row;
      const bool ok = (tile < n_tiles) & (row < KROWS) & (gm < M);
      const unsigned voff = ok ? (unsigned)((gm * K + 4 * k4) * 4) : GOOB;
      const unsigned dst = __builtin_amdgcn_readfirstlane(raw_lds + slot * slot_bytes + (wave * 64 + nthreads * i) * 16);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(voff), "s"(dst), "s"(rx)
                   : "memory");
    }
  };

  const int stride = gridDim.x;
  int tile = blockIdx.x;
  if (tile >= n_tiles) return;
  for (int r = 0; r < ring; ++r) dma_tile(tile + r * stride, r);    // `ring` tiles of x per CU in flight (two left HBM latency exposed: 3.8 TB/s)
  // zero the planes once: the k range [K, 32 KS) and the row padding are never written again
  for (int i = tid; i < 4 * KROWS * LDX / 8; i += nthreads) reinterpret_cast<u32x4k*>(xs)[i] = u32x4k{0, 0, 0, 0};

  // both planes of this wavefront's F_in x 16 weight block, as A operands
  f16x8k wf[KS][2];
  {
    const u16* src = packed + ((int64_t)ct * KS * 2 * 64 + lane) * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      wf[s][0] = *reinterpret_cast<const f16x8k*>(src + (s * 2) * 64 * 8);
      wf[s][1] = *reinterpret_cast<const f16x8k*>(src + (s * 2 + 1) * 64 * 8);
    }
  }
  // output addressing: lane (j, quad) holds, per row block, row 16 rb + j and virtual columns 16 ct + 4 quad .. + 3
  const bool to_bases = ct < c.TB;
  const int out_ld = to_bases ? c.ldb : c.W;
  const int col0 = to_bases ? 16 * ct + 4 * quad : 16 * (ct - c.TB) + 4 * quad;
  const int lim = to_bases ? c.ldb : c.W;
  float* outp = to_bases ? bases : weightings;
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, (unsigned)(M * out_ld * 4), 0x00020000);
  const float* inv_tab = reinterpret_cast<const float*>(packed + (int64_t)c.NT * KS * 2 * 64 * 8);
  // inverse column scales and biases live in LDS (8 registers less next to the 8 KS weight registers)
  for (int v = tid; v < 16 * c.NT; v += nthreads) {
    const int w = v - 16 * c.TB;
    colinfo[2 * v] = inv_tab[v];
    colinfo[2 * v + 1] = (w >= 0 && w < c.W && bcat != nullptr) ? bcat[w] : 0.f;
  }
  const bool vec_store = (out_ld & 3) == 0;
  // `addend` (the host passes it only where the bases rows are stored as 16-byte pieces): out = x W + addend on the bases columns --
  // the residual branch's gradient joining d x in the d x GEMM's store (one 16-byte load per lane and tile, requested right after
  // the barrier, ahead of the tile's DMA requests, by inline assembly: the compiler, which cannot see the DMAs, would wait for
  // everything)
  const bool add = addend != nullptr && to_bases;    // wave-uniform
  const u32x4k ra = {(unsigned)(uintptr_t)addend, (unsigned)((uintptr_t)addend >> 32) & 0xffffu, add ? (unsigned)(M * out_ld * 4) : 0u, 0x00020000u};
  vmwait_k<0>();
  // the compiler counts only its own loads: let it retire the weight loads HERE
#pragma unroll
  for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(wf[s][0]), "+v"(wf[s][1]));
  // the first tiles were requested before the weight loads: all have landed by now (vmcnt(0) above)
  lds_barrier_k();

  const int hw = tid >> 5;                    // half-wavefront index: one row of the tile per half-wavefront and pass
  const int hl = tid & 31;
  const int n_hw = nthreads >> 5;
  const int n_stores = vec_store ? 1 : 4;     // store instructions per tile and wavefront
  // raw fp32 rows of ring slot `rslot` -> the two fp16 planes of buffer `pbuf` + row scales
  auto split = [&](int rslot, int pbuf) {
    const char* rs = raw + rslot * slot_bytes;
    u16* xp = xs + pbuf * 2 * KROWS * LDX;
    for (int row = hw; row < KROWS; row += n_hw) {
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int k4 = hl + 32 * i;
        if (k4 < K4) {
          const float4 v = *reinterpret_cast<const float4*>(rs + ((size_t)row * K4 + k4) * 16);
          float mi;
          asm("v_max3_f32 %0, |%1|, |%2|, |%3|\n\tv_max_f32 %0, |%4|, %0" : "=&v"(mi) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
          m = fmaxf(m, mi);
        }
      }
      unsigned a = __float_as_uint(m);
      a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
      a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
      a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x141, 0xf, 0xf, true));  // row_half_mirror
      a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x140, 0xf, 0xf, true));  // row_mirror
      a = max(a, (unsigned)__builtin_amdgcn_ds_swizzle((int)a, 0x401F));                    // lane ^ 16
      unsigned e = a & 0x7f800000u;
      e = min(max(e, 13u << 23), 253u << 23);
      const float sc = __uint_as_float(0x7f000000u - e);                  // 2^-e
      const float sc2k = __uint_as_float(0x7f000000u + (11u << 23) - e);  // 2^(11-e)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int k4 = hl + 32 * i;
        if (k4 < K4) {
          const float4 v = *reinterpret_cast<const float4*>(rs + ((size_t)row * K4 + k4) * 16);   // second read: 12 registers less
          const f16x2k h01 = __builtin_convertvector(f32x2k{v.x * sc, v.y * sc}, f16x2k);
          const f16x2k h23 = __builtin_convertvector(f32x2k{v.z * sc, v.w * sc}, f16x2k);
          f16x2k l01, l23;
          l01[0] = (_Float16)__builtin_fmaf((float)h01[0], -2048.f, v.x * sc2k);
          l01[1] = (_Float16)__builtin_fmaf((float)h01[1], -2048.f, v.y * sc2k);
          l23[0] = (_Float16)__builtin_fmaf((float)h23[0], -2048.f, v.z * sc2k);
          l23[1] = (_Float16)__builtin_fmaf((float)h23[1], -2048.f, v.w * sc2k);
          u16* dst = xp + row * LDX + 4 * k4;
          *reinterpret_cast<u32x2k*>(dst) = u32x2k{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
          *reinterpret_cast<u32x2k*>(dst + KROWS * LDX) = u32x2k{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
        }
      }
      if (hl == 0) row_inv[pbuf * KROWS + row] = __uint_as_float(e);  // 2^e
    }
  };
  split(0, 0);
#ifdef EGC_GEMMK_STAMPS
  unsigned long long kt0, ksum[6] = {0, 0, 0, 0, 0, 0};
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(kt0) :: "memory");
#endif
  // One barrier per tile.  At the top of iteration t (tile t of this workgroup): ring slots (t + 1) % ring ... hold or await
  // the raw tiles t + 1 ... t + ring - 1, plane buffer t % 2 the split tile t.  After the barrier the slot t % ring is free
  // (everybody has split tile t) and is re-armed with tile t + ring at once; then the wavefronts split tile t + 1 into the
  // other plane buffer and multiply tile t, in whatever order they get there: the vector work of one wavefront runs beside
  // the matrix work of another.  The wait at the top is for this wavefront's own pieces of tile t + 1: behind them in the
  // (in-order) counter are the pieces of tiles t + 2 ... t + ring - 1 and the last tile's stores.
  int cur = 0, slot = 0;
  const int behind = (ring - 2) * (R + (add ? 1 : 0)) + n_stores;
  for (bool first = true; tile < n_tiles; tile += stride, first = false) {
    if (first) vmwait_k<0>(); else vmwait_rt(behind);
    KST(0)
    lds_barrier_k();
    KST(1)
    f32x4k av;                  // (no initial value: a second definition would be a copy the compiler may place ahead of the wait)
    asm volatile("" : "=v"(av));
    if (add) {
      const int64_t arow = (int64_t)tile * KROWS + j;
      const unsigned aoff = (arow < M && col0 + 3 < lim) ? (unsigned)((arow * out_ld + col0) * 4) : GOOB;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(av) : "v"(aoff), "s"(ra) : "memory");
    }
    const int nslot = slot + 1 == ring ? 0 : slot + 1;
    split(nslot, cur ^ 1);
    KST(3)
    dma_tile(tile + ring * stride, slot);
    slot = nslot;
    KST(2)
    {
      f32x4k acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const u16* xb = xs + cur * 2 * KROWS * LDX + j * LDX + 8 * quad;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const f16x8k xh = *reinterpret_cast<const f16x8k*>(xb + 32 * s);
        const f16x8k xl = *reinterpret_cast<const f16x8k*>(xb + KROWS * LDX + 32 * s);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[s][0], xh, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[s][0], xl, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[s][1], xh, acc1, 0, 0, 0);
      }
      const float ri = row_inv[cur * KROWS + j];
      f32x4k cinv, cbias;
      {
        const float* ci = colinfo + 2 * (16 * ct + 4 * quad);
        const f32x4k c01 = *reinterpret_cast<const f32x4k*>(ci), c23 = *reinterpret_cast<const f32x4k*>(ci + 4);
        cinv = f32x4k{c01[0], c01[2], c23[0], c23[2]};
        cbias = f32x4k{c01[1], c01[3], c23[1], c23[3]};
      }
      const int64_t grow = (int64_t)tile * KROWS + j;
      f32x4k o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = __builtin_fmaf(__builtin_fmaf(acc1[r], 1.f / 2048.f, acc0[r]), cinv[r] * ri, cbias[r]);
      // Every store instruction is always issued (masked lanes go out of the buffer's range): the counted wait at
      // the top of the loop relies on a fixed number of vector-memory operations per tile.
      const bool row_ok = grow < M;
      const unsigned off = (unsigned)((grow * out_ld + col0) * 4);
      if (add) {        // behind the addend's load in the counter: this iteration's R DMA requests
        vmwait_rt(R);
        asm volatile("" : "+v"(av));
        o += av;
      }
      if (vec_store) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4k, o), ro, (row_ok && col0 + 3 < lim) ? off : GOOB, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[r]), ro, (row_ok && col0 + r < lim) ? off + 4u * r : GOOB, 0, 0);
      }
    }
    KST(5)
    cur ^= 1;
  }
  vmwait_k<0>();  // no DMA may still be writing this block's LDS when it is handed to the next block
#ifdef EGC_GEMMK_STAMPS
  if (lane == 0)
    for (int k = 0; k < 6; ++k) atomicAdd(&egc_stampk[k], ksum[k]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// The same GEMM with the wavefronts' ROLES separated (round 4; the form the ogbn-mag layer runs on).  In the kernel above
// every wavefront requests its pieces of x, splits a row and multiplies: next to the 8 KS registers of its weight tile there
// is no register left to request the next k-step's operands ahead, so every k-step pays the LDS latency (98 cycles per MFMA
// measured at 352 -> 176 + 32, against 16 of matrix pipe), and the split, the DMA issue and the products of a wavefront are
// serial (6,400 cycles per 16-row tile and wavefront: 3,200 products, 1,050 split, 570 DMA issue, 1,100 barrier).  Here
// wavefronts [0, nmf) only multiply -- the operands of k-step s + 1 in flight during the products of k-step s -- and
// `nh` further wavefronts only move x: LDS-DMA requests `ring` tiles ahead, the fp16x2 split of tile t + 1 while tile t is
// being multiplied (their registers are free: each row's pieces are read once and kept).  Same numerics, same packed weights.
// ---------------------------------------------------------------------------------------------------------------------
// TPW (round 6): column tiles per multiplier wavefront.  Two (32 columns, 16 KS weight registers: K <= 224 at twelve wavefronts) let ONE
// launch cover the 17 - 20 column tiles of the reference's 224-wide forward GEMM (224 + 48 columns) and 296-wide d x GEMM, which
// sixteen single-tile wavefronts cannot: x is read once, the per-tile pipeline (a barrier, a split, the LDS round trips of the
// operands) runs once instead of once per launch, and both tiles' products share every operand read.  `ntl`: column tiles of
// this launch; the multipliers are wavefronts [0, nmf), nmf = ceil(ntl / TPW).  Per tile the same products in the same order.
template <int KS, int WAVES, int TPW = 1>
__global__ void __launch_bounds__(WAVES * 64) basis_gemm_f16x2k_spec_kernel(const float* __restrict__ x, const u16* __restrict__ packed,
                                                                           const float* __restrict__ bcat, int64_t M, int K,
                                                                           KCols c, float* __restrict__ bases,
                                                                           float* __restrict__ weightings, int n_tiles, int LDX,
                                                                           int R, int slot_bytes, int tile0, int ring, int nmf,
                                                                           const float* __restrict__ addend, int ntl) {
  extern __shared__ __attribute__((aligned(16))) char smem_k[];
  char* raw = smem_k;                                                    // [ring][slot_bytes] raw fp32 tiles (DMA ring)
  u16* xs = reinterpret_cast<u16*>(smem_k + ring * slot_bytes);          // [2 buffers][2 planes][KROWS][LDX] fp16
  float* row_inv = reinterpret_cast<float*>(xs + 4 * KROWS * LDX);       // [2 buffers][KROWS]
  float* colinfo = row_inv + 2 * KROWS;                                  // [16 NT][2]: inverse column scale, bias
  const int tid = threadIdx.x;
  const int nthreads = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_helper = wave >= nmf;
  const int K4 = K >> 2;                                                 // 16-byte pieces per row
  const int stride = gridDim.x;
  int tile = blockIdx.x;
  if (tile >= n_tiles) return;
  // zero the planes once: the k range [K, 32 KS) and the row padding are never written again
  for (int i = tid; i < 4 * KROWS * LDX / 8; i += nthreads) reinterpret_cast<u32x4k*>(xs)[i] = u32x4k{0, 0, 0, 0};
  {
    const float* inv_tab = reinterpret_cast<const float*>(packed + (int64_t)c.NT * KS * 2 * 64 * 8);
    for (int v = tid; v < 16 * c.NT; v += nthreads) {
      const int w = v - 16 * c.TB;
      colinfo[2 * v] = inv_tab[v];
      colinfo[2 * v + 1] = (w >= 0 && w < c.W && bcat != nullptr) ? bcat[w] : 0.f;
    }
  }

  if (is_helper) {
    // ================= the x movers =================
    const int ht = tid - nmf * 64;                 // thread index among the helpers
    const int hthreads = nthreads - nmf * 64;
    const int hwave = wave - nmf;
    constexpr unsigned GOOB = 0xFFFFFFF0u;
    const u32x4k rx = {(unsigned)(uintptr_t)x, (unsigned)((uintptr_t)x >> 32) & 0xffffu, (unsigned)(M * K * 4), 0x00020000u};
    const unsigned raw_lds = (unsigned)(uintptr_t)raw;
    const unsigned magic_K4 = (unsigned)(((uint64_t)1 << 32) / (uint64_t)K4) + 1u;
    // this thread's (up to 16) pieces of a tile: byte offset inside the tile, or out of range -- computed once (per tile and
    // piece the index arithmetic was a third of the helpers' time)
    unsigned poff[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int pc = ht + hthreads * i;
      const int row = (int)__umulhi((unsigned)pc, magic_K4);   // pc / K4 for pc < 2^16
      const int k4 = pc - row * K4;
      poff[i] = (i < R && row < KROWS) ? (unsigned)((row * K + 4 * k4) * 4) : GOOB;
    }
    const int64_t tile_bytes = (int64_t)KROWS * K * 4;
    auto dma_tile = [&](int tl, int slot) {
      // rows beyond M fall outside the buffer (its range check drops them); a tile beyond the last one is not requested
      const bool tile_ok = tl < n_tiles;
      const unsigned tbase = (unsigned)((int64_t)tl * tile_bytes);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i < R) {     // wave-uniform
          const unsigned voff = (tile_ok && poff[i] != GOOB) ? tbase + poff[i] : GOOB;
          const unsigned dst = __builtin_amdgcn_readfirstlane(raw_lds + slot * slot_bytes + (hwave * 64 + hthreads * i) * 16);
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep)
                       : "v"(voff), "s"(dst), "s"(rx)
                       : "memory");
        }
      }
    };
    const int hw = ht >> 5, hl = ht & 31, n_hw = hthreads >> 5;
    // raw fp32 rows of ring slot `rslot` -> the two fp16 planes of buffer `pbuf` + row scales; a half-wavefront per row,
    // the lane's (up to) three 16-byte pieces read once, together, and kept
    // (the rows of a half-wavefront -- up to four -- are processed TOGETHER: one row at a time is a chain of ~60 dependent
    // instructions on a wavefront that has its SIMD's vector unit almost to itself: 1,300 cycles per row measured)
    auto split = [&](int rslot, int pbuf) {
      const char* rs = raw + rslot * slot_bytes;
      u16* xp = xs + pbuf * 2 * KROWS * LDX;
      constexpr int RPH = 4;
      float4 v[RPH][3];
#pragma unroll
      for (int r = 0; r < RPH; ++r) {
        const int row = hw + r * n_hw;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int k4 = hl + 32 * i;
          // (unconditional reads of clamped addresses: a conditional LDS read becomes a branch with its own wait, and the
          // first form of this function spent its time in 112 branches; a clamped piece repeats one the maximum already holds)
          v[r][i] = *reinterpret_cast<const float4*>(rs + ((size_t)(row < KROWS ? row : KROWS - 1) * K4 + (k4 < K4 ? k4 : K4 - 1)) * 16);
        }
      }
      unsigned a[RPH];
#pragma unroll
      for (int r = 0; r < RPH; ++r) {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          float mi;
          asm("v_max3_f32 %0, |%1|, |%2|, |%3|\n\tv_max_f32 %0, |%4|, %0" : "=&v"(mi) : "v"(v[r][i].x), "v"(v[r][i].y), "v"(v[r][i].z), "v"(v[r][i].w));
          m = fmaxf(m, mi);
        }
        a[r] = __float_as_uint(m);
      }
#pragma unroll
      for (int r = 0; r < RPH; ++r) a[r] = max(a[r], (unsigned)__builtin_amdgcn_update_dpp(0, (int)a[r], 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
#pragma unroll
      for (int r = 0; r < RPH; ++r) a[r] = max(a[r], (unsigned)__builtin_amdgcn_update_dpp(0, (int)a[r], 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
#pragma unroll
      for (int r = 0; r < RPH; ++r) a[r] = max(a[r], (unsigned)__builtin_amdgcn_update_dpp(0, (int)a[r], 0x141, 0xf, 0xf, true));  // row_half_mirror
#pragma unroll
      for (int r = 0; r < RPH; ++r) a[r] = max(a[r], (unsigned)__builtin_amdgcn_update_dpp(0, (int)a[r], 0x140, 0xf, 0xf, true));  // row_mirror
#pragma unroll
      for (int r = 0; r < RPH; ++r) a[r] = max(a[r], (unsigned)__builtin_amdgcn_ds_swizzle((int)a[r], 0x401F));                    // lane ^ 16
#pragma unroll
      for (int r = 0; r < RPH; ++r) {
        const int row = hw + r * n_hw;
        if (row < KROWS) {     // uniform in the half-wavefront
          unsigned e = a[r] & 0x7f800000u;
          e = min(max(e, 13u << 23), 253u << 23);
          const float sc = __uint_as_float(0x7f000000u - e);                  // 2^-e
          const float sc2k = __uint_as_float(0x7f000000u + (11u << 23) - e);  // 2^(11-e)
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int k4 = hl + 32 * i;
            if (k4 < K4) {
              const float4 w = v[r][i];
              const f16x2k h01 = __builtin_convertvector(f32x2k{w.x * sc, w.y * sc}, f16x2k);
              const f16x2k h23 = __builtin_convertvector(f32x2k{w.z * sc, w.w * sc}, f16x2k);
              f16x2k l01, l23;
              l01[0] = (_Float16)__builtin_fmaf((float)h01[0], -2048.f, w.x * sc2k);
              l01[1] = (_Float16)__builtin_fmaf((float)h01[1], -2048.f, w.y * sc2k);
              l23[0] = (_Float16)__builtin_fmaf((float)h23[0], -2048.f, w.z * sc2k);
              l23[1] = (_Float16)__builtin_fmaf((float)h23[1], -2048.f, w.w * sc2k);
              u16* dst = xp + row * LDX + 4 * k4;
              *reinterpret_cast<u32x2k*>(dst) = u32x2k{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
              *reinterpret_cast<u32x2k*>(dst + KROWS * LDX) = u32x2k{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
            }
          }
          if (hl == 0) row_inv[pbuf * KROWS + row] = __uint_as_float(e);  // 2^e
        }
      }
    };
    for (int r = 0; r < ring; ++r) dma_tile(tile + r * stride, r);
    vmwait_k<0>();
    lds_barrier_k();           // every helper's pieces of the first tiles have landed; the planes are zeroed, colinfo written
    split(0, 0);
    // At the top of iteration t: this wavefront's pieces of tile t + 1 must have landed; behind them in the (in-order)
    // counter are only the pieces of tiles t + 2 ... t + ring - 1.
    int cur = 0, slot = 0;
    const int behind = (ring - 2) * R;
#ifdef EGC_GEMMK_STAMPS
    unsigned long long kt0, ksum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(kt0) :: "memory");
#endif
    for (; tile < n_tiles; tile += stride) {
      vmwait_rt(behind);
      KST(0)
      lds_barrier_k();         // tile t split (all helpers), tile t + 1 landed, the workers done with plane buffer cur ^ 1
      KST(1)
      const int nslot = slot + 1 == ring ? 0 : slot + 1;
      split(nslot, cur ^ 1);
      KST(2)
      dma_tile(tile + ring * stride, slot);     // (slot held tile t: split in the previous iteration, before this barrier)
      KST(3)
      slot = nslot;
      cur ^= 1;
    }
    vmwait_k<0>();  // no DMA may still be writing this block's LDS when it is handed to the next block
#ifdef EGC_GEMMK_STAMPS
    if (tid == nmf * 64) for (int k = 0; k < 4; ++k) atomicAdd(&egc_stampk[k], ksum[k]);
#endif
    return;
  }

  // ================= the multipliers =================
  const int j = lane & 15, quad = lane >> 4;
  constexpr unsigned GOOB = 0xFFFFFFF0u;
  int ct[TPW];
  bool live[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    live[u] = wave * TPW + u < ntl;                                    // (an odd tile count leaves the last wavefront's second tile idle:
    ct[u] = tile0 + (live[u] ? wave * TPW + u : wave * TPW);           //  it repeats its first tile and stores nothing)
  }
  f16x8k wf[TPW][KS][2];
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    const u16* src = packed + ((int64_t)ct[u] * KS * 2 * 64 + lane) * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      wf[u][s][0] = *reinterpret_cast<const f16x8k*>(src + (s * 2) * 64 * 8);
      wf[u][s][1] = *reinterpret_cast<const f16x8k*>(src + (s * 2 + 1) * 64 * 8);
    }
  }
  bool to_bases[TPW], vec_store[TPW], add[TPW];
  int out_ld[TPW], col0[TPW], lim[TPW];
  __amdgpu_buffer_rsrc_t ro[TPW], ra[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    to_bases[u] = ct[u] < c.TB;
    out_ld[u] = to_bases[u] ? c.ldb : c.W;
    col0[u] = to_bases[u] ? 16 * ct[u] + 4 * quad : 16 * (ct[u] - c.TB) + 4 * quad;
    lim[u] = live[u] ? (to_bases[u] ? c.ldb : c.W) : 0;                // (an idle tile: every column out of range)
    float* outp = to_bases[u] ? bases : weightings;
    ro[u] = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, (unsigned)(M * out_ld[u] * 4), 0x00020000);
    vec_store[u] = (out_ld[u] & 3) == 0;
    // `addend`: as in the kernel above; the multipliers issue no DMA, so the compiler's own count of this load is right
    add[u] = addend != nullptr && to_bases[u] && live[u];              // wave-uniform
    ra[u] = __builtin_amdgcn_make_buffer_rsrc((void*)(add[u] ? addend : outp), 0, add[u] ? (unsigned)(M * out_ld[u] * 4) : 0u, 0x00020000);
  }
  vmwait_k<0>();
#pragma unroll
  for (int u = 0; u < TPW; ++u)
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(wf[u][s][0]), "+v"(wf[u][s][1]));
  lds_barrier_k();
  int cur = 0;
#ifdef EGC_GEMMK_STAMPS
  unsigned long long kt0, ksum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rt0, rt1, ct0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(kt0) :: "memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0) :: "memory");
  ct0 = kt0;
#endif
  for (; tile < n_tiles; tile += stride) {
    lds_barrier_k();
    KST(4)
    f32x4k av[TPW], acc0[TPW], acc1[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      av[u] = acc0[u] = acc1[u] = f32x4k{0.f, 0.f, 0.f, 0.f};
      if (add[u]) {
        const int64_t arow = (int64_t)tile * KROWS + j;
        av[u] = __builtin_bit_cast(f32x4k, __builtin_amdgcn_raw_buffer_load_b128(
                                               ra[u], (arow < M && col0[u] + 3 < lim[u]) ? (unsigned)((arow * out_ld[u] + col0[u]) * 4) : GOOB, 0, 0));
      }
    }
    const u16* xb = xs + cur * 2 * KROWS * LDX + j * LDX + 8 * quad;
    // PF operand sets in rotation: the operands of k-step s + PF - 1 are requested before the products of k-step s, and the
    // order is pinned (left alone the scheduler folds the sets back into one and waits for a read two MFMAs after issuing it)
    // (two tiles: six products per operand pair, and the registers are the weights'; K > 320 at sixteen wavefronts: the third set
    // was paid for in spilled weight registers -- 12 / 20 at KS = 11 / 12 --, 570 -> 540 us at the ogbn-mag shape without it)
    constexpr int PF = (TPW == 1 && KS <= 10) ? 3 : 2;
    f16x8k xh[PF], xl[PF];
#pragma unroll
    for (int p = 0; p < PF - 1; ++p)
      if (p < KS) {
        xh[p] = *reinterpret_cast<const f16x8k*>(xb + 32 * p);
        xl[p] = *reinterpret_cast<const f16x8k*>(xb + KROWS * LDX + 32 * p);
      }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + PF - 1 < KS) {
        xh[(s + PF - 1) % PF] = *reinterpret_cast<const f16x8k*>(xb + 32 * (s + PF - 1));
        xl[(s + PF - 1) % PF] = *reinterpret_cast<const f16x8k*>(xb + KROWS * LDX + 32 * (s + PF - 1));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < TPW; ++u) acc0[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u][s][0], xh[s % PF], acc0[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < TPW; ++u) acc1[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u][s][0], xl[s % PF], acc1[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < TPW; ++u) acc1[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u][s][1], xh[s % PF], acc1[u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    KST(5)
    const float ri = row_inv[cur * KROWS + j];
    const int64_t grow = (int64_t)tile * KROWS + j;
    const bool row_ok = grow < M;
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      f32x4k cinv, cbias;
      {
        const float* ci = colinfo + 2 * (16 * ct[u] + 4 * quad);
        const f32x4k c01 = *reinterpret_cast<const f32x4k*>(ci), c23 = *reinterpret_cast<const f32x4k*>(ci + 4);
        cinv = f32x4k{c01[0], c01[2], c23[0], c23[2]};
        cbias = f32x4k{c01[1], c01[3], c23[1], c23[3]};
      }
      f32x4k o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = __builtin_fmaf(__builtin_fmaf(acc1[u][r], 1.f / 2048.f, acc0[u][r]), cinv[r] * ri, cbias[r]);
      const unsigned off = (unsigned)((grow * out_ld[u] + col0[u]) * 4);
      if (add[u]) o += av[u];
      if (vec_store[u]) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4k, o), ro[u], (row_ok && col0[u] + 3 < lim[u]) ? off : GOOB, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[r]), ro[u], (row_ok && col0[u] + r < lim[u]) ? off + 4u * r : GOOB, 0, 0);
      }
    }
    KST(6)
    cur ^= 1;
  }
#ifdef EGC_GEMMK_STAMPS
  if (tid == 0) for (int k = 4; k < 7; ++k) atomicAdd(&egc_stampk[k], ksum[k]);
  if (tid == 0 && blockIdx.x == 0) {
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1) :: "memory");
    egc_stampk[7] = ((kt0 - ct0) << 24) | ((rt1 - rt0) & 0xffffff);     // loop cycles | 100 MHz ticks of block 0
  }
#endif
}

bool f16x2k_shape(int f_in, int f_g, int ldb, int w_cols) {
  if (f_in <= 128 || f_in > 384 || (f_in & 3) != 0) return false;
  const int NT = (ldb + 15) / 16 + (w_cols + 15) / 16;
  const int per_launch = NT <= 16 ? NT : (NT + 1) / 2;   // two launches beyond 16 column tiles
  if ((f_in + 31) / 32 == 12 && per_launch > 12) return false;  // 96 weight registers do not fit four wavefronts per SIMD
  if (NT < 1 || NT > 32) return false;
  // launch_k's staging limits, on the narrower launch: a thread stages at most 16 pieces of 16 bytes of the x tile,
  // and the tile, its planes and the bias rows fit the LDS (a narrow GEMM over a long k -- the dx GEMM of a layer with
  // few input features -- has too few wavefronts to stage its tile: bf16x3 kernels then)
  const int narrow = NT <= 16 ? NT : NT / 2;
  const int KS = (f_in + 31) / 32, threads = narrow * 64;
  const int R = (KROWS * (f_in / 4) + threads - 1) / threads;
  const size_t lds = (size_t)2 * R * threads * 16 + (size_t)4 * KROWS * (32 * KS + 16) * sizeof(u16) + (2 * KROWS + 2 * 16 * NT) * sizeof(float);
  return R <= 16 && lds <= 160 * 1024;
}

static KCols kcols(int f_g, int ldb, int w_cols) {
  KCols c;
  c.F_g = f_g; c.ldb = ldb; c.W = w_cols;
  c.TB = (ldb + 15) / 16;
  c.NT = c.TB + (w_cols + 15) / 16;
  return c;
}

size_t f16x2k_pack_bytes(int f_in, int f_g, int ldb, int w_cols) {
  const KCols c = kcols(f_g, ldb, w_cols);
  const int KS = (f_in + 31) / 32;
  return (size_t)c.NT * KS * 2 * 64 * 8 * sizeof(u16) + (size_t)c.NT * 16 * sizeof(float);
}

int f16x2k_pack(const float* wcat, int64_t rs, int64_t cs, int f_in, int f_g, int ldb, int w_cols, void* packed,
                hipStream_t stream) {
  const KCols c = kcols(f_g, ldb, w_cols);
  const int KS = (f_in + 31) / 32;
  pack_f16x2k_kernel<<<c.NT * 16, 64, 0, stream>>>(wcat, rs, cs, f_in, c, KS, (u16*)packed);
  EGC_LAUNCH_CHECK("pack_f16x2k_kernel");
  return EGC_OK;
}

template <int KS, int WAVES>
static int launch_k(const float* x, const u16* packed, const float* bcat, int64_t M, int K, const KCols& c, float* bases,
                    float* weightings, hipStream_t stream, int tile0, int ntl, const float* addend) {
  const int64_t n_tiles64 = ceil_div(M, KROWS);
  if (n_tiles64 >= ((int64_t)1 << 31)) return EGC_ERR_INVALID;
  const int n_tiles = (int)n_tiles64;
  // row stride of the planes in fp16: the B fragments are ds_read_b128 of lanes (row j = lane & 15, k piece lane >> 4);
  // with 32 KS + 16 every one of the instruction's four 16-lane groups touches 16 distinct 4-bank slots
  // (32 KS + 8 leaves five 2-way conflicts per group)
  const int LDX = 32 * KS + 16;
  const size_t fixed = (size_t)4 * KROWS * LDX * sizeof(u16) + (2 * KROWS + 2 * 16 * c.NT) * sizeof(float);
  // ---- separated roles: `nh` helper wavefronts next to the ntl multipliers (this launch: column tiles tile0 .. tile0 + ntl - 1)
  const int nh = std::min(WAVES - ntl, 4);
  // (where two workgroups of the everything-in-every-wavefront kernel share a CU -- short k, few column tiles -- that form
  // stays: 52.7 against 62.3 us at 184 -> 96 + 32, N = 169,343)
  bool two_per_cu = false;
  {
    const int threads = ntl * 64;
    const int R0 = (int)ceil_div((int64_t)KROWS * (K / 4), threads);
    const size_t lds0 = (size_t)2 * R0 * threads * 16 + fixed;
    two_per_cu = KS <= 9 && std::min<size_t>((size_t)160 * 1024 / lds0, (size_t)(20 / ntl)) >= 2;
  }
  if (nh >= 2 && !two_per_cu) {
    const int hthreads = nh * 64;
    const int R = (int)ceil_div((int64_t)KROWS * (K / 4), hthreads);
    const int slot_bytes = R * hthreads * 16;
    int ring = (int)std::min<size_t>(4, ((size_t)160 * 1024 - fixed) / (size_t)slot_bytes);
    while (ring > 2 && (ring - 2) * R > 32) --ring;
    if (ring >= 2 && R <= 16) {
      const size_t lds = (size_t)ring * slot_bytes + fixed;
      auto kern = &basis_gemm_f16x2k_spec_kernel<KS, WAVES>;
      static bool attr_set = false;
      if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_last_error("hipFuncSetAttribute(f16x2k spec)", e); return EGC_ERR_HIP; }
        attr_set = true;
      }
      int grid = 256;
      if (grid > n_tiles) grid = n_tiles;
      kern<<<grid, (ntl + nh) * 64, lds, stream>>>(x, packed, bcat, M, K, c, bases, weightings, n_tiles, LDX, R, slot_bytes, tile0, ring,
                                                   ntl, addend, ntl);
      EGC_LAUNCH_CHECK("basis_gemm_f16x2k_spec_kernel");
#ifdef EGC_GEMMK_STAMPS
      {
        static int calls = 0;
        if (++calls == 10) {
          hipDeviceSynchronize();
          unsigned long long h[8];
          hipMemcpyFromSymbol(h, HIP_SYMBOL(egc_stampk), sizeof(h));
          const double per = (double)calls * n_tiles;   // one helper wavefront and one multiplier per workgroup report
          fprintf(stderr, "[gemmk spec stamps] K=%d ntl=%d nh=%d R=%d ring=%d per tile (cycles): helper: dma-wait %.0f barrier %.0f split %.0f dma-issue %.0f | multiplier: barrier %.0f products %.0f scale+store %.0f\n",
                  K, ntl, nh, R, ring, h[0] / per, h[1] / per, h[2] / per, h[3] / per, h[4] / per, h[5] / per, h[6] / per);
          fprintf(stderr, "[gemmk spec stamps] block 0 loop: %llu shader cycles in %.1f us (s_memrealtime) => %.2f GHz\n", h[7] >> 24, (h[7] & 0xffffff) * 0.01,
                  (double)(h[7] >> 24) / ((h[7] & 0xffffff) * 10.0));
        }
      }
#endif
      return EGC_OK;
    }
  }
  const int threads = ntl * 64;
  const int R = (int)ceil_div((int64_t)KROWS * (K / 4), threads);
  const int slot_bytes = R * threads * 16;
  size_t lds = (size_t)2 * slot_bytes + fixed;
  if (lds > 160 * 1024 || R > 16) return EGC_ERR_UNSUPPORTED;
  auto kern = &basis_gemm_f16x2k_kernel<KS, WAVES>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_last_error("hipFuncSetAttribute(f16x2k)", e); return EGC_ERR_HIP; }
    attr_set = true;
  }
  // workgroups per CU: as many as the LDS holds, within about five wavefronts per SIMD (the KS <= 9 kernels use up to
  // 102 registers).  Short k leaves room for two (F_in = 192, 8 column tiles: 59 KB of LDS each), and the second one's
  // matrix work covers the first one's barrier and split: 66.9 -> 50.1 us for the 192 -> 128 gradient GEMM at
  // N = 169,343 (a third workgroup that does not fit measured 57 us: uneven CUs).
  int per_cu = (int)std::min<size_t>((size_t)160 * 1024 / lds, (size_t)(20 / ntl));
  if (KS > 9 || per_cu < 1) per_cu = 1;
  int grid = 256 * per_cu;
  if (grid > n_tiles) grid = n_tiles;
  // depth of the raw-tile ring: what the LDS share of a workgroup holds next to the planes, at most 4 (and within the
  // counted wait's range); a workgroup that shares its CU keeps 2
  int ring = 2;
  if (per_cu == 1) {
    ring = (int)std::min<size_t>(4, ((size_t)160 * 1024 - fixed) / (size_t)slot_bytes);
    while (ring > 2 && ((ring - 2) * (R + 1) + 4 > 32 || (size_t)ring * slot_bytes + fixed > (size_t)160 * 1024)) --ring;
    if (ring < 2) ring = 2;
    lds = (size_t)ring * slot_bytes + fixed;
  }
  kern<<<grid, threads, lds, stream>>>(x, packed, bcat, M, K, c, bases, weightings, n_tiles, LDX, R, slot_bytes, tile0, ring, addend);
  EGC_LAUNCH_CHECK("basis_gemm_f16x2k_kernel");
  return EGC_OK;
}

// One launch over ALL column tiles with two tiles per multiplier wavefront (K <= 224: 16 KS weight registers fit twelve wavefronts):
// ceil(NT / 2) multipliers + the helpers that move and split x.  EGC_ERR_UNSUPPORTED where the tile ring does not fit.
template <int KS>
static int launch_k2(const float* x, const u16* packed, const float* bcat, int64_t M, int K, const KCols& c, float* bases,
                     float* weightings, hipStream_t stream, const float* addend) {
  constexpr int WAVES = 12;
  const int64_t n_tiles64 = ceil_div(M, KROWS);
  if (n_tiles64 >= ((int64_t)1 << 31)) return EGC_ERR_INVALID;
  const int n_tiles = (int)n_tiles64;
  const int LDX = 32 * KS + 16;
  const size_t fixed = (size_t)4 * KROWS * LDX * sizeof(u16) + (2 * KROWS + 2 * 16 * c.NT) * sizeof(float);
  const int nmf = (c.NT + 1) / 2, nh = WAVES - nmf;
  if (nh < 2) return EGC_ERR_UNSUPPORTED;
  const int hthreads = nh * 64;
  const int R = (int)ceil_div((int64_t)KROWS * (K / 4), hthreads);
  const int slot_bytes = R * hthreads * 16;
  int ring = (int)std::min<size_t>(4, ((size_t)160 * 1024 - fixed) / (size_t)slot_bytes);
  while (ring > 2 && (ring - 2) * R > 32) --ring;
  if (ring < 2 || R > 16) return EGC_ERR_UNSUPPORTED;
  const size_t lds = (size_t)ring * slot_bytes + fixed;
  auto kern = &basis_gemm_f16x2k_spec_kernel<KS, WAVES, 2>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_last_error("hipFuncSetAttribute(f16x2k spec, two tiles)", e); return EGC_ERR_HIP; }
    attr_set = true;
  }
  int grid = 256;
  if (grid > n_tiles) grid = n_tiles;
  kern<<<grid, WAVES * 64, lds, stream>>>(x, packed, bcat, M, K, c, bases, weightings, n_tiles, LDX, R, slot_bytes, 0, ring, nmf, addend, c.NT);
  EGC_LAUNCH_CHECK("basis_gemm_f16x2k_spec_kernel (two tiles per wavefront)");
  return EGC_OK;
}

template <int KS>
static int launch_ks(const float* x, const u16* packed, const float* bcat, int64_t M, int K, const KCols& c, float* bases,
                     float* weightings, hipStream_t stream, const float* addend) {
  if constexpr (KS <= 7) {
    // 17 - 20 column tiles (224 / H4 / B4 with three aggregators forward: 14 + 3; the d x GEMM of 296 / H8 / B4: 19): ONE launch.
    // From nine tiles on the two-tile form is also the faster single launch (half the operand reads per product, five to eight
    // multipliers and as many wavefronts left to move and split x: 3 - 20 % at 9 - 16 tiles and 169 k rows, same bits; at eight tiles
    // and fewer two workgroups of the all-in-one kernel per CU win: 60.6 against 67.0 us at 192 -> 128).
    if (c.NT >= 9 && c.NT <= 20) {
      const int st = launch_k2<KS>(x, packed, bcat, M, K, c, bases, weightings, stream, addend);
      if (st != EGC_ERR_UNSUPPORTED) return st;
    }
  }
  // more than 16 column tiles (e.g. 224/H4/B4 with three aggregators: 14 + 3; 300/H4/B4: 19 + 3; 304/H8/B8: 20 + 4): two
  // launches over half of the tiles each -- x is read twice, which still beats the LDS-staged bf16x3 kernel 2 x
  const int launches = c.NT <= 16 ? 1 : 2;
  for (int l = 0, t0 = 0; l < launches; ++l) {
    const int ntl = (c.NT - t0 + (launches - l) - 1) / (launches - l);
    // up to 9 column tiles: 12 wavefronts (168 registers), 3-4 of them helpers; more: 16 wavefronts (128 registers) with
    // 16 - ntl helpers (none at 16 tiles: the kernel in which every wavefront does everything)
    const int st = ntl <= 9 ? launch_k<KS, 12>(x, packed, bcat, M, K, c, bases, weightings, stream, t0, ntl, addend)
                            : launch_k<KS, 16>(x, packed, bcat, M, K, c, bases, weightings, stream, t0, ntl, addend);
    if (st != EGC_OK) return st;
    t0 += ntl;
  }
  return EGC_OK;
}

int f16x2k_launch(const float* x, const void* packed, const float* bcat, int64_t M, int K, int f_g, int ldb, int W,
                  float* bases, float* weightings, hipStream_t stream, const float* addend) {
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (reinterpret_cast<uintptr_t>(bases) & 15) != 0 ||
      (W > 0 && (W & 3) == 0 && (reinterpret_cast<uintptr_t>(weightings) & 15) != 0))
    return EGC_ERR_UNSUPPORTED;
  if (addend != nullptr && ((reinterpret_cast<uintptr_t>(addend) & 15) != 0 || (ldb & 3) != 0)) return EGC_ERR_UNSUPPORTED;
  const KCols c = kcols(f_g, ldb, W);
  const u16* pk = (const u16*)packed;
  const int64_t widest = std::max(std::max(K, ldb), W);
  int64_t max_rows = ((int64_t)0x7FFFFFF0 / (4 * widest)) & ~(int64_t)(KROWS - 1);
  if (const char* e = getenv("EGC_GEMM_MAX_ROWS")) max_rows = std::max<int64_t>(KROWS, atoll(e) & ~(int64_t)(KROWS - 1));  // tests
  for (int64_t r0 = 0; r0 < M; r0 += max_rows) {
    const int64_t rows = std::min(max_rows, M - r0);
    const float* xr = x + r0 * K;
    float* br = bases + r0 * ldb;
    float* wr = weightings != nullptr ? weightings + r0 * W : nullptr;
    const float* ar = addend != nullptr ? addend + r0 * ldb : nullptr;
    int st;
    switch ((K + 31) / 32) {
      case 5: st = launch_ks<5>(xr, pk, bcat, rows, K, c, br, wr, stream, ar); break;
      case 6: st = launch_ks<6>(xr, pk, bcat, rows, K, c, br, wr, stream, ar); break;
      case 7: st = launch_ks<7>(xr, pk, bcat, rows, K, c, br, wr, stream, ar); break;
      case 8: st = launch_ks<8>(xr, pk, bcat, rows, K, c, br, wr, stream, ar); break;
      case 9: st = launch_ks<9>(xr, pk, bcat, rows, K, c, br, wr, stream, ar); break;
      case 10: st = launch_ks<10>(xr, pk, bcat, rows, K, c, br, wr, stream, ar); break;
      case 11: st = launch_ks<11>(xr, pk, bcat, rows, K, c, br, wr, stream, ar); break;
      case 12: st = launch_ks<12>(xr, pk, bcat, rows, K, c, br, wr, stream, ar); break;
      default: return EGC_ERR_UNSUPPORTED;
    }
    if (st != EGC_OK) return st;
  }
  return EGC_OK;
}

}  // namespace egc
