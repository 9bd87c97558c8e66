// Basis transform + weightings Linear on the bf16 matrix cores with fp32-level accuracy (gfx950), for every
// shape the fp16x2 kernel (egc_gemm_f16x2.hip: the north-star layer) does not take.
//
//     [bases | weightings] = x[N,F_in] @ [bases_weight | comb.weight^T]  (+ comb.bias)
//
// Reference behaviour replaced: torch.matmul(x, bases_weight) (experiments/layers.py:97-101,
// optimized_layers.py:180) and comb_weights(x) (layers.py:110, optimized_layers.py:182).
//
// Why not the fp32 MFMA: v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate (157 TF), which makes this
// 8.3-GFLOP GEMM the largest item of the layer (~125 us); on bf16 MFMA (2.5 PF dense) it is bound by its
// 217 MB of HBM traffic instead.  Accuracy is kept by splitting BOTH operands into three bf16 planes,
//     x = xh + xm + xl,   w = wh + wm + wl     (each plane = round-to-nearest of the running remainder,
//                                               so the three planes carry ~24 significand bits)
// and accumulating the six products  xl*wh, xh*wl, xm*wm, xm*wh, xh*wm, xh*wh  (smallest first) in the
// MFMA's fp32 accumulators.  The dropped terms (xm*wl, xl*wm, xl*wl) are < 2^-23 relative to |x||w|, the
// same order as one fp32 rounding, so the result is within GEMM-reordering distance of the reference's
// fp32 GEMM (parity tests: <= 1e-5, measured ~2e-7).
//
// The weight planes are split once per parameter update by egc_basis_pack into the staging
// layout [k-step][plane][virtual column][32 k] (bf16), so the kernel copies them to LDS as 16-byte
// pieces.  The virtual column space is the one of egc_gemm.hip: [0,F_g) bases, [F_g,ldb) zero pad,
// [ldb,ldb+W) weightings, zero-padded to a multiple of 32.
//
// Tiling: 128 x 192 block tile (all columns of the north-star shape in one pass over x), K walked in
// steps of 32; 4 wavefronts stacked along M, each 32 rows x 6 column tiles = 6 accumulators of 32x32.
#include <stdlib.h>

#include <algorithm>

#include "egc_common.h"
#include <stdio.h>

#include "egc_gemm_split.h"

namespace egc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int XBM = 128;      // rows per block
constexpr int XBN = 192;      // virtual columns per block (6 MFMA tiles)
constexpr int XKT = 32;       // k per staging step
constexpr int XLD = 40;       // LDS row stride in bf16 (80 B: conflict-free ds_read_b128 of 16-byte k-runs)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding
// GLOBAL access of the wavefront (s_waitcnt vmcnt(0)): in a streaming kernel that drains the stores of the
// tile just finished and the prefetch of the next one at every barrier -- microseconds of HBM latency per
// tile.  The tiles exchanged between wavefronts live in LDS, so lgkmcnt(0) + s_barrier is all that is needed.
__device__ inline void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ inline u16 bf16_rn(float f) {
  const unsigned u = __float_as_uint(f);
  return (u16)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ inline float bf16_f(u16 h) { return __uint_as_float((unsigned)h << 16); }

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Two floats -> three packed bf16 pairs (h, m, l planes) with v_cvt_pk_bf16_f32 (round to nearest even):
// 9 VALU instructions per pair.  Element 0 sits in the low half of each packed word.
__device__ inline void split3_pk(f32x2 v, unsigned& h, unsigned& m, unsigned& l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 r1 = v - f32x2{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
  const f32x2 r2 = r1 - f32x2{__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

// f = h + m + l up to ~2^-25 |f|
__device__ inline void split3(float f, u16& h, u16& m, u16& l) {
  h = bf16_rn(f);
  const float r1 = f - bf16_f(h);
  m = bf16_rn(r1);
  l = bf16_rn(r1 - bf16_f(m));
}

// packed[ks][plane][v][32] : plane p of w[k = 32*ks + kk][source column of v]
__global__ void __launch_bounds__(256) pack_bf16x3_kernel(const float* __restrict__ wcat, int64_t rs, int64_t cs, int K, int F_g,
                                                          int W, int ldb, int NV, int KS, u16* __restrict__ packed) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (ks, v, kk)
  const int total = KS * NV * XKT;
  if (idx >= total) return;
  const int kk = idx % XKT;
  const int v = (idx / XKT) % NV;
  const int ks = idx / (XKT * NV);
  const int k = ks * XKT + kk;
  const int src = (v < F_g) ? v : ((v < ldb || v >= ldb + W) ? -1 : v - ldb + F_g);
  const float w = (src >= 0 && k < K) ? wcat[k * rs + src * cs] : 0.f;
  u16 h, m, l;
  split3(w, h, m, l);
  const int64_t base = ((int64_t)ks * 3 * NV + v) * XKT + kk;
  packed[base] = h;
  packed[base + (int64_t)NV * XKT] = m;
  packed[base + 2 * (int64_t)NV * XKT] = l;
}

#ifdef EGC_GEMM3_STAMPS
__device__ unsigned long long egc_stamp3[8];  // diagnostic build only: cycles per phase, summed over wavefronts
#define EGC_ST3(k) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); st3[k] += _t - st3_t0; st3_t0 = _t; }
#else
#define EGC_ST3(k)
#endif

// LDS operand read with an immediate offset (one address register for the whole tile loop)
constexpr unsigned A_PLANE = XBM * XLD * 2, B_TILE = 32 * XLD * 2, SUB = 16 * 2;
template <unsigned OFF>
__device__ inline void lds_rd(bf16x8& dst, unsigned addr) {
  static_assert(OFF < 65536, "ds_read offset field");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// Step IDX = (substep IDX / NT, tile IDX % NT) of a block of NT column tiles: issue the B reads of step IDX + 1, wait
// (counted) for this step's operands, six MFMAs.
template <int IDX, int NT, int NACC>
__device__ inline void mfma_step(f32x16 (&acc)[NACC], bf16x8 (&a3)[2][3], bf16x8 (&bq)[2][3], unsigned b_lds) {
  constexpr int s = IDX / NT, t = IDX % NT, cur = IDX & 1, nxt = (IDX + 1) & 1;
  constexpr unsigned B_PLANE = (NT == 7 ? 224 : XBN) * XLD * 2;  // the 7-tile block keeps 224 columns of B in LDS
  if constexpr (IDX + 1 < 2 * NT) {
    constexpr unsigned off = ((IDX + 1) % NT) * B_TILE + ((IDX + 1) / NT) * SUB;
    lds_rd<off>(bq[nxt][0], b_lds);
    lds_rd<off + B_PLANE>(bq[nxt][1], b_lds);
    lds_rd<off + 2 * B_PLANE>(bq[nxt][2], b_lds);
  }
  if constexpr (IDX == 0)
    asm volatile("s_waitcnt lgkmcnt(3)"
                 : "+v"(a3[0][0]), "+v"(a3[0][1]), "+v"(a3[0][2]), "+v"(a3[1][0]), "+v"(a3[1][1]), "+v"(a3[1][2]),
                   "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[0][2]));
  else if constexpr (IDX + 1 < 2 * NT)
    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(bq[cur][0]), "+v"(bq[cur][1]), "+v"(bq[cur][2]));
  else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[cur][0]), "+v"(bq[cur][1]), "+v"(bq[cur][2]));
  const bf16x8 ah = a3[s][0], am = a3[s][1], al = a3[s][2];
  const bf16x8 bh = bq[cur][0], bm = bq[cur][1], bl = bq[cur][2];
  f32x16 c = acc[t];
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
  acc[t] = c;
  __builtin_amdgcn_sched_barrier(0);
}

// NT > 0: every block of the launch spans NT column tiles (6 = the full XBN columns, 4 = a 128-column remainder or
// a 128-column GEMM) -> the pipelined straight-line MFMA loop; NT = 0: the block's tile count is a run-time value.
template <bool A_VEC4, int NT>
__global__ void __launch_bounds__(256) basis_gemm_bf16x3_kernel(const float* __restrict__ x, const u16* __restrict__ packed,
                                                                const float* __restrict__ bcat, int64_t M, int K,
                                                                int W, float* __restrict__ bases, int ldb,
                                                                float* __restrict__ weightings, int NV, int KS,
                                                                int vblock0) {
  // one LDS allocation, carved explicitly (the epilogue re-uses it as the transpose buffer)
  // NT = 7: ONE block of 224 columns (1 block per CU: 84 KB of LDS) instead of a 192-column pass plus a second
  // pass over x for the last 32 -- the ogbn-mag layers (176 bases + 32 weightings columns).
  constexpr int BN = NT == 7 ? 224 : XBN, NACC = NT == 7 ? 7 : 6, WJ = NT == 7 ? 4 : 3;
  constexpr unsigned B_PLANE = BN * XLD * 2;
  constexpr int A_ELEMS = 3 * XBM * XLD, B_ELEMS = 3 * BN * XLD;
  __shared__ __attribute__((aligned(16))) u16 lds_raw[A_ELEMS + B_ELEMS];
  u16 (*As)[XBM][XLD] = reinterpret_cast<u16 (*)[XBM][XLD]>(lds_raw);
  u16 (*Bs)[BN][XLD] = reinterpret_cast<u16 (*)[BN][XLD]>(lds_raw + A_ELEMS);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * XBM;
  const int v0 = (blockIdx.y + vblock0) * XBN;
  const int nvb = NT > 0 ? 32 * NT : min(XBN, NV - v0);  // virtual columns of this block (multiple of 32)
  static_assert(NT == 0 || NT == 4 || NT == 6 || NT == 7, "pipelined MFMA loop: 4, 6 or 7 column tiles");
  const int ntile = nvb >> 5;

  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // x tile of one k-step: 4 x float4 per thread, loaded unconditionally (clamped address, masked after)
  // so the four loads are in flight together; the tile of step ks+1 is fetched before the MFMAs of step ks.
  float4 xv[4];
  auto load_x = [&](int ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const int k4 = (tid & 7) * 4;
      const int64_t gm = m0 + row;
      const bool ok = gm < M && ks * XKT + k4 < K;
      const float4 v = *reinterpret_cast<const float4*>(ok ? x + gm * K + ks * XKT + k4 : x);
      xv[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // packed weight planes of one k-step: 3 x [nvb x 32] bf16 as 16-byte pieces, nine per thread, all issued
  // back to back (clamped index instead of a guard)
  const int pieces = nvb * 4;  // 16-byte pieces per plane (<= 256 WJ)
  u32x4 wreg[3][WJ];
  auto load_w = [&](int ks) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const u16* src = packed + (((int64_t)ks * 3 + p) * NV + v0) * XKT;
#pragma unroll
      for (int j = 0; j < WJ; ++j) {
        const int i = min(tid + 256 * j, pieces - 1);
        wreg[p][j] = *reinterpret_cast<const u32x4*>(src + (int64_t)i * 8);
      }
    }
  };
  if (A_VEC4) load_x(0);
  load_w(0);
#ifdef EGC_GEMM3_STAMPS
  unsigned long long st3[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st3_t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st3_t0) :: "memory");
#endif

  for (int ks = 0; ks < KS; ++ks) {
    const int k0 = ks * XKT;
#ifdef EGC_GEMM3_STAMPS
    if (A_VEC4) { asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); }  // x tile landed (the 9 weight pieces may still fly)
#endif
    EGC_ST3(0)
    // ---- stage x[m0 .. m0+128, k0 .. k0+32) as three bf16 planes
    if (A_VEC4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        const int k4 = (tid & 7) * 4;
        unsigned h0, m0_, l0, h1, m1, l1;
        split3_pk(f32x2{xv[i].x, xv[i].y}, h0, m0_, l0);
        split3_pk(f32x2{xv[i].z, xv[i].w}, h1, m1, l1);
        *reinterpret_cast<u32x2*>(&As[0][row][k4]) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(&As[1][row][k4]) = u32x2{m0_, m1};
        *reinterpret_cast<u32x2*>(&As[2][row][k4]) = u32x2{l0, l1};
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx >> 5;
        const int kk = idx & 31;
        const int64_t gm = m0 + row;
        const bool ok = gm < M && k0 + kk < K;
        const float val = ok ? x[gm * K + k0 + kk] : 0.f;
        u16 h, m, l;
        split3(val, h, m, l);
        As[0][row][kk] = h;
        As[1][row][kk] = m;
        As[2][row][kk] = l;
      }
    }
    EGC_ST3(1)
#ifdef EGC_GEMM3_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    EGC_ST3(2)
    // ---- weight planes of this k-step (fetched during the previous step's MFMAs) -> LDS
    {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
          const int i = tid + 256 * j;
          if (i < pieces) *reinterpret_cast<u32x4*>(&Bs[p][i >> 2][(i & 3) * 8]) = wreg[p][j];
        }
    }
    EGC_ST3(3)
    lds_barrier();
    EGC_ST3(4)
    if (ks + 1 < KS) {  // next step's operands: in flight during the MFMAs below
      if (A_VEC4) load_x(ks + 1);
      load_w(ks + 1);
    }
    // ---- 2 MFMA k-substeps of 16 x up to 6 column tiles
    const int arow = 32 * wave + (lane & 31);
    const int koff = 8 * (lane >> 5);
    if constexpr (NT > 0) {
      // Full-width block: straight-line code in which the B operand of the NEXT (substep, tile) is read from LDS
      // before the six MFMAs of the current one are issued.  Reads and counted waits are inline assembly: the
      // compiler sinks its own ds_reads next to their first use and, when pinned, still waits for lgkmcnt(0) --
      // either way every tile sits out an LDS round trip.  LDS operations complete in order, so "at most 3
      // outstanding" after issuing the next tile's three reads means everything older has landed.
      bf16x8 a3[2][3], bq[2][3];
      const unsigned a_lds = (unsigned)(uintptr_t)&As[0][arow][koff];        // LDS byte addresses: plane 0, substep 0
      const unsigned b_lds = (unsigned)(uintptr_t)&Bs[0][lane & 31][koff];   // ... and tile 0
      lds_rd<0>(a3[0][0], a_lds); lds_rd<A_PLANE>(a3[0][1], a_lds); lds_rd<2 * A_PLANE>(a3[0][2], a_lds);
      lds_rd<SUB>(a3[1][0], a_lds); lds_rd<A_PLANE + SUB>(a3[1][1], a_lds); lds_rd<2 * A_PLANE + SUB>(a3[1][2], a_lds);
      lds_rd<0>(bq[0][0], b_lds); lds_rd<B_PLANE>(bq[0][1], b_lds); lds_rd<2 * B_PLANE>(bq[0][2], b_lds);
      mfma_step<0, NT, NACC>(acc, a3, bq, b_lds);
      mfma_step<1, NT, NACC>(acc, a3, bq, b_lds);
      mfma_step<2, NT, NACC>(acc, a3, bq, b_lds);
      mfma_step<3, NT, NACC>(acc, a3, bq, b_lds);
      mfma_step<4, NT, NACC>(acc, a3, bq, b_lds);
      mfma_step<5, NT, NACC>(acc, a3, bq, b_lds);
      mfma_step<6, NT, NACC>(acc, a3, bq, b_lds);
      mfma_step<7, NT, NACC>(acc, a3, bq, b_lds);
      if constexpr (NT >= 6) {
        mfma_step<8, NT, NACC>(acc, a3, bq, b_lds);
        mfma_step<9, NT, NACC>(acc, a3, bq, b_lds);
        mfma_step<10, NT, NACC>(acc, a3, bq, b_lds);
        mfma_step<11, NT, NACC>(acc, a3, bq, b_lds);
      }
      if constexpr (NT == 7) {
        mfma_step<12, NT, NACC>(acc, a3, bq, b_lds);
        mfma_step<13, NT, NACC>(acc, a3, bq, b_lds);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&As[0][arow][16 * s + koff]);
        const bf16x8 am = *reinterpret_cast<const bf16x8*>(&As[1][arow][16 * s + koff]);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(&As[2][arow][16 * s + koff]);
#pragma unroll
        for (int t = 0; t < NACC; ++t) {
          if (t < ntile) {  // block-uniform
            const int bcol = 32 * t + (lane & 31);
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bs[0][bcol][16 * s + koff]);
            const bf16x8 bm = *reinterpret_cast<const bf16x8*>(&Bs[1][bcol][16 * s + koff]);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bs[2][bcol][16 * s + koff]);
            f32x16 c = acc[t];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
            acc[t] = c;
          }
        }
      }
    }
    EGC_ST3(5)
    lds_barrier();
    EGC_ST3(6)
  }
#ifdef EGC_GEMM3_STAMPS
  if (lane == 0)
    for (int k = 0; k < 7; ++k) atomicAdd(&egc_stamp3[k], st3[k]);
#endif

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
  const bool wide = (ntile >= 6) && (W % 4 == 0) && (bcat == nullptr || (reinterpret_cast<uintptr_t>(bcat) & 15) == 0);
  if (wide) {
    // Transpose the wavefront's 32 x 192 tile through the (now idle) LDS in two halves of 96 columns and
    // write it as 16-byte pieces of contiguous rows: 24 dwordx4 stores per lane instead of 96 dword stores.
    constexpr int CLD = 100;  // floats per staged row (96 + 4: rows land on different banks)
    float* cs = reinterpret_cast<float*>(lds_raw) + wave * (32 * CLD);  // 12.8 KB per wavefront
    static_assert(4 * 32 * CLD * 4 <= (A_ELEMS + B_ELEMS) * 2, "epilogue staging must fit the operand LDS");
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) {
        const int t = 3 * half + tt;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          cs[row * CLD + 32 * tt + (lane & 31)] = acc[t][r];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // per-wavefront region: LDS ops of a wave complete in order
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const int idx = lane + 64 * j;   // 32 rows x 24 pieces
        const int row = idx / 24;
        const int c4 = (idx - row * 24) * 4;
        const int64_t gm = m0 + 32 * wave + row;
        const int vc = v0 + 96 * half + c4;
        float4 val = *reinterpret_cast<const float4*>(cs + row * CLD + c4);
        if (gm < M && vc < ldb + W) {
          if (vc < ldb) {
            *reinterpret_cast<float4*>(bases + gm * ldb + vc) = val;
          } else {
            if (bcat != nullptr) {
              const float4 bb = *reinterpret_cast<const float4*>(bcat + (vc - ldb));
              val.x += bb.x; val.y += bb.y; val.z += bb.z; val.w += bb.w;
            }
            *reinterpret_cast<float4*>(weightings + gm * (int64_t)W + (vc - ldb)) = val;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (NACC == 6) return;  // a 7-tile block: the last tile goes out element by element below
  }
#pragma unroll
  for (int t = 0; t < NACC; ++t) {
    if (t >= ntile) break;
    if (wide && t < 6) continue;
    const int vc = v0 + 32 * t + (lane & 31);
    if (vc >= ldb + W) continue;
    const bool to_bases = vc < ldb;
    const float badd = (!to_bases && bcat != nullptr) ? bcat[vc - ldb] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int64_t gm = m0 + 32 * wave + row;
      if (gm >= M) continue;
      if (to_bases) bases[gm * ldb + vc] = acc[t][r];
      else weightings[gm * (int64_t)W + (vc - ldb)] = acc[t][r] + badd;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Weight-stationary form for F_in <= 128 (the common case: hidden sizes up to 128).
//
// C^T = W^T x^T: wavefront w of a block owns virtual columns [32w, 32w+32) and keeps that weight tile --
// all three bf16 planes, every k -- in registers as the MFMA A operand for the whole kernel.  The block
// streams 32-row tiles of x (persistent loop over tiles), splits each into bf16 planes once, and shares it
// through double-buffered LDS as the B operand of all its wavefronts: one barrier per tile, weights never
// re-staged, global loads of tile t+1 in flight during the MFMAs of tile t.  In the transposed result a
// lane holds, for ONE x row, columns {8j + 4*(lane>>5) + 0..3}: four 16-byte stores per tile, no LDS
// transpose; comb.bias is preloaded into the accumulators.
// ---------------------------------------------------------------------------------------------
constexpr int WS_ROWS = 32;

template <int KSUB>  // number of 16-k MFMA sub-steps kept in registers: F_in <= 16 * KSUB
__global__ void __launch_bounds__(512) basis_gemm_ws_kernel(const float* __restrict__ x, const u16* __restrict__ packed,
                                                             const float* __restrict__ bcat, int64_t M, int K, int W,
                                                             float* __restrict__ bases, int ldb,
                                                             float* __restrict__ weightings, int NV, int n_tiles,
                                                             int x_vec4) {
  constexpr int KP = 16 * KSUB;  // padded K held per row
  constexpr int LDX = KP + 8;    // bf16 per staged x row (+16 B: conflict-free ds_read_b128)
  extern __shared__ __attribute__((aligned(16))) u16 xs[];  // [2 buffers][3 planes][32 rows][LDX]
  const int tid = threadIdx.x;
  const int nthreads = blockDim.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int hh = lane >> 5;
  const int cb = 32 * wave;  // first virtual column of this wavefront
  const bool bias_vec4 = bcat == nullptr || (reinterpret_cast<uintptr_t>(bcat) & 15) == 0;

  // ---- this wavefront's weight tile -> registers (A operand: lane holds W[k = 16s + 8hh + j][cb + (lane&31)])
  bf16x8 wf[KSUB][3];
#pragma unroll
  for (int s = 0; s < KSUB; ++s)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const u16* src = packed + ((((int64_t)(s >> 1) * 3 + p) * NV + cb + (lane & 31)) * XKT + 16 * (s & 1) + 8 * hh);
      wf[s][p] = *reinterpret_cast<const bf16x8*>(src);
    }
  // ---- x tile staging: 32 rows x KP floats = 8 * KP float4 pieces, spread over the block
  constexpr int PIECES = WS_ROWS * KP / 4;
  constexpr int PPT = 4;  // pieces per thread (needs nthreads * PPT >= PIECES; host guarantees)
  auto load_tile = [&](int tile, float4 (&xr)[PPT]) {
    const int64_t m0 = (int64_t)tile * WS_ROWS;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int pc = tid + nthreads * i;
      const int row = pc / (KP / 4);
      const int k4 = (pc - row * (KP / 4)) * 4;
      const int64_t gm = m0 + row;
      const bool ok = tile < n_tiles && pc < PIECES && gm < M && k4 < K;
      if (x_vec4) {
        const float4 v = *reinterpret_cast<const float4*>(ok ? x + gm * K + k4 : x);
        xr[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
          const float* px = x + gm * K + k4;
          v.x = px[0];
          if (k4 + 1 < K) v.y = px[1];
          if (k4 + 2 < K) v.z = px[2];
          if (k4 + 3 < K) v.w = px[3];
        }
        xr[i] = v;
      }
    }
  };
  auto stage_tile = [&](int buf, const float4 (&xr)[PPT]) {
    u16* base = xs + buf * (3 * WS_ROWS * LDX);
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int pc = tid + nthreads * i;
      if (pc < PIECES) {
        const int row = pc / (KP / 4);
        const int k4 = (pc - row * (KP / 4)) * 4;
        unsigned h0, m0_, l0, h1, m1, l1;
        split3_pk(f32x2{xr[i].x, xr[i].y}, h0, m0_, l0);
        split3_pk(f32x2{xr[i].z, xr[i].w}, h1, m1, l1);
        u16* dst = base + row * LDX + k4;
        *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(dst + WS_ROWS * LDX) = u32x2{m0_, m1};
        *reinterpret_cast<u32x2*>(dst + 2 * WS_ROWS * LDX) = u32x2{l0, l1};
      }
    }
  };
  // MFMAs + stores of the tile staged in LDS buffer `buf`
  auto compute_tile = [&](int tile, int buf) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const u16* xb = xs + buf * (3 * WS_ROWS * LDX) + (lane & 31) * LDX + 8 * hh;
#pragma unroll
    for (int s = 0; s < KSUB; ++s) {
      const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xb + 16 * s);
      const bf16x8 xm = *reinterpret_cast<const bf16x8*>(xb + 16 * s + WS_ROWS * LDX);
      const bf16x8 xl = *reinterpret_cast<const bf16x8*>(xb + 16 * s + 2 * WS_ROWS * LDX);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][0], xl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][2], xh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][1], xm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][0], xm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][1], xh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][0], xh, acc, 0, 0, 0);
    }
    // lane (row i = lane&31, half hh) owns columns cb + 8j + 4hh + 0..3, j = 0..3
    const int64_t gm = (int64_t)tile * WS_ROWS + (lane & 31);
    if (gm < M) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int vc = cb + 8 * j + 4 * hh;
        const float4 val = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
        if (vc < ldb) {
          *reinterpret_cast<float4*>(bases + gm * ldb + vc) = val;
        } else if (vc + 3 < ldb + W && (W & 3) == 0 && bias_vec4) {
          float4 o = val;
          if (bcat != nullptr) {  // comb.bias, 16 bytes from L1 (row-invariant)
            const float4 bb = *reinterpret_cast<const float4*>(bcat + (vc - ldb));
            o.x += bb.x; o.y += bb.y; o.z += bb.z; o.w += bb.w;
          }
          *reinterpret_cast<float4*>(weightings + gm * (int64_t)W + (vc - ldb)) = o;
        } else {
          float* wrow = weightings + gm * (int64_t)W;
          const float v4[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (vc + e < ldb + W) wrow[vc + e - ldb] = v4[e] + (bcat != nullptr ? bcat[vc + e - ldb] : 0.f);
        }
      }
    }
  };

  // Persistent loop: while tile i is multiplied out of one LDS buffer, tile i+1 is in flight / being split
  // into the other; two blocks per CU (3 wavefronts on every SIMD) cover each other's barriers and waits.
  const int stride = gridDim.x;
  int tile = blockIdx.x;
  if (tile >= n_tiles) return;
  float4 xr[PPT];
  load_tile(tile, xr);
  stage_tile(0, xr);
  lds_barrier();
  int buf = 0;
  for (; tile < n_tiles; tile += stride) {
    load_tile(tile + stride, xr);  // masked past the end
    compute_tile(tile, buf);
    if (tile + stride < n_tiles) stage_tile(buf ^ 1, xr);
    lds_barrier();
    buf ^= 1;
  }
}

template <int KSUB>
static int launch_ws(const float* x, const u16* packed, const float* bcat, int64_t M, int K, int W, float* bases, int ldb,
                     float* weightings, int NV, hipStream_t stream) {
  const int nt = NV / 32;               // wavefronts per block (<= 16)
  const int threads = 64 * nt;
  const int pieces = WS_ROWS * (16 * KSUB) / 4;
  if (threads * 4 < pieces) return EGC_ERR_UNSUPPORTED;
  const int64_t n_tiles64 = ceil_div(M, WS_ROWS);
  if (n_tiles64 >= ((int64_t)1 << 31)) return EGC_ERR_INVALID;
  const int n_tiles = (int)n_tiles64;
  const size_t lds = (size_t)2 * 3 * WS_ROWS * (16 * KSUB + 8) * sizeof(u16);
  // persistent grid: enough blocks to fill the chip a few times over, each walks tiles with stride gridDim
  int blocks_per_cu = nt <= 4 ? 4 : (nt <= 8 ? 2 : 1);
  int grid = 256 * blocks_per_cu;
  if (grid > n_tiles) grid = n_tiles;
  const int x_vec4 = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  basis_gemm_ws_kernel<KSUB><<<grid, threads, lds, stream>>>(x, packed, bcat, M, K, W, bases, ldb, weightings, NV, n_tiles,
                                                             x_vec4);
  EGC_LAUNCH_CHECK("basis_gemm_ws_kernel");
  return EGC_OK;
}

static inline int round_up32(int v) { return (v + 31) & ~31; }

}  // namespace egc

using namespace egc;

extern "C" {

size_t egc_basis_pack_bytes(int32_t f_in, int32_t f_g, int32_t w_cols) {
  if (f_in <= 0 || f_g <= 0 || w_cols < 0) return 0;
  const int ldb = (f_g + 3) & ~3;
  const int NV = round_up32(ldb + w_cols);
  const int KS = (f_in + XKT - 1) / XKT;
  return std::max(std::max((size_t)KS * 3 * NV * XKT * sizeof(u16), f16x2_pack_bytes(KS, NV)),
                  f16x2k_pack_bytes(f_in, f_g, ldb, w_cols));
}

// flags & EGC_GEMM_24BIT: operands split into THREE bf16 planes (24 significand bits: nothing of an fp32 operand is
// dropped) whatever the shape -- the fp16x2 forms keep 22 bits, which layers with std / var amplify (egc_hip.h)
static bool use_f16x2(int f_in, int ldb, int NV, int flags, int w_cols) {
  return (flags & EGC_GEMM_24BIT) == 0 && f16x2_shape(f_in, ldb, NV, w_cols);
}
static bool use_f16x2k(int f_in, int f_g, int ldb, int w_cols, int flags) {
  return (flags & EGC_GEMM_24BIT) == 0 && f16x2k_shape(f_in, f_g, ldb, w_cols);
}

static int basis_pack_strided(const float* wcat, int64_t rs, int64_t cs, int32_t f_in, int32_t f_g, int32_t w_cols,
                              void* packed, size_t packed_bytes, hipStream_t stream, int flags = 0) {
  if (wcat == nullptr || packed == nullptr || f_in <= 0 || f_g <= 0 || w_cols < 0) return EGC_ERR_INVALID;
  if (packed_bytes < egc_basis_pack_bytes(f_in, f_g, w_cols)) return EGC_ERR_WORKSPACE;
  const int ldb = (f_g + 3) & ~3;
  const int NV = round_up32(ldb + w_cols);
  const int KS = (f_in + XKT - 1) / XKT;
  if (use_f16x2(f_in, ldb, NV, flags, w_cols)) return f16x2_pack(wcat, rs, cs, f_in, f_g, w_cols, ldb, NV, KS, packed, stream);
  if (use_f16x2k(f_in, f_g, ldb, w_cols, flags)) return f16x2k_pack(wcat, rs, cs, f_in, f_g, ldb, w_cols, packed, stream);
  const int total = KS * NV * XKT;
  pack_bf16x3_kernel<<<(total + 255) / 256, 256, 0, stream>>>(wcat, rs, cs, f_in, f_g, w_cols, ldb, NV, KS, (u16*)packed);
  EGC_LAUNCH_CHECK("pack_bf16x3_kernel");
  return EGC_OK;
}

int egc_basis_pack(const float* wcat, int32_t f_in, int32_t f_g, int32_t w_cols, void* packed, size_t packed_bytes,
                   egc_stream_t stream_) {
  return basis_pack_strided(wcat, (int64_t)f_g + w_cols, 1, f_in, f_g, w_cols, packed, packed_bytes, (hipStream_t)stream_);
}

int egc_basis_pack_ex(const float* wcat, int32_t f_in, int32_t f_g, int32_t w_cols, int32_t flags, void* packed,
                      size_t packed_bytes, egc_stream_t stream_) {
  return basis_pack_strided(wcat, (int64_t)f_g + w_cols, 1, f_in, f_g, w_cols, packed, packed_bytes, (hipStream_t)stream_, flags);
}

int egc_basis_pack_transposed(const float* wt, int64_t ld, int32_t f_in, int32_t f_g, int32_t w_cols, void* packed,
                              size_t packed_bytes, egc_stream_t stream_) {
  if (ld < f_in) return EGC_ERR_INVALID;
  return basis_pack_strided(wt, 1, ld, f_in, f_g, w_cols, packed, packed_bytes, (hipStream_t)stream_);
}

int egc_basis_transform_packed(const float* x, const void* packed, const float* bcat, int64_t n_nodes, int32_t f_in,
                               int32_t f_g, int32_t w_cols, float* bases, int32_t ldb, float* weightings,
                               egc_stream_t stream_) {
  return egc_basis_transform_packed_ex(x, packed, bcat, n_nodes, f_in, f_g, w_cols, 0, bases, ldb, weightings, stream_);
}

int egc_basis_transform_packed_add(const float* x, const void* packed, const float* bcat, int64_t n_nodes, int32_t f_in,
                                   int32_t f_g, int32_t w_cols, int32_t flags, const float* bases_addend, float* bases, int32_t ldb,
                                   float* weightings, egc_stream_t stream_) {
  if (bases_addend == nullptr) return egc_basis_transform_packed_ex(x, packed, bcat, n_nodes, f_in, f_g, w_cols, flags, bases, ldb, weightings, stream_);
  if (n_nodes < 0 || f_in <= 0 || f_g <= 0 || w_cols < 0 || ldb != ((f_g + 3) & ~3)) return EGC_ERR_INVALID;
  if (n_nodes == 0) return EGC_OK;
  if (x == nullptr || packed == nullptr || bases == nullptr || (w_cols > 0 && weightings == nullptr)) return EGC_ERR_INVALID;
  if (!use_f16x2k(f_in, f_g, ldb, w_cols, flags)) return EGC_ERR_UNSUPPORTED;   // only the long-k kernels carry the addend
  return f16x2k_launch(x, packed, bcat, n_nodes, f_in, f_g, ldb, w_cols, bases, weightings, (hipStream_t)stream_, bases_addend);
}

int egc_basis_transform_packed_ex(const float* x, const void* packed, const float* bcat, int64_t n_nodes, int32_t f_in,
                                  int32_t f_g, int32_t w_cols, int32_t flags, float* bases, int32_t ldb, float* weightings,
                                  egc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_nodes < 0 || f_in <= 0 || f_g <= 0 || w_cols < 0 || ldb != ((f_g + 3) & ~3)) return EGC_ERR_INVALID;
  if (n_nodes == 0) return EGC_OK;
  if (x == nullptr || packed == nullptr || bases == nullptr || (w_cols > 0 && weightings == nullptr)) return EGC_ERR_INVALID;
  const int NV = round_up32(ldb + w_cols);
  const int KS = (f_in + XKT - 1) / XKT;
  if (use_f16x2(f_in, ldb, NV, flags, w_cols))  // the planes were packed for this kernel: no other form can read them
    return f16x2_launch(x, packed, bcat, n_nodes, f_in, w_cols, bases, ldb, weightings, NV, stream);
  if (use_f16x2k(f_in, f_g, ldb, w_cols, flags))  // likewise: its planes are in its own fragment order
    return f16x2k_launch(x, packed, bcat, n_nodes, f_in, f_g, ldb, w_cols, bases, weightings, stream);
  if (f_in <= 128 && NV <= 256) {  // weight-stationary form (<= 8 wavefronts)
    const u16* pk = (const u16*)packed;
    int st;
    if (f_in <= 32) st = launch_ws<2>(x, pk, bcat, n_nodes, f_in, w_cols, bases, ldb, weightings, NV, stream);
    else if (f_in <= 64) st = launch_ws<4>(x, pk, bcat, n_nodes, f_in, w_cols, bases, ldb, weightings, NV, stream);
    else if (f_in <= 96) st = launch_ws<6>(x, pk, bcat, n_nodes, f_in, w_cols, bases, ldb, weightings, NV, stream);
    else st = launch_ws<8>(x, pk, bcat, n_nodes, f_in, w_cols, bases, ldb, weightings, NV, stream);
    if (st != EGC_ERR_UNSUPPORTED) return st;  // too few wavefronts to stage a tile: use the LDS-staged kernel
  }
  const int64_t mblocks = ceil_div(n_nodes, XBM);
  if (mblocks >= ((int64_t)1 << 31)) return EGC_ERR_INVALID;
  const bool vec4 = (f_in % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  const int full = NV / XBN;  // column blocks of the full 192 columns; a narrower remainder block follows
  const u16* pk = (const u16*)packed;
  if (NV == 224 && vec4) {  // 193..224 columns: one 7-tile block, one pass over x
    dim3 grid((unsigned)mblocks, 1);
    basis_gemm_bf16x3_kernel<true, 7><<<grid, 256, 0, stream>>>(x, pk, bcat, n_nodes, f_in, w_cols, bases, ldb, weightings, NV, KS, 0);
    EGC_LAUNCH_CHECK("basis_gemm_bf16x3_kernel");
    return EGC_OK;
  }
  if (full > 0) {
    dim3 grid((unsigned)mblocks, (unsigned)full);
    if (vec4)
      basis_gemm_bf16x3_kernel<true, 6><<<grid, 256, 0, stream>>>(x, pk, bcat, n_nodes, f_in, w_cols, bases, ldb, weightings, NV, KS, 0);
    else
      basis_gemm_bf16x3_kernel<false, 6><<<grid, 256, 0, stream>>>(x, pk, bcat, n_nodes, f_in, w_cols, bases, ldb, weightings, NV, KS, 0);
    EGC_LAUNCH_CHECK("basis_gemm_bf16x3_kernel");
  }
#ifdef EGC_GEMM3_STAMPS
  {
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(egc_stamp3), sizeof(h));
    static int calls = 0;
    if (++calls % 10 == 0) {
      const double waves = (double)mblocks * full * 4 * calls;
      fprintf(stderr, "[gemm3 stamps] K=%d NV=%d per wavefront per k-step (cycles of s_memtime @100MHz x?): xwait %.0f split %.0f wwait %.0f wstage %.0f bar1 %.0f mfma %.0f bar2 %.0f\n",
              f_in, NV, h[0] / waves / KS, h[1] / waves / KS, h[2] / waves / KS, h[3] / waves / KS, h[4] / waves / KS, h[5] / waves / KS, h[6] / waves / KS);
    }
  }
#endif
  if (NV % XBN != 0) {
    dim3 grid((unsigned)mblocks, 1);
    const bool four = (NV - full * XBN) == 128;  // a 128-column remainder (or a 128-column GEMM): pipelined too
    if (vec4 && four)
      basis_gemm_bf16x3_kernel<true, 4><<<grid, 256, 0, stream>>>(x, pk, bcat, n_nodes, f_in, w_cols, bases, ldb, weightings, NV, KS, full);
    else if (vec4)
      basis_gemm_bf16x3_kernel<true, 0><<<grid, 256, 0, stream>>>(x, pk, bcat, n_nodes, f_in, w_cols, bases, ldb, weightings, NV, KS, full);
    else
      basis_gemm_bf16x3_kernel<false, 0><<<grid, 256, 0, stream>>>(x, pk, bcat, n_nodes, f_in, w_cols, bases, ldb, weightings, NV, KS, full);
    EGC_LAUNCH_CHECK("basis_gemm_bf16x3_kernel");
  }
  return EGC_OK;
}

}  // extern "C"
