// Weight gradient of the layer's two Linear maps: out[F][K] = x^T @ d, with x [N][F] the layer input and
// d [N][K] = [d_bases | d_weightings] (what autograd computes for the reference's `torch.matmul(x, bases_weight)` and
// `comb_weights(x)`: experiments/optimized_layers.py:177-178), and in the same pass the column sums of d (the
// gradient of comb_weights' bias).
//
// Two kernels, same decomposition.  Outputs of up to 128 x 192 (the north-star layer): xt_gemm_bf16x3_kernel, split
// bf16 on the 16-bit matrix cores (further down, with the optional column sums of a third array riding along).  Larger
// outputs, and EGC_GEMM_EXACT=1: xt_gemm_kernel, exact fp32 on the fp32 matrix cores, described here.
//
// The reduction runs over the N rows, the output is tiny (128 x 192 at config 2): a split over row ranges.  Every
// workgroup keeps one whole output tile (up to 128 x 192) in accumulators, streams its row range through a
// double-buffered LDS tile of 32 rows (global -> registers -> LDS, the loads of two tiles in flight while one is
// multiplied), and multiplies on v_mfma_f32_16x16x4_f32: exact fp32 products and sums, no operand splitting and --
// because that form takes ONE element per lane (A[m = lane % 16][k = lane / 16]) -- no transposition of x on the way
// from its row-major rows.  Partial tiles (and the partial column sums behind them) go to a workspace; a second
// small launch adds them in a fixed order (deterministic, unlike float atomics).
//
// Bound: matrix cores.  2 N F K flops at 64 FLOP/clk/SIMD (MI355X_MICROARCH.md: 157 TFLOP/s fp32 MFMA) = 53 us at
// config 2 (N = 169,343, F = 128, K = 192); the operands are 217 MB = 27 us of HBM time, so the stream hides behind
// the MFMAs.  Measured on MI355X: 87 us + 6 us for the reduction (96 TFLOP/s in the main kernel; the same loop on
// LDS-resident data without loads reaches 122-132, a bare MFMA loop 150), against 85 + 8 + 15 + 8 us for the
// library's split batched GEMM, its sum, and the separate column-sum pass this call replaces.  Shapes that need
// several output tiles (F > 128 or K > 192) re-read the operands once per tile; the host routes the large ones to
// the library GEMM (egc_amd/functional.py).
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "egc_common.h"
#include "egc_pack_map.h"

namespace egc {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: the global loads that are meant
// to stay in flight across two tiles would be waited for at every barrier (measured: tile period = compute + HBM
// latency instead of their maximum).
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int XT_WM = 2;  // wavefronts along the output rows (F) of a block tile

constexpr int xt_pitch(int cols) { return cols + ((cols % 64) == 0 ? 16 : 48); }  // pitch % 64 == 16: the 4 k-rows of a fragment hit 64 distinct banks

// MT x NT: 16 x 16 output tiles per wavefront; XT_WM x WN wavefronts per workgroup (block tile 32 MT rows of F by
// 16 NT WN columns of K); XT_ROWS rows of the reduction per LDS stage
template <int MT, int NT, int WN, int XT_ROWS>
__global__ void __launch_bounds__(64 * XT_WM * WN) xt_gemm_kernel(const float* __restrict__ x, int64_t ldx, int F,
                                                             const float* __restrict__ d, int64_t ldd, int K,
                                                             int64_t n_rows, int64_t rows_per_chunk, int n_chunks,
                                                             int out_m, int out_n,
                                                             float* __restrict__ partial, int want_sums) {
  constexpr int XT_THREADS = 64 * XT_WM * WN;
  constexpr int TM = 16 * XT_WM * MT, TN = 16 * WN * NT, COLS = TM + TN, P = xt_pitch(COLS);
  extern __shared__ float lds[];  // [2][XT_ROWS][P]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave % XT_WM, wn = wave / XT_WM;
  const int m = lane & 15, kk = lane >> 4;
  // workgroup id -> (row range, output tile): the tiles of one row range sit on ONE XCD (ids 8 apart) and are
  // dispatched together, so the range's rows are fetched from HBM once and re-read from that XCD's L2
  const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
  const int tile_id = rest % (out_m * out_n), chunk = (rest / (out_m * out_n)) * 8 + xcd;
  if (chunk >= n_chunks) return;
  const int m_tile = tile_id % out_m, n_tile = tile_id / out_m;
  const int m0 = m_tile * TM, n0 = n_tile * TN;
  const int64_t row_begin = (int64_t)chunk * rows_per_chunk;
  const int64_t row_end = row_begin + rows_per_chunk < n_rows ? row_begin + rows_per_chunk : n_rows;
  const int n_tiles = row_end > row_begin ? (int)((row_end - row_begin + XT_ROWS - 1) / XT_ROWS) : 0;

  // what this thread stages per tile: LOADS float4 pieces (row r, column piece c4) of [x tile | d tile], through
  // buffer loads over the workgroup's row range: rows past the range and masked pieces read as zeros without a
  // branch, so the loads of two tiles stay in flight behind counted waits
  const int64_t range_rows = row_end > row_begin ? row_end - row_begin : 0;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(x + row_begin * ldx), 0, (unsigned)(range_rows * ldx * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(d + row_begin * ldd), 0, (unsigned)(range_rows * ldd * 4), 0x00020000);
  constexpr unsigned XT_OOB = 0xFFFFFFF0u;
  constexpr int AX4 = TM / 4, BD4 = TN / 4;  // float4 pieces per row of the x tile / the d tile
  constexpr int LX = (XT_ROWS * AX4 + XT_THREADS - 1) / XT_THREADS, LD = (XT_ROWS * BD4 + XT_THREADS - 1) / XT_THREADS;
  unsigned off_x[LX], off_d[LD];
  int lds_x[LX], lds_d[LD];
#pragma unroll
  for (int i = 0; i < LX; ++i) {
    const int idx = t + i * XT_THREADS, r = idx / AX4, c4 = idx - r * AX4;
    const bool in_tile = idx < XT_ROWS * AX4;
    off_x[i] = (in_tile && m0 + 4 * c4 < F) ? (unsigned)((r * ldx + m0 + 4 * c4) * 4) : XT_OOB;
    lds_x[i] = in_tile ? r * P + 4 * c4 : -1;
  }
#pragma unroll
  for (int i = 0; i < LD; ++i) {
    const int idx = t + i * XT_THREADS, r = idx / BD4, c4 = idx - r * BD4;
    const bool in_tile = idx < XT_ROWS * BD4;
    off_d[i] = (in_tile && n0 + 4 * c4 < K) ? (unsigned)((r * ldd + n0 + 4 * c4) * 4) : XT_OOB;
    lds_d[i] = in_tile ? r * P + TM + 4 * c4 : -1;
  }
  const unsigned step_x = (unsigned)(XT_ROWS * ldx * 4), step_d = (unsigned)(XT_ROWS * ldd * 4);
  // staging registers: two tiles ahead of the one being multiplied (an HBM round trip under load is longer than one
  // tile's MFMAs).  A masked piece keeps its out-of-range offset; live ones stay below 4 GiB (checked by the host).
  struct Stage { f4 x[LX]; f4 d[LD]; };
  Stage st0, st1;
  auto fetch = [&](int tile, Stage& st) {
#pragma unroll
    for (int i = 0; i < LX; ++i) {
      const unsigned o = off_x[i] == XT_OOB ? XT_OOB : off_x[i] + (unsigned)tile * step_x;
      st.x[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, o, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const unsigned o = off_d[i] == XT_OOB ? XT_OOB : off_d[i] + (unsigned)tile * step_d;
      st.d[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rd, o, 0, 0));
    }
  };
  auto put = [&](int buf, const Stage& st) {
    float* base = lds + buf * (XT_ROWS * P);
#pragma unroll
    for (int i = 0; i < LX; ++i)
      if (lds_x[i] >= 0) *reinterpret_cast<f4*>(base + lds_x[i]) = st.x[i];
#pragma unroll
    for (int i = 0; i < LD; ++i)
      if (lds_d[i] >= 0) *reinterpret_cast<f4*>(base + lds_d[i]) = st.d[i];
  };

  f4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  float cs[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) cs[j] = 0.f;

  const int a_col = wm * (16 * MT) + m, b_col = TM + wn * (16 * NT) + m;
  auto multiply = [&](int bufi) {
    const float* buf = lds + bufi * (XT_ROWS * P) + kk * P;
#pragma unroll
    for (int s = 0; s < XT_ROWS / 4; ++s) {
      float a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = buf[s * 4 * P + a_col + 16 * i];
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = buf[s * 4 * P + b_col + 16 * j];
#pragma unroll
      for (int j = 0; j < NT; ++j) cs[j] += b[j];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  fetch(0, st0);
  put(0, st0);
  fetch(1, st0);
  fetch(2, st1);
  lds_barrier();
  for (int tile = 0; tile < n_tiles; tile += 2) {  // n_tiles is uniform over the workgroup
    multiply(0);
    put(1, st0);
    fetch(tile + 3, st0);
    lds_barrier();
    if (tile + 1 < n_tiles) multiply(1);
    put(0, st1);
    fetch(tile + 4, st1);
    lds_barrier();
  }

  // partial tile: accumulator register r of tile (i, j) is output row 16 i + 4 (lane / 16) + r, column 16 j + lane % 16
  const int64_t record = (int64_t)F * K + K;  // the chunk's tile, then its column sums
  float* out = partial + (int64_t)chunk * record;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn * (16 * NT) + 16 * j + m;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * (16 * MT) + 16 * i + 4 * kk + r;
        if (row < F && col < K) out[(int64_t)row * K + col] = acc[i][j][r];
      }
    }
  if (want_sums && m_tile == 0 && wm == 0) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float v = cs[j];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      const int col = n0 + wn * (16 * NT) + 16 * j + m;
      if (kk == 0 && col < K) out[(int64_t)F * K + col] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same reduction on the 16-bit matrix cores, for outputs of up to 128 x 192 (one accumulator tile per workgroup:
// the north-star layer).  fp32 MFMA caps this GEMM at 53 us (157 TFLOP/s) while its operands are 27 us of HBM; the
// bf16 pipe is 16 x faster, so the operands are split into three bf16 planes IN REGISTERS on the way from LDS to the
// MFMA (v = h + m + l, each plane the top 16 bits of what the previous ones left: 24 significant bits, no scaling --
// bf16 has fp32's exponent range, so gradients of any magnitude survive) and six products are accumulated in fp32
// (hh, hm, mh, mm, hl, lh: what is dropped is below 2^-24 of a product) -- the scheme of egc_gemm_bf16x3.hip.
//   * v_mfma_f32_32x32x16_bf16 wants 8 consecutive elements of the REDUCTION index per lane; the reduction runs over
//     rows, so the tile is transposed on its way through LDS: the thread that splits a value reads 8 consecutive rows
//     of one column of the row-major fp32 stage (8 ds_read_b32, pitch 324 floats) and writes 16 bytes per plane;
//   * 8 wavefronts = 4 (rows of the output: 32 columns of x each) x 2 (3 x 32 columns of d each), 18 MFMAs per
//     wavefront and 16 rows.
// Measured on MI355X at config 2: 68 us + 6 us for the reduction against 87 + 6 for the fp32-MFMA kernel below (and
// 85 + 8 + 15 + 8 for the library path).  The matrix work is 20 us and the HBM stream alone 37 us (the kernel with the
// multiply removed), yet three organisations of the split -- in the consuming wavefronts from the fp32 stage; once per
// tile behind its own barrier; once per tile inside the MFMA interval (this one) -- all land within 3 us of each
// other: per 16 rows a CU moves ~180 KB through LDS (stage write, transposing read, plane write, and 96 KB of fragment
// reads because every B fragment is read by four wavefronts), which is what the three have in common.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

struct Planes3 { bf16x8 h, m, l; };

// eight fp32 values -> three bf16x8 planes (element j of a plane <- v[j]; truncation splits: every step exact)
__device__ inline Planes3 split3(const float (&v)[8]) {
  u32x4v ph, pm, pl;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned a0 = __float_as_uint(v[2 * i]), a1 = __float_as_uint(v[2 * i + 1]);
    const float r0 = v[2 * i] - __uint_as_float(a0 & 0xffff0000u), r1 = v[2 * i + 1] - __uint_as_float(a1 & 0xffff0000u);
    const unsigned b0 = __float_as_uint(r0), b1 = __float_as_uint(r1);
    const float s0 = r0 - __uint_as_float(b0 & 0xffff0000u), s1 = r1 - __uint_as_float(b1 & 0xffff0000u);
    ph[i] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);                       // high halves: [v1.hi16 | v0.hi16]
    pm[i] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    pl[i] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
  Planes3 o;
  o.h = __builtin_bit_cast(bf16x8, ph);
  o.m = __builtin_bit_cast(bf16x8, pm);
  o.l = __builtin_bit_cast(bf16x8, pl);
  return o;
}

constexpr int X3_ROWS = 16, X3_TM = 128, X3_TN = 192, X3_COLS = X3_TM + X3_TN, X3_P = X3_COLS + 4, X3_THREADS = 512;
constexpr int X3_RP = X3_ROWS + 8;   // row pitch of the transposed planes, in halves (48 bytes: 16-byte aligned pieces)
constexpr int X3_STAGE_FLOATS = X3_ROWS * X3_P, X3_PLANE_HALVES = X3_COLS * X3_RP;
constexpr int X3_LDS_BYTES = 2 * X3_STAGE_FLOATS * 4 + 2 * 3 * X3_PLANE_HALVES * 2;
constexpr int X3_AHEAD = 4;   // register stages: global loads run this many 16-row sub-tiles ahead of their LDS write

// Pipeline over 16-row sub-tiles s, ONE barrier per sub-tile; between two barriers every wavefront
//     multiplies sub-tile s        (bf16 planes [s % 2]: 18 MFMAs of 32x32x16, three output tiles, product-major order)
//     splits     sub-tile s + 1    (fp32 stage [(s+1) % 2] -> planes [(s+1) % 2]: its share of 640 column x 8-row tasks)
//     stages     sub-tile s + 2    (registers -> fp32 stage [s % 2], row-major as loaded)
//     requests   sub-tile s + 6    (buffer loads into the register set that was just written out)
// so the vector work of one wavefront (split) runs beside the matrix work of the other wavefront of its SIMD, and
// 64 rows (80 KB per CU) are always in flight.  Every value is split ONCE per tile, by the thread that transposes it
// (8 consecutive rows of one column -> one 16-byte write per plane); an MFMA fragment is one 16-byte read per plane.
// Splitting in the consuming wavefronts instead costs 4 x the vector work (every B fragment is used by four wavefronts)
// and a split phase behind its own barrier does not overlap the MFMAs (both forms were built and measured: header).
__global__ void __launch_bounds__(X3_THREADS) xt_gemm_bf16x3_kernel(const float* __restrict__ x, int64_t ldx, int F,
                                                                    const float* __restrict__ d, int64_t ldd, int K,
                                                                    int64_t n_rows, int64_t rows_per_chunk, int n_chunks,
                                                                    float* __restrict__ partial, int want_sums,
                                                                    const float* __restrict__ e, int64_t lde, int E) {
  extern __shared__ float lds[];
  float* stage = lds;                                                                  // [2][X3_ROWS][X3_P] fp32
  unsigned short* planes = reinterpret_cast<unsigned short*>(lds + 2 * X3_STAGE_FLOATS);  // [2][3][X3_COLS][X3_RP] bf16
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int c32 = lane & 31, g = lane >> 5;
  const int chunk = blockIdx.x;
  if (chunk >= n_chunks) return;
  const int64_t row_begin = (int64_t)chunk * rows_per_chunk;
  const int64_t row_end = row_begin + rows_per_chunk < n_rows ? row_begin + rows_per_chunk : n_rows;
  const int n_sub = row_end > row_begin ? (int)((row_end - row_begin + X3_ROWS - 1) / X3_ROWS) : 0;

  // staging: branch-free buffer loads over the workgroup's row range (rows past it and masked columns read as zeros)
  const int64_t range_rows = row_end > row_begin ? row_end - row_begin : 0;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(x + row_begin * ldx), 0, (unsigned)(range_rows * ldx * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(d + row_begin * ldd), 0, (unsigned)(range_rows * ldd * 4), 0x00020000);
  constexpr unsigned XT_OOB = 0xFFFFFFF0u;
  constexpr int AX4 = X3_TM / 4, BD4 = X3_TN / 4;            // 32, 48 sixteen-byte pieces per row
  static_assert(X3_ROWS * AX4 == X3_THREADS, "one x piece per thread and sub-tile");
  constexpr int LD = (X3_ROWS * BD4 + X3_THREADS - 1) / X3_THREADS;   // 2 (the second for half of the threads)
  unsigned off_x, off_d[LD];
  int lds_x, lds_d[LD];
  {
    const int r = t / AX4, c4 = t - r * AX4;
    off_x = (4 * c4 < F) ? (unsigned)((r * ldx + 4 * c4) * 4) : XT_OOB;
    lds_x = r * X3_P + 4 * c4;
  }
#pragma unroll
  for (int i = 0; i < LD; ++i) {
    const int idx = t + i * X3_THREADS, r = idx / BD4, c4 = idx - r * BD4;
    const bool in_tile = idx < X3_ROWS * BD4;
    off_d[i] = (in_tile && 4 * c4 < K) ? (unsigned)((r * ldd + 4 * c4) * 4) : XT_OOB;
    lds_d[i] = in_tile ? r * X3_P + X3_TM + 4 * c4 : -1;
  }
  const unsigned step_x = (unsigned)(X3_ROWS * ldx * 4), step_d = (unsigned)(X3_ROWS * ldd * 4);
  // optional third array e [n_rows][E <= 128]: only its column sums are wanted (the layer's bias gradient = column sums
  // of grad_out): one more 16-byte load per thread and sub-tile, added up in registers, never staged
  const bool has_e = e != nullptr;
  const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(has_e ? e + row_begin * lde : x), 0, has_e ? (unsigned)(range_rows * lde * 4) : 0u, 0x00020000);
  const unsigned off_e = (has_e && 4 * (t % AX4) < E) ? (unsigned)(((t / AX4) * lde + 4 * (t % AX4)) * 4) : XT_OOB;
  const unsigned step_e = (unsigned)(X3_ROWS * lde * 4);
  f4 esum = f4{0.f, 0.f, 0.f, 0.f};
  struct Regs { f4 x; f4 d[LD]; f4 e; };
  Regs rg[X3_AHEAD];
  auto fetch = [&](int sub, Regs& st) {
    const unsigned ox = off_x == XT_OOB ? XT_OOB : off_x + (unsigned)sub * step_x;
    st.x = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, ox, 0, 0));
    if (has_e)   // uniform
      st.e = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(re, off_e == XT_OOB ? XT_OOB : off_e + (unsigned)sub * step_e, 0, 0));
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const unsigned o = off_d[i] == XT_OOB ? XT_OOB : off_d[i] + (unsigned)sub * step_d;
      st.d[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rd, o, 0, 0));
    }
  };
  auto put = [&](int sb, const Regs& st) {
#ifdef EGC_XT_NO_PUT
    { float sink = st.x.x + st.d[0].x + st.d[LD - 1].x; asm volatile("" :: "v"(sink)); return; }
#endif
    float* base = stage + sb * X3_STAGE_FLOATS;
    if (has_e) esum += st.e;
    *reinterpret_cast<f4*>(base + lds_x) = st.x;
#pragma unroll
    for (int i = 0; i < LD; ++i)
      if (lds_d[i] >= 0) *reinterpret_cast<f4*>(base + lds_d[i]) = st.d[i];
  };
  // column sums of d (from the fp32 stage): wavefront w owns d columns 24 w .. 24 w + 23, two lanes (8 rows each) per column
  const bool sums = want_sums != 0;
  const int cs_col = wave * 24 + (lane % 24), cs_part = lane / 24;   // lanes 48..63 idle
  float colsum = 0.f;
  auto split = [&](int sb) {   // fp32 stage sb -> planes sb (both indexed by the sub-tile's parity)
#ifdef EGC_XT_NO_SPLIT
    return;
#endif
    const float* src = stage + sb * X3_STAGE_FLOATS;
    unsigned short* pl = planes + sb * 3 * X3_PLANE_HALVES;
    if (sums && cs_part < 2) {
#pragma unroll
      for (int j = 0; j < X3_ROWS / 2; ++j) colsum += src[(cs_part * (X3_ROWS / 2) + j) * X3_P + X3_TM + cs_col];
    }
    // task = (column c, row group of 8): lanes take consecutive columns (conflict-free fp32 reads at pitch X3_P)
#pragma unroll
    for (int it = 0; it < (X3_COLS * (X3_ROWS / 8) + X3_THREADS - 1) / X3_THREADS; ++it) {
      const int task = t + it * X3_THREADS;
      if (task < X3_COLS * (X3_ROWS / 8)) {
        const int rgp = task / X3_COLS, c = task - rgp * X3_COLS;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(8 * rgp + j) * X3_P + c];
        const Planes3 p = split3(v);
        unsigned short* dst = pl + c * X3_RP + 8 * rgp;
        *reinterpret_cast<bf16x8*>(dst) = p.h;
        *reinterpret_cast<bf16x8*>(dst + X3_PLANE_HALVES) = p.m;
        *reinterpret_cast<bf16x8*>(dst + 2 * X3_PLANE_HALVES) = p.l;
      }
    }
  };

  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int a_row = wm * 32 + c32, b_row = X3_TM + wn * 96 + c32;   // "rows" of the transposed planes = tile columns
  auto multiply = [&](int sb) {
#ifdef EGC_XT_NO_MFMA
    return;
#endif
    const unsigned short* base = planes + sb * 3 * X3_PLANE_HALVES + 8 * g;
    Planes3 a, b[3];
    a.h = *reinterpret_cast<const bf16x8*>(base + a_row * X3_RP);
    a.m = *reinterpret_cast<const bf16x8*>(base + a_row * X3_RP + X3_PLANE_HALVES);
    a.l = *reinterpret_cast<const bf16x8*>(base + a_row * X3_RP + 2 * X3_PLANE_HALVES);
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
      const unsigned short* bp = base + (b_row + 32 * tt) * X3_RP;
      b[tt].h = *reinterpret_cast<const bf16x8*>(bp);
      b[tt].m = *reinterpret_cast<const bf16x8*>(bp + X3_PLANE_HALVES);
      b[tt].l = *reinterpret_cast<const bf16x8*>(bp + 2 * X3_PLANE_HALVES);
    }
    // product-major order: consecutive MFMAs go to different accumulators (a chain on one accumulator would issue at
    // the result latency instead of the pipe rate); small terms first
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b[tt].h, acc[tt], 0, 0, 0);
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b[tt].l, acc[tt], 0, 0, 0);
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b[tt].m, acc[tt], 0, 0, 0);
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b[tt].h, acc[tt], 0, 0, 0);
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b[tt].m, acc[tt], 0, 0, 0);
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b[tt].h, acc[tt], 0, 0, 0);
  };

  // prologue: sub-tiles 0 and 1 staged, 0 split; register sets hold sub-tiles 2 .. 5 (set = sub-tile % 4)
  fetch(0, rg[0]);
  fetch(1, rg[1]);
  put(0, rg[0]);
  put(1, rg[1]);
  fetch(2, rg[2]);
  fetch(3, rg[3]);
  fetch(4, rg[0]);
  fetch(5, rg[1]);
  lds_barrier();
  split(0);
  lds_barrier();
  // interval s: multiply s | split s + 1 | stage s + 2 | request s + 6        (unrolled by four: register sets are static)
  // The two wavefronts of a SIMD (w and w + 4) take the interval's two halves in OPPOSITE order: a wavefront is held at the
  // issue of its own MFMAs while they run, so in lock step both multiplied (sharing the one matrix pipe) and then both split
  // (sharing the vector pipe) -- the SIMD's two pipes one after the other instead of side by side.
#ifdef EGC_XT_LOCKSTEP
  const bool split_first = false;
#elif defined(EGC_XT_STAGGER_BIT)
  const bool split_first = ((wave >> EGC_XT_STAGGER_BIT) & 1) != 0;
#else
  const bool split_first = wave >= 4;
#endif
#define X3_INTERVAL(S, SET)                      \
  {                                              \
    if (split_first) {                           \
      split(((S) + 1) & 1);                      \
      if ((S) < n_sub) multiply((S) & 1);        \
    } else {                                     \
      if ((S) < n_sub) multiply((S) & 1);        \
      split(((S) + 1) & 1);                      \
    }                                            \
    put((S) & 1, rg[SET]);                       \
    fetch((S) + 6, rg[SET]);                     \
    lds_barrier();                               \
  }
  for (int s = 0; s < n_sub; s += 4) {   // n_sub is uniform over the workgroup
    X3_INTERVAL(s, 2)
    X3_INTERVAL(s + 1, 3)
    X3_INTERVAL(s + 2, 0)
    X3_INTERVAL(s + 3, 1)
  }
#undef X3_INTERVAL

  // C/D layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int64_t record = (int64_t)F * K + K + E;
  float* out = partial + (int64_t)chunk * record;
#pragma unroll
  for (int tt = 0; tt < 3; ++tt) {
    const int col = wn * 96 + 32 * tt + c32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
      if (row < F && col < K) out[(int64_t)row * K + col] = acc[tt][r];
    }
  }
  if (sums) {   // the two row halves of a column meet through LDS (nothing reads the stage any more)
    float* red = stage;
    if (cs_part < 2) red[cs_part * X3_TN + cs_col] = colsum;
    lds_barrier();
    if (t < X3_TN && t < K) out[(int64_t)F * K + t] = red[t] + red[X3_TN + t];
  }
  if (has_e) {  // the sixteen row lanes of a 16-byte column piece
    f4* red4 = reinterpret_cast<f4*>(planes);
    lds_barrier();
    red4[t] = esum;
    lds_barrier();
    if (t < AX4 && 4 * t < E) {
      f4 v = red4[t];
#pragma unroll
      for (int r = 1; r < X3_ROWS; ++r) v += red4[r * AX4 + t];
      *reinterpret_cast<f4*>(out + (int64_t)F * K + K + 4 * t) = v;
    }
  }
}

// out[i] = sum over chunks of partial[c][i], four floats per thread, 16 threads per output piece each adding every
// 16th chunk (all loads of a thread in flight at once), then a tree over the 16.  A chunk's record is the F x K tile
// followed by the K column sums; pieces past `fk` floats go to `sums`.
// SCATTER: the F x K tile is the gradient of a layer's GEMM operand [bases | comb_weight^T] and `sums` that of its bias: they
// are written straight into the PARAMETERS' gradients through the pack's index map (egc_pack_map.h) -- no d wcat array, no
// unpack launch (a training step of the batched nets is bound by its launches).
struct XtScatter {
  PackPtrs bases;      // gradients of the basis matrices
  float* comb_w;       // gradient of the combination Linear's weight
  float* comb_b;       // gradient of its bias (rows permuted with the weight's), or nullptr
  float* bcat;         // ... or the bias gradient in the operand's order, or nullptr
  PackDims d;
  int k_cols;          // columns of the tile = B Ls + W
};
template <bool SCATTER = false>
__global__ void __launch_bounds__(256) xt_reduce_kernel(const float* __restrict__ partial, int64_t record, int64_t fk,
                                                        int chunks, float* __restrict__ out, float* __restrict__ sums,
                                                        int sums_cols = 0, float* __restrict__ sums2 = nullptr,
                                                        XtScatter sc = XtScatter()) {
  __shared__ f4 red[256];
  const int t = threadIdx.x, o = t & 15, g = t >> 4;
  const int64_t piece = (int64_t)blockIdx.x * 16 + o;
  // record = [fk floats -> out][sums_cols floats -> sums][the rest -> sums2]; a null destination ends the record early
  const bool has_sums = SCATTER || sums != nullptr;      // (scatter mode: the column sums go to the bias gradient)
  const int64_t used = !has_sums ? fk : (sums2 == nullptr && sums_cols > 0 ? fk + sums_cols : record);
  const int64_t pieces = used / 4, stride = record / 4;
  f4 s = f4{0.f, 0.f, 0.f, 0.f};
  if (piece < pieces) {
    const f4* p = reinterpret_cast<const f4*>(partial) + piece;
#pragma unroll 8
    for (int c = g; c < chunks; c += 16) s += p[(int64_t)c * stride];
  }
  red[t] = s;
  __syncthreads();
#pragma unroll
  for (int w = 128; w >= 16; w >>= 1) {
    if (t < w) red[t] += red[t + w];
    __syncthreads();
  }
  if (t < 16 && piece < pieces) {
    if (SCATTER && piece < (fk + sums_cols) / 4) {
      const f4 v = red[t];
      const float vv[4] = {v.x, v.y, v.z, v.w};
      if (piece < fk / 4) {
        const int64_t idx = 4 * piece;
        const int k = (int)(idx / sc.k_cols), c0 = (int)(idx - (int64_t)k * sc.k_cols);   // (k_cols % 4 == 0: one row per piece)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float* p = pack_param_ptr(sc.bases, sc.comb_w, sc.d, k, c0 + i);
          if (p != nullptr) *p = vv[i];
        }
      } else {
        const int F_g = sc.d.B * sc.d.Ls;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int j = (int)(4 * (piece - fk / 4)) + i - F_g;    // column of the weightings block
          if (j < 0) continue;                                      // column sums of d bases: nobody's gradient
          if (sc.comb_b != nullptr) sc.comb_b[pack_comb_row(sc.d, j)] = vv[i];
          else if (sc.bcat != nullptr) sc.bcat[j] = vv[i];
        }
      }
    } else if (piece < fk / 4) reinterpret_cast<f4*>(out)[piece] = red[t];
    else if (sums2 == nullptr || piece < (fk + sums_cols) / 4) reinterpret_cast<f4*>(sums)[piece - fk / 4] = red[t];
    else reinterpret_cast<f4*>(sums2)[piece - (fk + sums_cols) / 4] = red[t];
  }
}

struct XtPlan {
  int mt, nt, wn, m_tiles, n_tiles, chunks;
  int64_t rows_per_chunk;
};

constexpr int XT_STAGE = 32;  // rows of the reduction per LDS stage

// The compiled block tiles of xt_gemm_kernel: 32 mt rows of F by 16 wn nt columns of K, 2 x wn wavefronts (one workgroup per CU).
// The tall ones (mt 5 / 7: 160 / 224 rows, 60 - 84 accumulator registers per lane) are for the reference's batched nets, whose
// outputs a grid of 128 x 192 tiles pads by half (224 x 272 of molhiv EGC-M -> 256 x 384; 296 x 180 of EGC-S re-read d five times
// as 64-row tiles): round 6, DESIGN.md 3.7.
struct XtShape { int mt, nt, wn; };
constexpr XtShape XT_SHAPES[] = {{1, 1, 4}, {1, 2, 4}, {1, 3, 4}, {2, 1, 4}, {2, 2, 4}, {2, 3, 4}, {4, 1, 4}, {4, 2, 4}, {4, 3, 4},
                                 {5, 2, 4}, {5, 3, 4}, {7, 2, 4}, {7, 3, 3}, {7, 3, 4}, {7, 5, 4}, {7, 3, 6}};

// EGC_GEMM_EXACT=1 (the switch of the forward GEMMs): exact fp32 products on the fp32 MFMA
// instead of the split-bf16 form
// Read per call (a getenv is nothing next to a launch): the host side decides from the same variables at every call
// (functional._weight_grads), and a value cached here at first use would disagree with it once the environment changes.
bool xt_fp32_only() {
  const char* e = getenv("EGC_GEMM_EXACT");
  return e != nullptr && e[0] != '\0' && !(e[0] == '0' && e[1] == '\0');
}

XtPlan xt_plan(int64_t n_rows, int F, int K) {
  // Tile shape and row split together, by modelled time: a workgroup's time per row of the reduction is the larger of its tile's MFMA
  // time on one CU's share of the fp32 matrix rate (157 flop/ps over 256 CUs; six wavefronts leave two of the four SIMDs with twice
  // the work) and of the operand bytes it stages (6 B/ps over 256 CUs: the tiles of one row range re-read them from their XCD's
  // L2), plus a quarter of the smaller; times the rows of a workgroup and the rounds of workgroups on the fullest XCD; plus the partial tiles'
  // way through the workspace and back.  Short reductions (a batch of 128 molecules: 3 k rows) want many small tiles, long ones
  // the tallest tile that fits the registers.
  XtPlan p{};
  double best = 1e300;
  const bool one_tile = F <= 128 && K <= 192 && !xt_fp32_only();   // xt_gemm_bf16x3_kernel
  const int64_t n = n_rows > 0 ? n_rows : 1;
  const int64_t max_chunks = ceil_div(n, XT_STAGE);
  // EGC_XT_TILE="mt,nt,wn": this tile of XT_SHAPES for every multi-tile call (tools/xt_wide_time.py: how the model above was fitted)
  int force[3] = {0, 0, 0};
  if (const char* e = getenv("EGC_XT_TILE")) {
    if (sscanf(e, "%d,%d,%d", &force[0], &force[1], &force[2]) != 3) force[0] = 0;
  }
  for (const XtShape& sh : XT_SHAPES) {
    if (one_tile && (sh.mt != 4 || sh.nt != 3 || sh.wn != 4)) continue;
    if (!one_tile && force[0] != 0 && (sh.mt != force[0] || sh.nt != force[1] || sh.wn != force[2])) continue;
    const int tm = 32 * sh.mt, tn = 16 * sh.wn * sh.nt;
    const int64_t mtl = ceil_div(F, tm), ntl = ceil_div(K, tn), tiles = mtl * ntl;
    // row ranges: the tiles of a range share an XCD (xt_gemm_kernel's id mapping), 32 CUs each -- as many ranges as keep every
    // XCD at one round of workgroups (86 ranges of three tiles were 33 workgroups on some XCDs: two rounds, 206 instead of 99 us
    // at 136 x 184 and 169 k rows)
    int64_t chunks = std::min<int64_t>(8 * std::max<int64_t>(1, 32 / tiles), max_chunks);
    const int64_t rows = ceil_div(ceil_div(n, chunks), XT_STAGE) * XT_STAGE;
    chunks = ceil_div(n, rows);
    const int64_t rounds = ceil_div(ceil_div(chunks, 8) * tiles, 32);
    const double mfma = 2.0 * tm * tn * 256.0 / 157.0 * (sh.wn == 3 ? 4.0 / 3.0 : 1.0);
    const double mem = 4.0 * (tm + tn) * 256.0 / 6.0;
    const double per_row = (mfma > mem ? mfma : mem) + 0.25 * (mfma > mem ? mem : mfma);
    const double cost = (double)rounds * (double)rows * per_row + 2.0 * (double)chunks * F * K * 4.0 / 6.0;
    if (cost < best) {
      best = cost;
      p.mt = sh.mt;
      p.nt = sh.nt;
      p.wn = sh.wn;
      p.m_tiles = (int)mtl;
      p.n_tiles = (int)ntl;
      p.rows_per_chunk = rows;
      p.chunks = (int)chunks;
    }
  }
  return p;
}

template <int MT, int NT, int WN, int ROWS>
int launch_xt(const XtPlan& p, const float* x, int64_t ldx, int F, const float* d, int64_t ldd, int K, int64_t n_rows,
              float* partial, int want_sums, hipStream_t stream) {
  constexpr int lds_bytes = 2 * ROWS * xt_pitch(16 * XT_WM * MT + 16 * WN * NT) * 4;
  static bool configured = false;
  if (!configured) {
    EGC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&xt_gemm_kernel<MT, NT, WN, ROWS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    configured = true;
  }
  const unsigned grid = (unsigned)(ceil_div(p.chunks, 8) * 8 * p.m_tiles * p.n_tiles);
  xt_gemm_kernel<MT, NT, WN, ROWS><<<grid, 64 * XT_WM * WN, lds_bytes, stream>>>(
      x, ldx, F, d, ldd, K, n_rows, p.rows_per_chunk, p.chunks, p.m_tiles, p.n_tiles, partial, want_sums);
  EGC_LAUNCH_CHECK("xt_gemm_kernel");
  return EGC_OK;
}

}  // namespace
}  // namespace egc

extern "C" {

int64_t egc_weight_grad_ex_workspace_bytes(int64_t n_rows, int32_t f_in, int32_t k_cols, int32_t e_cols) {
  if (n_rows < 0 || f_in <= 0 || k_cols <= 0 || e_cols < 0) return 0;
  const egc::XtPlan p = egc::xt_plan(n_rows, f_in, k_cols);
  return (int64_t)p.chunks * ((int64_t)f_in * k_cols + k_cols + e_cols) * 4;
}

int egc_weight_grad_plan(int64_t n_rows, int32_t f_in, int32_t k_cols, int32_t* plan8) {
  if (n_rows < 0 || f_in <= 0 || k_cols <= 0 || plan8 == nullptr) return EGC_ERR_INVALID;
  const egc::XtPlan p = egc::xt_plan(n_rows, f_in, k_cols);
  if (p.mt == 0) return EGC_ERR_UNSUPPORTED;   // (EGC_XT_TILE names a tile that is not compiled)
  const bool one_tile = !egc::xt_fp32_only() && f_in <= egc::X3_TM && k_cols <= egc::X3_TN;
  const int32_t v[8] = {one_tile ? 1 : 0, one_tile ? egc::X3_TM : 32 * p.mt, one_tile ? egc::X3_TN : 16 * p.wn * p.nt,
                        one_tile ? 1 : p.m_tiles, one_tile ? 1 : p.n_tiles, p.chunks, (int32_t)p.rows_per_chunk,
                        one_tile ? egc::X3_THREADS : 64 * egc::XT_WM * p.wn};
  for (int i = 0; i < 8; ++i) plan8[i] = v[i];
  return EGC_OK;
}

int64_t egc_weight_grad_workspace_bytes(int64_t n_rows, int32_t f_in, int32_t k_cols) {
  return egc_weight_grad_ex_workspace_bytes(n_rows, f_in, k_cols, 0);
}

static int weight_grad_impl(const float* x, int64_t ldx, const float* d, int64_t ldd, int64_t n_rows, int32_t f_in,
                            int32_t k_cols, float* out, float* col_sums, const float* e, int64_t lde, int32_t e_cols,
                            float* e_sums, void* workspace, int64_t workspace_bytes, void* stream_, const egc::XtScatter* sc) {
  using namespace egc;
  hipStream_t stream = (hipStream_t)stream_;
  if (n_rows < 0 || f_in <= 0 || k_cols <= 0 || (out == nullptr && sc == nullptr)) return EGC_ERR_INVALID;
  if (e == nullptr || e_sums == nullptr) { e = nullptr; e_sums = nullptr; e_cols = 0; }
  if ((f_in % 4) || (k_cols % 4) || (ldx % 4) || (ldd % 4) || ((uintptr_t)x % 16) || ((uintptr_t)d % 16) ||
      ((uintptr_t)out % 16) || ((uintptr_t)col_sums % 16) || ((uintptr_t)workspace % 16) || (e_cols % 4) || (lde % 4) ||
      ((uintptr_t)e % 16) || ((uintptr_t)e_sums % 16))
    return EGC_ERR_UNSUPPORTED;
  const bool fp32_only = xt_fp32_only();
  const bool one_tile = !fp32_only && f_in <= X3_TM && k_cols <= X3_TN;
  // the third array rides along only in the one-tile kernel, next to the column sums of d, 128 columns at most
  const int want_sums = (col_sums != nullptr || sc != nullptr) ? 1 : 0;   // scatter mode: the sums are the bias gradient
  if (e != nullptr && (!one_tile || e_cols > X3_TM || !want_sums)) return EGC_ERR_UNSUPPORTED;
  if (workspace_bytes < egc_weight_grad_ex_workspace_bytes(n_rows, f_in, k_cols, e_cols) || workspace == nullptr)
    return EGC_ERR_INVALID;
  const XtPlan p = xt_plan(n_rows, f_in, k_cols);
  if (p.mt == 0) return EGC_ERR_UNSUPPORTED;
  // 32-bit buffer offsets inside a workgroup's row range (plus the look-ahead past its end)
  const int64_t widest = std::max(std::max(ldx, ldd), e != nullptr ? lde : (int64_t)0);
  if ((double)(p.rows_per_chunk + 160) * (double)widest * 4.0 >= 4.0e9) return EGC_ERR_UNSUPPORTED;
  const int64_t fk = (int64_t)f_in * k_cols;
  float* partial = static_cast<float*>(workspace);
  int rc = EGC_ERR_UNSUPPORTED;
  if (one_tile) {   // one accumulator tile: the bf16x3 kernel
    constexpr int lds_bytes = X3_LDS_BYTES;
    static bool configured = false;
    if (!configured) {
      EGC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&xt_gemm_bf16x3_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
      configured = true;
    }
    xt_gemm_bf16x3_kernel<<<(unsigned)p.chunks, X3_THREADS, lds_bytes, stream>>>(
        x, ldx, f_in, d, ldd, k_cols, n_rows, p.rows_per_chunk, p.chunks, partial, want_sums, e, lde, e_cols);
    EGC_LAUNCH_CHECK("xt_gemm_bf16x3_kernel");
    rc = EGC_OK;
  } else {
#define EGC_XT_CASE(MT, NT, WN) \
  if (p.mt == MT && p.nt == NT && p.wn == WN) rc = launch_xt<MT, NT, WN, ((MT * NT > 21 || WN > 4) ? XT_STAGE / 2 : XT_STAGE)>(p, x, ldx, f_in, d, ldd, k_cols, n_rows, partial, want_sums, stream);
  EGC_XT_CASE(1, 1, 4) EGC_XT_CASE(1, 2, 4) EGC_XT_CASE(1, 3, 4)
  EGC_XT_CASE(2, 1, 4) EGC_XT_CASE(2, 2, 4) EGC_XT_CASE(2, 3, 4)
  EGC_XT_CASE(4, 1, 4) EGC_XT_CASE(4, 2, 4) EGC_XT_CASE(4, 3, 4)
  EGC_XT_CASE(5, 2, 4) EGC_XT_CASE(5, 3, 4)
  EGC_XT_CASE(7, 2, 4) EGC_XT_CASE(7, 3, 3) EGC_XT_CASE(7, 3, 4) EGC_XT_CASE(7, 5, 4) EGC_XT_CASE(7, 3, 6)
#undef EGC_XT_CASE
  }
  if (rc != EGC_OK) return rc;
  const int64_t record = fk + k_cols + e_cols;
  const int64_t used = !want_sums ? fk : (e != nullptr ? record : fk + k_cols);
  if (sc != nullptr)
    xt_reduce_kernel<true><<<(unsigned)ceil_div(used / 4, 16), 256, 0, stream>>>(partial, record, fk, p.chunks, out, col_sums, k_cols,
                                                                                 e_sums, *sc);
  else
    xt_reduce_kernel<false><<<(unsigned)ceil_div(used / 4, 16), 256, 0, stream>>>(partial, record, fk, p.chunks, out, col_sums, k_cols,
                                                                                  e_sums);
  EGC_LAUNCH_CHECK("xt_reduce_kernel");
  return EGC_OK;
}

int egc_weight_grad_ex_f32(const float* x, int64_t ldx, const float* d, int64_t ldd, int64_t n_rows, int32_t f_in,
                           int32_t k_cols, float* out, float* col_sums, const float* e, int64_t lde, int32_t e_cols,
                           float* e_sums, void* workspace, int64_t workspace_bytes, void* stream_) {
  return weight_grad_impl(x, ldx, d, ldd, n_rows, f_in, k_cols, out, col_sums, e, lde, e_cols, e_sums, workspace, workspace_bytes,
                          stream_, nullptr);
}

int egc_weight_grad_params_f32(const float* x, int64_t ldx, const float* d, int64_t ldd, int64_t n_rows, int32_t f_in,
                               int32_t num_heads, int32_t num_aggrs, int32_t num_bases, int32_t basis_len, int32_t basis_stride,
                               int32_t permute_hab, float* const* d_bases_parts, int32_t n_parts, float* d_comb_weight,
                               float* d_comb_bias, float* d_bcat, const float* e, int64_t lde, int32_t e_cols, float* e_sums,
                               void* workspace, int64_t workspace_bytes, void* stream_) {
  using namespace egc;
  if (f_in <= 0 || num_heads <= 0 || num_aggrs <= 0 || num_bases <= 0 || basis_len <= 0 || basis_stride < basis_len ||
      d_bases_parts == nullptr || d_comb_weight == nullptr || (n_parts != 1 && n_parts != num_bases))
    return EGC_ERR_INVALID;
  if (n_parts > PACK_MAX_PARTS) return EGC_ERR_UNSUPPORTED;
  XtScatter sc;
  for (int i = 0; i < PACK_MAX_PARTS; ++i) sc.bases.part[i] = i < n_parts ? d_bases_parts[i] : nullptr;
  for (int i = 0; i < n_parts; ++i)
    if (sc.bases.part[i] == nullptr) return EGC_ERR_INVALID;
  sc.comb_w = d_comb_weight;
  sc.comb_b = d_comb_bias;
  sc.bcat = d_bcat;
  sc.d = PackDims{f_in, num_heads, num_aggrs, num_bases, basis_len, basis_stride, n_parts, permute_hab != 0};
  sc.k_cols = num_bases * basis_stride + num_heads * num_bases * num_aggrs;
  // the column sums are always formed (the bias gradient rides on them); they land in the parameters, not in an array
  return weight_grad_impl(x, ldx, d, ldd, n_rows, f_in, sc.k_cols, nullptr, nullptr, e, lde, e_cols, e_sums, workspace,
                          workspace_bytes, stream_, &sc);
}

int egc_weight_grad_f32(const float* x, int64_t ldx, const float* d, int64_t ldd, int64_t n_rows, int32_t f_in,
                        int32_t k_cols, float* out, float* col_sums, void* workspace, int64_t workspace_bytes,
                        void* stream_) {
  return egc_weight_grad_ex_f32(x, ldx, d, ldd, n_rows, f_in, k_cols, out, col_sums, nullptr, 0, 0, nullptr, workspace,
                                workspace_bytes, stream_);
}

/* out[c] = sum over p of partials[p][c] (p < n_partials, c < cols): the second step of egc_column_sums_f32 (and of any
 * other [P][cols] float32 partial layout), in a fixed order. */
int egc_sum_partials_f32(const float* partials, int32_t n_partials, int32_t cols, float* out, void* stream_) {
  using namespace egc;
  hipStream_t stream = (hipStream_t)stream_;
  if (n_partials <= 0 || cols <= 0 || partials == nullptr || out == nullptr) return EGC_ERR_INVALID;
  if ((cols & 3) != 0 || ((uintptr_t)partials & 15) != 0 || ((uintptr_t)out & 15) != 0) return EGC_ERR_UNSUPPORTED;
  xt_reduce_kernel<<<(unsigned)ceil_div(cols / 4, 16), 256, 0, stream>>>(partials, cols, cols, n_partials, out, nullptr);
  EGC_LAUNCH_CHECK("xt_reduce_kernel");
  return EGC_OK;
}

}  // extern "C"
