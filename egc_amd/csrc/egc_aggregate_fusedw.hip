// Fused aggregate + combine WITH the weightings Linear inside the launch (gfx950) -- SURVEY.md 8(f) rank 3:
// the [N, H*B*A] `weightings` array is never written to or read from memory.
//
// Reference behaviour replaced, in one launch: comb_weights(x) / comb_weight(x) (experiments/layers.py:110,
// optimized_layers.py:182), the weight nonlinearity (layers.py:112-125, optimized_layers.py:183-184) and
// everything egc_aggregate_fast.hip replaces (gather, per-aggregator scatter / spmm, stack, weighted sum, bias:
// layers.py:109-138,191-225; optimized_layers.py:186-278).
//
// Why it is a producer / consumer kernel.  The register-resident aggregation wants 24 wavefronts per CU at 80
// registers each (measured on the north-star launch: 16 wavefronts per CU +6 %, 12 +18 %, 8 +45 %), and a 16-row
// tile of weightings is 32 registers per lane: a wavefront cannot hold both.  So the roles are split INSIDE a
// workgroup of 5 wavefronts that owns one 16-row tile:
//   * wavefront 4, the producer, computes the tile's weightings on the fp32 matrix cores
//     (v_mfma_f32_16x16x4_f32: exact fp32; the weights streamed as pre-arranged A fragments from L2 by a hand-counted
//     software pipeline, x rows straight from memory as the B operand -- no LDS staging, no operand splitting) into
//     8 KB of LDS;
//   * wavefronts 0-3, the consumers, run the unchanged lane-group-per-row aggregation of
//     egc_aggregate_fast_dev.h on one row group each, meet the producer at ONE s_barrier right before the combine
//     and read their rows' weightings from LDS.
// Long rows keep their chunk scheme (leading workgroups); the wavefront that finishes a long row computes that one
// row's weightings itself (all 16 MFMA columns carry the same row).
//
// STATUS: parity-green, but SLOWER than the two-launch path on the north-star shape (163 us
// against 102 us for the aggregate launch, MI355X): every producer wavefront takes a slot the gather needs -- the
// launch is bound by the number of row groups in flight (about 14 us per row group whatever runs beside it), 16
// consumer wavefronts per CU instead of 24.  A persistent form with a device-wide work queue and an LDS ring
// (2 producers + 6 consumers per workgroup) was built and measured as well: 325 us, gather latency growing
// linearly with the consumers per CU; it is not kept.  The path is therefore opt-in (EGC_FUSEDW=1).
#include <stdlib.h>

#include <algorithm>

#include "egc_aggregate_fast_dev.h"

namespace egc {

// weights of the combination Linear as MFMA A fragments + its bias as [H][4][4]
__global__ void __launch_bounds__(256) fusedw_pack_kernel(const float* __restrict__ wcat, const float* __restrict__ bcat,
                                                          int F_in, int ldw_cat, int col0, int H, int A, int M,
                                                          float* __restrict__ frag, float* __restrict__ bias4) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = H * M * 64 * 4;
  if (idx < total) {
    const int t = idx & 3, lane = (idx >> 2) & 63, hm = idx >> 8;
    const int h = hm % H, m = hm / H;   // fragment order [m][h][lane]: a producer streams it front to back
    const int i = lane & 15, kq = lane >> 4;
    const int k = 16 * m + 4 * kq + t;
    const int b = i >> 2, aa = i & 3;
    float v = 0.f;
    if (k < F_in && aa < A) v = wcat[(int64_t)k * ldw_cat + col0 + (h * 4 + b) * A + aa];
    frag[idx] = v;
  }
  if (idx < H * 16) {
    const int h = idx >> 4, b = (idx >> 2) & 3, aa = idx & 3;
    bias4[idx] = (bcat != nullptr && aa < A) ? bcat[(h * 4 + b) * A + aa] : 0.f;
  }
}

// A workgroup of 5 wavefronts owns one 16-row tile.  Wavefront 4 computes the tile's weightings into LDS while
// wavefronts 0-3 already gather their row groups; one s_barrier right before the combine is the only hand-off.
// Leading workgroups take the long-row chunks (wavefronts 0-3).
template <int HPB, int NH, class C>
__global__ void __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(6))) agg_fusedw_tile_kernel(AggArgs a) {
  constexpr int LPR_LOG2 = 4, LPR = 16, G = 4, NEED = 0;
  extern __shared__ float smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int g = lane >> LPR_LOG2;
  const int q = lane & (LPR - 1);
  const int WROW = a.w_lds_stride;
  const bool post = a.post_scale != nullptr;
  float* lds_bias = smem;
  float* lds_scale = lds_bias + a.bias_lds_floats;
  float* tile = lds_bias + (post ? 2 : 1) * a.bias_lds_floats;   // [16 rows][WROW]; chunk role: [4 waves][4 strips][WROW]
  for (int o = threadIdx.x; o < C::H(a) * C::Ls(a); o += 320) {
    float bv = a.bias != nullptr ? a.bias[o] : 0.f;
    if (post) {
      const float sc = a.post_scale[o];
      bv = fmaf(bv, sc, a.post_shift[o]);
      lds_scale[o] = sc;
    }
    lds_bias[o] = bv;
  }
  FastRsrc R;
  R.bases = bases_rsrc(a);
  const int F_out = C::F_out(a);
  R.out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (unsigned)a.n_nodes * (unsigned)F_out * 4u, 0x00020000);
  R.res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual != nullptr ? a.residual : a.out), 0,
                                            (unsigned)a.n_nodes * (unsigned)F_out * 4u, 0x00020000);
  if ((int)blockIdx.x < a.chunk_blocks) {
    __syncthreads();  // bias strip
    if (wave == 4) return;
    const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    long_row_chunk<LPR_LOG2, HPB, NEED, C, true>(a, R, c, lane, tile + wave * G * WROW, lds_bias, lds_scale);
    return;
  }
  const int n_end = a.row_end;
  const int r0t = a.row_begin + (((int)blockIdx.x - a.chunk_blocks) << 4);
  if (wave == 4) {
    // ---- producer: the tile's weightings, four heads per pass ----
    const __amdgpu_buffer_rsrc_t xr = x_rsrc(a);
    const int j = lane & 15, quad = lane >> 4;
    float* dst = tile + j * WROW + quad * 4;
    for (int h0 = 0; h0 < NH; h0 += 4) {
      f4 acc[4];
      w_tile<4>(a, xr, lane, r0t + j, r0t + j < n_end, h0, NH, acc);
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) *reinterpret_cast<f4*>(dst + (h0 + hh) * 16) = w_act<C>(a, acc[hh]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    return;
  }
  // ---- consumers: wavefront k owns row group k of the tile ----
  const bool looped_any = C::xl(a) || C::yl(a);
  const unsigned slot_off = (unsigned)q * 16u;
  const unsigned row_bytes = (unsigned)a.ldb * 4u;
  const bool lane_live = q < C::slots(a);
  const int grp_addr = (g << LPR_LOG2) << 2;
  const int r0 = r0t + 4 * wave;
  const int rp = a.rowptr[min(r0 + lane, a.n_nodes)];
  const int start = bperm(g << 2, rp);
  const int deg_all = bperm((g + 1) << 2, rp) - start;
  const int row = r0 + g;
  const bool row_ok = row < n_end;
  const int nd = (row_ok && deg_all <= EGC_LONG_ROW_THRESHOLD) ? deg_all : 0;
  int jj = q < nd ? a.col[start + q] : 0;
  int maxd = nd;
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) maxd = max(maxd, bperm((lane ^ off) << 2, maxd));
  maxd = __builtin_amdgcn_readfirstlane(maxd);
  f4 wpre[2], vself;
  bool has_self;
  const float dis_i = (a.dis != nullptr && row_ok) ? a.dis[row] : 0.f;
  load_row_operands<LPR_LOG2, C, true>(a, R, lane, row, row_ok, wpre, vself, has_self);
  FAcc<NEED> acc;
  acc.init();
  int nself = 0;
  for (int ts = 0; ts < maxd; ts += LPR) {
    if (ts > 0) jj = (ts + q < nd) ? a.col[start + ts + q] : 0;
    const bool pv = ts + q < nd;
    const float dd = !pv ? 0.f : a.edis != nullptr ? a.edis[start + ts + q] : a.dis != nullptr ? a.dis[jj] : 0.f;
    if (looped_any) {
      const unsigned long long sb = __ballot(pv && jj == row);
      nself += __popcll((sb >> (g << LPR_LOG2)) & ((1ull << LPR) - 1ull));
    }
    const int cnt = min(LPR, maxd - ts);
    for (int t0 = 0; t0 < cnt; t0 += FU)
      gather_batch<NEED, C>(a, R, acc, grp_addr + (t0 << 2), 4, row, jj, dd, dis_i, lane_live ? nd : 0, ts + t0, 1, row_bytes,
                            slot_off, start);
  }
  int ln = lane;
  asm volatile("" : "+v"(ln));
  const bool is_short = deg_all <= EGC_LONG_ROW_THRESHOLD;
  __syncthreads();  // the tile's weightings (and the bias strip) are in LDS
  finish_group<LPR_LOG2, HPB, NEED, C, true>(a, R, ln, row, row_ok, acc, nd, nself, dis_i, vself, has_self, wpre, is_short,
                                             tile + wave * G * WROW, lds_bias, lds_scale);
}

template <int HPB, int NH, class C>
static int launch_fw_tile(const AggArgs& a, unsigned grid, size_t lds, hipStream_t stream) {
  agg_fusedw_tile_kernel<HPB, NH, C><<<grid, 320, lds, stream>>>(a);
  EGC_LAUNCH_CHECK("agg_fusedw_tile_kernel");
  return EGC_OK;
}

bool fusedw_supported(const AggArgs& a, int layout) {
  if (!fast_path_supported(a, layout, 1)) return false;
  if (a.B != 4 || a.Ls != 16 || a.L != 16 || a.slots != 16) return false;     // lane group of 16, [b = 4][a <= 4] per head
  if (a.H != 4 && a.H != 8) return false;
  if (a.stats != nullptr || a.arg_max != nullptr || a.arg_min != nullptr) return false;  // inference form only
  if (a.F_in <= 0 || (a.F_in & 3) != 0) return false;                                   // 16-byte x pieces
  if ((uint64_t)a.n_nodes * (uint64_t)a.F_in * 4ull > (uint64_t)OOB) return false;      // 32-bit buffer offsets into x
  for (int t = 0; t < a.A; ++t)
    if (a.aggr[t] == EGC_AGGR_VAR || a.aggr[t] == EGC_AGGR_STD || a.aggr[t] == EGC_AGGR_MIN) return false;  // NEED == 0 variants
  return true;
}

static inline int fusedw_ksteps(int f_in) { return (((f_in + 15) / 16) + 1) & ~1; }  // k-steps of 16, an even number

size_t fusedw_pack_floats(int H, int f_in) { return (size_t)H * fusedw_ksteps(f_in) * 64 * 4 + (size_t)H * 16; }

int fusedw_pack(const float* wcat, const float* bcat, int f_in, int ldw_cat, int col0, int H, int A, float* packed,
                hipStream_t stream) {
  const int M = fusedw_ksteps(f_in);
  const int total = H * M * 64 * 4;
  fusedw_pack_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, stream>>>(wcat, bcat, f_in, ldw_cat, col0, H, A, M, packed,
                                                                             packed + total);
  EGC_LAUNCH_CHECK("fusedw_pack_kernel");
  return EGC_OK;
}

int launch_fusedw(AggArgs a, const PlanCaps& caps, hipStream_t stream) {
  a.lanes_pb = a.Ls / 4;
  a.lpb_log2 = 2;
  a.magic_P = (unsigned)(((uint64_t)1 << 32) / (uint64_t)a.lanes_pb) + 1u;
  a.rows_per_wave = 1;
  a.need_mean = a.need_var = 0;
  for (int t = 0; t < a.A; ++t)
    if (a.aggr[t] == EGC_AGGR_MEAN) a.need_mean = 1;
  a.w_lds_stride = a.H * 16;
  a.bias_lds_floats = (a.H * a.Ls + 3) & ~3;
  a.lds_floats_per_wave = 0;
  a.M = fusedw_ksteps(a.F_in);
  a.chunk_blocks = (int)ceil_div(a.n_chunks_hint >= 0 ? a.n_chunks_hint : caps.cap_chunks, 4);
  const int64_t n_tiles = ceil_div((int64_t)a.row_end - a.row_begin, 16);
  const size_t lds = ((size_t)(a.post_scale != nullptr ? 2 : 1) * a.bias_lds_floats + (size_t)16 * a.w_lds_stride) * sizeof(float);
  if (lds > 64 * 1024) return EGC_ERR_UNSUPPORTED;
  const unsigned grid = (unsigned)(a.chunk_blocks + n_tiles);
  constexpr int S = EGC_AGGR_SUM, M = EGC_AGGR_MEAN, X = EGC_AGGR_MAX, Y = EGC_AGGR_SYMNORM;
  unsigned pk = 0;
  for (int t = 0; t < a.A; ++t) pk |= (unsigned)a.aggr[t] << (3 * t);
  if (getenv("EGC_NO_STATIC_CFG") == nullptr && a.H == 8 && a.act == EGC_ACT_NONE && a.loops_all != 0) {
    // EGConv / EGC-M north star: d = 128, H = 8, B = 4, sum+mean+max+symnorm on the gcn_norm edge set
    if (a.A == 4 && pk == agg_pack(S, M, X, Y) && a.x_looped && a.y_looped)
      return launch_fw_tile<2, 8, StCfg<8, 4, 16, 4, agg_pack(S, M, X, Y), EGC_ACT_NONE, true, true, true>>(a, grid, lds, stream);
    // EfficientGraphConv EGC-M at d = 128: symadd looped, max / mean raw (layers.py:166-193)
    if (a.A == 3 && pk == agg_pack(Y, X, M) && !a.x_looped && a.y_looped)
      return launch_fw_tile<2, 8, StCfg<8, 4, 16, 3, agg_pack(Y, X, M), EGC_ACT_NONE, false, true, true>>(a, grid, lds, stream);
  }
  if (a.H == 8) return launch_fw_tile<2, 8, RtCfg>(a, grid, lds, stream);
  return launch_fw_tile<1, 4, RtCfg>(a, grid, lds, stream);
}

}  // namespace egc
