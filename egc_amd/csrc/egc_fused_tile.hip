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
#include "egc_pack_map.h"
#include "egc_fused_tile_dev.h"

namespace egc {

// packed[column tile][k-step of 32][plane][lane][8]: lane 16 (k % 32 / 8) + column % 16 holds k = 32 s + 8 (lane / 16) ..+7 of
// column 16 ct + lane % 16 -- the B operand of v_mfma_f32_16x16x32_f16 as it is loaded, one KiB per (ct, s, plane).
// Column scales and the two planes as pack_f16x2_kernel (egc_gemm_f16x2.hip).
// (the operand's source: wcat [K][F_g + W] + bcat [W], or -- FtParamSrc -- the layer's parameters through the pack's index map)
struct FtWcatSrc {
  const float* wcat;
  const float* bcat;
  int ncol;
  __device__ inline float w(int k, int c) const { return wcat[(int64_t)k * ncol + c]; }
  __device__ inline float b(int j) const { return bcat != nullptr ? bcat[j] : 0.f; }
};
struct FtParamSrc {
  PackPtrs bases;
  const float* comb_w;
  const float* comb_b;      // the combination Linear's bias (its rows permuted as the weight's), or nullptr
  const float* bcat;        // or a bias already in the operand's order, or nullptr
  PackDims d;
  __device__ inline float w(int k, int c) const {
    const float* p = pack_param_ptr(bases, const_cast<float*>(comb_w), d, k, c);
    return p != nullptr ? *p : 0.f;
  }
  __device__ inline float b(int j) const { return comb_b != nullptr ? comb_b[pack_comb_row(d, j)] : (bcat != nullptr ? bcat[j] : 0.f); }
};
template <class S>
__device__ inline void ft_pack_column(int v, const S& src_of, int K, int F_g, int W, int ldb, ft_u16* __restrict__ packed) {
  const int lane = threadIdx.x;
  const int src = (v < F_g) ? v : ((v < ldb || v >= ldb + W) ? -1 : v - ldb + F_g);
  unsigned amax = 0;
  if (src >= 0)
    for (int k = lane; k < K; k += 64) amax = max(amax, __float_as_uint(src_of.w(k, src)) & 0x7fffffffu);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, d));
  unsigned be = amax >> 23;
  be = be > 253u ? 253u : be;
  const float scale = __uint_as_float((254u - be) << 23);
  const float inv = __uint_as_float(be << 23);
  for (int k = lane; k < FT_KP; k += 64) {
    const float w = (src >= 0 && k < K) ? src_of.w(k, src) * scale : 0.f;
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
    tail[FT_NV + v] = (wcol >= 0 && wcol < W) ? src_of.b(wcol) : 0.f;
  }
}
__global__ void __launch_bounds__(64) ft_pack_kernel(const float* __restrict__ wcat, const float* __restrict__ bcat, int K,
                                                      int F_g, int W, int ldb, ft_u16* __restrict__ packed) {
  ft_pack_column(blockIdx.x, FtWcatSrc{wcat, bcat, F_g + W}, K, F_g, W, ldb, packed);
}

// WIDE form: packed[32-column tile][k-step of 16][plane][lane][8] -- lane 32 (k % 16 / 8) + column % 32 holds k = 16 s + 8 (lane / 32) ..+7
// of column 32 ct + lane % 32: the B operand of v_mfma_f32_32x32x16_f16 as it is loaded, one KiB per (ct, s, plane), the k-steps
// of a tile contiguous (the kernel streams them in order).  Virtual columns: [bases 0 .. ldb) | padding to a multiple of 32 |
// weightings ldbp .. ldbp + W).  Tail: float col_inv[384], col_bias[384].
__global__ void __launch_bounds__(64) ft_pack_wide_kernel(const float* __restrict__ wcat, const float* __restrict__ bcat, int K,
                                                           int F_g, int W, int ldb, int ldbp, int n_ct, int k16,
                                                           ft_u16* __restrict__ packed) {
  const int v = blockIdx.x;
  const int lane = threadIdx.x;
  const int ncol = F_g + W;
  const int src = (v < F_g) ? v : ((v < ldbp || v >= ldbp + W) ? -1 : v - ldbp + F_g);
  unsigned amax = 0;
  if (src >= 0)
    for (int k = lane; k < K; k += 64) amax = max(amax, __float_as_uint(wcat[(int64_t)k * ncol + src]) & 0x7fffffffu);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, d));
  unsigned be = amax >> 23;
  be = be > 253u ? 253u : be;
  const float scale = __uint_as_float((254u - be) << 23);
  const float inv = __uint_as_float(be << 23);
  for (int k = lane; k < k16 * 16; k += 64) {
    const float w = (src >= 0 && k < K) ? wcat[(int64_t)k * ncol + src] * scale : 0.f;
    const _Float16 h = (_Float16)w;
    const _Float16 l = (_Float16)((w - (float)h) * 2048.f);
    const int64_t base = ((((int64_t)(v >> 5) * k16 + (k >> 4)) * 2) * 64 + 32 * ((k & 15) >> 3) + (v & 31)) * 8 + (k & 7);
    packed[base] = __builtin_bit_cast(ft_u16, h);
    packed[base + 64 * 8] = __builtin_bit_cast(ft_u16, l);
  }
  if (lane == 0) {
    float* tail = reinterpret_cast<float*>(packed + (int64_t)n_ct * k16 * 2 * 64 * 8);
    tail[v] = inv;
    const int wcol = v - ldbp;
    tail[FTW_MAX_CT * 32 + v] = (bcat != nullptr && wcol >= 0 && wcol < W) ? bcat[wcol] : 0.f;
  }
}

static bool ft_narrow_shape(const AggArgs& a, int f_in) {
  return f_in >= 4 && f_in <= FT_KP && (f_in & 3) == 0 && a.ldb + a.W <= FT_NV && a.slots <= 64 && a.A <= AMAX;
}
// the WIDE form's envelope: F_in <= 320, at most 12 column tiles of 32 (bases padded to a multiple of 32, then the weightings)
static bool ft_wide_shape(const AggArgs& a, int f_in) {
  if (a.slots > 64 && (a.slots > 128 || a.B * (((a.Ls >> 2) + 1) / 2) > 64)) return false;    // two passes of at most 64 lanes
  return f_in >= 4 && f_in <= FTW_MAX_FIN && (f_in & 3) == 0 && ((a.ldb + 31) & ~31) + a.W <= FTW_MAX_CT * 32 && a.A <= AMAX;
}
static inline int ftw_k16(int f_in) { return (((f_in + 15) / 16) + 3) & ~3; }   // k-steps of 16, padded (zero fragments) to the kernel's ring of four
static inline int ftw_n_ct(const AggArgs& a) { return (((a.ldb + 31) & ~31) + a.W + 31) / 32; }

size_t fused_tile_pack_bytes(const AggArgs& a, int f_in) {
  if (ft_narrow_shape(a, f_in)) return (size_t)FT_MFMA_WAVES * 4 * 2 * 64 * 8 * sizeof(ft_u16) + 2 * FT_NV * sizeof(float);
  if (ft_wide_shape(a, f_in)) return (size_t)ftw_n_ct(a) * ftw_k16(f_in) * 2 * 64 * 8 * sizeof(ft_u16) + 2 * FTW_MAX_CT * 32 * sizeof(float);
  return 0;
}

int fused_tile_pack(const AggArgs& a, const float* wcat, const float* bcat, int f_in, int f_g, int w_cols, int ldb, void* packed,
                    hipStream_t stream) {
  if (ft_narrow_shape(a, f_in)) {
    ft_pack_kernel<<<FT_NV, 64, 0, stream>>>(wcat, bcat, f_in, f_g, w_cols, ldb, (ft_u16*)packed);
    EGC_LAUNCH_CHECK("ft_pack_kernel");
    return EGC_OK;
  }
  if (!ft_wide_shape(a, f_in)) return EGC_ERR_UNSUPPORTED;
  const int n_ct = ftw_n_ct(a);
  ft_pack_wide_kernel<<<n_ct * 32, 64, 0, stream>>>(wcat, bcat, f_in, f_g, w_cols, ldb, (ldb + 31) & ~31, n_ct, ftw_k16(f_in),
                                                    (ft_u16*)packed);
  EGC_LAUNCH_CHECK("ft_pack_wide_kernel");
  return EGC_OK;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
constexpr size_t FT_LDS_BUDGET = 160 * 1024 - 256;

struct FtLds {
  size_t total;
  int off_rec, off_planes, off_rowinv, off_bases, off_wt;
  int off_col, off_rowptr, off_cnt, off_dis;   // the CSR areas of an even tile; csr_stride bytes further: those of an odd tile
  int csr_stride;
  int off_db, off_rowinv2;                     // backward form
};

static FtLds ft_lds(const AggArgs& a, int wl_floats, int tcap, int emax, bool with_post, bool wide = false, bool bwd = false) {
  FtLds L = {};
  auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const int bias_floats = (a.H * a.Ls + 3) & ~3;
  size_t at = up16((size_t)(with_post ? 2 : 1) * bias_floats * sizeof(float));
  L.off_rec = (int)at; at += 128;
  L.off_planes = (int)at; at += wide ? FTW_PLANES_BYTES : (bwd ? std::max(FT_PLANES_BYTES, FTB_PLANES_BYTES) : FT_PLANES_BYTES);
  L.off_rowinv = (int)at; at += wide ? up16(2 * FTW_CH * sizeof(float)) : up16(FT_PBUF * FT_CHUNK * sizeof(float));
  if (bwd) { L.off_rowinv2 = (int)at; at += up16(2 * FT_CHUNK * sizeof(float)); }
  L.off_bases = (int)at; at += up16((size_t)(tcap + 1) * a.ldb * 4);   // (+ the all-zero row absent entries read)
  L.off_wt = (int)at; at += up16((size_t)tcap * wl_floats * 4);
  if (bwd) { L.off_db = (int)at; at += up16((size_t)(tcap + 1) * a.ldb * 8); }   // d bases as 64-bit fixed point (+ a row absent entries would address)
  const size_t csr0 = at;
  L.off_col = (int)at; at += up16((size_t)emax * 2);
  L.off_rowptr = (int)at; at += up16((size_t)(tcap + 1) * 4);
  L.off_cnt = (int)at; at += up16((size_t)tcap * 4);      // (with rowptr: one 16-bit counter, then cursor, per CSR wavefront and row)
  L.off_dis = (int)at; at += up16((size_t)tcap * 4);
  L.csr_stride = (int)(at - csr0);
  at += L.csr_stride;       // the second set: tile it + 1's CSR is built while tile it's is being read
  L.total = at;
  return L;
}

bool fused_tile_shape(const AggArgs& a, int f_in) { return ft_narrow_shape(a, f_in) || ft_wide_shape(a, f_in); }

int fused_tile_quantum(const AggArgs& a, int f_in) { return ft_narrow_shape(a, f_in) ? FT_CHUNK : (ft_wide_shape(a, f_in) ? FTW_CH : 0); }

static inline int ftw_aw(const AggArgs& a) { return a.A >= 3 ? 4 : a.A; }

// rows of a tile whose image (bases + weightings + CSR areas for max_tile_edges entries) fits the LDS of a CU; 0 = none
int fused_tile_capacity(const AggArgs& a, int f_in, int max_tile_edges, bool with_post) {
  if (max_tile_edges < 0) return 0;
  int best = 0;
  if (ft_narrow_shape(a, f_in)) {
    const int wl = a.H * a.B * 4;
    for (int tcap = FT_CHUNK; tcap <= FT_CHUNK * FT_RING; tcap += FT_CHUNK) {
      if (ft_lds(a, wl, tcap, max_tile_edges, with_post).total <= FT_LDS_BUDGET) best = tcap; else break;
    }
  } else if (ft_wide_shape(a, f_in)) {
    const int wl = a.H * a.B * ftw_aw(a);
    for (int tcap = FTW_CH; tcap <= FTW_CH * FTW_MAXCH; tcap += FTW_CH) {
      if (ft_lds(a, wl, tcap, max_tile_edges, with_post, true).total <= FT_LDS_BUDGET) best = tcap; else break;
    }
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
#ifdef EGC_FT_STAMPS
  static unsigned long long* dbuf = nullptr;
  if (dbuf == nullptr) {
    hipMalloc(&dbuf, (256 * 9 + 96) * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(egc_ft_stamp_buf), &dbuf, sizeof(dbuf));
  }
  hipMemset(dbuf, 0, (256 * 9 + 96) * 8);
#endif
  fused_tile_kernel<LPR_LOG2, HPB, NEED, C><<<grid, FT_THREADS, lds, stream>>>(a, t);
  EGC_LAUNCH_CHECK("fused_tile_kernel");
#ifdef EGC_FT_STAMPS
  {
    hipDeviceSynchronize();
    static int calls = 0;
    if ((++calls % 40) == 0) {
      unsigned long long h[256 * 9 + 96];
      hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);
      double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tmax = 0;
      for (unsigned b = 0; b < grid; ++b) {
        for (int k = 0; k < 8; ++k) sum[k] += (double)h[b * 8 + k];
        tmax = std::max(tmax, (double)h[b * 8 + 7]);
      }
      for (int i = 0; i < 10; ++i)
        fprintf(stderr, "[ft tile %d of block 7] start %llu | helpers after the GEMM: plan (wavefront 15) %llu, counts %llu, rows requested %llu, CSR done %llu | GEMM %llu rows %llu end %llu\n", i, h[256 * 9 + i * 8], h[256 * 9 + i * 8 + 7],
                h[256 * 9 + i * 8 + 1], h[256 * 9 + i * 8 + 2], h[256 * 9 + i * 8 + 3], h[256 * 9 + i * 8 + 4], h[256 * 9 + i * 8 + 5], h[256 * 9 + i * 8 + 6]);
      double pro = 0;
      for (unsigned b = 0; b < grid; ++b) pro += (double)h[256 * 8 + b];
      fprintf(stderr, "[ft stamps] prologue %.0f; ", pro / grid);
      fprintf(stderr, "[ft stamps] grid %u, n_nodes %d: per workgroup (shader cycles): start %.0f  (-) %.0f  (-) %.0f  "
              "(-) %.0f  GEMM %.0f  rows %.0f  end barrier %.0f | total avg %.0f max %.0f\n", grid, a.n_nodes,
              sum[0] / grid, sum[1] / grid, sum[2] / grid, sum[3] / grid, sum[4] / grid, sum[5] / grid, sum[6] / grid, sum[7] / grid, tmax);
    }
  }
#endif
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
  const bool two_sets = a.slots > 64;
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
  const bool wide = !ft_narrow_shape(a, f_in);
  t.n_ct = wide ? ftw_n_ct(a) : (a.ldb + a.W + 15) / 16;
  t.tcap = tcap; t.emax = emax;
  a.w_aw = wide ? ftw_aw(a) : 4;
  t.w_aw = a.w_aw;
  t.wl_floats = a.H * a.B * a.w_aw;
  t.n_slabs = (f_in + FTW_SLAB - 1) / FTW_SLAB;
  t.k16 = ftw_k16(f_in);
  t.ldbp = (a.ldb + 31) & ~31;
  t.nsets = two_sets ? 2 : 1;
  t.p0 = ((a.Ls >> 2) + 1) / 2;
  t.magic0 = (unsigned)(((uint64_t)1 << 32) / (uint64_t)t.p0) + 1u;
  t.magic1 = (unsigned)(((uint64_t)1 << 32) / (uint64_t)std::max(1, (a.Ls >> 2) - t.p0)) + 1u;
  if (const char* e = getenv("EGC_FT_DBG")) t.dbg = atoi(e);     // (read by diagnostic builds of the kernel only: -DEGC_FT_STAMPS)
  if (tcap < FT_CHUNK || tcap > FT_CHUNK * FT_RING || (tcap % (wide ? FTW_CH : FT_CHUNK)) != 0 || emax < 0 || emax > 65535) return EGC_ERR_INVALID;   // (16-bit cursors of the CSR build)
  const FtLds L = ft_lds(a, t.wl_floats, tcap, emax, a.post_scale != nullptr, wide);
  if (L.total > FT_LDS_BUDGET) return EGC_ERR_UNSUPPORTED;
  t.off_rec = L.off_rec; t.off_planes = L.off_planes; t.off_rowinv = L.off_rowinv; t.off_bases = L.off_bases; t.off_wt = L.off_wt;
  t.off_col = L.off_col; t.off_rowptr = L.off_rowptr; t.off_cnt = L.off_cnt; t.off_dis = L.off_dis; t.csr_stride = L.csr_stride;
  // one workgroup per CU at most; fewer when the batch is small (a workgroup's share: at least ~16 nodes, at least one graph)
  int64_t grid = 256;
  if (const char* e = getenv("EGC_FT_GRID")) grid = std::max(1, atoi(e));
  grid = std::min<int64_t>(grid, std::max<int64_t>(1, n_graphs));
  grid = std::min<int64_t>(grid, std::max<int64_t>(1, (int64_t)a.n_nodes / 16));
  if (wide) {
    switch (t.n_slabs) {
      case 1: return launch_fused_tile_wide1(a, t, lpr, need, (unsigned)grid, L.total, stream);
      case 2: return launch_fused_tile_wide2(a, t, lpr, need, (unsigned)grid, L.total, stream);
      default: return launch_fused_tile_wide3(a, t, lpr, need, (unsigned)grid, L.total, stream);
    }
  }
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


// ---------------------------------------------------------------------------------------------
// The BACKWARD of the layer on batches of whole graphs, tile-local (fused_tile_kernel<..., MODE = 1>).  What autograd derives
// through layers.py:89-140 / optimized_layers.py:177-210 for a PyG batch -- in the reference: the backward of two Linears, of
// propagate's gathers and of the per-aggregator scatters -- as ONE launch per layer plus the weight gradient x^T d:
//   x rows -> [bases | w'] on the matrix cores -> LDS (as the forward) -> CSR of the tile -> per destination row the
//   aggregates again (with the entry attaining each maximum), d w' = <g, agg>, d agg = w' g, scattered to the sources' rows
//   of a d bases image kept as 64-bit fixed point (integer LDS atomics: order-independent sums) -> d x = [d bases | d w'] [bases_weight | comb_weight^T]^T on the matrix cores.
// x, grad_out and the edge list in; d x and d_cat = the gradient of [bases | pre-activation weightings] out.  No CSR, no
// transposed CSR, no `bases` / `weightings` / statistics in memory.  Envelope: the d = 128 / 64 layers (B = 4 bases of 16
// channels, H = 4 or 8, F_in <= 128), aggregators of sum / mean / max / symnorm, no weight nonlinearity.
// ---------------------------------------------------------------------------------------------
// packed_t[feature tile of 16][k-step of 32][plane][lane][8]: the B operand of d x = d W^T -- lane 16 (k % 32 / 8) + f % 16 holds
// W[f][k], k = 32 s + 8 (lane / 16) ..+7, where k runs over the LDS images' columns: [d bases 0 .. ldb) | d w' as [h][b][4]
// (a = 0 .. A - 1 real, the rest zero).  Scale per output feature f; tail: float col_inv[128].
template <class S>
__device__ inline void ft_pack_t_feature(int f, const S& src_of, int K, int F_g, int W, int A, int ldb, int k2,
                                         ft_u16* __restrict__ packed) {
  const int lane = threadIdx.x;       // f: output feature (row of wcat), 0 .. 127
  auto src_col = [&](int k) -> int {  // image column k -> column of wcat, or -1
    if (k < ldb) return k < F_g ? k : -1;
    const int j = k - ldb, hb = j >> 2, aa = j & 3;
    return (aa < A && hb * A + aa < W) ? F_g + hb * A + aa : -1;
  };
  unsigned amax = 0;
  if (f < K)
    for (int k = lane; k < k2; k += 64) {
      const int c = src_col(k);
      if (c >= 0) amax = max(amax, __float_as_uint(src_of.w(f, c)) & 0x7fffffffu);
    }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, d));
  unsigned be = amax >> 23;
  be = be > 253u ? 253u : be;
  const float scale = __uint_as_float((254u - be) << 23);
  const float inv = __uint_as_float(be << 23);
  for (int k = lane; k < 192; k += 64) {
    const int c = k < k2 ? src_col(k) : -1;
    const float w = (f < K && c >= 0) ? src_of.w(f, c) * scale : 0.f;
    const _Float16 h = (_Float16)w;
    const _Float16 l = (_Float16)((w - (float)h) * 2048.f);
    const int64_t base = ((((int64_t)(f >> 4) * 6 + (k >> 5)) * 2) * 64 + 16 * ((k & 31) >> 3) + (f & 15)) * 8 + (k & 7);
    packed[base] = __builtin_bit_cast(ft_u16, h);
    packed[base + 64 * 8] = __builtin_bit_cast(ft_u16, l);
  }
  if (lane == 0) reinterpret_cast<float*>(packed + (int64_t)8 * 6 * 2 * 64 * 8)[f] = inv;
}
__global__ void __launch_bounds__(64) ft_pack_t_kernel(const float* __restrict__ wcat, int K, int F_g, int W, int A, int ldb,
                                                        int k2, ft_u16* __restrict__ packed) {
  ft_pack_t_feature(blockIdx.x, FtWcatSrc{wcat, nullptr, F_g + W}, K, F_g, W, A, ldb, k2, packed);
}
// both operands of a training step in one launch: blocks 0 .. FT_NV - 1 the forward's columns, the next 128 the backward's features
__global__ void __launch_bounds__(64) ft_pack_both_kernel(const float* __restrict__ wcat, const float* __restrict__ bcat, int K,
                                                           int F_g, int W, int A, int ldb, int k2, ft_u16* __restrict__ packed,
                                                           ft_u16* __restrict__ packed_t) {
  const FtWcatSrc src{wcat, bcat, F_g + W};
  if (blockIdx.x < FT_NV) ft_pack_column(blockIdx.x, src, K, F_g, W, ldb, packed);
  else ft_pack_t_feature(blockIdx.x - FT_NV, src, K, F_g, W, A, ldb, k2, packed_t);
}
// ... straight from the layer's parameters (no wcat / bcat arrays, no egc_weights_pack_f32 launch in front)
__global__ void __launch_bounds__(64) ft_pack_both_params_kernel(FtParamSrc src, int K, int F_g, int W, int A, int ldb, int k2,
                                                                  ft_u16* __restrict__ packed, ft_u16* __restrict__ packed_t) {
  if (blockIdx.x < FT_NV) ft_pack_column(blockIdx.x, src, K, F_g, W, ldb, packed);
  else ft_pack_t_feature(blockIdx.x - FT_NV, src, K, F_g, W, A, ldb, k2, packed_t);
}

bool fused_tile_bwd_shape(const AggArgs& a, int f_in) {
  if (!ft_narrow_shape(a, f_in) || a.act != EGC_ACT_NONE) return false;
  if (a.B != 4 || a.L != 16 || a.Ls != 16 || a.ldb != 64 || (a.H != 4 && a.H != 8)) return false;
  for (int k = 0; k < a.A; ++k)
    if (a.aggr[k] != EGC_AGGR_SUM && a.aggr[k] != EGC_AGGR_MEAN && a.aggr[k] != EGC_AGGR_MAX && a.aggr[k] != EGC_AGGR_SYMNORM) return false;
  return true;
}

size_t fused_tile_bwd_pack_bytes() { return (size_t)8 * 6 * 2 * 64 * 8 * sizeof(ft_u16) + 128 * sizeof(float); }

int fused_tile_bwd_pack(const AggArgs& a, const float* wcat, int f_in, void* packed, hipStream_t stream) {
  ft_pack_t_kernel<<<128, 64, 0, stream>>>(wcat, f_in, a.B * a.Ls, a.W, a.A, a.ldb, a.ldb + a.H * a.B * 4, (ft_u16*)packed);
  EGC_LAUNCH_CHECK("ft_pack_t_kernel");
  return EGC_OK;
}

int fused_tile_train_pack(const AggArgs& a, const float* wcat, const float* bcat, int f_in, int f_g, int w_cols, int ldb, void* packed,
                          void* packed_t, hipStream_t stream) {
  if (!fused_tile_bwd_shape(a, f_in)) return EGC_ERR_UNSUPPORTED;
  ft_pack_both_kernel<<<FT_NV + 128, 64, 0, stream>>>(wcat, bcat, f_in, f_g, w_cols, a.A, ldb, a.ldb + a.H * a.B * 4, (ft_u16*)packed,
                                                      (ft_u16*)packed_t);
  EGC_LAUNCH_CHECK("ft_pack_both_kernel");
  return EGC_OK;
}

int fused_tile_train_pack_params(const AggArgs& a, const PackPtrs& bases, const float* comb_w, const float* comb_b, const float* bcat,
                                 const PackDims& d, int f_g, int w_cols, int ldb, void* packed, void* packed_t, hipStream_t stream) {
  if (!fused_tile_bwd_shape(a, d.F_in)) return EGC_ERR_UNSUPPORTED;
  FtParamSrc src{bases, comb_w, comb_b, bcat, d};
  ft_pack_both_params_kernel<<<FT_NV + 128, 64, 0, stream>>>(src, d.F_in, f_g, w_cols, a.A, ldb, a.ldb + a.H * a.B * 4, (ft_u16*)packed,
                                                             (ft_u16*)packed_t);
  EGC_LAUNCH_CHECK("ft_pack_both_params_kernel");
  return EGC_OK;
}

int fused_tile_bwd_capacity(const AggArgs& a, int f_in, int max_tile_edges) {
  if (!fused_tile_bwd_shape(a, f_in) || max_tile_edges < 0) return 0;
  int best = 0;
  // (the kernel keeps eight 16-row chunks of x in flight; six at H = 8, where the LDS image -- d bases next to bases and w' -- holds
  // no more than 96 rows anyway and the static configuration's helpers give the registers of the other two to their working set)
  for (int tcap = FT_CHUNK; tcap <= FT_CHUNK * (a.H == 8 ? 6 : 8); tcap += FT_CHUNK) {
    if (ft_lds(a, a.H * a.B * 4, tcap, max_tile_edges, false, false, true).total <= FT_LDS_BUDGET) best = tcap; else break;
  }
  return best;
}

template <class C>
static int launch_ftb_one(const AggArgs& a, const FusedTileArgs& t, unsigned grid, size_t lds, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_tile_kernel<4, 1, 0, C, 0, 1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_last_error("hipFuncSetAttribute(fused_tile_kernel, backward)", e); return EGC_ERR_HIP; }
    attr_set = true;
  }
#ifdef EGC_FT_STAMPS
  static unsigned long long* dbuf = nullptr;
  if (dbuf == nullptr) {
    hipMalloc(&dbuf, (256 * 9 + 96) * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(egc_ft_stamp_buf), &dbuf, sizeof(dbuf));
  }
  hipMemset(dbuf, 0, (256 * 9 + 96) * 8);
#endif
  fused_tile_kernel<4, 1, 0, C, 0, 1><<<grid, FT_THREADS, lds, stream>>>(a, t);
  EGC_LAUNCH_CHECK("fused_tile_kernel (backward)");
#ifdef EGC_FT_STAMPS
  {
    hipDeviceSynchronize();
    static int calls = 0;
    if ((++calls % 40) == 0) {
      unsigned long long h[256 * 9 + 96];
      hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);
      double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tmax = 0;
      for (unsigned b = 0; b < grid; ++b) {
        for (int k = 0; k < 8; ++k) sum[k] += (double)h[b * 8 + k];
        tmax = std::max(tmax, (double)h[b * 8 + 7]);
      }
      fprintf(stderr, "[ft bwd stamps] grid %u, n_nodes %d: per workgroup (shader cycles): start %.0f  GEMM1 %.0f  scale %.0f  "
              "rows (backward; wavefront 0) %.0f  B1 + GEMM2 %.0f | total avg %.0f max %.0f\n", grid, a.n_nodes,
              sum[0] / grid, sum[4] / grid, sum[1] / grid, sum[5] / grid, sum[6] / grid, sum[7] / grid, tmax);
    }
  }
#endif
  return EGC_OK;
}

int launch_fused_tile_bwd(AggArgs a, const int64_t* ptr, const int64_t* edge_ptr, int64_t n_graphs, const int64_t* src,
                          const int64_t* dst, int64_t n_edges, const int* max_index, const float* x, int f_in, const void* packed,
                          const void* packed_t, const float* grad_out, float* d_x, const float* d_x_add, float* d_cat, int ld_dcat, int tcap, int emax,
                          int32_t* status, int32_t* host_flag, hipStream_t stream) {
  if (!fused_tile_bwd_shape(a, f_in)) return EGC_ERR_UNSUPPORTED;
  a.lanes_pb = a.Ls / 4;
  a.magic_P = (unsigned)(((uint64_t)1 << 32) / (uint64_t)a.lanes_pb) + 1u;
  a.lpb_log2 = 2;
  a.need_mean = a.need_var = 0;
  for (int k = 0; k < a.A; ++k)
    if (a.aggr[k] == EGC_AGGR_MEAN) a.need_mean = 1;
  a.w_lds_stride = 0;
  a.bias_lds_floats = (a.H * a.Ls + 3) & ~3;
  a.lds_floats_per_wave = 0;
  a.w_aw = 4;
  FusedTileArgs t = {};
  t.ptr = ptr; t.edge_ptr = edge_ptr; t.n_graphs = n_graphs; t.src = src; t.dst = dst; t.n_edges = n_edges;
  t.max_index = max_index; t.status = status; t.host_flag = host_flag; t.x = x; t.packed = (const ft_u16*)packed;
  t.F_in = f_in;
  t.n_ct = (a.ldb + a.W + 15) / 16;
  t.tcap = tcap; t.emax = emax;
  t.w_aw = 4;
  t.wl_floats = a.H * a.B * 4;
  t.nsets = 1;
  t.grad_out = grad_out; t.d_x = d_x; t.d_x_add = d_x_add; t.d_cat = d_cat; t.ld_dcat = ld_dcat; t.packed_t = (const ft_u16*)packed_t;
  if (const char* e = getenv("EGC_FT_DBG")) t.dbg = atoi(e);     // (read by diagnostic builds of the kernel only: -DEGC_FT_STAMPS)
  if (tcap < FT_CHUNK || tcap > FT_CHUNK * (a.H == 8 ? 6 : 8) || (tcap % FT_CHUNK) != 0 || emax < 0 || emax > 16384) return EGC_ERR_INVALID;
  const FtLds L = ft_lds(a, t.wl_floats, tcap, emax, false, false, true);
  if (L.total > FT_LDS_BUDGET) return EGC_ERR_UNSUPPORTED;
  t.off_rec = L.off_rec; t.off_planes = L.off_planes; t.off_rowinv = L.off_rowinv; t.off_bases = L.off_bases; t.off_wt = L.off_wt;
  t.off_col = L.off_col; t.off_rowptr = L.off_rowptr; t.off_cnt = L.off_cnt; t.off_dis = L.off_dis; t.csr_stride = L.csr_stride;
  t.off_db = L.off_db; t.off_rowinv2 = L.off_rowinv2;
  int64_t grid = 256;
  if (const char* e = getenv("EGC_FT_GRID")) grid = std::max(1, atoi(e));
  grid = std::min<int64_t>(grid, std::max<int64_t>(1, n_graphs));
  grid = std::min<int64_t>(grid, std::max<int64_t>(1, (int64_t)a.n_nodes / 16));
  if (getenv("EGC_NO_STATIC_CFG") == nullptr && a.Ls == a.L) {
    // the row pass is bound by its vector instructions: with the layer's constants compiled in the aggregator switches, the
    // head count and the edge-set tests fold away (the run-time form is 2,000 instructions per turn, a quarter of them moves)
    constexpr int S = EGC_AGGR_SUM, M = EGC_AGGR_MEAN, X = EGC_AGGR_MAX, Y = EGC_AGGR_SYMNORM;
    unsigned pk = 0;
    for (int k = 0; k < a.A; ++k) pk |= (unsigned)a.aggr[k] << (3 * k);
    if (a.H == 8 && a.A == 4 && pk == agg_pack(S, M, X, Y) && a.x_looped && a.y_looped && a.loops_all)       // EGConv EGC-M north star
      return launch_ftb_one<StCfg<8, 4, 16, 4, agg_pack(S, M, X, Y), EGC_ACT_NONE, true, true, true>>(a, t, (unsigned)grid, L.total, stream);
    if (a.H == 8 && a.A == 3 && pk == agg_pack(Y, X, M) && !a.x_looped && a.y_looped && a.loops_all)          // EfficientGraphConv EGC-M at d = 128
      return launch_ftb_one<StCfg<8, 4, 16, 3, agg_pack(Y, X, M), EGC_ACT_NONE, false, true, true>>(a, t, (unsigned)grid, L.total, stream);
  }
  return launch_ftb_one<RtCfg>(a, t, (unsigned)grid, L.total, stream);
}

}  // namespace egc
