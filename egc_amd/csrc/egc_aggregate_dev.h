// Device-side building blocks shared by the generic and the register-resident EGC aggregate kernels.
#pragma once
#include <math.h>

#include "egc_common.h"

// hipcc defaults to -ffp-contract=fast, which would fuse the reference's separately rounded
// x*x / mean*mean products into FMAs (var of a single neighbour must be EXACTLY 0).  Every fused
// multiply-add in the aggregate kernels is an explicit fmaf().
#pragma clang fp contract(off)

namespace egc {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0xFFFFFFF0u;  // any offset >= num_records makes a buffer load return 0

// In-row arg positions in 8 bits (training forward -> backward).  The backward's source side compares, per transposed
// entry, the arg positions of the destination row with the entry's own position: as int32 that is a 256-byte row per
// entry at the north star; relative to the row's first entry a position fits one byte for all but hub rows and the
// gathered row shrinks to 64 bytes.  Encoding: 0 .. 253 = position - rowptr[row]; ARG8_NONE = no entry of the row
// can match (appended self-loop, empty row); ARG8_FAR = position >= rowptr[row] + ARG8_NONE: entries that far into a
// (hub) row, and only they, still compare against the int32 table.  The tables live behind the raw aggregates in
// the `stats` buffer of egc_aggregate_combine_train_f32: [N][stat_k][ldb] floats, then [N][ldb] bytes for max, then
// [N][ldb] bytes for min (each only if the layer has that aggregator).
constexpr unsigned ARG8_NONE = 254u, ARG8_FAR = 255u;
__device__ inline unsigned arg8_pack(int4 a, int start, int n_edges) {
  auto enc = [&](int p) -> unsigned {
    if (p < 0 || p >= n_edges) return ARG8_NONE;
    const unsigned rel = (unsigned)(p - start);
    return rel < ARG8_NONE ? rel : ARG8_FAR;
  };
  return enc(a.x) | (enc(a.y) << 8) | (enc(a.z) << 16) | (enc(a.w) << 24);
}

struct AggArgs {
  const int* rowptr;
  const int* col;
  const float* dis;        // deg^-1/2 of the symnorm edge set, or nullptr
  const float* edis;       // dis[col[p]] per CSR entry (egc_graph.edge_dis_*), or nullptr: gather dis[col[p]] instead
  const int* max_index;    // device scalar (used when !loops_all)
  const int* plan;
  const float* bases;
  const float* weightings;
  const float* bias;
  float* out;
  float* partial;          // [cap_chunks][5 (7 with arg tracking)][slots] float4
  int* partial_nself;      // [cap_chunks]
  int* counters;           // [cap_long] arrival counters of the fused kernel (zero on entry, zero on exit)
  int n_nodes;
  int row_begin, row_end;  // rows this launch finishes ([0, n_nodes) unless the caller splits the rows)
  int ldb, slots;          // slots = ldb / 4
  int F_out, W, H, B, A, L;
  int ldw;                 // floats between consecutive rows of `weightings` (>= W)
  int Ls;                  // floats between consecutive bases in a row (>= L; == L when contiguous)
  int aggr[EGC_MAX_AGGRS];
  int x_looped, y_looped, loops_all;
  int sa, sb;              // strides of (aggregator, basis) inside one head's weight block
  int act;
  int lpr_log2;
  unsigned magic_L;        // floor(2^32 / L) + 1: o / L == umulhi(o, magic_L) for o, L < 2^16 (L > 1)
  unsigned bases_bytes;
  int lds_floats_per_wave;
  // register-resident ("fast") kernel family only
  int lpb_log2;            // log2(lanes per basis block) = log2(L / 4), or -1: not a power-of-two layout
  int lanes_pb;            // L / 4
  unsigned magic_P;        // floor(2^32 / lanes_pb) + 1
  unsigned magic_P1;       // two-slots-per-lane kernel: the same for the second set (wide_p1)
  int hpg;                 // heads per lane group = ceil(H / G)
  int rows_per_wave;
  int chunk_blocks;        // leading blocks of the grid that take long-row chunks
  int need_mean, need_var;
  int var_ref;             // EGC_STDVAR_REFERENCE=1: var as the reference's float32 E[x^2] - E[x]^2 (no shift; egc_aggregate.hip)
  int n_chunks_hint;       // host-known number of long-row chunks, or -1 (launch for the capacity)
  int l4_off;              // two-slots-per-lane kernel: first slot (inside a basis) of the set being finished (else 0)
  int wide_p0, wide_p1;    // two-slots-per-lane kernel: slots per basis of the lane's first / second set (P0 + P1 = Ls / 4)
  int w_lds_stride;        // floats between the per-group weight strips in LDS
  int bias_lds_floats;     // floats of the per-wavefront bias strip in LDS
  // training forward only (egc_aggregate_combine_train_f32): the row's raw running aggregates, after the
  // self-loop term, as [n_nodes][stat_k][ldb], and its entry count -- what the backward needs instead of a
  // second gather.  stat_slot[s] = position of statistic s (STAT_*) inside a row's block, or -1.
  // fused caller-side epilogue (egc_post; all optional): out = act((z + bias) * post_scale + post_shift) + residual
  const float* post_scale;   // [F_out]
  const float* post_shift;   // [F_out]
  const float* residual;     // [n_nodes, F_out]
  int post_relu;
  float* stats;
  int* cnt_out;
  int stat_slot[5];
  int stat_k;
  // register-resident kernels, training forward of a layer with max / min: CSR position of the first entry
  // attaining the row's extremum per bases column ([n_nodes, ldb]; self_pos = the appended self-loop, -1 = empty
  // row), tracked inside the aggregation.  The generic kernels leave them to arg_extrema().
  int* arg_max;
  int* arg_min;
  unsigned* arg8_max;      // the same positions as bytes relative to the row's first entry (arg8_pack), or nullptr
  unsigned* arg8_min;
  int self_pos;              // = n_edges
  int w_aw;                  // egc_fused_tile_wide.hip: floats per (h, b) block of the weightings row it keeps in LDS (4, or A when A < 3)
  // (fields of the deleted fused-weightings launch of rounds 2-4; kept so that the aggregate workspace layout is unchanged)
  const float* x;            // [n_nodes, F_in]
  const float* wfrag;        // comb_weight^T as A fragments of v_mfma_f32_16x16x4_f32: [H][M][64 lanes][4]
  const float* wbias4;       // comb_bias laid out [H][B = 4][4] (zero beyond A)
  int F_in, M;               // M = ceil(F_in / 16) k-steps of 16
  int* queue;                // [FUSEDW_QUEUES + 1] work counters + exit counter (zero on entry, zero on exit)
};

enum { STAT_SUM = 0, STAT_SQ = 1, STAT_MX = 2, STAT_MN = 3, STAT_WS = 4 };

// Which raw statistics a layer's aggregator list needs (shared by the forward store and the backward load).
static inline int stat_layout(const int* aggr, int A, int (&slot)[5]) {
  bool need[5] = {false, false, false, false, false};
  for (int t = 0; t < A; ++t) {
    switch (aggr[t]) {
      case EGC_AGGR_SUM: case EGC_AGGR_MEAN: need[STAT_SUM] = true; break;
      case EGC_AGGR_VAR: case EGC_AGGR_STD: need[STAT_SUM] = need[STAT_SQ] = true; break;
      case EGC_AGGR_MAX: need[STAT_MX] = true; break;
      case EGC_AGGR_MIN: need[STAT_MN] = true; break;
      default: need[STAT_WS] = true; break;
    }
  }
  int k = 0;
  for (int s = 0; s < 5; ++s) slot[s] = need[s] ? k++ : -1;
  return k;
}

template <int CHUNKS>
struct Acc {
  f4 sum[CHUNKS], sq[CHUNKS], mx[CHUNKS], mn[CHUNKS], ws[CHUNKS];
  // sq = sum of (x - sh)^2 with sh = the row's first entry (set_shift, the same for every partial accumulator of a row):
  // var = E[(x - sh)^2] - (E[x] - sh)^2 -- layers.py:203-214's E[x^2] - E[x]^2 without its cancellation (FAcc::sh in
  // egc_aggregate_fast_dev.h has the reasoning)
  f4 sh[CHUNKS];
  __device__ inline void init() {
#pragma unroll
    for (int k = 0; k < CHUNKS; ++k) {
      sum[k] = 0.f; sq[k] = 0.f; ws[k] = 0.f; sh[k] = 0.f;
      mx[k] = -INFINITY; mn[k] = INFINITY;
    }
  }
};

__device__ inline f4 f4_max(f4 a, f4 b) {
  return f4{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)};
}
__device__ inline f4 f4_min(f4 a, f4 b) {
  return f4{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), fminf(a.w, b.w)};
}
__device__ inline f4 f4_fma(f4 a, f4 b, f4 c) {
  return f4{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)};
}
__device__ inline f4 f4_shfl_xor(f4 v, int off) {
  return f4{__shfl_xor(v.x, off), __shfl_xor(v.y, off), __shfl_xor(v.z, off), __shfl_xor(v.w, off)};
}

__device__ inline f4 load_slot(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0));
}

// Fold one gathered neighbour slot into the lane's running aggregates.
__device__ inline void fold(f4& sum, f4& sq, f4& mx, f4& mn, f4& ws, f4 v, bool in_x, bool in_y, float w, f4 sh) {
  const f4 vx = in_x ? v : f4{0.f, 0.f, 0.f, 0.f};
  sum += vx;
  // (x - sh)^2 rounded on its own, then added (as scatter(inputs * inputs) adds its squares, layers.py:206-212)
  const f4 d = in_x ? v - sh : f4{0.f, 0.f, 0.f, 0.f};
  sq += f4{__fmul_rn(d.x, d.x), __fmul_rn(d.y, d.y), __fmul_rn(d.z, d.z), __fmul_rn(d.w, d.w)};
  mx = f4_max(mx, in_x ? v : f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY});
  mn = f4_min(mn, in_x ? v : f4{INFINITY, INFINITY, INFINITY, INFINITY});
  const float wy = in_y ? w : 0.f;
  ws = f4_fma(f4{wy, wy, wy, wy}, v, ws);
}


__device__ inline f4 f4_div(f4 a, float d) { return f4{a.x / d, a.y / d, a.z / d, a.w / d}; }

// var = E[x^2] - E[x]^2 with separately rounded product and difference (layers.py:203-214).
__device__ inline f4 f4_var(f4 mean_sq, f4 mean) {
  return f4{__fsub_rn(mean_sq.x, __fmul_rn(mean.x, mean.x)), __fsub_rn(mean_sq.y, __fmul_rn(mean.y, mean.y)),
            __fsub_rn(mean_sq.z, __fmul_rn(mean.z, mean.z)), __fsub_rn(mean_sq.w, __fmul_rn(mean.w, mean.w))};
}
__device__ inline f4 f4_std(f4 var) {
  return f4{sqrtf(fmaxf(var.x, 0.f) + 1e-5f), sqrtf(fmaxf(var.y, 0.f) + 1e-5f), sqrtf(fmaxf(var.z, 0.f) + 1e-5f),
            sqrtf(fmaxf(var.w, 0.f) + 1e-5f)};
}

__device__ inline __amdgpu_buffer_rsrc_t bases_rsrc(const AggArgs& a) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)a.bases, 0, a.bases_bytes, 0x00020000);
}

// Launches the register-resident kernel family (egc_aggregate_fast.hip) if the layer shape qualifies.
// Returns EGC_OK, an error, or EGC_ERR_UNSUPPORTED when the generic path must be used instead.
bool fast_path_supported(const AggArgs& a, int layout, int chunks);
int launch_fast(AggArgs a, int64_t n_nodes, const PlanCaps& caps, hipStream_t stream);
// rows of 65..128 slots (the two ogbg-code nets): two slots per lane, short rows only (egc_aggregate_fast.hip)
bool wide_path_supported(const AggArgs& a, int layout);
int launch_wide_rows(AggArgs a, const PlanCaps& caps, hipStream_t stream);   // short rows AND long-row chunks: one launch

constexpr int FUSEDW_QUEUE_INTS = 16;  // (reserved words of the aggregate workspace: layout of rounds 2-4)

// egc_aggregate_tile.hip: batches of small graphs, tiles of whole graphs with the CSR built in LDS
int tile_capacity(const AggArgs& a, int tmax, int emax);
int launch_tile_plan(const int64_t* ptr, int64_t n_graphs, const int64_t* dst, int64_t n_edges, int64_t n_nodes, int slot,
                     int n_slots, int4* tiles, int* count, const int64_t* edge_ptr, hipStream_t stream);
int launch_tile_simple(AggArgs a, const int4* tiles, const int* n_tiles_dev, int n_tiles_bound, int tlds, int tmax, int emax,
                       const int64_t* src, const int64_t* dst, const int* max_index, int32_t* status, int32_t* host_flag,
                       hipStream_t stream);

// egc_fused_tile.hip: the whole layer on tiles of whole graphs in ONE launch (plan + GEMM + CSR + aggregate + combine)
bool fused_tile_shape(const AggArgs& a, int f_in);
int fused_tile_capacity(const AggArgs& a, int f_in, int max_tile_edges, bool with_post);
int fused_tile_quantum(const AggArgs& a, int f_in);      // rows per GEMM chunk: tile_nodes is a multiple of it (16 / 32; 0 = outside)
size_t fused_tile_pack_bytes(const AggArgs& a, int f_in);
int fused_tile_pack(const AggArgs& a, const float* wcat, const float* bcat, int f_in, int f_g, int w_cols, int ldb, void* packed,
                    hipStream_t stream);
int launch_fused_tile(AggArgs a, const int64_t* ptr, const int64_t* edge_ptr, int64_t n_graphs, const int64_t* src,
                      const int64_t* dst, int64_t n_edges, const int* max_index, const float* x, int f_in, const void* packed,
                      int tcap, int emax, int32_t* status, int32_t* host_flag, hipStream_t stream);

// the tile-local BACKWARD of the same launch (egc_fused_tile.hip: fused_tile_kernel<..., MODE = 1>)
bool fused_tile_bwd_shape(const AggArgs& a, int f_in);
int fused_tile_bwd_capacity(const AggArgs& a, int f_in, int max_tile_edges);
size_t fused_tile_bwd_pack_bytes();
int fused_tile_bwd_pack(const AggArgs& a, const float* wcat, int f_in, void* packed, hipStream_t stream);
struct PackPtrs;
struct PackDims;
int fused_tile_train_pack_params(const AggArgs& a, const PackPtrs& bases, const float* comb_w, const float* comb_b, const float* bcat,
                                 const PackDims& d, int f_g, int w_cols, int ldb, void* packed, void* packed_t, hipStream_t stream);
int fused_tile_train_pack(const AggArgs& a, const float* wcat, const float* bcat, int f_in, int f_g, int w_cols, int ldb, void* packed,
                          void* packed_t, hipStream_t stream);
int launch_fused_tile_bwd(AggArgs a, const int64_t* ptr, const int64_t* edge_ptr, int64_t n_graphs, const int64_t* src,
                          const int64_t* dst, int64_t n_edges, const int* max_index, const float* x, int f_in, const void* packed,
                          const void* packed_t, const float* grad_out, float* d_x, const float* d_x_add, float* d_cat, int ld_dcat, int tcap, int emax,
                          int32_t* status, int32_t* host_flag, hipStream_t stream);

}  // namespace egc
