// Register-resident EGC aggregate+combine kernel family (gfx950).
//
// Same contract as the generic kernels in egc_aggregate.hip (and the same reference call sites:
// layers.py:109-138,191-225; optimized_layers.py:183-278), specialised for
//     L a multiple of 4,  B a power of two,  S = B*L/4 <= 64 basis slots per row,  weightings laid out
//     [h][b][a],  A <= 4,  weight nonlinearity in {none, sigmoid, hardtanh}
// -- the north-star shape (d=128, H=8, B=4, A=4: S = 16, L = 16) and e.g. the reference's ogbn-mag layer
// (352/H8/B4: L = 44, S = 44) and its molhiv EGC-M layer (224/H4/B4: L = 56, S = 56).  A row occupies a lane
// group of LPR = 16 / 32 / 64 lanes (the power of two >= S; lanes S..LPR-1 idle); when L/4 is a power of
// two the lane <-> (basis, channel) arithmetic is shifts and the sum over bases a DPP / xor butterfly,
// otherwise a division and a rotation butterfly over the S live lanes.
//
// Work decomposition -- ONE launch, two roles selected by blockIdx:
//   * short rows (<= EGC_LONG_ROW_THRESHOLD entries): one LANE GROUP per row.  A basis row is LPR
//     16-byte slots, so a wavefront holds G = 64/LPR rows at once (4 at the north-star shape); every
//     wave-instruction gathers one neighbour row for each of its G rows, the running aggregates of a row
//     never leave its lane group (no cross-group merge), and the per-row fixed work (self-loop term,
//     finalisation, combine, store) is paid once per G rows.
//   * long rows: leading blocks reduce EGC_LONG_ROW_CHUNK-entry chunks with all G groups of a wavefront
//     splitting the entries; the wavefront that completes a row's last chunk (agent-scope release ->
//     arrival counter -> acquire; cdna guide Guideline 16) merges the partial records in chunk order
//     (deterministic) and finishes the row with the same epilogue.
// Epilogue (no LDS round trip for the aggregates): a lane holds 4 columns (one basis b, channels
// l..l+3) of every aggregator; for head h it forms sum_a w[h][b][a] * agg_a, a butterfly over the lanes
// that share l sums over b (DPP row rotations at L = 16), and the lane with b == h mod B keeps the
// result -> each lane ends up with ceil(H/B) 16-byte pieces of the output row.  The weightings row is
// staged through LDS once per row.
//
// The kernel is VALU-issue bound (~15 neighbours per row leave little to amortise the per-row work), so
//   * folds are straight-line: out-of-range buffer offsets return 0 (neutral for the sums) and the
//     extrema are updated under EXEC masking -- no select chains, no register copies at merges;
//   * every layer constant is read through a config accessor `C`.  `StCfg<...>` makes them compile-time
//     constants (aggregator list, head/basis counts, nonlinearity, edge-set flags) for a curated list of
//     layer configurations -- everything generic folds away; `RtCfg` reads them from the kernel
//     arguments and serves every other qualifying layer with the same code.
#include <stdlib.h>

#include "egc_aggregate_dev.h"

namespace egc {

constexpr int AMAX = 4;     // aggregators supported by the register-resident combine
// `out` is written once and read by a LATER kernel: non-temporal stores (aux bit 1) keep the rows from piling up
// as dirty lines in the XCDs' L2s, whose write-back at the end of the kernel otherwise costs ~3 us per launch.
constexpr int OUT_NT = 2;
#ifndef EGC_AGG_FU
#define EGC_AGG_FU 4
#endif
constexpr int FU = EGC_AGG_FU;  // neighbour-row loads in flight per lane group
constexpr int HPB_MAX = 4;  // ceil(H / B) supported
#ifndef EGC_AGG_WAVES
#define EGC_AGG_WAVES 6   // wavefronts per SIMD the inference variants are register-limited to
#endif

// Which optional running aggregates a layer needs (template mask: unused ones cost no registers).
constexpr int NEED_SQ = 1;   // sum of squares  (var, std)
constexpr int NEED_MN = 2;   // running minimum (min)
constexpr int NEED_ARG = 4;  // training forward: CSR position of the first entry attaining max (and min with NEED_MN)
constexpr int ARG_NONE = 0x7fffffff;  // "no entry yet": loses every position comparison

// ---------------------------------------------------------------------------------------------
// layer-constant accessors
// ---------------------------------------------------------------------------------------------
struct RtCfg {
  static __device__ inline int H(const AggArgs& a) { return a.H; }
  static __device__ inline int B(const AggArgs& a) { return a.B; }
  static __device__ inline int L(const AggArgs& a) { return a.L; }
  static __device__ inline int A(const AggArgs& a) { return a.A; }
  static __device__ inline int W(const AggArgs& a) { return a.W; }
  static __device__ inline int F_out(const AggArgs& a) { return a.F_out; }
  static __device__ inline int lpb_log2(const AggArgs& a) { return a.lpb_log2; }
  static __device__ inline bool pow2(const AggArgs& a) { return a.lpb_log2 >= 0; }
  static __device__ inline int slots(const AggArgs& a) { return a.slots; }
  static __device__ inline bool padded(const AggArgs& a) { return a.Ls != a.L; }  // bases padded to whole slots
  static __device__ inline int Ls(const AggArgs& a) { return a.Ls; }
  static __device__ inline int lanes_pb(const AggArgs& a) { return a.lanes_pb; }
  static __device__ inline int basis_of(const AggArgs& a, int q) { return (int)__umulhi((unsigned)q, a.magic_P); }
  static __device__ inline int act(const AggArgs& a) { return a.act; }
  static __device__ inline bool xl(const AggArgs& a) { return a.x_looped != 0; }
  static __device__ inline bool yl(const AggArgs& a) { return a.y_looped != 0; }
  static __device__ inline bool loops_all(const AggArgs& a) { return a.loops_all != 0; }
  static __device__ inline bool need_mean(const AggArgs& a) { return a.need_mean != 0; }
  static __device__ inline int aggr(const AggArgs& a, int t) { return a.aggr[t]; }
};

constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x >> 1); }

// AGG packs the aggregator codes, 3 bits each, first aggregator in the low bits.
// LS_ = floats between consecutive bases in a row (L_ rounded up to 4 when the layer pads them).
template <int H_, int B_, int L_, int A_, unsigned AGG, int ACT_, bool XL_, bool YL_, bool LOOPS_ALL_, int LS_ = L_>
struct StCfg {
  static constexpr int P_ = LS_ / 4;                       // lanes per basis
  static constexpr bool POW2_ = (P_ & (P_ - 1)) == 0;
  static constexpr int agg_at(int t) { return (int)((AGG >> (3 * t)) & 7u); }
  static constexpr bool has(int code) {
    for (int t = 0; t < A_; ++t)
      if (agg_at(t) == code) return true;
    return false;
  }
  static __device__ inline constexpr int H(const AggArgs&) { return H_; }
  static __device__ inline constexpr int B(const AggArgs&) { return B_; }
  static __device__ inline constexpr int L(const AggArgs&) { return L_; }
  static __device__ inline constexpr int A(const AggArgs&) { return A_; }
  static __device__ inline constexpr int W(const AggArgs&) { return H_ * B_ * A_; }
  static __device__ inline constexpr int F_out(const AggArgs&) { return H_ * L_; }
  static __device__ inline constexpr int lpb_log2(const AggArgs&) { return ilog2(P_); }
  static __device__ inline constexpr bool pow2(const AggArgs&) { return POW2_; }
  static __device__ inline constexpr int slots(const AggArgs&) { return B_ * P_; }
  static __device__ inline constexpr bool padded(const AggArgs&) { return LS_ != L_; }
  static __device__ inline constexpr int Ls(const AggArgs&) { return LS_; }
  static __device__ inline constexpr int lanes_pb(const AggArgs&) { return P_; }
  static __device__ inline constexpr int basis_of(const AggArgs&, int q) { return q / P_; }
  static __device__ inline constexpr int act(const AggArgs&) { return ACT_; }
  static __device__ inline constexpr bool xl(const AggArgs&) { return XL_; }
  static __device__ inline constexpr bool yl(const AggArgs&) { return YL_; }
  static __device__ inline constexpr bool loops_all(const AggArgs&) { return LOOPS_ALL_; }
  static __device__ inline constexpr bool need_mean(const AggArgs&) {
    return has(EGC_AGGR_MEAN) || has(EGC_AGGR_VAR) || has(EGC_AGGR_STD);
  }
  static __device__ inline constexpr int aggr(const AggArgs&, int t) { return agg_at(t); }
};

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ inline float vmax_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline float vmin_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline f4 f4_vmax(f4 a, f4 b) {
  return f4{vmax_raw(a.x, b.x), vmax_raw(a.y, b.y), vmax_raw(a.z, b.z), vmax_raw(a.w, b.w)};
}
__device__ inline f4 f4_vmin(f4 a, f4 b) {
  return f4{vmin_raw(a.x, b.x), vmin_raw(a.y, b.y), vmin_raw(a.z, b.z), vmin_raw(a.w, b.w)};
}
__device__ inline f4 f4_sqr_rn(f4 v) {
  return f4{__fmul_rn(v.x, v.x), __fmul_rn(v.y, v.y), __fmul_rn(v.z, v.z), __fmul_rn(v.w, v.w)};
}
__device__ inline f4 splat(float w) { return f4{w, w, w, w}; }

__device__ inline float bperm(int byte_addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}
__device__ inline int bperm(int byte_addr, int v) { return __builtin_amdgcn_ds_bpermute(byte_addr, v); }
__device__ inline f4 bperm(int byte_addr, f4 v) {
  return f4{bperm(byte_addr, v.x), bperm(byte_addr, v.y), bperm(byte_addr, v.z), bperm(byte_addr, v.w)};
}
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ inline i4 bperm(int byte_addr, i4 v) {
  return i4{bperm(byte_addr, v.x), bperm(byte_addr, v.y), bperm(byte_addr, v.z), bperm(byte_addr, v.w)};
}

// v += (v rotated by 8 lanes) ; v += (v rotated by 4 lanes), inside each 16-lane DPP row: afterwards
// every lane holds the sum over the 4 lanes {q, q^4, q^8, q^12}.  One VALU instruction per component
// and stage (the compiler otherwise emits v_mov + v_mov_dpp + v_add).  The leading s_nop covers the
// VALU-write -> DPP-read hazard of the inputs; inside the block 4 instructions separate each write
// from its DPP read.
__device__ inline void dpp_sum_over_4_bases(f4& v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf"
      : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

template <int NEED>
struct FAcc {
  f4 sum, mx, ws;
  f4 sq, mn;  // only touched when NEED says so; dead otherwise
  i4 ax, an;  // NEED_ARG: positions of the running max / min
  __device__ inline void init() {
    sum = 0.f; ws = 0.f; mx = -INFINITY;
    if constexpr (NEED & NEED_SQ) sq = 0.f;
    if constexpr (NEED & NEED_MN) mn = INFINITY;
    if constexpr (NEED & NEED_ARG) ax = an = ARG_NONE;
  }
};

// Running extremum with its position.  Strictly-better updates keep the FIRST entry attaining the extremum as
// long as positions arrive in increasing order (torch_scatter's arg rule); merges of independently built
// candidates compare (value, position) lexicographically.
__device__ inline void take_gt4(f4& m, i4& ar, f4 v, int pos) {
  const bool cx = v.x > m.x, cy = v.y > m.y, cz = v.z > m.z, cw = v.w > m.w;
  m = f4{cx ? v.x : m.x, cy ? v.y : m.y, cz ? v.z : m.z, cw ? v.w : m.w};
  ar = i4{cx ? pos : ar.x, cy ? pos : ar.y, cz ? pos : ar.z, cw ? pos : ar.w};
}
__device__ inline void take_lt4(f4& m, i4& ar, f4 v, int pos) {
  const bool cx = v.x < m.x, cy = v.y < m.y, cz = v.z < m.z, cw = v.w < m.w;
  m = f4{cx ? v.x : m.x, cy ? v.y : m.y, cz ? v.z : m.z, cw ? v.w : m.w};
  ar = i4{cx ? pos : ar.x, cy ? pos : ar.y, cz ? pos : ar.z, cw ? pos : ar.w};
}
__device__ inline void merge_gt4(f4& m, i4& ar, f4 om, i4 oa) {
  const bool cx = om.x > m.x || (om.x == m.x && oa.x < ar.x), cy = om.y > m.y || (om.y == m.y && oa.y < ar.y);
  const bool cz = om.z > m.z || (om.z == m.z && oa.z < ar.z), cw = om.w > m.w || (om.w == m.w && oa.w < ar.w);
  m = f4{cx ? om.x : m.x, cy ? om.y : m.y, cz ? om.z : m.z, cw ? om.w : m.w};
  ar = i4{cx ? oa.x : ar.x, cy ? oa.y : ar.y, cz ? oa.z : ar.z, cw ? oa.w : ar.w};
}
__device__ inline void merge_lt4(f4& m, i4& ar, f4 om, i4 oa) {
  const bool cx = om.x < m.x || (om.x == m.x && oa.x < ar.x), cy = om.y < m.y || (om.y == m.y && oa.y < ar.y);
  const bool cz = om.z < m.z || (om.z == m.z && oa.z < ar.z), cw = om.w < m.w || (om.w == m.w && oa.w < ar.w);
  m = f4{cx ? om.x : m.x, cy ? om.y : m.y, cz ? om.z : m.z, cw ? om.w : m.w};
  ar = i4{cx ? oa.x : ar.x, cy ? oa.y : ar.y, cz ? oa.z : ar.z, cw ? oa.w : ar.w};
}

// Fold one gathered slot.  `v` is 0 where the entry is absent or excluded (out-of-range buffer offset),
// which is neutral for the sums, so only the extrema need the lane mask -- applied through EXEC
// (a divergent `if`), which costs two scalar instructions and no register copies.
template <int NEED>
__device__ inline void fold(FAcc<NEED>& acc, f4 v, float w, bool in_x, int pos) {
  acc.sum += v;
  acc.ws = f4_fma(splat(w), v, acc.ws);
  if constexpr (NEED & NEED_SQ) acc.sq += f4_sqr_rn(v);
  if (in_x) {
    if constexpr (NEED & NEED_ARG) {
      take_gt4(acc.mx, acc.ax, v, pos);
      if constexpr (NEED & NEED_MN) take_lt4(acc.mn, acc.an, v, pos);
    } else {
      acc.mx = f4_vmax(acc.mx, v);
      if constexpr (NEED & NEED_MN) acc.mn = f4_vmin(acc.mn, v);
    }
  }
}

// Merge the G lane groups (every lane ends with the aggregates of its slot over all entries).
template <int LPR_LOG2, int NEED>
__device__ inline void all_reduce_groups(FAcc<NEED>& acc, int lane) {
#pragma unroll
  for (int off = 1 << LPR_LOG2; off < 64; off <<= 1) {
    const int addr = (lane ^ off) << 2;
    acc.sum += bperm(addr, acc.sum);
    acc.ws += bperm(addr, acc.ws);
    if constexpr (NEED & NEED_SQ) acc.sq += bperm(addr, acc.sq);
    if constexpr (NEED & NEED_ARG) {
      merge_gt4(acc.mx, acc.ax, bperm(addr, acc.mx), bperm(addr, acc.ax));
      if constexpr (NEED & NEED_MN) merge_lt4(acc.mn, acc.an, bperm(addr, acc.mn), bperm(addr, acc.an));
    } else {
      acc.mx = f4_vmax(acc.mx, bperm(addr, acc.mx));
      if constexpr (NEED & NEED_MN) acc.mn = f4_vmin(acc.mn, bperm(addr, acc.mn));
    }
  }
}

// 16-byte load that bypasses L1/L2 residency (sc0 sc1): used for records other wavefronts just published.
__device__ inline f4 load_slot_wt(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0x11));
}

struct FastRsrc {
  __amdgpu_buffer_rsrc_t bases, out, res;
};

// Issue + fold one batch of FU wave-instructions.  `addr0` is the ds_bpermute byte address of the lane
// holding this lane's first entry of the batch, `step` the byte distance to the next one; entry u of
// the batch is valid iff first + u * vstep < n_valid (lane-dependent), and its CSR position is
// pos_base + first + u * vstep; jj / dd are the staged source ids / deg^-1/2.
template <int NEED, class C>
__device__ inline void gather_batch(const AggArgs& a, const FastRsrc& R, FAcc<NEED>& acc, int addr0, int step, int row,
                                    int jj, float dd, float dis_i, int n_valid, int first, int vstep,
                                    unsigned row_bytes, unsigned slot_off, int pos_base) {
  f4 v[FU];
  float w[FU];
  bool in_x[FU];
#pragma unroll
  for (int u = 0; u < FU; ++u) {
    const int addr = addr0 + u * step;
    const int j = bperm(addr, jj);
    const bool is_self = j == row;
    in_x[u] = (first + u * vstep < n_valid) && !(C::xl(a) && is_self);
    v[u] = load_slot(R.bases, in_x[u] ? (unsigned)j * row_bytes + slot_off : OOB);
    w[u] = bperm(addr, dd) * dis_i;
    if (C::yl(a) && !C::xl(a)) w[u] = is_self ? 0.f : w[u];  // mixed sets: self-entry counts for sum/max only
  }
#pragma unroll
  for (int u = 0; u < FU; ++u) fold<NEED>(acc, v[u], w[u], in_x[u], pos_base + first + u * vstep);
}

// ---------------------------------------------------------------------------------------------
// Epilogue shared by both roles.  Per lane group: `row` (same in all lanes of the group), the group's
// merged aggregates, its entry count `deg` and self-entry count `nself`; `store` masks the output.
// ---------------------------------------------------------------------------------------------
template <int LPR_LOG2, int HPB, int NEED, class C>
__device__ inline void finish_group(const AggArgs& a, const FastRsrc& R, int lane, int row, bool row_ok, FAcc<NEED>& acc,
                                    int deg, int nself, float dis_i, f4 vself, bool has_self, const f4 (&wpre)[2],
                                    bool store, float* lds_w, const float* lds_bias, const float* lds_scale) {
  constexpr int LPR = 1 << LPR_LOG2;
  const int g = lane >> LPR_LOG2;
  const int q = lane & (LPR - 1);
  const int A = C::A(a), B = C::B(a), H = C::H(a), W = C::W(a);
  // lane q <-> (basis b, channels 4 l4 .. 4 l4 + 3); lanes q >= S hold no slot
  int b, l4;
  const bool live = q < C::slots(a);
  if (C::pow2(a)) {
    b = min(q >> C::lpb_log2(a), B - 1);
    l4 = q & ((1 << C::lpb_log2(a)) - 1);
  } else {
    b = min(C::basis_of(a, q), B - 1);
    l4 = q - b * C::lanes_pb(a);
  }

  // (1) the row's weightings (nonlinearity applied) -> this group's LDS strip, 32 bytes per lane
  float* wl = lds_w + g * a.w_lds_stride;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c0 = (q + k * LPR) * 4;
    if (c0 < W) {
      f4 t = wpre[k];
      if (C::act(a) == EGC_ACT_SIGMOID) {
        t = f4{1.0f / (1.0f + expf(-t.x)), 1.0f / (1.0f + expf(-t.y)), 1.0f / (1.0f + expf(-t.z)),
               1.0f / (1.0f + expf(-t.w))};
      } else if (C::act(a) == EGC_ACT_HARDTANH) {
        t = f4{fminf(fmaxf(t.x, -1.f), 1.f), fminf(fmaxf(t.y, -1.f), 1.f), fminf(fmaxf(t.z, -1.f), 1.f),
               fminf(fmaxf(t.w, -1.f), 1.f)};
      }
      *reinterpret_cast<f4*>(wl + c0) = t;
    }
  }

  // (2) self-loop term and aggregator finalisation
  int cnt = deg;
  if (C::xl(a)) cnt = deg - nself + (has_self ? 1 : 0);
  if (C::xl(a)) {
    fold<NEED>(acc, vself, dis_i * dis_i, has_self, a.self_pos);  // vself is 0 where the row has no self-loop
  } else if (C::yl(a)) {
    acc.ws = f4_fma(splat(dis_i * dis_i), vself, acc.ws);
  }
  if (a.stats != nullptr && store && row_ok && live) {  // training forward: keep the raw aggregates for the backward
    float* st = a.stats + ((int64_t)row * a.stat_k) * a.ldb + 4 * q;
    if (a.stat_slot[STAT_SUM] >= 0) __builtin_nontemporal_store(acc.sum, reinterpret_cast<f4*>(st + a.stat_slot[STAT_SUM] * a.ldb));
    if (a.stat_slot[STAT_MX] >= 0) __builtin_nontemporal_store(acc.mx, reinterpret_cast<f4*>(st + a.stat_slot[STAT_MX] * a.ldb));
    if (a.stat_slot[STAT_WS] >= 0) __builtin_nontemporal_store(acc.ws, reinterpret_cast<f4*>(st + a.stat_slot[STAT_WS] * a.ldb));
    if constexpr (NEED & NEED_SQ)
      if (a.stat_slot[STAT_SQ] >= 0) __builtin_nontemporal_store(acc.sq, reinterpret_cast<f4*>(st + a.stat_slot[STAT_SQ] * a.ldb));
    if constexpr (NEED & NEED_MN)
      if (a.stat_slot[STAT_MN] >= 0) __builtin_nontemporal_store(acc.mn, reinterpret_cast<f4*>(st + a.stat_slot[STAT_MN] * a.ldb));
    if (q == 0) a.cnt_out[row] = cnt;
    if constexpr (NEED & NEED_ARG) {
      // first position attaining the extremum; self_pos = the appended self-loop; -1 for an empty row
      const int none = cnt > 0 ? a.self_pos : -1;
      const int64_t ao = (int64_t)row * a.ldb + 4 * q;
      if (a.arg_max != nullptr) {
        const i4 r = i4{acc.ax.x == ARG_NONE ? none : acc.ax.x, acc.ax.y == ARG_NONE ? none : acc.ax.y,
                        acc.ax.z == ARG_NONE ? none : acc.ax.z, acc.ax.w == ARG_NONE ? none : acc.ax.w};
        __builtin_nontemporal_store(r, reinterpret_cast<i4*>(a.arg_max + ao));
      }
      if constexpr (NEED & NEED_MN)
        if (a.arg_min != nullptr) {
          const i4 r = i4{acc.an.x == ARG_NONE ? none : acc.an.x, acc.an.y == ARG_NONE ? none : acc.an.y,
                          acc.an.z == ARG_NONE ? none : acc.an.z, acc.an.w == ARG_NONE ? none : acc.an.w};
          __builtin_nontemporal_store(r, reinterpret_cast<i4*>(a.arg_min + ao));
        }
    }
  }
  const float cntf = (float)max(cnt, 1);
  const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
  f4 mean = zero, var = zero;
  if constexpr (NEED & NEED_SQ) {
    // exact divisions: var of identical neighbours must cancel to exactly 0 (see egc_aggregate_dev.h)
    mean = f4_div(acc.sum, cntf);
    var = f4_var(f4_div(acc.sq, cntf), mean);
  } else if (C::need_mean(a)) {
    // no var/std in this layer: one reciprocal instead of four IEEE divisions (<= 1 ulp from sum / cnt)
    mean = acc.sum * splat(__builtin_amdgcn_rcpf(cntf));
  }
  const bool nonempty = cnt > 0;
  f4 val[AMAX];
#pragma unroll
  for (int t = 0; t < AMAX; ++t) {
    val[t] = zero;
    if (t < A) {  // wave-uniform
      switch (C::aggr(a, t)) {
        case EGC_AGGR_SUM: val[t] = acc.sum; break;
        case EGC_AGGR_MEAN: val[t] = mean; break;
        case EGC_AGGR_MAX: val[t] = nonempty ? acc.mx : zero; break;
        case EGC_AGGR_MIN: if constexpr (NEED & NEED_MN) val[t] = nonempty ? acc.mn : zero; break;
        case EGC_AGGR_VAR: val[t] = var; break;
        case EGC_AGGR_STD: val[t] = f4_std(var); break;
        default: val[t] = acc.ws; break;  // EGC_AGGR_SYMNORM
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // weight strip visible to the whole wavefront

  // (3) combine: for head h = hb*B + bb every lane forms its (b, l..l+3) share, the butterfly sums over
  //     b, and the lane with b == bb keeps the head's 4 channels
  f4 o[HPB];
#pragma unroll
  for (int hb = 0; hb < HPB; ++hb) {
    o[hb] = zero;
#pragma unroll 4
    for (int bb = 0; bb < B; ++bb) {
      const int h = hb * B + bb;
      if (h >= H) break;  // wave-uniform
      const float* wp = wl + (h * B + b) * A;
      f4 part;
      if (A == 4) {  // wave-uniform
        const f4 wv = *reinterpret_cast<const f4*>(wp);
        part = val[0] * splat(wv.x);
        part = f4_fma(splat(wv.y), val[1], part);
        part = f4_fma(splat(wv.z), val[2], part);
        part = f4_fma(splat(wv.w), val[3], part);
      } else {
        part = val[0] * splat(wp[0]);
        if (A > 1) part = f4_fma(splat(wp[1]), val[1], part);
        if (A > 2) part = f4_fma(splat(wp[2]), val[2], part);
      }
      if (!C::pow2(a)) {
        // rotation butterfly over the S live lanes of the group: after log2(B) steps every lane holds the sum
        // over the B lanes that share its l4
        for (int rot = C::lanes_pb(a); rot < C::slots(a); rot <<= 1) {
          int src = q + rot;
          src = src >= C::slots(a) ? src - C::slots(a) : src;
          part += bperm(((g << LPR_LOG2) + (live ? src : q)) << 2, part);
        }
      } else if (LPR == 16 && C::lpb_log2(a) == 2 && C::slots(a) == 16) {
        dpp_sum_over_4_bases(part);  // 4 bases x 4 slots inside one 16-lane DPP row: no LDS traffic
      } else {
        for (int off = 1 << C::lpb_log2(a); off < C::slots(a); off <<= 1) part += bperm((lane ^ off) << 2, part);
      }
      if (b == bb) o[hb] = part;
    }
  }
  // (4) bias + store: lane (b, l4) owns out[row, (hb*B + b)*L + 4*l4 ..+3]
  const unsigned orow = (unsigned)row * (unsigned)C::F_out(a) * 4u;
#pragma unroll
  for (int hb = 0; hb < HPB; ++hb) {
    const int h = hb * B + b;
    const bool mine = store && row_ok && live && h < H;
    const int oc = h * C::L(a) + 4 * l4;
    if (!C::padded(a)) {
      f4 r = o[hb];
      if (a.post_scale != nullptr) r = r * *reinterpret_cast<const f4*>(lds_scale + (mine ? oc : 0));
      r = r + *reinterpret_cast<const f4*>(lds_bias + (mine ? oc : 0));
      if (a.post_relu) r = f4{fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f), fmaxf(r.w, 0.f)};
      if (a.residual != nullptr) r = r + load_slot(R.res, mine ? orow + (unsigned)oc * 4u : OOB);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, r), R.out, mine ? orow + (unsigned)oc * 4u : OOB, 0, OUT_NT);
    } else {
      // padded bases (L % 4 != 0): the bias strip is padded the same way, head rows are only 4-byte aligned
      // and the last slot of a head is ragged -> four dword stores, out-of-range where the channel does not exist
      f4 r = o[hb];
      if (a.post_scale != nullptr) r = r * *reinterpret_cast<const f4*>(lds_scale + (mine ? h * C::Ls(a) + 4 * l4 : 0));
      r = r + *reinterpret_cast<const f4*>(lds_bias + (mine ? h * C::Ls(a) + 4 * l4 : 0));
      if (a.post_relu) r = f4{fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f), fmaxf(r.w, 0.f)};
      const int left = mine ? C::L(a) - 4 * l4 : 0;
      const unsigned base = orow + (unsigned)oc * 4u;
      float rx = r.x, ry = r.y, rz = r.z, rw = r.w;  // (bit_cast of a vector-element expression picks element 0)
      if (a.residual != nullptr) {
        const float* rr = a.residual + (int64_t)(mine ? row : 0) * C::F_out(a) + (mine ? oc : 0);
        rx += left > 0 ? rr[0] : 0.f; ry += left > 1 ? rr[1] : 0.f; rz += left > 2 ? rr[2] : 0.f; rw += left > 3 ? rr[3] : 0.f;
      }
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rx), R.out, left > 0 ? base : OOB, 0, OUT_NT);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ry), R.out, left > 1 ? base + 4u : OOB, 0, OUT_NT);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rz), R.out, left > 2 ? base + 8u : OOB, 0, OUT_NT);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rw), R.out, left > 3 ? base + 12u : OOB, 0, OUT_NT);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // strip reads done before the next row overwrites it
}

// Row-only operands of the epilogue (weightings row as 2 x 16 bytes per lane, own basis slot).
template <int LPR_LOG2, class C>
__device__ inline void load_row_operands(const AggArgs& a, const FastRsrc& R, int lane, int row, bool row_ok,
                                         f4 (&wpre)[2], f4& vself, bool& has_self) {
  constexpr int LPR = 1 << LPR_LOG2;
  const int q = lane & (LPR - 1);
  const int W = C::W(a);
  // row_ok already says row < n_nodes: with a loop on every node there is nothing left to compare
  has_self = row_ok && (C::loops_all(a) || row <= *a.max_index);
  const bool want_self = (C::xl(a) || C::yl(a)) && has_self;
  vself = load_slot(R.bases, (want_self && q < C::slots(a)) ? (unsigned)row * (unsigned)a.ldb * 4u + (unsigned)q * 16u : OOB);
  const float* wrow = a.weightings + (int64_t)(row_ok ? row : 0) * a.ldw;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c0 = (q + k * LPR) * 4;
    wpre[k] = f4{0.f, 0.f, 0.f, 0.f};
    if (c0 + 3 < W) wpre[k] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(wrow + c0));  // read once: leave L2 to `bases`
    else if (c0 < W) {  // W not a multiple of 4: ragged last piece
      wpre[k].x = wrow[c0];
      if (c0 + 1 < W) wpre[k].y = wrow[c0 + 1];
      if (c0 + 2 < W) wpre[k].z = wrow[c0 + 2];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
template <int LPR_LOG2, int HPB, int NEED, class C>
// Inference variants (NEED == 0) fit 80 VGPRs without spilling when asked to, which buys the sixth wavefront per
// SIMD; the variants carrying more running aggregates are left to the register allocator.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NEED == 0 ? EGC_AGG_WAVES : ((NEED & NEED_ARG) && (NEED & NEED_SQ)) ? 2 : 4)))
agg_fast_kernel(AggArgs a) {
  constexpr int LPR = 1 << LPR_LOG2, G = 64 / LPR;
  extern __shared__ float smem[];
  if ((int)blockIdx.x < a.chunk_blocks && (int)blockIdx.x * 4 >= a.plan[1]) return;  // unused chunk slots
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int g = lane >> LPR_LOG2;
  const int q = lane & (LPR - 1);
  const unsigned slot_off = (unsigned)q * 16u;
  const unsigned row_bytes = (unsigned)a.ldb * 4u;
  const bool lane_live = q < C::slots(a);  // lanes beyond the row's S slots gather nothing
  const int F_out = C::F_out(a);
  // per-wavefront LDS: [bias F_out][G weight strips]
  float* lds_bias = smem + wave * a.lds_floats_per_wave;
  // with a fused post-op the strip holds bias * scale + shift and a second strip the scale itself
  const bool post = a.post_scale != nullptr;
  float* lds_scale = lds_bias + a.bias_lds_floats;
  float* lds_w = lds_bias + (post ? 2 : 1) * a.bias_lds_floats;
  for (int o = lane; o < C::H(a) * C::Ls(a); o += 64) {  // padded head layout [h][Ls] (== [F_out] when contiguous)
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
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  FastRsrc R;
  R.bases = bases_rsrc(a);
  R.out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (unsigned)a.n_nodes * (unsigned)F_out * 4u, 0x00020000);
  R.res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual != nullptr ? a.residual : a.out), 0,
                                            (unsigned)a.n_nodes * (unsigned)F_out * 4u, 0x00020000);
  const bool looped_any = C::xl(a) || C::yl(a);

  if ((int)blockIdx.x < a.chunk_blocks) {
    // ---------------- long-row chunk role: the G groups split one chunk's entries ----------------
    const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (c >= a.plan[1]) return;
    const int cap_long = a.plan[2], cap_chunks = a.plan[3];
    const int* long_row = a.plan + 4;
    const int* long_chunk0 = long_row + cap_long;
    const int* chunk_slot = long_chunk0 + cap_long;
    const int* chunk_begin = chunk_slot + cap_chunks;
    const int slot = __builtin_amdgcn_readfirstlane(chunk_slot[c]);
    const int row = __builtin_amdgcn_readfirstlane(long_row[slot]);
    if (row < a.row_begin || row >= a.row_end) return;  // every chunk of a row takes the same exit
    const int start = __builtin_amdgcn_readfirstlane(chunk_begin[c]);
    const int row_start = __builtin_amdgcn_readfirstlane(a.rowptr[row]);
    const int row_end = __builtin_amdgcn_readfirstlane(a.rowptr[row + 1]);
    const int end = min(start + EGC_LONG_ROW_CHUNK, row_end);
    const int deg = row_end - row_start;
    const int nch = (deg + EGC_LONG_ROW_CHUNK - 1) / EGC_LONG_ROW_CHUNK;
    const float dis_i = a.dis != nullptr ? a.dis[row] : 0.f;
    FAcc<NEED> acc;
    acc.init();
    int nself = 0;
    for (int base = start; base < end; base += 64) {
      const int p = base + lane;
      const bool pv = p < end;
      const int jj = pv ? a.col[p] : row;
      // source-side deg^-1/2: streamed per entry when the graph carries it, else gathered
      const float dd = a.edis != nullptr ? (pv ? a.edis[p] : 0.f) : (a.dis != nullptr ? a.dis[jj] : 0.f);
      if (looped_any) nself += __popcll(__ballot(pv && jj == row));
      const int cnt = min(64, end - base);
      for (int t0 = 0; t0 < cnt; t0 += FU * G)
        gather_batch<NEED, C>(a, R, acc, (g + t0) << 2, G << 2, row, jj, dd, dis_i, lane_live ? cnt : 0, t0 + g, G, row_bytes,
                              slot_off, base);
    }
    all_reduce_groups<LPR_LOG2, NEED>(acc, lane);
    if (nch > 1) {
      // Publish this chunk's record with write-through (sc0 sc1) stores and drain them before the
      // arrival counter is bumped: the data then sits at the memory side without an agent-scope release
      // fence -- a whole-L2 write-back that costs tens of microseconds when hundreds of chunks publish
      // (cdna guide, Guideline 16 "valid forms": sc1 stores + drained + counter; consumer keeps its acquire).
      constexpr int WT = 0x11;  // aux bits: sc0 | sc1
      constexpr int REC = (NEED & NEED_ARG) ? 7 : 5;  // 16-byte slots per lane in a chunk record (workspace holds 7)
      const __amdgpu_buffer_rsrc_t pw = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(reinterpret_cast<f4*>(a.partial) + (int64_t)c * REC * LPR), 0, (unsigned)REC * LPR * 16u, 0x00020000);
      const unsigned po = (g == 0 && lane_live) ? (unsigned)q * 16u : OOB;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, acc.sum), pw, po, 0 * LPR * 16, WT);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, acc.mx), pw, po, 2 * LPR * 16, WT);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, acc.ws), pw, po, 4 * LPR * 16, WT);
      if constexpr (NEED & NEED_SQ)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, acc.sq), pw, po, 1 * LPR * 16, WT);
      if constexpr (NEED & NEED_MN)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, acc.mn), pw, po, 3 * LPR * 16, WT);
      if constexpr (NEED & NEED_ARG) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, acc.ax), pw, po, 5 * LPR * 16, WT);
        if constexpr (NEED & NEED_MN)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, acc.an), pw, po, 6 * LPR * 16, WT);
      }
      const __amdgpu_buffer_rsrc_t pn =
          __builtin_amdgcn_make_buffer_rsrc((void*)(a.partial_nself + c), 0, 4u, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b32(nself, pn, lane == 0 ? 0u : OOB, 0, WT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int arrived = 0;
      if (lane == 0) arrived = __hip_atomic_fetch_add(&a.counters[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      arrived = __builtin_amdgcn_readfirstlane(arrived);
      if (arrived != nch - 1) return;
      // last arriver: acquire, reset the counter for the next launch, merge in chunk order
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(&a.counters[slot], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int c0 = __builtin_amdgcn_readfirstlane(long_chunk0[slot]);
      acc.init();
      // MU records per group in flight (a hub row has hundreds of chunks)
      constexpr int MU = 4;
      const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(reinterpret_cast<const f4*>(a.partial) + (int64_t)c0 * REC * LPR), 0,
          (unsigned)nch * (unsigned)REC * LPR * 16u, 0x00020000);
      for (int k0 = g; k0 < nch; k0 += G * MU) {
        f4 rs[MU], rm[MU], rw[MU], rq[MU], rn[MU], rax[MU], ran[MU];
#pragma unroll
        for (int m = 0; m < MU; ++m) {
          const int kk = k0 + m * G;
          const unsigned off = (kk < nch && lane_live) ? ((unsigned)kk * (unsigned)REC * LPR + (unsigned)q) * 16u : OOB;
          rs[m] = load_slot_wt(prs, off);
          rm[m] = load_slot_wt(prs, off == OOB ? OOB : off + 2u * LPR * 16u);
          rw[m] = load_slot_wt(prs, off == OOB ? OOB : off + 4u * LPR * 16u);
          if constexpr (NEED & NEED_SQ) rq[m] = load_slot_wt(prs, off == OOB ? OOB : off + 1u * LPR * 16u);
          if constexpr (NEED & NEED_MN) rn[m] = load_slot_wt(prs, off == OOB ? OOB : off + 3u * LPR * 16u);
          if constexpr (NEED & NEED_ARG) {
            rax[m] = load_slot_wt(prs, off == OOB ? OOB : off + 5u * LPR * 16u);
            if constexpr (NEED & NEED_MN) ran[m] = load_slot_wt(prs, off == OOB ? OOB : off + 6u * LPR * 16u);
          }
        }
#pragma unroll
        for (int m = 0; m < MU; ++m) {
          acc.sum += rs[m];  // out-of-range records read as 0: neutral for the sums
          acc.ws += rw[m];
          if constexpr (NEED & NEED_SQ) acc.sq += rq[m];
          if (k0 + m * G < nch) {
            if constexpr (NEED & NEED_ARG) {
              merge_gt4(acc.mx, acc.ax, rm[m], __builtin_bit_cast(i4, rax[m]));
              if constexpr (NEED & NEED_MN) merge_lt4(acc.mn, acc.an, rn[m], __builtin_bit_cast(i4, ran[m]));
            } else {
              acc.mx = f4_vmax(acc.mx, rm[m]);
              if constexpr (NEED & NEED_MN) acc.mn = f4_vmin(acc.mn, rn[m]);
            }
          }
        }
      }
      all_reduce_groups<LPR_LOG2, NEED>(acc, lane);
      nself = 0;
      for (int k = lane; k < nch; k += 64)
        nself += __hip_atomic_load(&a.partial_nself[c0 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) nself += bperm((lane ^ off) << 2, nself);
    }
    // every group now holds the whole row: all run the epilogue, group 0 stores
    f4 wpre[2], vself;
    bool has_self;
    load_row_operands<LPR_LOG2, C>(a, R, lane, row, true, wpre, vself, has_self);
    finish_group<LPR_LOG2, HPB, NEED, C>(a, R, lane, row, true, acc, deg, nself, dis_i, vself, has_self, wpre, g == 0,
                                         lds_w, lds_bias, lds_scale);
    return;
  }

  // ---------------- short-row role: one lane group per row, G rows per wavefront at a time ----------------
  const int Q = a.rows_per_wave;  // row groups (of G rows) per wavefront; Q * G <= 63
  const int gw = ((int)blockIdx.x - a.chunk_blocks) * 4 + wave;
  const int r0 = __builtin_amdgcn_readfirstlane(a.row_begin + gw * Q * G);
  const int n_end = a.row_end;
  if (r0 >= n_end) return;
  const int rp = a.rowptr[min(r0 + lane, a.n_nodes)];  // lanes 0..Q*G hold this wavefront's row pointers
  const int grp_addr = (g << LPR_LOG2) << 2;            // ds_bpermute byte address of the group's lane 0
  // stage the first row group's column indices (lane q of group g <- entry q of row r0 + g)
  int start_n = bperm(g << 2, rp);
  int nd_n;  // entries of the group's row handled here (0 for long or out-of-range rows)
  {
    const int deg = bperm((g + 1) << 2, rp) - start_n;
    nd_n = (r0 + g < n_end && deg <= EGC_LONG_ROW_THRESHOLD) ? deg : 0;
  }
  int jj_n = q < nd_n ? a.col[start_n + q] : 0;
  for (int k = 0; k < Q; ++k) {
    const int rbase = r0 + k * G;
    if (rbase >= n_end) break;
    const int row = rbase + g;
    const bool row_ok = row < n_end;
    const int start = start_n, nd = nd_n;
    int jj = jj_n;
    if (k + 1 < Q) {  // prefetch the next row group's bounds and first LPR column indices
      start_n = bperm(((k + 1) * G + g) << 2, rp);
      const int deg = bperm(((k + 1) * G + g + 1) << 2, rp) - start_n;
      nd_n = (rbase + G + g < n_end && deg <= EGC_LONG_ROW_THRESHOLD) ? deg : 0;
      jj_n = q < nd_n ? a.col[start_n + q] : 0;
    }
    // wave-uniform trip count: entries still valid in any group
    int maxd = nd;
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) maxd = max(maxd, bperm((lane ^ off) << 2, maxd));
    maxd = __builtin_amdgcn_readfirstlane(maxd);
    // row-only operands, issued ahead of the gathers
    f4 wpre[2], vself;
    bool has_self;
    const float dis_i = (a.dis != nullptr && row_ok) ? a.dis[row] : 0.f;
    load_row_operands<LPR_LOG2, C>(a, R, lane, row, row_ok, wpre, vself, has_self);

    FAcc<NEED> acc;
    acc.init();
    int nself = 0;
    for (int ts = 0; ts < maxd; ts += LPR) {
      if (ts > 0) jj = (ts + q < nd) ? a.col[start + ts + q] : 0;  // rows of more than LPR entries
      const bool pv = ts + q < nd;
      const float dd = !pv ? 0.f : a.edis != nullptr ? a.edis[start + ts + q] : a.dis != nullptr ? a.dis[jj] : 0.f;
      if (looped_any) {  // self-entries are excluded from LOOPED sets: count them per group
        const unsigned long long sb = __ballot(pv && jj == row);
        nself += __popcll((sb >> (g << LPR_LOG2)) & ((LPR == 64) ? ~0ull : ((1ull << LPR) - 1ull)));
      }
      const int cnt = min(LPR, maxd - ts);  // wave-uniform
      for (int t0 = 0; t0 < cnt; t0 += FU)
        gather_batch<NEED, C>(a, R, acc, grp_addr + (t0 << 2), 4, row, jj, dd, dis_i, lane_live ? nd : 0, ts + t0, 1, row_bytes,
                              slot_off, start);
    }
    // Opaque copy of the lane id: keeps the compiler from hoisting the epilogue's lane arithmetic out
    // of the row loop, where it would stay live across the gathers and cost occupancy.
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int deg_all = bperm((k * G + g + 1) << 2, rp) - start;
    const bool is_short = deg_all <= EGC_LONG_ROW_THRESHOLD;
    finish_group<LPR_LOG2, HPB, NEED, C>(a, R, ln, row, row_ok, acc, nd, nself, dis_i, vself, has_self, wpre, is_short,
                                         lds_w, lds_bias, lds_scale);
  }
}

bool fast_path_supported(const AggArgs& a, int layout, int chunks) {
  if (chunks != 1 || layout != EGC_LAYOUT_HBA || a.act == EGC_ACT_SOFTMAX) return false;
  if (a.x_looped && !a.y_looped) return false;  // never produced by either layer class
  if (a.slots < 1 || a.slots > 64) return false;
  if ((a.Ls & 3) != 0 || a.ldb != a.B * a.Ls) return false;  // every 16-byte slot belongs to one basis
  if ((a.B & (a.B - 1)) != 0) return false;
  if (a.A < 1 || a.A > AMAX) return false;
  if ((a.H + a.B - 1) / a.B > HPB_MAX) return false;
  if (a.W > 8 * (a.slots <= 16 ? 16 : a.slots <= 32 ? 32 : 64)) return false;  // weightings row: 2 x 16 bytes per lane of a group
  // the buffer descriptor addresses `out` with 32-bit byte offsets
  if ((uint64_t)a.n_nodes * (uint64_t)a.F_out * 4ull > (uint64_t)OOB) return false;
  return true;
}

template <int LPR_LOG2, int HPB, int NEED, class C>
static int launch_one(const AggArgs& a, unsigned grid, size_t lds, hipStream_t stream) {
  agg_fast_kernel<LPR_LOG2, HPB, NEED, C><<<grid, 256, lds, stream>>>(a);
  EGC_LAUNCH_CHECK("agg_fast_kernel");
  return EGC_OK;
}

template <int LPR_LOG2, int HPB>
static int launch_need(const AggArgs& a, int need, unsigned grid, size_t lds, hipStream_t stream) {
  if (a.arg_max != nullptr || a.arg_min != nullptr) {  // training forward of a layer with max / min
    if (need == 0) return launch_one<LPR_LOG2, HPB, NEED_ARG, RtCfg>(a, grid, lds, stream);
    return launch_one<LPR_LOG2, HPB, NEED_SQ | NEED_MN | NEED_ARG, RtCfg>(a, grid, lds, stream);
  }
  if (need == 0) return launch_one<LPR_LOG2, HPB, 0, RtCfg>(a, grid, lds, stream);
  return launch_one<LPR_LOG2, HPB, NEED_SQ | NEED_MN, RtCfg>(a, grid, lds, stream);
}

template <int LPR_LOG2>
static int launch_rt(const AggArgs& a, int need, unsigned grid, size_t lds, hipStream_t stream) {
  const int hpb = (a.H + a.B - 1) / a.B;
  if (hpb <= 1) return launch_need<LPR_LOG2, 1>(a, need, grid, lds, stream);
  if (hpb <= 2) return launch_need<LPR_LOG2, 2>(a, need, grid, lds, stream);
  return launch_need<LPR_LOG2, 4>(a, need, grid, lds, stream);
}

constexpr unsigned agg_pack(int a0, int a1 = 0, int a2 = 0, int a3 = 0) {
  return (unsigned)a0 | ((unsigned)a1 << 3) | ((unsigned)a2 << 6) | ((unsigned)a3 << 9);
}

// Statically specialised configurations (H, B, L, aggregator list, nonlinearity, edge sets).  Adding a
// line to launch_fast() buys the constant-folded kernel for that layer; everything else runs RtCfg.
template <class C, int LPR_LOG2, int HPB, int NEED>
static bool try_static(const AggArgs& a, int h, int b, int l, int ls, int na, unsigned agg, int act, bool xl, bool yl,
                       bool loops_all, unsigned grid, size_t lds, hipStream_t stream, int* status) {
  unsigned packed = 0;
  for (int t = 0; t < a.A; ++t) packed |= (unsigned)a.aggr[t] << (3 * t);
  if (a.H != h || a.B != b || a.L != l || a.A != na || packed != agg || a.act != act ||
      (a.x_looped != 0) != xl || (a.y_looped != 0) != yl || (a.loops_all != 0) != loops_all ||
      a.Ls != ls || a.slots > (1 << LPR_LOG2) || 2 * a.slots <= (1 << LPR_LOG2))
    return false;
  if constexpr (C::has(EGC_AGGR_MAX)) {
    if (a.arg_max != nullptr) {  // training forward: the variant that also tracks the arg positions
      *status = launch_one<LPR_LOG2, HPB, NEED | NEED_ARG, C>(a, grid, lds, stream);
      return true;
    }
  }
  *status = launch_one<LPR_LOG2, HPB, NEED, C>(a, grid, lds, stream);
  return true;
}

constexpr int lpr_log2_of(int slots) { return slots <= 16 ? 4 : slots <= 32 ? 5 : 6; }
#define EGC_STATIC_CFG(H, B, L, A, AGG, ACT, XL, YL, LA, NEED)                                                      \
  if (try_static<StCfg<H, B, L, A, AGG, ACT, XL, YL, LA, ((L) + 3) / 4 * 4>,                                        \
                 lpr_log2_of((B) * (((L) + 3) / 4)), ((H) + (B)-1) / (B), NEED>(                                   \
          a, H, B, L, ((L) + 3) / 4 * 4, A, AGG, ACT, XL, YL, LA, grid, lds, stream, &status))                      \
    return status;

int launch_fast(AggArgs a, int64_t n_nodes, const PlanCaps& caps, hipStream_t stream) {
  const int lpr = a.slots <= 16 ? 16 : a.slots <= 32 ? 32 : 64;
  const int G = 64 / lpr;
  // lanes per basis: shifts and an xor butterfly when L / 4 is a power of two, else division + rotation butterfly
  a.lanes_pb = a.Ls / 4;
  a.magic_P = (unsigned)(((uint64_t)1 << 32) / (uint64_t)a.lanes_pb) + 1u;  // q / lanes_pb == umulhi(q, magic_P), q < 64
  if ((a.lanes_pb & (a.lanes_pb - 1)) == 0) {
    int lg = 0;
    while ((4 << lg) < a.Ls) ++lg;
    a.lpb_log2 = lg;
  } else {
    a.lpb_log2 = -1;
  }
  if (a.rows_per_wave <= 0) a.rows_per_wave = 1;
  if (a.rows_per_wave * G > 60) a.rows_per_wave = 60 / G;
  a.chunk_blocks = (int)ceil_div(a.n_chunks_hint >= 0 ? a.n_chunks_hint : caps.cap_chunks, 4);
  a.need_mean = a.need_var = 0;
  int need = 0;
  for (int t = 0; t < a.A; ++t) {
    if (a.aggr[t] == EGC_AGGR_MEAN || a.aggr[t] == EGC_AGGR_VAR || a.aggr[t] == EGC_AGGR_STD) a.need_mean = 1;
    if (a.aggr[t] == EGC_AGGR_VAR || a.aggr[t] == EGC_AGGR_STD) { a.need_var = 1; need |= NEED_SQ; }
    if (a.aggr[t] == EGC_AGGR_MIN) need |= NEED_MN;
  }
  a.w_lds_stride = (a.W + 3) & ~3;
  a.bias_lds_floats = (a.H * a.Ls + 3) & ~3;  // >= F_out: the bias strip follows the (padded) head layout
  a.lds_floats_per_wave = (a.post_scale != nullptr ? 2 : 1) * a.bias_lds_floats + G * a.w_lds_stride;
  size_t lds = (size_t)4 * a.lds_floats_per_wave * sizeof(float);
  if (const char* e = getenv("EGC_AGG_LDS_PAD")) lds += (size_t)atoi(e);  // experiments: caps the blocks per CU
  if (lds > 64 * 1024) return EGC_ERR_UNSUPPORTED;
  const int64_t row_blocks = ceil_div((int64_t)a.row_end - a.row_begin, (int64_t)4 * a.rows_per_wave * G);
  const unsigned grid = (unsigned)(a.chunk_blocks + row_blocks);

  if (getenv("EGC_NO_STATIC_CFG") == nullptr) {
    int status = EGC_OK;
    constexpr int S = EGC_AGGR_SUM, M = EGC_AGGR_MEAN, X = EGC_AGGR_MAX, Y = EGC_AGGR_SYMNORM;
    // EGConv / EGC-M north star: d=128, H=8, B=4, sum+mean+max+symnorm, gcn_norm self-loops on every node
    EGC_STATIC_CFG(8, 4, 16, 4, agg_pack(S, M, X, Y), EGC_ACT_NONE, true, true, true, 0)
    // EGConv / EGC-S default: symnorm only (optimized_layers.py:77)
    EGC_STATIC_CFG(8, 4, 16, 1, agg_pack(Y), EGC_ACT_NONE, true, true, true, 0)
    // EfficientGraphConv EGC-M / EGC-S flavours at d=128 (symadd looped, the others raw): layers.py:166-193
    EGC_STATIC_CFG(8, 4, 16, 3, agg_pack(Y, X, M), EGC_ACT_NONE, false, true, true, 0)
    EGC_STATIC_CFG(8, 4, 16, 1, agg_pack(Y), EGC_ACT_NONE, false, true, true, 0)
    // the reference's trained nets (run_pretrained.sh / output/pretrained.txt), EfficientGraphConv:
    EGC_STATIC_CFG(8, 4, 23, 1, agg_pack(Y), EGC_ACT_NONE, false, true, true, 0)            // arxiv EGC-S 184/H8/B4 symadd
    EGC_STATIC_CFG(4, 4, 34, 3, agg_pack(Y, X, M), EGC_ACT_NONE, false, true, true, 0)      // arxiv EGC-M 136/H4/B4
    EGC_STATIC_CFG(8, 4, 21, 1, agg_pack(Y), EGC_ACT_NONE, false, true, true, 0)            // zinc EGC-S 168/H8/B4 symadd
    // EGConv on ogbn-mag (mag/models.py:24-53; train_main_table.sh:53-54): 352/H8/B4, symnorm or mean
    EGC_STATIC_CFG(8, 4, 44, 1, agg_pack(Y), EGC_ACT_NONE, true, true, true, 0)
    EGC_STATIC_CFG(8, 4, 44, 1, agg_pack(M), EGC_ACT_NONE, true, true, true, 0)
    // relational EGC (rmag/models.py:75-148): mean+max over a relation's raw rectangular adjacency, and the
    // root term (sum over an identity adjacency), at 128/H8/B4 and 64/H4/B4
    EGC_STATIC_CFG(8, 4, 16, 2, agg_pack(M, X), EGC_ACT_NONE, false, false, true, 0)
    EGC_STATIC_CFG(8, 4, 16, 1, agg_pack(S), EGC_ACT_NONE, false, false, true, 0)
    EGC_STATIC_CFG(4, 4, 16, 2, agg_pack(M, X), EGC_ACT_NONE, false, false, true, 0)
    EGC_STATIC_CFG(4, 4, 16, 1, agg_pack(S), EGC_ACT_NONE, false, false, true, 0)
  }
  switch (lpr) {
    case 16: return launch_rt<4>(a, need, grid, lds, stream);
    case 32: return launch_rt<5>(a, need, grid, lds, stream);
    default: return launch_rt<6>(a, need, grid, lds, stream);
  }
}

}  // namespace egc
