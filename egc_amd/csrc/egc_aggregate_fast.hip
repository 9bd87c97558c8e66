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

#include "egc_aggregate_fast_dev.h"

namespace egc {

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
template <int LPR_LOG2, int HPB, int NEED, class C>
// Inference variants (NEED == 0) fit 80 VGPRs without spilling when asked to, which buys the sixth wavefront per
// SIMD; the variants carrying more running aggregates are left to the register allocator.
#ifndef EGC_AGG_WAVES_SQ
#define EGC_AGG_WAVES_SQ 5   // wavefronts per SIMD of the std / var (or min) variants: six spill
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NEED == 0 ? EGC_AGG_WAVES : ((NEED & NEED_ARG) && (NEED & NEED_SQ)) ? 2 : (NEED == NEED_SQ || NEED == NEED_MN) ? EGC_AGG_WAVES_SQ : 4)))
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
    long_row_chunk<LPR_LOG2, HPB, NEED, C>(a, R, c, lane, lds_w, lds_bias, lds_scale);
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
    if constexpr (NEED & NEED_SQ) {   // the variance's shift: the row's first entry (lane 0 of the group staged its column), also
      // when the layer's x-part excludes that entry (a self-entry under add_self_loops: its value is as good a shift, and 0 --
      // what the masked gather returns for it -- is none: tiny graphs full of self-entries, tools/tile_fuzz.py seed 77)
      const int first = bperm(grp_addr, jj);
      acc.sh = load_slot(R.bases, (lane_live && nd > 0) ? (unsigned)first * row_bytes + slot_off : OOB);
    }
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

// ---------------------------------------------------------------------------------------------
// Rows of 65 .. 128 slots (the reference's two ogbg-code nets: 300/H4/B4 -> 76 slots, 304/H8/B8 -> 80 slots with padded
// bases; run_pretrained.sh:47-48, code/models.py:226-234): one row per wavefront, TWO slots per lane.  The slots of every
// basis are cut into a first set of P0 and a second set of P1 = P - P0: lane q holds slot (b = q / P0, l4 = q % P0) of
// the first set and slot (b = q / P1, P0 + q % P1) of the second.  Each set is a complete sub-layer over its own
// channels (every basis present, <= 64 lanes), so the register epilogue of the fast kernels runs once per set, unchanged
// but for the channel offset (l4_off); a neighbour row is gathered whole, as two 16-byte loads per lane.  Short rows only:
// long rows keep the chunk + merge kernels of egc_aggregate.hip.  Inference form.
// ---------------------------------------------------------------------------------------------
constexpr int WFU = 2;   // neighbour rows in flight per wavefront (two slots each)

// The per-set views of a layer description: the set's lanes per basis and its first slot inside a basis.  Over a
// compile-time configuration (StCfg) both are constants; over RtCfg they come from the kernel arguments
// (wide_p0 / wide_p1), selected by SET.
template <class Base, int SET, int PS = 0, int OFF = 0>
struct WideSet : Base {
  static constexpr bool stat = PS > 0;
  static __device__ inline int lanes_pb(const AggArgs& a) { return stat ? PS : (SET == 0 ? a.wide_p0 : a.wide_p1); }
  static __device__ inline int slots(const AggArgs& a) { return Base::B(a) * lanes_pb(a); }
  static __device__ inline bool pow2(const AggArgs& a) { return stat ? ((PS & (PS - 1)) == 0 && (Base::B(a) & (Base::B(a) - 1)) == 0) : false; }
  static __device__ inline int lpb_log2(const AggArgs&) { return stat ? ilog2(PS > 0 ? PS : 1) : -1; }
  static __device__ inline int basis_of(const AggArgs& a, int q) {
    return stat ? q / (PS > 0 ? PS : 1) : (int)__umulhi((unsigned)q, SET == 0 ? a.magic_P : a.magic_P1);
  }
  static __device__ inline int l4_off(const AggArgs& a) { return stat ? OFF : (SET == 0 ? 0 : a.wide_p0); }
};

struct WideLane {   // what a lane needs to address its two slots
  bool live0, live1;
  unsigned so0, so1;
};

template <class C0, class C1>
__device__ inline WideLane wide_lane(const AggArgs& a, int lane) {
  WideLane w;
  const int P = a.Ls >> 2, B = C0::B(a);
  const int b0 = min(C0::basis_of(a, lane), B - 1), b1 = min(C1::basis_of(a, lane), B - 1);
  w.live0 = lane < C0::slots(a);
  w.live1 = lane < C1::slots(a);
  w.so0 = (unsigned)(b0 * P + (lane - b0 * C0::lanes_pb(a))) * 16u;
  w.so1 = (unsigned)(b1 * P + C1::l4_off(a) + (lane - b1 * C1::lanes_pb(a))) * 16u;
  return w;
}

// 64 staged entries (jj, dd in the lanes; `cnt` of them valid) folded into the two sets
template <int NEED, class C0>
__device__ inline void wide_gather(const AggArgs& a, const FastRsrc& R, const WideLane& wl_, FAcc<NEED>& acc0, FAcc<NEED>& acc1,
                                   int row, int jj, float dd, float dis_i, int cnt, unsigned row_bytes, int pos_base) {
  for (int t0 = 0; t0 < cnt; t0 += WFU) {
    f4 v0[WFU], v1[WFU];
    float w[WFU];
    bool in_x[WFU];
#pragma unroll
    for (int u = 0; u < WFU; ++u) {
      const int addr = (t0 + u) << 2;
      const int j = bperm(addr, jj);
      const bool is_self = j == row;
      in_x[u] = (t0 + u < cnt) && !(C0::xl(a) && is_self);
      const unsigned base = (unsigned)j * row_bytes;
      v0[u] = load_slot(R.bases, (in_x[u] && wl_.live0) ? base + wl_.so0 : OOB);
      v1[u] = load_slot(R.bases, (in_x[u] && wl_.live1) ? base + wl_.so1 : OOB);
      w[u] = bperm(addr, dd) * dis_i;
      if (C0::yl(a) && !C0::xl(a)) w[u] = is_self ? 0.f : w[u];
    }
#pragma unroll
    for (int u = 0; u < WFU; ++u) {
      fold<NEED>(acc0, v0[u], w[u], in_x[u] && wl_.live0, pos_base + t0 + u);
      fold<NEED>(acc1, v1[u], w[u], in_x[u] && wl_.live1, pos_base + t0 + u);
    }
  }
}

// row-only operands + the two epilogues
template <int HPB, int NEED, class C0, class C1>
__device__ inline void wide_finish(const AggArgs& a, const FastRsrc& R, const WideLane& wl_, int lane, int row, FAcc<NEED>& acc0,
                                   FAcc<NEED>& acc1, int deg, int nself, float dis_i, unsigned row_bytes, float* lds_w,
                                   const float* lds_bias, const float* lds_scale) {
  const bool looped_any = C0::xl(a) || C0::yl(a);
  const bool has_self = C0::loops_all(a) || row <= *a.max_index;
  const bool want_self = looped_any && has_self;
  const f4 vself0 = load_slot(R.bases, (want_self && wl_.live0) ? (unsigned)row * row_bytes + wl_.so0 : OOB);
  const f4 vself1 = load_slot(R.bases, (want_self && wl_.live1) ? (unsigned)row * row_bytes + wl_.so1 : OOB);
  f4 wpre[2];
  const float* wrow = a.weightings + (int64_t)row * a.ldw;
  const int W = C0::W(a);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int c0 = (lane + kk * 64) * 4;
    wpre[kk] = f4{0.f, 0.f, 0.f, 0.f};
    if (c0 + 3 < W) wpre[kk] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(wrow + c0));
    else if (c0 < W) {
      wpre[kk].x = wrow[c0];
      if (c0 + 1 < W) wpre[kk].y = wrow[c0 + 1];
      if (c0 + 2 < W) wpre[kk].z = wrow[c0 + 2];
    }
  }
  int ln = lane;
  asm volatile("" : "+v"(ln));
  finish_group<6, HPB, NEED, C0>(a, R, ln, row, true, acc0, deg, nself, dis_i, vself0, has_self, wpre, true, lds_w, lds_bias, lds_scale);
  finish_group<6, HPB, NEED, C1>(a, R, ln, row, true, acc1, deg, nself, dis_i, vself1, has_self, wpre, true, lds_w, lds_bias, lds_scale);
}

// chunk record of the two-slots-per-lane kernel: [5 aggregates][2 sets][64 lanes] float4 (the workspace holds 7 x 128)
constexpr int WREC = 5 * 2 * 64;

template <int HPB, int NEED, class C0, class C1>
__device__ inline void wide_long_row_chunk(const AggArgs& a, const FastRsrc& R, const WideLane& wl_, int c, int lane,
                                           unsigned row_bytes, float* lds_w, const float* lds_bias, const float* lds_scale) {
  if (c >= a.plan[1]) return;
  const int cap_long = a.plan[2], cap_chunks = a.plan[3];
  const int* long_row = a.plan + 4;
  const int* long_chunk0 = long_row + cap_long;
  const int* chunk_slot = long_chunk0 + cap_long;
  const int* chunk_begin = chunk_slot + cap_chunks;
  const int slot = __builtin_amdgcn_readfirstlane(chunk_slot[c]);
  const int row = __builtin_amdgcn_readfirstlane(long_row[slot]);
  if (row < a.row_begin || row >= a.row_end) return;
  const int start = __builtin_amdgcn_readfirstlane(chunk_begin[c]);
  const int row_start = __builtin_amdgcn_readfirstlane(a.rowptr[row]);
  const int row_end = __builtin_amdgcn_readfirstlane(a.rowptr[row + 1]);
  const int end = min(start + EGC_LONG_ROW_CHUNK, row_end);
  const int deg = row_end - row_start;
  const int nch = (deg + EGC_LONG_ROW_CHUNK - 1) / EGC_LONG_ROW_CHUNK;
  const float dis_i = a.dis != nullptr ? a.dis[row] : 0.f;
  const bool looped_any = C0::xl(a) || C0::yl(a);
  FAcc<NEED> acc0, acc1;
  acc0.init();
  acc1.init();
  f4 shift0 = f4{0.f, 0.f, 0.f, 0.f}, shift1 = shift0;   // NEED_SQ: the row's first entry (FAcc::sh), for every chunk and the merge
  if constexpr (NEED & NEED_SQ) {
    const int first = __builtin_amdgcn_readfirstlane(a.col[row_start]);
    shift0 = load_slot(R.bases, wl_.live0 ? (unsigned)first * row_bytes + wl_.so0 : OOB);
    shift1 = load_slot(R.bases, wl_.live1 ? (unsigned)first * row_bytes + wl_.so1 : OOB);
    acc0.sh = shift0;
    acc1.sh = shift1;
  }
  int nself = 0;
  for (int base = start; base < end; base += 64) {
    const int p = base + lane;
    const bool pv = p < end;
    const int jj = pv ? a.col[p] : row;
    const float dd = a.edis != nullptr ? (pv ? a.edis[p] : 0.f) : (a.dis != nullptr ? a.dis[jj] : 0.f);
    if (looped_any) nself += __popcll(__ballot(pv && jj == row));
    wide_gather<NEED, C0>(a, R, wl_, acc0, acc1, row, jj, dd, dis_i, min(64, end - base), row_bytes, base);
  }
  if (nch > 1) {
    // publication as in long_row_chunk (egc_aggregate_fast_dev.h): write-through stores, drained, then the counter
    constexpr int WT = 0x11;
    const __amdgpu_buffer_rsrc_t pw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<f4*>(a.partial) + (int64_t)c * WREC), 0, (unsigned)WREC * 16u, 0x00020000);
    const unsigned po = (unsigned)lane * 16u;
    auto put = [&](int k, f4 v0, f4 v1) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v0), pw, po, (k * 2 + 0) * 64 * 16, WT);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v1), pw, po, (k * 2 + 1) * 64 * 16, WT);
    };
    put(0, acc0.sum, acc1.sum);
    put(2, acc0.mx, acc1.mx);
    put(4, acc0.ws, acc1.ws);
    if constexpr (NEED & NEED_SQ) put(1, acc0.sq, acc1.sq);
    if constexpr (NEED & NEED_MN) put(3, acc0.mn, acc1.mn);
    const __amdgpu_buffer_rsrc_t pn = __builtin_amdgcn_make_buffer_rsrc((void*)(a.partial_nself + c), 0, 4u, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(nself, pn, lane == 0 ? 0u : OOB, 0, WT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int arrived = 0;
    if (lane == 0) arrived = __hip_atomic_fetch_add(&a.counters[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived != nch - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&a.counters[slot], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int c0 = __builtin_amdgcn_readfirstlane(long_chunk0[slot]);
    acc0.init();
    acc1.init();
    if constexpr (NEED & NEED_SQ) { acc0.sh = shift0; acc1.sh = shift1; }
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const f4*>(a.partial) + (int64_t)c0 * WREC), 0, (unsigned)nch * (unsigned)WREC * 16u, 0x00020000);
    for (int kk = 0; kk < nch; ++kk) {   // chunk order: deterministic
      const unsigned off = ((unsigned)kk * WREC + (unsigned)lane) * 16u;
      auto get = [&](int k, int s_) { return load_slot_wt(prs, off + (unsigned)((k * 2 + s_) * 64 * 16)); };
      acc0.sum += get(0, 0); acc1.sum += get(0, 1);
      acc0.ws += get(4, 0); acc1.ws += get(4, 1);
      acc0.mx = f4_vmax(acc0.mx, get(2, 0)); acc1.mx = f4_vmax(acc1.mx, get(2, 1));
      if constexpr (NEED & NEED_SQ) { acc0.sq += get(1, 0); acc1.sq += get(1, 1); }
      if constexpr (NEED & NEED_MN) { acc0.mn = f4_vmin(acc0.mn, get(3, 0)); acc1.mn = f4_vmin(acc1.mn, get(3, 1)); }
    }
    nself = 0;
    for (int k = lane; k < nch; k += 64)
      nself += __hip_atomic_load(&a.partial_nself[c0 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nself += bperm((lane ^ off) << 2, nself);
  }
  wide_finish<HPB, NEED, C0, C1>(a, R, wl_, lane, row, acc0, acc1, deg, nself, dis_i, row_bytes, lds_w, lds_bias, lds_scale);
}

template <int HPB, int NEED, class C0, class C1>
__global__ void __launch_bounds__(256) agg_wide_kernel(AggArgs a) {
  extern __shared__ float smem[];
  if ((int)blockIdx.x < a.chunk_blocks && (int)blockIdx.x * 4 >= a.plan[1]) return;  // unused chunk slots
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const unsigned row_bytes = (unsigned)a.ldb * 4u;
  const int F_out = C0::F_out(a);
  float* lds_bias = smem + wave * a.lds_floats_per_wave;
  const bool post = a.post_scale != nullptr;
  float* lds_scale = lds_bias + a.bias_lds_floats;
  float* lds_w = lds_bias + (post ? 2 : 1) * a.bias_lds_floats;
  for (int o = lane; o < C0::H(a) * C0::Ls(a); o += 64) {
    const int h = o / C0::Ls(a), l = o - h * C0::Ls(a);
    const int c = h * C0::L(a) + l;
    const bool real = l < C0::L(a);
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
  const WideLane wl_ = wide_lane<C0, C1>(a, lane);

  if ((int)blockIdx.x < a.chunk_blocks) {   // long-row chunk role
    const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    wide_long_row_chunk<HPB, NEED, C0, C1>(a, R, wl_, c, lane, row_bytes, lds_w, lds_bias, lds_scale);
    return;
  }
  const bool looped_any = C0::xl(a) || C0::yl(a);
  const int Q = a.rows_per_wave;
  const int gw = ((int)blockIdx.x - a.chunk_blocks) * 4 + wave;
  const int r0 = __builtin_amdgcn_readfirstlane(a.row_begin + gw * Q);
  const int n_end = a.row_end;
  if (r0 >= n_end) return;
  const int rp = a.rowptr[min(r0 + lane, a.n_nodes)];
  for (int k = 0; k < Q; ++k) {
    const int row = r0 + k;
    if (row >= n_end) break;
    const int start = __builtin_amdgcn_readfirstlane(bperm(k << 2, rp));
    const int deg = __builtin_amdgcn_readfirstlane(bperm((k + 1) << 2, rp)) - start;
    if (deg > EGC_LONG_ROW_THRESHOLD) continue;     // chunk role
    const bool pv = lane < deg;
    const int jj = pv ? a.col[start + lane] : 0;
    const float dd = !pv ? 0.f : a.edis != nullptr ? a.edis[start + lane] : a.dis != nullptr ? a.dis[jj] : 0.f;
    const float dis_i = a.dis != nullptr ? a.dis[row] : 0.f;
    const int nself = looped_any ? __popcll(__ballot(pv && jj == row)) : 0;
    FAcc<NEED> acc0, acc1;
    acc0.init();
    acc1.init();
    if constexpr (NEED & NEED_SQ) {
      const int first = __builtin_amdgcn_readfirstlane(jj);       // lane 0 holds the row's first entry
      acc0.sh = load_slot(R.bases, (wl_.live0 && deg > 0) ? (unsigned)first * row_bytes + wl_.so0 : OOB);
      acc1.sh = load_slot(R.bases, (wl_.live1 && deg > 0) ? (unsigned)first * row_bytes + wl_.so1 : OOB);
    }
    wide_gather<NEED, C0>(a, R, wl_, acc0, acc1, row, jj, dd, dis_i, deg, row_bytes, start);
    wide_finish<HPB, NEED, C0, C1>(a, R, wl_, lane, row, acc0, acc1, deg, nself, dis_i, row_bytes, lds_w, lds_bias, lds_scale);
  }
}

bool wide_path_supported(const AggArgs& a, int layout) {
  if (layout != EGC_LAYOUT_HBA || a.act == EGC_ACT_SOFTMAX) return false;
  if (a.x_looped && !a.y_looped) return false;
  if (a.slots <= 64 || a.slots > 128) return false;
  if ((a.Ls & 3) != 0 || a.ldb != a.B * a.Ls) return false;          // every 16-byte slot belongs to one basis (padded bases)
  if ((a.B & (a.B - 1)) != 0) return false;
  if (a.A < 1 || a.A > AMAX) return false;
  if ((a.H + a.B - 1) / a.B > HPB_MAX) return false;
  if (a.W > 512) return false;                                        // weightings row: 2 x 16 bytes per lane
  const int P = a.Ls / 4, P0 = (P + 1) / 2;
  if (a.B * P0 > 64) return false;
  if (a.stats != nullptr || a.arg_max != nullptr || a.arg_min != nullptr) return false;   // inference form
  if ((uint64_t)a.n_nodes * (uint64_t)a.F_out * 4ull > (uint64_t)OOB) return false;
  // LDS of launch_wide_rows (bias strip [+ scale strip] + one weightings row per wavefront, four wavefronts)
  const size_t strips = (size_t)(a.post_scale != nullptr ? 2 : 1) * ((a.H * a.Ls + 3) & ~3) + ((a.W + 3) & ~3);
  if (4 * strips * sizeof(float) > 64 * 1024) return false;
  return true;
}

template <int HPB, int NEED, class C0, class C1>
static int launch_wide_one(const AggArgs& a, unsigned grid, size_t lds, hipStream_t stream) {
  agg_wide_kernel<HPB, NEED, C0, C1><<<grid, 256, lds, stream>>>(a);
  EGC_LAUNCH_CHECK("agg_wide_kernel");
  return EGC_OK;
}

template <int HPB>
static int launch_wide_need(const AggArgs& a, int need, unsigned grid, size_t lds, hipStream_t stream) {
  using R0 = WideSet<RtCfg, 0>;
  using R1 = WideSet<RtCfg, 1>;
  if (need == 0) return launch_wide_one<HPB, 0, R0, R1>(a, grid, lds, stream);
  return launch_wide_one<HPB, NEED_SQ | NEED_MN, R0, R1>(a, grid, lds, stream);
}

int launch_wide_rows(AggArgs a, const PlanCaps& caps, hipStream_t stream) {
  const int P = a.Ls / 4;
  a.wide_p0 = (P + 1) / 2;
  a.wide_p1 = P - a.wide_p0;
  a.magic_P = (unsigned)(((uint64_t)1 << 32) / (uint64_t)a.wide_p0) + 1u;
  a.magic_P1 = (unsigned)(((uint64_t)1 << 32) / (uint64_t)a.wide_p1) + 1u;
  a.lanes_pb = a.wide_p0;
  a.lpb_log2 = -1;
  a.l4_off = 0;
  a.rows_per_wave = 8;
  a.chunk_blocks = (int)ceil_div(a.n_chunks_hint >= 0 ? a.n_chunks_hint : caps.cap_chunks, 4);
  a.need_mean = a.need_var = 0;
  int need = 0;
  unsigned packed = 0;
  for (int t = 0; t < a.A; ++t) {
    if (a.aggr[t] == EGC_AGGR_MEAN || a.aggr[t] == EGC_AGGR_VAR || a.aggr[t] == EGC_AGGR_STD) a.need_mean = 1;
    if (a.aggr[t] == EGC_AGGR_VAR || a.aggr[t] == EGC_AGGR_STD) { a.need_var = 1; need |= NEED_SQ; }
    if (a.aggr[t] == EGC_AGGR_MIN) need |= NEED_MN;
    packed |= (unsigned)a.aggr[t] << (3 * t);
  }
  a.w_lds_stride = (a.W + 3) & ~3;
  a.bias_lds_floats = (a.H * a.Ls + 3) & ~3;
  a.lds_floats_per_wave = (a.post_scale != nullptr ? 2 : 1) * a.bias_lds_floats + a.w_lds_stride;   // G = 1
  const size_t lds = (size_t)4 * a.lds_floats_per_wave * sizeof(float);
  if (lds > 64 * 1024) return EGC_ERR_UNSUPPORTED;
  const unsigned grid = (unsigned)(a.chunk_blocks + ceil_div((int64_t)a.row_end - a.row_begin, (int64_t)4 * a.rows_per_wave));
  if (getenv("EGC_NO_STATIC_CFG") == nullptr && a.act == EGC_ACT_NONE && !a.x_looped && a.y_looped && a.loops_all) {
    constexpr int X = EGC_AGGR_MAX, N = EGC_AGGR_MIN, Y = EGC_AGGR_SYMNORM;
    // the reference's ogbg-code nets (run_pretrained.sh:47-48): EGC-M 300/H4/B4 symadd,min,max and EGC-S 304/H8/B8 symadd
    if (a.H == 4 && a.B == 4 && a.L == 75 && a.Ls == 76 && a.A == 3 && packed == agg_pack(Y, N, X)) {
      using St = StCfg<4, 4, 75, 3, agg_pack(Y, N, X), EGC_ACT_NONE, false, true, true, 76>;
      return launch_wide_one<1, NEED_MN, WideSet<St, 0, 10, 0>, WideSet<St, 1, 9, 10>>(a, grid, lds, stream);
    }
    if (a.H == 8 && a.B == 8 && a.L == 38 && a.Ls == 40 && a.A == 1 && packed == agg_pack(Y)) {
      using St = StCfg<8, 8, 38, 1, agg_pack(Y), EGC_ACT_NONE, false, true, true, 40>;
      return launch_wide_one<1, 0, WideSet<St, 0, 5, 0>, WideSet<St, 1, 5, 5>>(a, grid, lds, stream);
    }
  }
  const int hpb = (a.H + a.B - 1) / a.B;
  if (hpb <= 1) return launch_wide_need<1>(a, need, grid, lds, stream);
  if (hpb <= 2) return launch_wide_need<2>(a, need, grid, lds, stream);
  return launch_wide_need<4>(a, need, grid, lds, stream);
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
  // (squares without min and min without squares are kernels of their own: either accumulator alone leaves room for a fifth
  // wavefront per SIMD -- a std layer's aggregate at ogbn-arxiv size 137 -> see DESIGN.md 3.1)
  if (need == NEED_SQ) return launch_one<LPR_LOG2, HPB, NEED_SQ, RtCfg>(a, grid, lds, stream);
  if (need == NEED_MN) return launch_one<LPR_LOG2, HPB, NEED_MN, RtCfg>(a, grid, lds, stream);
  return launch_one<LPR_LOG2, HPB, NEED_SQ | NEED_MN, RtCfg>(a, grid, lds, stream);
}

template <int LPR_LOG2>
static int launch_rt(const AggArgs& a, int need, unsigned grid, size_t lds, hipStream_t stream) {
  const int hpb = (a.H + a.B - 1) / a.B;
  if (hpb <= 1) return launch_need<LPR_LOG2, 1>(a, need, grid, lds, stream);
  if (hpb <= 2) return launch_need<LPR_LOG2, 2>(a, need, grid, lds, stream);
  return launch_need<LPR_LOG2, 4>(a, need, grid, lds, stream);
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
  // Row groups per wavefront (the kernel requests group k + 1's bounds and first column indices under group k's gathers, and
  // stages its bias strip once).  One-row groups (33 - 64 slots: the 136 - 352-wide layers) on graphs that leave every CU
  // thousands of wavefronts take four: row pointers -> indices -> gathers are three dependent round trips per row otherwise --
  // ogbn-mag 352 / H8 / B4 1.46 -> 1.32 ms, molhiv b2048 296 / H8 / B4 53 -> 44 us (round 6; two: 1.33 ms / 45.5 us, eight: 49.5 us).
  // The 16- and 32-lane groups keep one (config 2: 101 -> 105 -> 110 us with two / four), and so do small batches, which need
  // the wavefronts (ZINC b128 at 124 / H4 / B4: 5.6 -> 7.6 -> 12.3 us).  (Two-row groups at CIFAR b2048: 168 / H8 / B4 160 -> 152 -> 157 us
  // with two / four, 128 / H4 / B4 symadd, std, max 294 -> 294 -> 291: left at one.)
  if (a.rows_per_wave <= 0) a.rows_per_wave = (G == 1 && (int64_t)a.row_end - a.row_begin >= 32768) ? 4 : 1;
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
  if (lds > 64 * 1024) return EGC_ERR_UNSUPPORTED;
  const int64_t row_blocks = ceil_div((int64_t)a.row_end - a.row_begin, (int64_t)4 * a.rows_per_wave * G);
  const unsigned grid = (unsigned)(a.chunk_blocks + row_blocks);

  if (getenv("EGC_NO_STATIC_CFG") == nullptr) {
    int status = EGC_OK;
    constexpr int S = EGC_AGGR_SUM, M = EGC_AGGR_MEAN, X = EGC_AGGR_MAX, Y = EGC_AGGR_SYMNORM;
    // EGConv / EGC-M north star: d=128, H=8, B=4, sum+mean+max+symnorm, gcn_norm self-loops on every node
    EGC_STATIC_CFG(8, 4, 16, 4, agg_pack(S, M, X, Y), EGC_ACT_NONE, true, true, true, 0)
    // the same layer with `std` in place of `mean` (bench.py: std_layer): the squares about the row's first entry
    EGC_STATIC_CFG(8, 4, 16, 4, agg_pack(S, EGC_AGGR_STD, X, Y), EGC_ACT_NONE, true, true, true, NEED_SQ)
    // EGConv / EGC-S default: symnorm only (optimized_layers.py:77)
    EGC_STATIC_CFG(8, 4, 16, 1, agg_pack(Y), EGC_ACT_NONE, true, true, true, 0)
    // EfficientGraphConv EGC-M / EGC-S flavours at d=128 (symadd looped, the others raw): layers.py:166-193
    EGC_STATIC_CFG(8, 4, 16, 3, agg_pack(Y, X, M), EGC_ACT_NONE, false, true, true, 0)
    EGC_STATIC_CFG(8, 4, 16, 1, agg_pack(Y), EGC_ACT_NONE, false, true, true, 0)
    // the reference's trained nets (run_pretrained.sh / output/pretrained.txt), EfficientGraphConv:
    EGC_STATIC_CFG(8, 4, 23, 1, agg_pack(Y), EGC_ACT_NONE, false, true, true, 0)            // arxiv EGC-S 184/H8/B4 symadd
    EGC_STATIC_CFG(4, 4, 34, 3, agg_pack(Y, X, M), EGC_ACT_NONE, false, true, true, 0)      // arxiv EGC-M 136/H4/B4
    EGC_STATIC_CFG(8, 4, 21, 1, agg_pack(Y), EGC_ACT_NONE, false, true, true, 0)            // zinc EGC-S 168/H8/B4 symadd
    EGC_STATIC_CFG(4, 4, 31, 3, agg_pack(S, EGC_AGGR_STD, X), EGC_ACT_NONE, false, true, true, NEED_SQ)  // zinc EGC-M 124/H4/B4 add,std,max
    EGC_STATIC_CFG(4, 4, 32, 3, agg_pack(Y, EGC_AGGR_STD, X), EGC_ACT_NONE, false, true, true, NEED_SQ)  // cifar EGC-M 128/H4/B4 symadd,std,max
    EGC_STATIC_CFG(8, 4, 37, 1, agg_pack(Y), EGC_ACT_NONE, false, true, true, 0)            // molhiv EGC-S 296/H8/B4 symadd
    EGC_STATIC_CFG(4, 4, 56, 3, agg_pack(S, M, X), EGC_ACT_NONE, false, true, true, 0)      // molhiv EGC-M 224/H4/B4 add,mean,max
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
