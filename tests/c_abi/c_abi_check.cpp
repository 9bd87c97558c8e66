// Stand-alone check of the C ABI (include/egc_hip.h) from a plain C++ host: no Python, no torch.  Device memory
// comes from hipMalloc, everything runs on a caller-created stream, and the result of one EGC layer forward
// (COO -> CSR -> degree statistics / plan -> GEMM -> fused aggregate + combine) is compared with a scalar
// double-precision restatement of the layer written here (test infrastructure, like oracle/).
//
// Layer: EGConv(128, 128, aggrs = sum, mean, max, symnorm, H = 8, B = 4) semantics, i.e. the gcn_norm edge set
// (self-loops replaced by one per node) for every aggregator (reference optimized_layers.py:127-208), in both
// weight layouts (HBA -> register-resident kernels, HAB -> generic kernels) and both GEMM forms (fp32, packed).
// Round 3: also the training forward and the sparse backward, with both workspace sizes (record path / arg-byte path).
// Exit code 0 and "c_abi_check: OK" on success.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "egc_hip.h"

#define HIP_OK(e)                                                                   \
  do {                                                                              \
    hipError_t _e = (e);                                                            \
    if (_e != hipSuccess) {                                                         \
      std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
      std::exit(2);                                                                 \
    }                                                                               \
  } while (0)
#define EGC_CHECK(e)                                                                                   \
  do {                                                                                                 \
    int _s = (e);                                                                                      \
    if (_s != EGC_OK) {                                                                                \
      std::fprintf(stderr, "%s:%d: egc status %d (%s)\n", __FILE__, __LINE__, _s, egc_last_error());   \
      std::exit(3);                                                                                    \
    }                                                                                                  \
  } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_u64() {
  rng_state ^= rng_state << 7;
  rng_state ^= rng_state >> 9;
  return rng_state * 0x2545F4914F6CDD1Dull;
}
static float next_unit() { return (float)((next_u64() >> 40) / (double)(1 << 24)) * 2.f - 1.f; }  // [-1, 1)

template <class T>
static T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  HIP_OK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
  if (!h.empty()) HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}
template <class T>
static T* device_zeros(size_t n) {
  T* d = nullptr;
  HIP_OK(hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(T)));
  HIP_OK(hipMemset(d, 0, std::max<size_t>(n, 1) * sizeof(T)));
  return d;
}

int main() {
  const int n = 1500, F_in = 128, F_out = 128, H = 8, B = 4, A = 4, L = F_out / H, F_g = B * L, W = H * B * A;
  const int aggrs[A] = {EGC_AGGR_SUM, EGC_AGGR_MEAN, EGC_AGGR_MAX, EGC_AGGR_SYMNORM};

  // ---- graph: random edges, a hub row far above EGC_LONG_ROW_THRESHOLD, duplicates, pre-existing self-loops
  std::vector<int64_t> src, dst;
  for (int k = 0; k < 12000; ++k) { src.push_back(next_u64() % n); dst.push_back(next_u64() % n); }
  for (int k = 0; k < 700; ++k) { src.push_back(next_u64() % n); dst.push_back(7); }
  for (int k = 0; k < 40; ++k) { int64_t v = next_u64() % n; src.push_back(v); dst.push_back(v); }
  const int64_t e = (int64_t)src.size();

  // ---- parameters and input
  std::vector<float> x((size_t)n * F_in), wb((size_t)F_in * F_g), wc((size_t)W * F_in), bc(W), bias(F_out);
  for (auto& v : x) v = next_unit();
  for (auto& v : wb) v = 0.2f * next_unit();
  for (auto& v : wc) v = 0.2f * next_unit();
  for (auto& v : bc) v = next_unit();
  for (auto& v : bias) v = next_unit();

  // ---- scalar restatement (double): comb rows in the reference's EGConv order h*A*B + a*B + b
  std::vector<std::vector<int>> nbr(n);
  for (int64_t k = 0; k < e; ++k)
    if (src[k] != dst[k]) nbr[dst[k]].push_back((int)src[k]);   // gcn_norm drops existing self-loops ...
  for (int i = 0; i < n; ++i) nbr[i].push_back(i);              // ... and appends exactly one per node
  std::vector<double> dis(n), bases((size_t)n * F_g), ref((size_t)n * F_out);
  for (int i = 0; i < n; ++i) dis[i] = 1.0 / std::sqrt((double)nbr[i].size());
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < F_g; ++c) {
      double s = 0;
      for (int k = 0; k < F_in; ++k) s += (double)x[(size_t)i * F_in + k] * wb[(size_t)k * F_g + c];
      bases[(size_t)i * F_g + c] = s;
    }
  for (int i = 0; i < n; ++i) {
    std::vector<double> wt(W);
    for (int r = 0; r < W; ++r) {
      double s = bc[r];
      for (int k = 0; k < F_in; ++k) s += (double)x[(size_t)i * F_in + k] * wc[(size_t)r * F_in + k];
      wt[r] = s;
    }
    std::vector<double> agg((size_t)A * F_g);
    for (int c = 0; c < F_g; ++c) {
      double sum = 0, mx = -INFINITY, sym = 0;
      for (int j : nbr[i]) {
        const double v = bases[(size_t)j * F_g + c];
        sum += v; mx = std::max(mx, v); sym += v * dis[j] * dis[i];
      }
      agg[0 * F_g + c] = sum; agg[1 * F_g + c] = sum / (double)nbr[i].size(); agg[2 * F_g + c] = mx; agg[3 * F_g + c] = sym;
    }
    for (int h = 0; h < H; ++h)
      for (int l = 0; l < L; ++l) {
        double s = bias[h * L + l];
        for (int a = 0; a < A; ++a)
          for (int b = 0; b < B; ++b) s += wt[h * A * B + a * B + b] * agg[(size_t)a * F_g + b * L + l];
        ref[(size_t)i * F_out + h * L + l] = s;
      }
  }

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));

  // ---- COO -> CSR, degree statistics, long-row plan, streamed deg^-1/2
  int64_t *d_src = to_device(src), *d_dst = to_device(dst);
  int32_t* d_rowptr = device_zeros<int32_t>(n + 1);
  int32_t* d_col = device_zeros<int32_t>(e);
  int32_t* d_eid = device_zeros<int32_t>(e);
  int32_t* d_maxidx = device_zeros<int32_t>(1);
  const size_t csr_ws_bytes = egc_coo_to_csr_workspace_bytes(n, e);
  char* d_csr_ws = device_zeros<char>(csr_ws_bytes);
  EGC_CHECK(egc_coo_to_csr(d_src, d_dst, e, n, d_rowptr, d_col, d_eid, d_maxidx, d_csr_ws, csr_ws_bytes, stream));
  float *d_dis_raw = device_zeros<float>(n), *d_dis_looped = device_zeros<float>(n);
  int32_t* d_plan = device_zeros<int32_t>((size_t)egc_plan_ints(n, e));
  EGC_CHECK(egc_csr_prepare(n, e, d_rowptr, d_col, d_dis_raw, d_dis_looped, d_plan, stream));
  float *d_edis_raw = device_zeros<float>(e), *d_edis_looped = device_zeros<float>(e);
  EGC_CHECK(egc_csr_edge_dis(e, d_col, d_dis_raw, d_dis_looped, d_edis_raw, d_edis_looped, stream));

  egc_graph g = {};
  g.n_nodes = n; g.n_edges = e; g.rowptr = d_rowptr; g.col = d_col; g.edge_id = d_eid;
  g.dis_raw = d_dis_raw; g.dis_looped = d_dis_looped; g.max_index = d_maxidx; g.plan = d_plan;
  g.n_chunks = -1; g.n_src_rows = 0; g.edge_dis_raw = d_edis_raw; g.edge_dis_looped = d_edis_looped;

  float* d_x = to_device(x);
  float* d_bias = to_device(bias);
  double worst = 0.0;
  int runs = 0;
  for (int layout = 0; layout < 2; ++layout) {
    egc_layer layer = {};
    layer.in_channels = F_in; layer.out_channels = F_out; layer.num_heads = H; layer.num_bases = B; layer.num_aggrs = A;
    for (int a = 0; a < A; ++a) layer.aggrs[a] = aggrs[a];
    layer.agg_set = EGC_SET_LOOPED; layer.sym_set = EGC_SET_LOOPED; layer.loops_all_nodes = 1;
    layer.weight_layout = layout == 0 ? EGC_LAYOUT_HBA : EGC_LAYOUT_HAB;
    layer.weight_act = EGC_ACT_NONE; layer.basis_stride = 0;
    const int ldb = egc_bases_ld(&layer);
    // wcat = [bases_weight | comb.weight^T] with the comb rows permuted into the layout handed to the kernels
    std::vector<float> wcat((size_t)F_in * (F_g + W)), bcat(W);
    for (int k = 0; k < F_in; ++k)
      for (int c = 0; c < F_g; ++c) wcat[(size_t)k * (F_g + W) + c] = wb[(size_t)k * F_g + c];
    for (int h = 0; h < H; ++h)
      for (int a = 0; a < A; ++a)
        for (int b = 0; b < B; ++b) {
          const int r_ref = h * A * B + a * B + b;
          const int r = layout == 0 ? h * B * A + b * A + a : r_ref;
          bcat[r] = bc[r_ref];
          for (int k = 0; k < F_in; ++k) wcat[(size_t)k * (F_g + W) + F_g + r] = wc[(size_t)r_ref * F_in + k];
        }
    float *d_wcat = to_device(wcat), *d_bcat = to_device(bcat);
    float* d_bases = device_zeros<float>((size_t)n * ldb);
    float* d_wt = device_zeros<float>((size_t)n * W);
    float* d_out = device_zeros<float>((size_t)n * F_out);
    const size_t ws_bytes = egc_aggregate_workspace_bytes(&layer, n, e);
    char* d_ws = device_zeros<char>(ws_bytes);   // contract: zero-filled before its first use
    const size_t pk_bytes = egc_basis_pack_bytes(F_in, F_g, W);
    char* d_packed = device_zeros<char>(pk_bytes);
    EGC_CHECK(egc_basis_pack(d_wcat, F_in, F_g, W, d_packed, pk_bytes, stream));
    for (int form = 0; form < 2; ++form) {
      HIP_OK(hipMemsetAsync(d_out, 0xff, (size_t)n * F_out * sizeof(float), stream));
      if (form == 0)
        EGC_CHECK(egc_layer_forward_f32(&g, &layer, d_x, d_wcat, d_bcat, d_bias, d_bases, ldb, d_wt, d_out, d_ws, ws_bytes, stream));
      else
        EGC_CHECK(egc_layer_forward_packed(&g, &layer, d_x, d_packed, d_bcat, d_bias, d_bases, ldb, d_wt, d_out, d_ws, ws_bytes, stream));
      std::vector<float> out((size_t)n * F_out);
      HIP_OK(hipMemcpyAsync(out.data(), d_out, out.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
      HIP_OK(hipStreamSynchronize(stream));
      double scale = 1.0, err = 0.0;
      for (double v : ref) scale = std::max(scale, std::fabs(v));
      for (size_t k = 0; k < out.size(); ++k) {
        const double d = std::fabs((double)out[k] - ref[k]);
        err = std::isnan(d) ? INFINITY : std::max(err, d);
      }
      std::printf("layout %s, %s GEMM: max |diff| / scale = %.3e\n", layout == 0 ? "HBA" : "HAB",
                  form == 0 ? "fp32" : "packed", err / scale);
      worst = std::max(worst, err / scale);
      ++runs;
    }
    HIP_OK(hipFree(d_wcat)); HIP_OK(hipFree(d_bcat)); HIP_OK(hipFree(d_bases)); HIP_OK(hipFree(d_wt));
    HIP_OK(hipFree(d_out)); HIP_OK(hipFree(d_ws)); HIP_OK(hipFree(d_packed));
  }
  // ---- training forward + sparse backward through the C ABI (round 3), twice: with the workspace of
  // egc_backward_workspace_bytes (arg-byte path) and of egc_backward_workspace_bytes_for (one 64-byte record per entry for the
  // max gradients) -- the two must agree up to the order of float additions
  double bwd_diff = 0.0;
  {
    egc_layer layer = {};
    layer.in_channels = F_in; layer.out_channels = F_out; layer.num_heads = H; layer.num_bases = B; layer.num_aggrs = A;
    for (int a = 0; a < A; ++a) layer.aggrs[a] = aggrs[a];
    layer.agg_set = EGC_SET_LOOPED; layer.sym_set = EGC_SET_LOOPED; layer.loops_all_nodes = 1;
    layer.weight_layout = EGC_LAYOUT_HBA; layer.weight_act = EGC_ACT_NONE; layer.basis_stride = 0;
    const int ldb = egc_bases_ld(&layer);
    std::vector<float> wcat((size_t)F_in * (F_g + W)), bcat(W), gout((size_t)n * F_out);
    for (int k = 0; k < F_in; ++k)
      for (int c = 0; c < F_g; ++c) wcat[(size_t)k * (F_g + W) + c] = wb[(size_t)k * F_g + c];
    for (int h = 0; h < H; ++h)
      for (int a = 0; a < A; ++a)
        for (int b = 0; b < B; ++b) {
          const int r_ref = h * A * B + a * B + b, r = h * B * A + b * A + a;
          bcat[r] = bc[r_ref];
          for (int k = 0; k < F_in; ++k) wcat[(size_t)k * (F_g + W) + F_g + r] = wc[(size_t)r_ref * F_in + k];
        }
    for (auto& v : gout) v = next_unit();
    float *d_wcat = to_device(wcat), *d_bcat = to_device(bcat), *d_gout = to_device(gout);
    float* d_bases = device_zeros<float>((size_t)n * ldb);
    float* d_wt = device_zeros<float>((size_t)n * W);
    float* d_out = device_zeros<float>((size_t)n * F_out);
    const size_t ws_bytes = egc_aggregate_workspace_bytes(&layer, n, e);
    char* d_ws = device_zeros<char>(ws_bytes);
    const size_t pk_bytes = egc_basis_pack_bytes(F_in, F_g, W);
    char* d_packed = device_zeros<char>(pk_bytes);
    EGC_CHECK(egc_basis_pack(d_wcat, F_in, F_g, W, d_packed, pk_bytes, stream));
    EGC_CHECK(egc_basis_transform_packed(d_x, d_packed, d_bcat, n, F_in, F_g, W, d_bases, ldb, d_wt, stream));
    float* d_stats = device_zeros<float>((size_t)n * (size_t)egc_train_stats_floats(&layer));
    int32_t* d_cnt = device_zeros<int32_t>(n);
    int32_t* d_argmax = device_zeros<int32_t>((size_t)n * ldb);
    EGC_CHECK(egc_aggregate_combine_train_f32(&g, &layer, d_bases, ldb, d_wt, d_bias, d_out, d_stats, d_cnt, d_argmax, nullptr, d_ws,
                                              ws_bytes, stream));
    // the transposed graph, from the edge list in destination-CSR order (its edge_id then names CSR positions)
    int64_t *d_tsrc = device_zeros<int64_t>(e), *d_tdst = device_zeros<int64_t>(e);
    EGC_CHECK(egc_csr_transposed_coo(n, e, d_rowptr, d_col, d_tsrc, d_tdst, stream));
    int32_t* d_trowptr = device_zeros<int32_t>(n + 1);
    int32_t *d_tcol = device_zeros<int32_t>(e), *d_teid = device_zeros<int32_t>(e), *d_tmax = device_zeros<int32_t>(1);
    HIP_OK(hipMemsetAsync(d_csr_ws, 0, csr_ws_bytes, stream));
    EGC_CHECK(egc_coo_to_csr(d_tsrc, d_tdst, e, n, d_trowptr, d_tcol, d_teid, d_tmax, d_csr_ws, csr_ws_bytes, stream));
    float *d_tdr = device_zeros<float>(n), *d_tdl = device_zeros<float>(n);
    int32_t* d_tplan = device_zeros<int32_t>((size_t)egc_plan_ints(n, e));
    EGC_CHECK(egc_csr_prepare(n, e, d_trowptr, d_tcol, d_tdr, d_tdl, d_tplan, stream));
    egc_graph tg = {};
    tg.n_nodes = n; tg.n_edges = e; tg.rowptr = d_trowptr; tg.col = d_tcol; tg.edge_id = d_teid; tg.dis_raw = d_tdr;
    tg.dis_looped = d_tdl; tg.max_index = d_tmax; tg.plan = d_tplan; tg.n_chunks = -1; tg.n_src_rows = 0;
    std::vector<float> res[2];
    const size_t sizes[2] = {egc_backward_workspace_bytes(&layer, n), egc_backward_workspace_bytes_for(&layer, &g)};
    if (sizes[1] != sizes[0] + (size_t)e * 64) { std::fprintf(stderr, "workspace sizes: %zu %zu\n", sizes[0], sizes[1]); return 4; }
    for (int v = 0; v < 2; ++v) {
      char* d_bws = device_zeros<char>(sizes[v]);
      float* d_db = device_zeros<float>((size_t)n * ldb);
      float* d_dw = device_zeros<float>((size_t)n * W);
      HIP_OK(hipMemsetAsync(d_db, 0xff, (size_t)n * ldb * sizeof(float), stream));   // square graph: every row is written
      EGC_CHECK(egc_aggregate_combine_backward_f32(&g, &tg, &layer, d_bases, ldb, d_wt, d_gout, d_stats, d_cnt, d_argmax, nullptr, d_db, 0,
                                                   d_dw, 0, d_bws, sizes[v], stream));
      res[v].resize((size_t)n * (ldb + W));
      HIP_OK(hipMemcpyAsync(res[v].data(), d_db, (size_t)n * ldb * sizeof(float), hipMemcpyDeviceToHost, stream));
      HIP_OK(hipMemcpyAsync(res[v].data() + (size_t)n * ldb, d_dw, (size_t)n * W * sizeof(float), hipMemcpyDeviceToHost, stream));
      HIP_OK(hipStreamSynchronize(stream));
      HIP_OK(hipFree(d_bws)); HIP_OK(hipFree(d_db)); HIP_OK(hipFree(d_dw));
    }
    double scale = 1.0;
    for (float v : res[0]) scale = std::max(scale, (double)std::fabs(v));
    for (size_t k = 0; k < res[0].size(); ++k) {
      const double d = std::fabs((double)res[0][k] - (double)res[1][k]);
      bwd_diff = std::isnan(d) ? INFINITY : std::max(bwd_diff, d / scale);
    }
    std::printf("sparse backward, record path vs arg-byte path: max |diff| / scale = %.3e\n", bwd_diff);
  }
  const bool ok = runs == 4 && worst <= 1e-5 && bwd_diff <= 3e-6;
  std::printf("%s %s\n", egc_version(), ok ? "c_abi_check: OK" : "c_abi_check: FAILED");
  return ok ? 0 : 1;
}
