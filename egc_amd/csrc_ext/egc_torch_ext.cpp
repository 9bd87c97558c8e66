// Compiled PyTorch binding of the layer call over the C ABI of libegc_hip.so (include/egc_hip.h): TORCH_LIBRARY operators
// registered for the HIP device type only (PyTorch-ROCm's "CUDA" dispatch key) -- north_star: "exposed as a PyTorch-ROCm
// C++/HIP extension".  Replaces, on the host side, the ctypes marshalling + torch.empty calls of egc_amd/functional.py
// for the inference forward of the layer modules (the same reference call sites as egc_layer_forward_packed:
// layers.py:97-138, optimized_layers.py:177-210): one dispatcher call allocates bases / weightings / out from the
// caching allocator and issues the library's launches on the caller's stream.  No device code here; nothing of the
// hot path is computed by torch.
//
// The graph and the layer description are the library's own structs (egc_graph, egc_layer), kept alive by their Python
// owners (CSRGraph.c_struct(), LayerSpec.c); they travel as integer addresses, the stream as the raw hipStream_t the
// ctypes path uses as well (torch._C._cuda_getCurrentRawStream).
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "egc_hip.h"

namespace {

const float* fptr(const c10::optional<at::Tensor>& t) { return t.has_value() ? t->data_ptr<float>() : nullptr; }

void check_f32(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), "egc_amd: ", name, " must live on a ROCm device (the EGC hot path is HIP-only, no CPU fallback)");
  TORCH_CHECK(t.scalar_type() == at::kFloat && t.is_contiguous(), "egc_amd: ", name, " must be a dense float32 tensor");
}

void check_status(int st, const char* what) {
  TORCH_CHECK(st == EGC_OK, "egc_amd: ", what, " failed with status ", st, " (", egc_last_error(), ")");
}

// Shapes of the optional operands and the workspace (the ctypes path checks them in Python; here a wrong-sized scale or
// residual would be read out of bounds on the device), and the device the pointers live on.
static void check_vec(const c10::optional<at::Tensor>& t, int64_t n, const char* what, const at::Tensor& x) {
  if (!t.has_value()) return;
  check_f32(*t, what);
  TORCH_CHECK(t->numel() == n && t->device() == x.device(), "egc_amd: ", what, " must hold ", n, " floats on x's device");
}
static void check_layer_operands(const at::Tensor& x, const egc_graph* g, const egc_layer* l, const at::Tensor& packed,
                                 const c10::optional<at::Tensor>& bcat, const c10::optional<at::Tensor>& bias,
                                 const at::Tensor& workspace, int64_t ldb, int64_t w_cols, int64_t f_out) {
  TORCH_CHECK(f_out == l->out_channels && ldb == egc_bases_ld(l) && w_cols == (int64_t)l->num_heads * l->num_bases * l->num_aggrs,
              "egc_amd: ldb / w_cols / f_out do not belong to this layer");
  check_vec(bcat, w_cols, "bcat", x);
  check_vec(bias, f_out, "bias", x);
  TORCH_CHECK(packed.device() == x.device() && workspace.device() == x.device(), "egc_amd: packed weights / workspace on another device");
  TORCH_CHECK((size_t)workspace.numel() >= egc_aggregate_workspace_bytes_for(l, g), "egc_amd: workspace too small for this graph and layer");
}

// out = layer(x) -- basis GEMM (split-precision planes `packed`) + fused aggregate/combine, two launches, one call
at::Tensor layer_forward(const at::Tensor& x, const at::Tensor& packed, const c10::optional<at::Tensor>& bcat,
                         const c10::optional<at::Tensor>& bias, int64_t graph, int64_t layer, const at::Tensor& workspace,
                         int64_t stream, int64_t ldb, int64_t w_cols, int64_t f_out) {
  check_f32(x, "x");
  const auto* g = reinterpret_cast<const egc_graph*>(graph);
  const auto* l = reinterpret_cast<const egc_layer*>(layer);
  TORCH_CHECK(x.dim() == 2 && x.size(0) == g->n_nodes && x.size(1) == l->in_channels, "egc_amd: x has the wrong shape");
  check_layer_operands(x, g, l, packed, bcat, bias, workspace, ldb, w_cols, f_out);
  const c10::OptionalDeviceGuard device_guard(x.device());   // (allocations below on x's device whatever the current one is)
  const int64_t n = x.size(0);
  const auto opts = x.options();
  at::Tensor bases = at::empty({n, ldb}, opts), weightings = at::empty({n, w_cols}, opts), out = at::empty({n, f_out}, opts);
  check_status(egc_layer_forward_packed(g, l, x.data_ptr<float>(), packed.data_ptr(), fptr(bcat), fptr(bias),
                                        bases.data_ptr<float>(), (int32_t)ldb, weightings.data_ptr<float>(),
                                        out.data_ptr<float>(), workspace.data_ptr(), (size_t)workspace.numel(),
                                        reinterpret_cast<egc_stream_t>(stream)),
               "egc_layer_forward_packed");
  return out;
}

// the same with the caller's tail fused into the store: out = act((z + bias) * scale + shift) + residual
at::Tensor layer_forward_post(const at::Tensor& x, const at::Tensor& packed, const c10::optional<at::Tensor>& bcat,
                              const c10::optional<at::Tensor>& bias, int64_t graph, int64_t layer, const at::Tensor& workspace,
                              int64_t stream, int64_t ldb, int64_t w_cols, int64_t f_out, int64_t gemm_flags,
                              const c10::optional<at::Tensor>& scale, const c10::optional<at::Tensor>& shift,
                              const c10::optional<at::Tensor>& residual, bool relu) {
  check_f32(x, "x");
  const auto* g = reinterpret_cast<const egc_graph*>(graph);
  const auto* l = reinterpret_cast<const egc_layer*>(layer);
  TORCH_CHECK(x.dim() == 2 && x.size(0) == g->n_nodes && x.size(1) == l->in_channels, "egc_amd: x has the wrong shape");
  check_layer_operands(x, g, l, packed, bcat, bias, workspace, ldb, w_cols, f_out);
  check_vec(scale, f_out, "post.scale", x);
  check_vec(shift, f_out, "post.shift", x);
  check_vec(residual, x.size(0) * f_out, "post.residual", x);
  TORCH_CHECK(scale.has_value() == shift.has_value(), "egc_amd: post.scale and post.shift come together");
  const c10::OptionalDeviceGuard device_guard(x.device());   // (allocations below on x's device whatever the current one is)
  const int64_t n = x.size(0);
  const auto opts = x.options();
  at::Tensor bases = at::empty({n, ldb}, opts), weightings = at::empty({n, w_cols}, opts), out = at::empty({n, f_out}, opts);
  const int32_t fg = l->num_bases * (l->basis_stride > 0 ? l->basis_stride : l->out_channels / l->num_heads);
  auto st = reinterpret_cast<egc_stream_t>(stream);
  check_status(egc_basis_transform_packed_ex(x.data_ptr<float>(), packed.data_ptr(), fptr(bcat), n, l->in_channels, fg,
                                             (int32_t)w_cols, (int32_t)gemm_flags, bases.data_ptr<float>(), (int32_t)ldb,
                                             weightings.data_ptr<float>(), st),
               "egc_basis_transform_packed_ex");
  egc_post post{fptr(scale), fptr(shift), fptr(residual), relu ? 1 : 0};
  check_status(egc_aggregate_combine_post_f32(g, l, bases.data_ptr<float>(), (int32_t)ldb, weightings.data_ptr<float>(), fptr(bias),
                                              &post, out.data_ptr<float>(), workspace.data_ptr(), (size_t)workspace.numel(), st),
               "egc_aggregate_combine_post_f32");
  return out;
}


// ---------------------------------------------------------------------------------------------------------------------
// Training (round 3): forward and backward of the layer FROM THE MODULE'S PARAMETERS, one dispatcher call each -- what
// egc_amd/functional.py's _EGCLayerParamsFunction does through ten Python helpers and as many ctypes calls (the eager
// small-batch training step is bound by exactly that host time: DESIGN.md section 5).  Same library entry points, same
// order, same buffers; square graphs without a halo, the dense-gradient shapes of egc_weight_grad_ex_f32 -- everything
// else stays on the Python path (egc_amd/functional.py decides).
// ---------------------------------------------------------------------------------------------------------------------
// -> [out, wcat, bcat (or empty), bases, weightings, stats, cnt, arg_max (or empty), arg_min (or empty)]
std::vector<at::Tensor> train_forward(const at::Tensor& x, const at::Tensor& comb_w, const c10::optional<at::Tensor>& comb_b,
                                      const c10::optional<at::Tensor>& bcat_direct, at::TensorList basis_parts,
                                      const c10::optional<at::Tensor>& bias, int64_t graph, int64_t layer,
                                      const at::Tensor& workspace, int64_t stream, int64_t H, int64_t A, int64_t B, int64_t L,
                                      int64_t Ls, bool permute_hab, int64_t gemm_flags) {
  check_f32(x, "x");
  check_f32(comb_w, "comb weight");
  const auto* g = reinterpret_cast<const egc_graph*>(graph);
  const auto* l = reinterpret_cast<const egc_layer*>(layer);
  auto st = reinterpret_cast<egc_stream_t>(stream);
  const int64_t n = x.size(0), f_in = x.size(1);
  TORCH_CHECK(x.dim() == 2 && n == g->n_nodes && f_in == l->in_channels, "egc_amd: x has the wrong shape");
  TORCH_CHECK(g->n_src_rows == 0 || g->n_src_rows == n, "egc_amd: the compiled training path takes square graphs");
  const auto opts = x.options();
  const int64_t f_g = B * Ls, W = H * B * A, ldb = egc_bases_ld(l), f_out = l->out_channels;
  // (1) the GEMM operand from the parameters
  at::Tensor wcat = at::empty({f_in, f_g + W}, opts);
  at::Tensor bcat = comb_b.has_value() ? at::empty({W}, opts) : at::empty({0}, opts);
  std::vector<const float*> parts;
  for (const auto& t : basis_parts) { check_f32(t, "basis matrix"); parts.push_back(t.data_ptr<float>()); }
  if (comb_b.has_value()) check_f32(*comb_b, "comb bias");
  check_status(egc_weights_pack_f32(parts.data(), (int32_t)parts.size(), comb_w.data_ptr<float>(), fptr(comb_b), (int32_t)f_in,
                                    (int32_t)H, (int32_t)A, (int32_t)B, (int32_t)L, (int32_t)Ls, permute_hab ? 1 : 0,
                                    wcat.data_ptr<float>(), comb_b.has_value() ? bcat.data_ptr<float>() : nullptr, 0, st),
               "egc_weights_pack_f32");
  // (2) its split-precision planes, (3) the GEMM
  const size_t pbytes = egc_basis_pack_bytes((int32_t)f_in, (int32_t)f_g, (int32_t)W);
  at::Tensor planes = at::empty({(int64_t)pbytes}, opts.dtype(at::kByte));
  check_status(egc_basis_pack_ex(wcat.data_ptr<float>(), (int32_t)f_in, (int32_t)f_g, (int32_t)W, (int32_t)gemm_flags,
                                 planes.data_ptr(), pbytes, st), "egc_basis_pack_ex");
  const float* bc = comb_b.has_value() ? bcat.data_ptr<float>() : fptr(bcat_direct);
  if (bcat_direct.has_value()) check_f32(*bcat_direct, "comb bias");
  at::Tensor bases = at::empty({n, ldb}, opts), weightings = at::empty({n, W}, opts);
  check_status(egc_basis_transform_packed_ex(x.data_ptr<float>(), planes.data_ptr(), bc, n, (int32_t)f_in, (int32_t)f_g,
                                             (int32_t)W, (int32_t)gemm_flags, bases.data_ptr<float>(), (int32_t)ldb,
                                             weightings.data_ptr<float>(), st), "egc_basis_transform_packed_ex");
  // (4) the training aggregate: output + what the backward consumes
  bool has_max = false, has_min = false;
  for (int t = 0; t < l->num_aggrs; ++t) {
    has_max = has_max || l->aggrs[t] == EGC_AGGR_MAX;
    has_min = has_min || l->aggrs[t] == EGC_AGGR_MIN;
  }
  at::Tensor out = at::empty({n, f_out}, opts);
  at::Tensor stats = at::empty({n, std::max<int64_t>(egc_train_stats_floats(l), 1)}, opts);
  at::Tensor cnt = at::empty({std::max<int64_t>(n, 1)}, opts.dtype(at::kInt));
  at::Tensor arg_max = has_max ? at::empty({n, ldb}, opts.dtype(at::kInt)) : at::empty({0}, opts.dtype(at::kInt));
  at::Tensor arg_min = has_min ? at::empty({n, ldb}, opts.dtype(at::kInt)) : at::empty({0}, opts.dtype(at::kInt));
  if (bias.has_value()) check_f32(*bias, "bias");
  check_status(egc_aggregate_combine_train_f32(g, l, bases.data_ptr<float>(), (int32_t)ldb, weightings.data_ptr<float>(), fptr(bias),
                                               out.data_ptr<float>(), stats.data_ptr<float>(), cnt.data_ptr<int32_t>(),
                                               has_max ? arg_max.data_ptr<int32_t>() : nullptr,
                                               has_min ? arg_min.data_ptr<int32_t>() : nullptr, workspace.data_ptr(),
                                               (size_t)workspace.numel(), st), "egc_aggregate_combine_train_f32");
  return {out, wcat, bcat, bases, weightings, stats, cnt, arg_max, arg_min};
}

// -> [dx (or empty), d comb_w, d comb_b or d bcat_direct (or empty), d bias (or empty), d basis matrices ...]
// Requires (checked by the caller): ldb == B Ls, (ldb + W) % 4 == 0, f_in % 4 == 0, f_out % 4 == 0; a bias and a combination bias
// present.  The one-pass dense-gradient kernel inside its envelope (f_in <= 128, ldb + W <= 192, f_out <= 128), the general
// sequence outside it (round 6).
// dx_addend (or nullptr): [N, f_in] added to d x -- the residual branch's gradient; joins the d x GEMM's store where the kernel
// carries an addend (egc_basis_transform_packed_add), else one in-place add.  bias_grad (or nullptr): the column sums of grad_out
// when the caller holds them already (the BatchNorm tail's backward: egc_bn_backward_stats_sums_f32) -- no pass over grad_out.
static std::vector<at::Tensor> train_backward_impl(const at::Tensor& grad_out, const at::Tensor& x, const at::Tensor& wcat,
                                                   const at::Tensor& bases, const at::Tensor& weightings, const at::Tensor& stats,
                                                   const at::Tensor& cnt, const at::Tensor& arg_max, const at::Tensor& arg_min, int64_t graph,
                                                   int64_t t_graph, int64_t layer, int64_t stream, int64_t H, int64_t A, int64_t B, int64_t L,
                                                   int64_t Ls, bool permute_hab, bool packed_bias, bool need_x, at::IntArrayRef comb_w_shape,
                                                   at::IntArrayRef comb_b_shape, int64_t n_parts, at::IntArrayRef part_shape,
                                                   const at::Tensor* dx_addend, const at::Tensor* bias_grad) {
  const auto* g = reinterpret_cast<const egc_graph*>(graph);
  const auto* tg = reinterpret_cast<const egc_graph*>(t_graph);
  const auto* l = reinterpret_cast<const egc_layer*>(layer);
  auto st = reinterpret_cast<egc_stream_t>(stream);
  at::Tensor go = grad_out.contiguous();
  check_f32(go, "grad_out");
  const int64_t n = x.size(0), f_in = x.size(1);
  const int64_t W = H * B * A, ldb = egc_bases_ld(l), k = ldb + W, f_out = l->out_channels;
  const auto opts = x.options();
  // (1) sparse backward into the column blocks of ONE [N, ldb + W] array: the left operand of both dense gradients
  at::Tensor d_cat = at::empty({n, k}, opts);
  const size_t wbytes = egc_backward_workspace_bytes_for(l, g);
  at::Tensor ws = at::empty({(int64_t)std::max<size_t>(wbytes, 1)}, opts.dtype(at::kByte));
  check_status(egc_aggregate_combine_backward_f32(g, tg, l, bases.data_ptr<float>(), (int32_t)ldb, weightings.data_ptr<float>(),
                                                  go.data_ptr<float>(), stats.data_ptr<float>(), cnt.data_ptr<int32_t>(),
                                                  arg_max.numel() ? arg_max.data_ptr<int32_t>() : nullptr,
                                                  arg_min.numel() ? arg_min.data_ptr<int32_t>() : nullptr, d_cat.data_ptr<float>(),
                                                  (int32_t)k, d_cat.data_ptr<float>() + ldb, (int32_t)k, ws.data_ptr(), wbytes, st),
               "egc_aggregate_combine_backward_f32");
  // (2) dx = d_cat wcat^T on the split-precision GEMM (wcat packed where it lies)
  at::Tensor dx = at::empty({0}, opts);
  if (need_x) {
    const size_t pb = egc_basis_pack_bytes((int32_t)k, (int32_t)f_in, 0);
    at::Tensor packed = at::empty({(int64_t)pb}, opts.dtype(at::kByte));
    dx = at::empty({n, f_in}, opts);
    check_status(egc_basis_pack_transposed(wcat.data_ptr<float>(), k, (int32_t)k, (int32_t)f_in, 0, packed.data_ptr(), pb, st),
                 "egc_basis_pack_transposed");
    int rc = EGC_ERR_UNSUPPORTED;
    if (dx_addend != nullptr && (f_in % 4) == 0)
      rc = egc_basis_transform_packed_add(d_cat.data_ptr<float>(), packed.data_ptr(), nullptr, n, (int32_t)k, (int32_t)f_in, 0, 0,
                                          dx_addend->data_ptr<float>(), dx.data_ptr<float>(), (int32_t)f_in, nullptr, st);
    if (rc == EGC_ERR_UNSUPPORTED) {
      check_status(egc_basis_transform_packed(d_cat.data_ptr<float>(), packed.data_ptr(), nullptr, n, (int32_t)k, (int32_t)f_in, 0,
                                              dx.data_ptr<float>(), (int32_t)f_in, nullptr, st), "egc_basis_transform_packed");
      if (dx_addend != nullptr) dx.add_(*dx_addend);       // autograd's own add, in place
    } else {
      check_status(rc, "egc_basis_transform_packed_add");
    }
  }
  // (3) the dense gradients of the parameters: x^T d_cat with the column sums of d_cat (combination bias), written straight into
  // the parameters' gradients through the pack's index map (egc_weight_grad_params_f32: no d wcat array, no unpack launch), at
  // every width (round 6: the reference's own batched nets -- 168 / 224 / 296 wide, run_pretrained.sh:7,12,23,24 -- go through
  // the exact-fp32 tile grid; same sums in the same order as egc_weight_grad_ex_f32 + egc_weights_pack_f32(grad = 1), the calls
  // of egc_amd/functional.py's _layer_train_backward + _unpack_param_grads).  The column sums of grad_out (the layer's bias) ride
  // along inside the one-tile kernel's envelope (f_in <= 128, ldb + W <= 192, f_out <= 128), come from the caller when it holds
  // them (the block node's BatchNorm step), or take their own pass.
  const bool have_es = bias_grad != nullptr;
  at::Tensor es = have_es ? *bias_grad : at::empty({f_out}, opts);
  at::Tensor dcw = at::empty(comb_w_shape, opts);
  at::Tensor dcb = packed_bias ? at::empty(comb_b_shape, opts) : at::empty({W}, opts);
  std::vector<at::Tensor> dparts;
  std::vector<float*> ptrs;
  for (int64_t i = 0; i < n_parts; ++i) {
    // (padding columns of a padded basis have no parameter behind them: every element of a part is written)
    dparts.push_back(at::empty(part_shape, opts));
    ptrs.push_back(dparts.back().data_ptr<float>());
  }
  // e rides along only in the one-tile kernel (f_in <= 128, ldb + W <= 192) and for at most 128 columns
  const bool ride = !have_es && f_in <= 128 && k <= 192 && f_out <= 128 && (f_out % 4) == 0;
  {
    const int64_t gbytes = (int64_t)egc_weight_grad_ex_workspace_bytes(n, (int32_t)f_in, (int32_t)k, ride ? (int32_t)f_out : 0);
    at::Tensor gws = at::empty({std::max<int64_t>(gbytes, 16)}, opts.dtype(at::kByte));
    check_status(egc_weight_grad_params_f32(x.data_ptr<float>(), f_in, d_cat.data_ptr<float>(), k, n, (int32_t)f_in, (int32_t)H,
                                            (int32_t)A, (int32_t)B, (int32_t)L, (int32_t)Ls, permute_hab ? 1 : 0, ptrs.data(),
                                            (int32_t)n_parts, dcw.data_ptr<float>(), packed_bias ? dcb.data_ptr<float>() : nullptr,
                                            packed_bias ? nullptr : dcb.data_ptr<float>(), ride ? go.data_ptr<float>() : nullptr, f_out,
                                            ride ? (int32_t)f_out : 0, ride ? es.data_ptr<float>() : nullptr, gws.data_ptr(), gws.numel(), st),
                 "egc_weight_grad_params_f32");
  }
  if (!have_es && !ride) {      // the layer's bias gradient on its own: the column sums of grad_out
    const int64_t parts = std::max<int64_t>(1, std::min<int64_t>(1024, (n + 127) / 128));
    at::Tensor psum = at::empty({parts, f_out}, opts);
    check_status(egc_column_sums_f32(go.data_ptr<float>(), n, (int32_t)f_out, (int32_t)f_out, psum.data_ptr<float>(), (int32_t)parts, st),
                 "egc_column_sums_f32");
    if (parts == 1) es = psum[0];
    else check_status(egc_sum_partials_f32(psum.data_ptr<float>(), (int32_t)parts, (int32_t)f_out, es.data_ptr<float>(), st), "egc_sum_partials_f32");
  }
  std::vector<at::Tensor> out{dx, dcw, dcb, es};
  out.insert(out.end(), dparts.begin(), dparts.end());
  return out;
}

std::vector<at::Tensor> train_backward(const at::Tensor& grad_out, const at::Tensor& x, const at::Tensor& wcat,
                                       const at::Tensor& bases, const at::Tensor& weightings, const at::Tensor& stats,
                                       const at::Tensor& cnt, const at::Tensor& arg_max, const at::Tensor& arg_min, int64_t graph,
                                       int64_t t_graph, int64_t layer, int64_t stream, int64_t H, int64_t A, int64_t B, int64_t L,
                                       int64_t Ls, bool permute_hab, bool packed_bias, bool need_x, at::IntArrayRef comb_w_shape,
                                       at::IntArrayRef comb_b_shape, int64_t n_parts, at::IntArrayRef part_shape) {
  return train_backward_impl(grad_out, x, wcat, bases, weightings, stats, cnt, arg_max, arg_min, graph, t_graph, layer, stream, H, A, B, L, Ls,
                             permute_hab, packed_bias, need_x, comb_w_shape, comb_b_shape, n_parts, part_shape, nullptr, nullptr);
}


// ---------------------------------------------------------------------------------------------------------------------
// Training on a BATCH of whole graphs (round 6): the reference's block  x -> x + relu(bn(conv(x)))  (zinc/models.py:66-73,
// mol/pna_style_models.py:71-78; its loop, zinc/configs.py:53-72, is an eager Python loop) as ONE autograd node written in C++:
//   forward   pack both launches' weight planes straight from the parameters (egc_batch_fused_train_pack_params) | the layer as one
//             launch (egc_layer_forward_batch_fused_f32) | BatchNorm's batch statistics + running statistics
//             (egc_bn_forward_stats_f32) | normalise, ReLU, + x (egc_affine_act_residual_f32)
//   backward  BatchNorm / ReLU backward (egc_bn_backward_stats_f32, egc_affine_act_backward_f32) | the layer's backward as one
//             launch with the residual branch's gradient added to d x in its store (egc_layer_backward_batch_fused_f32) | x^T d_cat
//             and both bias sums into the parameters' gradients (egc_weight_grad_params_f32)
// -- the same library calls, in the same order, as egc_amd/functional.py's _BatchFusedTrainFunction +
// _BatchNormActResidualFunction + ResidualLink, without their ctypes marshalling, the two Python autograd Functions and the
// hand-over between them (the eager ZINC step was ~1 ms of host time for 0.37 ms of kernels: profiles/r06_eager_step.md).
// `with_tail` false: the layer alone (a conv called outside a FusedEGCBlock).  Envelope: checked by the Python caller
// (functional._native_block_train); everything outside it stays on the Python path, same kernels.
// ---------------------------------------------------------------------------------------------------------------------
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

struct BlockStatic {
  at::Tensor ptr, edge_ptr, src, dst, max_index, status;      // the batch (kept alive for the backward): edge_ptr / max_index may be undefined
  at::Tensor running_mean, running_var, n_tracked;            // BatchNorm's buffers, updated in place by the forward (may be undefined)
  int64_t host_flag, layer, stream;
  int64_t f_in, H, A, B, L, Ls;
  bool permute_hab;
  int64_t tile_f, emax_f, tile_b, emax_b;
  double eps, momentum;                                        // momentum < 0: cumulative average over n_tracked
  bool relu, residual, with_tail;
};

// BlockStatic <-> AutogradContext::saved_data (plain IValues: tensor list with presence flags, ints, doubles)
static void save_static(AutogradContext* ctx, const BlockStatic& s);
static BlockStatic load_static(AutogradContext* ctx);


static const int64_t* iptr(const at::Tensor& t) { return t.defined() ? t.data_ptr<int64_t>() : nullptr; }

struct BatchBlockTrainFn : public torch::autograd::Function<BatchBlockTrainFn> {
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x, const c10::optional<at::Tensor>& bias, const at::Tensor& comb_w,
                            const c10::optional<at::Tensor>& comb_b, const c10::optional<at::Tensor>& bcat_direct,
                            const c10::optional<at::Tensor>& gamma, const c10::optional<at::Tensor>& beta, at::TensorList parts,
                            const BlockStatic& s) {
    check_f32(x, "x");
    check_f32(comb_w, "comb weight");
    const auto* l = reinterpret_cast<const egc_layer*>(s.layer);
    auto st = reinterpret_cast<egc_stream_t>(s.stream);
    const int64_t n = x.size(0), f_out = l->out_channels;
    TORCH_CHECK(x.dim() == 2 && x.size(1) == s.f_in && s.f_in == l->in_channels, "egc_amd: x has the wrong shape");
    TORCH_CHECK(reinterpret_cast<uintptr_t>(x.data_ptr()) % 16 == 0, "egc_amd: x must be 16-byte aligned");
    TORCH_CHECK(!(comb_b.has_value() && bcat_direct.has_value()), "egc_amd: one combination bias, not two");
    const c10::OptionalDeviceGuard device_guard(x.device());
    const auto opts = x.options();
    // (1) the weight planes of both launches from the parameters
    const int64_t nb = egc_batch_fused_pack_bytes(l), nbt = egc_batch_fused_bwd_pack_bytes(l);
    TORCH_CHECK(nb > 0 && nbt > 0, "egc_amd: layer outside the envelope of the one-launch training path");
    at::Tensor packed = at::empty({nb}, opts.dtype(at::kByte)), packed_t = at::empty({nbt}, opts.dtype(at::kByte));
    std::vector<const float*> pp;
    for (const auto& t : parts) { check_f32(t, "basis matrix"); pp.push_back(t.data_ptr<float>()); }
    if (comb_b.has_value()) check_f32(*comb_b, "comb bias");
    if (bcat_direct.has_value()) check_f32(*bcat_direct, "comb bias");
    if (bias.has_value()) check_f32(*bias, "bias");
    check_status(egc_batch_fused_train_pack_params(l, pp.data(), (int32_t)pp.size(), comb_w.data_ptr<float>(), fptr(comb_b), fptr(bcat_direct),
                                                   (int32_t)s.H, (int32_t)s.A, (int32_t)s.B, (int32_t)s.L, (int32_t)s.Ls, s.permute_hab ? 1 : 0,
                                                   packed.data_ptr(), nb, packed_t.data_ptr(), nbt, st), "egc_batch_fused_train_pack_params");
    // (2) the layer, one launch
    at::Tensor h = at::empty({n, f_out}, opts);
    const int64_t n_graphs = s.ptr.numel() - 1, n_edges = s.src.numel();
    check_status(egc_layer_forward_batch_fused_f32(iptr(s.ptr), iptr(s.edge_ptr), n_graphs, iptr(s.src), iptr(s.dst), n_edges, n,
                                                   s.max_index.defined() ? s.max_index.data_ptr<int32_t>() : nullptr, l, x.data_ptr<float>(),
                                                   packed.data_ptr(), fptr(bias), nullptr, h.data_ptr<float>(), (int32_t)s.tile_f, (int32_t)s.emax_f,
                                                   s.status.data_ptr<int32_t>(), reinterpret_cast<int32_t*>(s.host_flag), st),
                 "egc_layer_forward_batch_fused_f32");
    save_static(ctx, s);
    ctx->saved_data["has"] = std::vector<bool>{bias.has_value(), comb_b.has_value(), bcat_direct.has_value(), gamma.has_value(), beta.has_value()};
    std::vector<std::vector<int64_t>> shapes;
    shapes.push_back(comb_w.sizes().vec());
    shapes.push_back(comb_b.has_value() ? comb_b->sizes().vec() : std::vector<int64_t>{});
    for (const auto& t : parts) shapes.push_back(t.sizes().vec());
    ctx->saved_data["shapes"] = shapes;
    if (!s.with_tail) {
      ctx->save_for_backward({x, packed, packed_t});
      return h;
    }
    // (3) BatchNorm on batch statistics (running statistics updated as nn.BatchNorm1d does), (4) normalise -> ReLU -> + x
    const int64_t c = f_out;
    const int64_t n_parts = std::max<int64_t>(1, std::min<int64_t>(1024, (n + 127) / 128));
    at::Tensor partials = at::empty({n_parts, 2, c}, opts.dtype(at::kDouble)), stats = at::empty({3, c}, opts.dtype(at::kDouble));
    at::Tensor affine = at::empty({2, c}, opts), out = at::empty({n, c}, opts);
    at::Tensor gamma_c = gamma.has_value() ? gamma->detach() : at::Tensor();
    if (gamma.has_value()) check_f32(*gamma, "BatchNorm weight");
    if (beta.has_value()) check_f32(*beta, "BatchNorm bias");
    const bool track = s.running_mean.defined();
    int64_t* cnt = s.n_tracked.defined() ? s.n_tracked.data_ptr<int64_t>() : nullptr;
    check_status(egc_bn_forward_stats_f32(h.data_ptr<float>(), n, (int32_t)c, partials.data_ptr<double>(), (int32_t)n_parts,
                                          track ? cnt : nullptr, nullptr, fptr(gamma), fptr(beta), s.eps, stats.data_ptr<double>(),
                                          affine.data_ptr<float>(), track ? s.running_mean.data_ptr<float>() : nullptr,
                                          track ? s.running_var.data_ptr<float>() : nullptr, s.momentum, cnt, nullptr, st),
                 "egc_bn_forward_stats_f32");
    check_status(egc_affine_act_residual_f32(h.data_ptr<float>(), affine.data_ptr<float>(), affine.data_ptr<float>() + c,
                                             s.residual ? x.data_ptr<float>() : nullptr, s.relu ? 1 : 0, nullptr, 1.0f, n, (int32_t)c,
                                             out.data_ptr<float>(), nullptr, st), "egc_affine_act_residual_f32");
    ctx->save_for_backward({x, packed, packed_t, h, affine, stats, gamma_c});
    return out;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const BlockStatic s = load_static(ctx);
    const auto has = ctx->saved_data["has"].toBoolList();
    const auto shapes = ctx->saved_data["shapes"].to<std::vector<std::vector<int64_t>>>();
    const int64_t n_parts_w = (int64_t)shapes.size() - 2;
    variable_list out(8 + n_parts_w);            // x, bias, comb_w, comb_b, bcat_direct, gamma, beta, parts..., static
    if (!grads[0].defined()) return out;
    const at::Tensor& x = saved[0];
    const at::Tensor &packed = saved[1], &packed_t = saved[2];
    const auto* l = reinterpret_cast<const egc_layer*>(s.layer);
    auto st = reinterpret_cast<egc_stream_t>(s.stream);
    const c10::OptionalDeviceGuard device_guard(x.device());
    const auto opts = x.options();
    const int64_t n = x.size(0), f_in = s.f_in, f_out = l->out_channels, W = s.H * s.B * s.A, ldb = egc_bases_ld(l), k = ldb + W;
    at::Tensor go = grads[0].contiguous();
    check_f32(go, "grad_out");
    at::Tensor g_conv = go, dh_sums;
    if (s.with_tail) {
      const at::Tensor &h = saved[3], &affine = saved[4], &stats = saved[5], &gamma_c = saved[6];
      const int64_t c = f_out;
      const int64_t n_parts = std::max<int64_t>(1, std::min<int64_t>(1024, (n + 127) / 128));
      at::Tensor partials = at::empty({n_parts, 2, c}, opts.dtype(at::kDouble)), out5 = at::empty({5, c}, opts);
      const float* a0 = affine.data_ptr<float>();
      if (has[0]) dh_sums = at::empty({c}, opts);      // the layer's bias gradient = the column sums of dh: from this step's own sums
      check_status(egc_bn_backward_stats_sums_f32(go.data_ptr<float>(), h.data_ptr<float>(), a0, a0 + c, s.relu ? 1 : 0, nullptr, 1.0f, n, (int32_t)c,
                                                  partials.data_ptr<double>(), (int32_t)n_parts, nullptr, stats.data_ptr<double>(),
                                                  gamma_c.defined() ? gamma_c.data_ptr<float>() : nullptr, out5.data_ptr<float>(),
                                                  dh_sums.defined() ? dh_sums.data_ptr<float>() : nullptr, nullptr, st),
                   "egc_bn_backward_stats_sums_f32");
      at::Tensor dh = at::empty({n, c}, opts);
      const float* o5 = out5.data_ptr<float>();
      check_status(egc_affine_act_backward_f32(go.data_ptr<float>(), h.data_ptr<float>(), a0, a0 + c, s.relu ? 1 : 0, nullptr, 1.0f, o5 + 2 * c,
                                               o5 + 3 * c, o5 + 4 * c, n, (int32_t)c, dh.data_ptr<float>(), nullptr, st),
                   "egc_affine_act_backward_f32");
      if (has[3]) out[5] = out5[0];
      if (has[4]) out[6] = out5[1];
      g_conv = dh;
    }
    // the layer's backward, one launch; the residual branch's gradient (x = x + ...) joins d x in its store
    at::Tensor dx = at::empty({n, f_in}, opts), d_cat = at::empty({n, k}, opts);
    const int64_t n_graphs = s.ptr.numel() - 1, n_edges = s.src.numel();
    check_status(egc_layer_backward_batch_fused_f32(iptr(s.ptr), iptr(s.edge_ptr), n_graphs, iptr(s.src), iptr(s.dst), n_edges, n,
                                                    s.max_index.defined() ? s.max_index.data_ptr<int32_t>() : nullptr, l, x.data_ptr<float>(),
                                                    packed.data_ptr(), packed_t.data_ptr(), g_conv.data_ptr<float>(), dx.data_ptr<float>(),
                                                    (s.with_tail && s.residual) ? go.data_ptr<float>() : nullptr, d_cat.data_ptr<float>(),
                                                    (int32_t)k, (int32_t)s.tile_b, (int32_t)s.emax_b, s.status.data_ptr<int32_t>(),
                                                    reinterpret_cast<int32_t*>(s.host_flag), st), "egc_layer_backward_batch_fused_f32");
    out[0] = dx;
    // x^T d_cat + the column sums of d_cat's weightings part (combination bias) and of the layer's incoming gradient (its bias)
    at::Tensor es = at::empty({f_out}, opts), dcw = at::empty(shapes[0], opts);
    at::Tensor dcb = has[1] ? at::empty(shapes[1], opts) : at::empty({W}, opts);
    std::vector<at::Tensor> dparts;
    std::vector<float*> ptrs;
    for (int64_t i = 0; i < n_parts_w; ++i) {
      dparts.push_back(at::empty(shapes[2 + i], opts));
      ptrs.push_back(dparts.back().data_ptr<float>());
    }
    // (the layer's bias gradient: from the BatchNorm step above where there is one, else -- and when nobody asks -- no third stream)
    const bool ride = has[0] && !dh_sums.defined();
    const int64_t gbytes = egc_weight_grad_ex_workspace_bytes(n, (int32_t)f_in, (int32_t)k, ride ? (int32_t)f_out : 0);
    at::Tensor gws = at::empty({std::max<int64_t>(gbytes, 16)}, opts.dtype(at::kByte));
    check_status(egc_weight_grad_params_f32(x.data_ptr<float>(), f_in, d_cat.data_ptr<float>(), k, n, (int32_t)f_in, (int32_t)s.H, (int32_t)s.A,
                                            (int32_t)s.B, (int32_t)s.L, (int32_t)s.Ls, s.permute_hab ? 1 : 0, ptrs.data(), (int32_t)n_parts_w,
                                            dcw.data_ptr<float>(), has[1] ? dcb.data_ptr<float>() : nullptr, has[1] ? nullptr : dcb.data_ptr<float>(),
                                            ride ? g_conv.data_ptr<float>() : nullptr, f_out, ride ? (int32_t)f_out : 0,
                                            ride ? es.data_ptr<float>() : nullptr, gws.data_ptr(), gws.numel(),
                                            reinterpret_cast<void*>(s.stream)), "egc_weight_grad_params_f32");
    if (has[0]) out[1] = dh_sums.defined() ? dh_sums : es;
    out[2] = dcw;
    if (has[1]) out[3] = dcb;
    if (has[2]) out[4] = dcb;
    for (int64_t i = 0; i < n_parts_w; ++i) out[7 + i] = dparts[i];
    return out;
  }
};

static void save_static(AutogradContext* ctx, const BlockStatic& s) {
  // (the batch's tensors are kept alive for the backward; BatchNorm's buffers are not needed there)
  std::vector<at::Tensor> ts;
  std::vector<int64_t> present;
  for (const at::Tensor* t : {&s.ptr, &s.edge_ptr, &s.src, &s.dst, &s.max_index, &s.status}) {
    present.push_back(t->defined() ? 1 : 0);
    ts.push_back(t->defined() ? *t : s.ptr);
  }
  ctx->saved_data["batch"] = ts;
  ctx->saved_data["present"] = present;
  ctx->saved_data["ints"] = std::vector<int64_t>{s.host_flag, s.layer, s.stream, s.f_in, s.H, s.A, s.B, s.L, s.Ls, s.permute_hab ? 1 : 0, s.tile_f,
                                                s.emax_f, s.tile_b, s.emax_b, s.relu ? 1 : 0, s.residual ? 1 : 0, s.with_tail ? 1 : 0};
}
static BlockStatic load_static(AutogradContext* ctx) {
  BlockStatic s;
  const auto ts = ctx->saved_data["batch"].toTensorVector();
  const auto pr = ctx->saved_data["present"].toIntVector();
  at::Tensor* dst[6] = {&s.ptr, &s.edge_ptr, &s.src, &s.dst, &s.max_index, &s.status};
  for (int i = 0; i < 6; ++i) if (pr[i]) *dst[i] = ts[i];
  const auto v = ctx->saved_data["ints"].toIntVector();
  s.host_flag = v[0]; s.layer = v[1]; s.stream = v[2]; s.f_in = v[3]; s.H = v[4]; s.A = v[5]; s.B = v[6]; s.L = v[7]; s.Ls = v[8];
  s.permute_hab = v[9] != 0; s.tile_f = v[10]; s.emax_f = v[11]; s.tile_b = v[12]; s.emax_b = v[13];
  s.relu = v[14] != 0; s.residual = v[15] != 0; s.with_tail = v[16] != 0;
  s.eps = 0; s.momentum = 0;
  return s;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same block on the CSR path (round 6): x -> x + relu(bn(conv(x))) for layers and batches outside the one-launch training
// envelope -- the reference's own batched nets (168 / 224 / 296 / 300 / 304 wide: run_pretrained.sh:7-48), full graphs -- as ONE
// autograd node: train_forward (pack, planes, GEMM, training aggregate) + the BatchNorm tail forward; the tail's backward +
// train_backward (sparse backward, d x GEMM, dense gradients into the parameters) + the residual branch's gradient backward.
// The graph structs are COPIED into the node (their Python owners may be gone when the backward runs); `keep` holds the device
// tensors they point into.
// ---------------------------------------------------------------------------------------------------------------------
struct CsrStatic {
  egc_graph g, tg;
  at::Tensor workspace, running_mean, running_var, n_tracked;
  std::vector<at::Tensor> keep;
  int64_t layer, stream, H, A, B, L, Ls, gemm_flags;
  bool permute_hab;
  double eps, momentum;
  bool relu, residual, with_tail;
};

struct CsrBlockTrainFn : public torch::autograd::Function<CsrBlockTrainFn> {
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x, const c10::optional<at::Tensor>& bias, const at::Tensor& comb_w,
                            const c10::optional<at::Tensor>& comb_b, const c10::optional<at::Tensor>& bcat_direct,
                            const c10::optional<at::Tensor>& gamma, const c10::optional<at::Tensor>& beta, at::TensorList parts,
                            const CsrStatic& s) {
    const c10::OptionalDeviceGuard device_guard(x.device());
    auto r = train_forward(x, comb_w, comb_b, bcat_direct, parts, bias, reinterpret_cast<int64_t>(&s.g), s.layer, s.workspace, s.stream,
                           s.H, s.A, s.B, s.L, s.Ls, s.permute_hab, s.gemm_flags);
    const at::Tensor& h = r[0];
    const auto* l = reinterpret_cast<const egc_layer*>(s.layer);
    auto st = reinterpret_cast<egc_stream_t>(s.stream);
    const int64_t n = x.size(0), c = l->out_channels;
    const auto opts = x.options();
    ctx->saved_data["graphs"] = std::string(reinterpret_cast<const char*>(&s.g), sizeof(egc_graph)) +
                                std::string(reinterpret_cast<const char*>(&s.tg), sizeof(egc_graph));
    ctx->saved_data["keep"] = s.keep;
    ctx->saved_data["ints"] = std::vector<int64_t>{s.layer, s.stream, s.H, s.A, s.B, s.L, s.Ls, s.permute_hab ? 1 : 0, s.relu ? 1 : 0,
                                                  s.residual ? 1 : 0, s.with_tail ? 1 : 0, comb_b.has_value() ? 1 : 0,
                                                  bias.has_value() ? 1 : 0, gamma.has_value() ? 1 : 0, beta.has_value() ? 1 : 0,
                                                  (int64_t)parts.size()};
    ctx->saved_data["comb_w_shape"] = comb_w.sizes().vec();
    ctx->saved_data["comb_b_shape"] = comb_b.has_value() ? comb_b->sizes().vec() : std::vector<int64_t>{0};
    ctx->saved_data["part_shape"] = parts[0].sizes().vec();
    if (!s.with_tail) {
      ctx->save_for_backward({x, r[1], r[3], r[4], r[5], r[6], r[7], r[8]});
      return h;
    }
    const int64_t n_parts = std::max<int64_t>(1, std::min<int64_t>(1024, (n + 127) / 128));
    at::Tensor partials = at::empty({n_parts, 2, c}, opts.dtype(at::kDouble)), stats = at::empty({3, c}, opts.dtype(at::kDouble));
    at::Tensor affine = at::empty({2, c}, opts), out = at::empty({n, c}, opts);
    at::Tensor gamma_c = gamma.has_value() ? gamma->detach() : at::Tensor();
    if (gamma.has_value()) check_f32(*gamma, "BatchNorm weight");
    if (beta.has_value()) check_f32(*beta, "BatchNorm bias");
    const bool track = s.running_mean.defined();
    int64_t* cnt = s.n_tracked.defined() ? s.n_tracked.data_ptr<int64_t>() : nullptr;
    check_status(egc_bn_forward_stats_f32(h.data_ptr<float>(), n, (int32_t)c, partials.data_ptr<double>(), (int32_t)n_parts,
                                          track ? cnt : nullptr, nullptr, fptr(gamma), fptr(beta), s.eps, stats.data_ptr<double>(),
                                          affine.data_ptr<float>(), track ? s.running_mean.data_ptr<float>() : nullptr,
                                          track ? s.running_var.data_ptr<float>() : nullptr, s.momentum, cnt, nullptr, st),
                 "egc_bn_forward_stats_f32");
    check_status(egc_affine_act_residual_f32(h.data_ptr<float>(), affine.data_ptr<float>(), affine.data_ptr<float>() + c,
                                             s.residual ? x.data_ptr<float>() : nullptr, s.relu ? 1 : 0, nullptr, 1.0f, n, (int32_t)c,
                                             out.data_ptr<float>(), nullptr, st), "egc_affine_act_residual_f32");
    ctx->save_for_backward({x, r[1], r[3], r[4], r[5], r[6], r[7], r[8], h, affine, stats, gamma_c});
    return out;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const auto v = ctx->saved_data["ints"].toIntVector();
    const int64_t layer = v[0], stream = v[1], H = v[2], A = v[3], B = v[4], L = v[5], Ls = v[6], n_parts_w = v[15];
    const bool permute = v[7] != 0, relu = v[8] != 0, residual = v[9] != 0, with_tail = v[10] != 0, packed_bias = v[11] != 0;
    const bool has_bias = v[12] != 0, has_gamma = v[13] != 0, has_beta = v[14] != 0;
    variable_list out(8 + n_parts_w);            // x, bias, comb_w, comb_b, bcat_direct, gamma, beta, parts..., static
    if (!grads[0].defined()) return out;
    const std::string gs = ctx->saved_data["graphs"].toStringRef();
    egc_graph g, tg;
    std::memcpy(&g, gs.data(), sizeof(egc_graph));
    std::memcpy(&tg, gs.data() + sizeof(egc_graph), sizeof(egc_graph));
    const at::Tensor& x = saved[0];
    const c10::OptionalDeviceGuard device_guard(x.device());
    const auto opts = x.options();
    auto st = reinterpret_cast<egc_stream_t>(stream);
    const auto* l = reinterpret_cast<const egc_layer*>(layer);
    const int64_t n = x.size(0), c = l->out_channels;
    at::Tensor go = grads[0].contiguous();
    check_f32(go, "grad_out");
    at::Tensor g_conv = go, dh_sums;
    if (with_tail) {
      const at::Tensor &h = saved[8], &affine = saved[9], &stats = saved[10], &gamma_c = saved[11];
      const int64_t n_parts = std::max<int64_t>(1, std::min<int64_t>(1024, (n + 127) / 128));
      at::Tensor partials = at::empty({n_parts, 2, c}, opts.dtype(at::kDouble)), out5 = at::empty({5, c}, opts);
      const float* a0 = affine.data_ptr<float>();
      dh_sums = at::empty({c}, opts);                   // the layer's bias gradient = the column sums of dh: from this step's own sums
      check_status(egc_bn_backward_stats_sums_f32(go.data_ptr<float>(), h.data_ptr<float>(), a0, a0 + c, relu ? 1 : 0, nullptr, 1.0f, n, (int32_t)c,
                                                  partials.data_ptr<double>(), (int32_t)n_parts, nullptr, stats.data_ptr<double>(),
                                                  gamma_c.defined() ? gamma_c.data_ptr<float>() : nullptr, out5.data_ptr<float>(),
                                                  dh_sums.data_ptr<float>(), nullptr, st),
                   "egc_bn_backward_stats_sums_f32");
      at::Tensor dh = at::empty({n, c}, opts);
      const float* o5 = out5.data_ptr<float>();
      check_status(egc_affine_act_backward_f32(go.data_ptr<float>(), h.data_ptr<float>(), a0, a0 + c, relu ? 1 : 0, nullptr, 1.0f, o5 + 2 * c,
                                               o5 + 3 * c, o5 + 4 * c, n, (int32_t)c, dh.data_ptr<float>(), nullptr, st),
                   "egc_affine_act_backward_f32");
      if (has_gamma) out[5] = out5[0];
      if (has_beta) out[6] = out5[1];
      g_conv = dh;
    }
    const auto cws = ctx->saved_data["comb_w_shape"].toIntVector(), cbs = ctx->saved_data["comb_b_shape"].toIntVector();
    const auto ps = ctx->saved_data["part_shape"].toIntVector();
    // the residual branch's gradient (x = x + ...) joins d x in the d x GEMM's store where that kernel takes an addend
    auto r = train_backward_impl(g_conv, x, saved[1], saved[2], saved[3], saved[4], saved[5], saved[6], saved[7], reinterpret_cast<int64_t>(&g),
                                 reinterpret_cast<int64_t>(&tg), layer, stream, H, A, B, L, Ls, permute, packed_bias, true, cws, cbs, n_parts_w, ps,
                                 (with_tail && residual) ? &go : nullptr, dh_sums.defined() ? &dh_sums : nullptr);
    at::Tensor dx = r[0];
    out[0] = dx;
    if (has_bias) out[1] = r[3];
    out[2] = r[1];
    if (packed_bias) out[3] = r[2]; else out[4] = r[2];
    for (int64_t i = 0; i < n_parts_w; ++i) out[7 + i] = r[4 + i];
    return out;
  }
};

at::Tensor csr_block_train(const at::Tensor& x, const c10::optional<at::Tensor>& bias, const at::Tensor& comb_w,
                           const c10::optional<at::Tensor>& comb_b, const c10::optional<at::Tensor>& bcat_direct,
                           const c10::optional<at::Tensor>& gamma, const c10::optional<at::Tensor>& beta, at::TensorList parts,
                           const c10::optional<at::Tensor>& running_mean, const c10::optional<at::Tensor>& running_var,
                           const c10::optional<at::Tensor>& n_tracked, int64_t graph, int64_t t_graph, at::TensorList keep,
                           const at::Tensor& workspace, int64_t layer, int64_t stream, at::IntArrayRef dims, bool permute_hab,
                           int64_t gemm_flags, double eps, double momentum, bool relu, bool residual, bool with_tail) {
  TORCH_CHECK(dims.size() == 5, "egc_amd: dims = (H, A, B, L, Ls)");
  TORCH_CHECK(parts.size() >= 1, "egc_amd: at least one basis matrix");
  CsrStatic s;
  s.g = *reinterpret_cast<const egc_graph*>(graph);
  s.tg = *reinterpret_cast<const egc_graph*>(t_graph);
  s.keep = keep.vec();
  s.workspace = workspace;
  if (running_mean.has_value()) {
    TORCH_CHECK(running_var.has_value(), "egc_amd: running_mean and running_var come together");
    check_f32(*running_mean, "running_mean"); check_f32(*running_var, "running_var");
    s.running_mean = *running_mean; s.running_var = *running_var;
  }
  if (n_tracked.has_value()) { TORCH_CHECK(n_tracked->is_cuda() && n_tracked->scalar_type() == at::kLong && n_tracked->numel() == 1, "egc_amd: num_batches_tracked"); s.n_tracked = *n_tracked; }
  TORCH_CHECK(momentum >= 0 || !s.running_mean.defined() || s.n_tracked.defined(), "egc_amd: a cumulative average needs num_batches_tracked");
  s.layer = layer; s.stream = stream;
  s.H = dims[0]; s.A = dims[1]; s.B = dims[2]; s.L = dims[3]; s.Ls = dims[4];
  s.gemm_flags = gemm_flags; s.permute_hab = permute_hab;
  s.eps = eps; s.momentum = momentum; s.relu = relu; s.residual = residual; s.with_tail = with_tail;
  return CsrBlockTrainFn::apply(x, bias, comb_w, comb_b, bcat_direct, gamma, beta, parts, s);
}

at::Tensor batch_block_train(const at::Tensor& x, const c10::optional<at::Tensor>& bias, const at::Tensor& comb_w,
                             const c10::optional<at::Tensor>& comb_b, const c10::optional<at::Tensor>& bcat_direct,
                             const c10::optional<at::Tensor>& gamma, const c10::optional<at::Tensor>& beta, at::TensorList parts,
                             const c10::optional<at::Tensor>& running_mean, const c10::optional<at::Tensor>& running_var,
                             const c10::optional<at::Tensor>& n_tracked, const at::Tensor& ptr, const c10::optional<at::Tensor>& edge_ptr,
                             const at::Tensor& src, const at::Tensor& dst, const c10::optional<at::Tensor>& max_index, const at::Tensor& status,
                             int64_t host_flag, int64_t layer, int64_t stream, at::IntArrayRef dims, bool permute_hab, at::IntArrayRef setups,
                             double eps, double momentum, bool relu, bool residual, bool with_tail) {
  TORCH_CHECK(dims.size() == 6 && setups.size() == 4, "egc_amd: dims = (f_in, H, A, B, L, Ls), setups = (tile_f, emax_f, tile_b, emax_b)");
  auto i64 = [](const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kLong && t.is_contiguous(), "egc_amd: ", name, " must be a dense int64 tensor on the device");
  };
  i64(ptr, "ptr"); i64(src, "edge_index[0]"); i64(dst, "edge_index[1]");
  TORCH_CHECK(src.numel() == dst.numel() && ptr.numel() >= 1, "egc_amd: malformed batch");
  TORCH_CHECK(status.is_cuda() && status.scalar_type() == at::kInt, "egc_amd: status word");
  BlockStatic s;
  s.ptr = ptr; s.src = src; s.dst = dst; s.status = status;
  if (edge_ptr.has_value()) { i64(*edge_ptr, "edge_ptr"); TORCH_CHECK(edge_ptr->numel() == ptr.numel(), "egc_amd: edge_ptr"); s.edge_ptr = *edge_ptr; }
  if (max_index.has_value()) { TORCH_CHECK(max_index->is_cuda() && max_index->scalar_type() == at::kInt, "egc_amd: max_index"); s.max_index = *max_index; }
  if (running_mean.has_value()) {
    TORCH_CHECK(running_var.has_value(), "egc_amd: running_mean and running_var come together");
    check_f32(*running_mean, "running_mean"); check_f32(*running_var, "running_var");
    s.running_mean = *running_mean; s.running_var = *running_var;
  }
  if (n_tracked.has_value()) { TORCH_CHECK(n_tracked->is_cuda() && n_tracked->scalar_type() == at::kLong && n_tracked->numel() == 1, "egc_amd: num_batches_tracked"); s.n_tracked = *n_tracked; }
  TORCH_CHECK(momentum >= 0 || !s.running_mean.defined() || s.n_tracked.defined(), "egc_amd: a cumulative average needs num_batches_tracked");
  s.host_flag = host_flag; s.layer = layer; s.stream = stream;
  s.f_in = dims[0]; s.H = dims[1]; s.A = dims[2]; s.B = dims[3]; s.L = dims[4]; s.Ls = dims[5];
  s.permute_hab = permute_hab;
  s.tile_f = setups[0]; s.emax_f = setups[1]; s.tile_b = setups[2]; s.emax_b = setups[3];
  s.eps = eps; s.momentum = momentum; s.relu = relu; s.residual = residual; s.with_tail = with_tail;
  return BatchBlockTrainFn::apply(x, bias, comb_w, comb_b, bcat_direct, gamma, beta, parts, s);
}

}  // namespace

TORCH_LIBRARY(egc_amd_native, m) {
  m.def("layer_forward(Tensor x, Tensor packed, Tensor? bcat, Tensor? bias, int graph, int layer, Tensor workspace, int stream, "
        "int ldb, int w_cols, int f_out) -> Tensor");
  m.def("layer_forward_post(Tensor x, Tensor packed, Tensor? bcat, Tensor? bias, int graph, int layer, Tensor workspace, "
        "int stream, int ldb, int w_cols, int f_out, int gemm_flags, Tensor? scale, Tensor? shift, Tensor? residual, bool relu) "
        "-> Tensor");
  m.def("train_forward(Tensor x, Tensor comb_w, Tensor? comb_b, Tensor? bcat_direct, Tensor[] basis_parts, Tensor? bias, int graph, "
        "int layer, Tensor workspace, int stream, int H, int A, int B, int L, int Ls, bool permute_hab, int gemm_flags) -> Tensor[]");
  m.def("train_backward(Tensor grad_out, Tensor x, Tensor wcat, Tensor bases, Tensor weightings, Tensor stats, Tensor cnt, "
        "Tensor arg_max, Tensor arg_min, int graph, int t_graph, int layer, int stream, int H, int A, int B, int L, int Ls, "
        "bool permute_hab, bool packed_bias, bool need_x, int[] comb_w_shape, int[] comb_b_shape, int n_parts, int[] part_shape) "
        "-> Tensor[]");
  m.def("batch_block_train(Tensor x, Tensor? bias, Tensor comb_w, Tensor? comb_b, Tensor? bcat_direct, Tensor? gamma, Tensor? beta, "
        "Tensor[] parts, Tensor(a!)? running_mean, Tensor(b!)? running_var, Tensor(c!)? n_tracked, Tensor ptr, Tensor? edge_ptr, Tensor src, "
        "Tensor dst, Tensor? max_index, Tensor status, int host_flag, int layer, int stream, int[] dims, bool permute_hab, int[] setups, "
        "float eps, float momentum, bool relu, bool residual, bool with_tail) -> Tensor");
  m.def("csr_block_train(Tensor x, Tensor? bias, Tensor comb_w, Tensor? comb_b, Tensor? bcat_direct, Tensor? gamma, Tensor? beta, "
        "Tensor[] parts, Tensor(a!)? running_mean, Tensor(b!)? running_var, Tensor(c!)? n_tracked, int graph, int t_graph, Tensor[] keep, "
        "Tensor workspace, int layer, int stream, int[] dims, bool permute_hab, int gemm_flags, float eps, float momentum, bool relu, "
        "bool residual, bool with_tail) -> Tensor");
}

// HIP devices only (PyTorch-ROCm dispatches them under the CUDA key): a CPU tensor finds no kernel and the dispatcher raises
TORCH_LIBRARY_IMPL(egc_amd_native, CUDA, m) {
  m.impl("layer_forward", &layer_forward);
  m.impl("layer_forward_post", &layer_forward_post);
  m.impl("train_forward", &train_forward);
  m.impl("train_backward", &train_backward);
}

// the batch training block carries its own autograd node (BatchBlockTrainFn): registered at the Autograd key, which is where a call
// with parameters that require gradients enters; the node's forward issues the library's launches itself (no dispatcher re-entry)
TORCH_LIBRARY_IMPL(egc_amd_native, Autograd, m) {
  m.impl("batch_block_train", &batch_block_train);
  m.impl("csr_block_train", &csr_block_train);
}
