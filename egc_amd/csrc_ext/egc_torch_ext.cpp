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
#include <torch/library.h>

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

// out = layer(x) -- basis GEMM (split-precision planes `packed`) + fused aggregate/combine, two launches, one call
at::Tensor layer_forward(const at::Tensor& x, const at::Tensor& packed, const c10::optional<at::Tensor>& bcat,
                         const c10::optional<at::Tensor>& bias, int64_t graph, int64_t layer, const at::Tensor& workspace,
                         int64_t stream, int64_t ldb, int64_t w_cols, int64_t f_out) {
  check_f32(x, "x");
  const auto* g = reinterpret_cast<const egc_graph*>(graph);
  const auto* l = reinterpret_cast<const egc_layer*>(layer);
  TORCH_CHECK(x.dim() == 2 && x.size(0) == g->n_nodes && x.size(1) == l->in_channels, "egc_amd: x has the wrong shape");
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
  if (residual.has_value()) check_f32(*residual, "post.residual");
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

}  // namespace

TORCH_LIBRARY(egc_amd_native, m) {
  m.def("layer_forward(Tensor x, Tensor packed, Tensor? bcat, Tensor? bias, int graph, int layer, Tensor workspace, int stream, "
        "int ldb, int w_cols, int f_out) -> Tensor");
  m.def("layer_forward_post(Tensor x, Tensor packed, Tensor? bcat, Tensor? bias, int graph, int layer, Tensor workspace, "
        "int stream, int ldb, int w_cols, int f_out, int gemm_flags, Tensor? scale, Tensor? shift, Tensor? residual, bool relu) "
        "-> Tensor");
}

// HIP devices only (PyTorch-ROCm dispatches them under the CUDA key): a CPU tensor finds no kernel and the dispatcher raises
TORCH_LIBRARY_IMPL(egc_amd_native, CUDA, m) {
  m.impl("layer_forward", &layer_forward);
  m.impl("layer_forward_post", &layer_forward_post);
}
