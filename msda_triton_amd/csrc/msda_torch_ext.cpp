// msda_torch_ext.cpp — optional thin PyTorch binding over the C ABI (include/msda_hip.h).
//
// The product boundary is the C ABI; Python reaches it through ctypes (msda_triton_amd/_lib.py).  For
// Grounding-DINO-sized calls (B*Q ~ 10^3) the kernels take ~60 us while Python's autograd glue, ctypes
// marshalling and the engine's hop into a Python backward take ~100 us per forward+backward.  This extension is
// the same glue in C++: one torch::autograd::Function whose forward and backward call msda_fwd_<dtype> /
// msda_bwd_<dtype> directly.  It contains no kernels and no numerics; the argument validation stays in Python
// (msda_triton_amd/functional.py) and the same tests cover both routes.
// Reference counterpart: _TritonMultiscaleDeformableAttentionFunction, src/msda_triton/frontend.py:108-142.
#include <torch/extension.h>
#include <torch/csrc/autograd/functions/basic_ops.h>

#include <c10/hip/HIPStream.h>

#include "../../include/msda_hip.h"

namespace {

using FwdFn = int (*)(const void *, const int64_t *, const void *, const void *, void *, int64_t, int64_t, int64_t,
                      int64_t, int64_t, int64_t, int64_t, int, int, void *);
// (the level-size bound travels as an argument: include/msda_hip.h, max_level_cells)
using BwdFn = int (*)(const void *, const void *, const int64_t *, const void *, const void *, void *, void *, void *,
                      int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int, int64_t, void *, int64_t,
                      void *);

using FwdFusedFn = int (*)(const void *, const int64_t *, const void *, const void *, void *, int64_t, int64_t, int64_t,
                           int64_t, int64_t, int64_t, int64_t, int, int, int, void *);
using BwdFusedFn = int (*)(const void *, const void *, const int64_t *, const void *, const void *, void *, void *, void *,
                           int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int, int, int64_t, void *,
                           int64_t, void *);

struct Fns {
    FwdFn fwd;
    BwdFn bwd;
    FwdFusedFn fwd_fused;
    BwdFusedFn bwd_fused;
};

// `t`: dtype of the value pyramid (and its gradient); `c`: dtype of every other tensor — the same, or float next to a
// 16-bit pyramid (the mixed-storage entry points)
Fns fns_for(at::ScalarType t, at::ScalarType c)
{
    if (t != c) {
        TORCH_CHECK_VALUE(c == at::kFloat && (t == at::kBFloat16 || t == at::kHalf),
                          "unsupported dtype combination: value ", t, " with ", c);
        if (t == at::kBFloat16)
            return {msda_fwd_f32_vbf16, msda_bwd_f32_vbf16, msda_fwd_fused_f32_vbf16, msda_bwd_fused_f32_vbf16};
        return {msda_fwd_f32_vf16, msda_bwd_f32_vf16, msda_fwd_fused_f32_vf16, msda_bwd_fused_f32_vf16};
    }
    switch (t) {
    case at::kFloat: return {msda_fwd_f32, msda_bwd_f32, msda_fwd_fused_f32, msda_bwd_fused_f32};
    case at::kHalf: return {msda_fwd_f16, msda_bwd_f16, msda_fwd_fused_f16, msda_bwd_fused_f16};
    case at::kBFloat16: return {msda_fwd_bf16, msda_bwd_bf16, msda_fwd_fused_bf16, msda_bwd_fused_bf16};
    case at::kDouble: return {msda_fwd_f64, msda_bwd_f64, msda_fwd_fused_f64, msda_bwd_fused_f64};
    default: TORCH_CHECK_VALUE(false, "unsupported dtype ", t);
    }
}

void check_rc(int rc, const char *what)
{
    if (rc == 0) return;
    TORCH_CHECK_VALUE(rc > 0, what, ": rejected arguments (", rc, "): ", msda_last_error());
    TORCH_CHECK(false, what, ": HIP error ", rc, ": ", msda_last_error());
}

// torch.autograd.function.once_differentiable (frontend.py:130 of the reference) for a C++ Function: when the
// backward itself runs under grad mode (create_graph=True) and was handed a differentiable gradient, the results
// are routed through a DelayedError node, so a second differentiation raises instead of silently treating them as
// constants.
torch::autograd::variable_list once_differentiable(const torch::autograd::variable_list &grads_in,
                                                   torch::autograd::variable_list outs)
{
    if (!at::GradMode::is_enabled()) return outs;
    bool any = false;
    for (const auto &g : grads_in) any = any || (g.defined() && g.requires_grad());
    if (!any) return outs;
    for (auto &o : outs)
        if (o.defined()) o = o.detach().requires_grad_(true);
    auto err = std::make_shared<torch::autograd::DelayedError>(
        "trying to differentiate twice a function that was marked with @once_differentiable", (int64_t)outs.size());
    return (*err)(std::move(outs));
}

void *current_stream(const at::Tensor &t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

class MSDAFunction : public torch::autograd::Function<MSDAFunction> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &img_, const at::Tensor &shapes_,
                              const at::Tensor &pts_, const at::Tensor &att_, int64_t padding_mode, bool align_corners,
                              int64_t level_cells)
    {
        const at::Tensor img = img_.contiguous(), pts = pts_.contiguous(), att = att_.contiguous();
        const at::Tensor shapes = shapes_.to(at::kLong).contiguous();  // stays on the device
        const int64_t B = img.size(0), I = img.size(1), H = img.size(2), D = img.size(3);
        const int64_t Q = pts.size(1), L = pts.size(3), P = pts.size(4);
        at::Tensor out = at::empty({B, Q, H, D}, pts.options());
        const c10::DeviceGuard guard(img.device());
        check_rc(fns_for(img.scalar_type(), pts.scalar_type())
                     .fwd(img.data_ptr(), shapes.data_ptr<int64_t>(), pts.data_ptr(), att.data_ptr(), out.data_ptr(), B, I,
                          H, D, Q, L, P, (int)padding_mode, align_corners ? 1 : 0, current_stream(img)),
                 "msda_fwd");
        ctx->save_for_backward({img, shapes, pts, att});
        ctx->saved_data["padding_mode"] = padding_mode;
        ctx->saved_data["align_corners"] = align_corners;
        ctx->saved_data["level_cells"] = level_cells;
        return out;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx,
                                                   torch::autograd::variable_list grads)
    {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor &img = saved[0], &shapes = saved[1], &pts = saved[2], &att = saved[3];
        const int padding_mode = (int)ctx->saved_data["padding_mode"].toInt();
        const bool align_corners = ctx->saved_data["align_corners"].toBool();
        const int64_t level_cells = ctx->saved_data["level_cells"].toInt();  // bound on the largest level's cells (0: unknown)
        at::Tensor gout = grads[0].contiguous();
        if (gout.scalar_type() != pts.scalar_type()) gout = gout.to(pts.scalar_type());
        const bool want_value = ctx->needs_input_grad(0);
        const bool want_sample = ctx->needs_input_grad(2) || ctx->needs_input_grad(3);
        const int64_t B = img.size(0), I = img.size(1), H = img.size(2), D = img.size(3);
        const int64_t Q = pts.size(1), L = pts.size(3), P = pts.size(4);
        at::Tensor g_img, g_pts, g_att, ws;
        int64_t ws_bytes = 0;
        if (want_sample) {
            g_pts = at::empty_like(pts);
            g_att = at::empty_like(att);
        }
        if (want_value) {
            g_img = at::empty_like(img);
            // all three gradients in one call: the sorted records may use the gradient buffers themselves (the library's
            // own conditions: 16-byte aligned buffers, no forced side-stream fork)
            const bool in_grads = want_sample && reinterpret_cast<uintptr_t>(g_pts.data_ptr()) % 16 == 0 &&
                                  reinterpret_cast<uintptr_t>(g_att.data_ptr()) % 16 == 0 &&
                                  reinterpret_cast<uintptr_t>(g_img.data_ptr()) % 16 == 0 && msda_get_option("overlap") != 1;
            ws_bytes = msda_bwd_workspace_bytes(B, I, H, D, Q, L, P, (int)pts.element_size(), (int)img.element_size(),
                                                level_cells, in_grads ? MSDA_WS_RECORDS_IN_GRADS : 0);
            ws = at::empty({ws_bytes}, img.options().dtype(at::kByte));  // scratch: no initialisation needed
        }
        if (want_value || want_sample) {
            const c10::DeviceGuard guard(img.device());
            check_rc(fns_for(img.scalar_type(), pts.scalar_type())
                         .bwd(gout.data_ptr(), img.data_ptr(), shapes.data_ptr<int64_t>(), pts.data_ptr(), att.data_ptr(),
                              want_value ? g_img.data_ptr() : nullptr, want_sample ? g_pts.data_ptr() : nullptr,
                              want_sample ? g_att.data_ptr() : nullptr, B, I, H, D, Q, L, P, padding_mode,
                              align_corners ? 1 : 0, level_cells, ws.defined() ? ws.data_ptr() : nullptr, ws_bytes,
                              current_stream(img)),
                     "msda_bwd");
        }
        return once_differentiable(grads, {g_img, at::Tensor(), ctx->needs_input_grad(2) ? g_pts : at::Tensor(),
                                           ctx->needs_input_grad(3) ? g_att : at::Tensor(), at::Tensor(), at::Tensor(),
                                           at::Tensor()});
    }
};

// The module core with its prologue fused in (msda_fwd_fused_ / msda_bwd_fused_<dtype>).  The caller has checked
// L*P <= msda_fused_lp_limit(D, element size): the library then never declines.
class MSDAFusedFunction : public torch::autograd::Function<MSDAFusedFunction> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &img_, const at::Tensor &shapes_,
                              const at::Tensor &proj_, const at::Tensor &ref_, int64_t padding_mode, bool align_corners,
                              int64_t level_cells)
    {
        const at::Tensor img = img_.contiguous(), proj = proj_.contiguous(), ref = ref_.contiguous();
        const at::Tensor shapes = shapes_.to(at::kLong).contiguous();
        const int64_t B = img.size(0), I = img.size(1), H = img.size(2), D = img.size(3);
        const int64_t Q = proj.size(1), L = proj.size(3), P = proj.size(4);
        at::Tensor out = at::empty({B, Q, H, D}, proj.options());
        const c10::DeviceGuard guard(img.device());
        check_rc(fns_for(img.scalar_type(), proj.scalar_type())
                     .fwd_fused(img.data_ptr(), shapes.data_ptr<int64_t>(), proj.data_ptr(), ref.data_ptr(), out.data_ptr(),
                                B, I, H, D, Q, L, P, (int)ref.size(-1), (int)padding_mode, align_corners ? 1 : 0,
                                current_stream(img)),
                 "msda_fwd_fused");
        ctx->save_for_backward({img, shapes, proj, ref});
        ctx->saved_data["padding_mode"] = padding_mode;
        ctx->saved_data["align_corners"] = align_corners;
        ctx->saved_data["level_cells"] = level_cells;
        return out;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx,
                                                   torch::autograd::variable_list grads)
    {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor &img = saved[0], &shapes = saved[1], &proj = saved[2], &ref = saved[3];
        const int padding_mode = (int)ctx->saved_data["padding_mode"].toInt();
        const bool align_corners = ctx->saved_data["align_corners"].toBool();
        const int64_t level_cells = ctx->saved_data["level_cells"].toInt();
        at::Tensor gout = grads[0].contiguous();
        if (gout.scalar_type() != proj.scalar_type()) gout = gout.to(proj.scalar_type());
        const bool want_value = ctx->needs_input_grad(0);
        const int64_t B = img.size(0), I = img.size(1), H = img.size(2), D = img.size(3);
        const int64_t Q = proj.size(1), L = proj.size(3), P = proj.size(4);
        const int64_t ref_dim = ref.size(-1);
        at::Tensor g_img, ws;
        at::Tensor g_proj = at::empty_like(proj), g_ref_part = at::empty({B, Q, H, ref_dim}, proj.options());
        int64_t ws_bytes = 0;
        if (want_value) {
            g_img = at::empty_like(img);
            ws_bytes = msda_bwd_fused_workspace_bytes(B, I, H, D, Q, L, P, (int)proj.element_size(),
                                                      (int)img.element_size(), level_cells);
            ws = at::empty({ws_bytes}, img.options().dtype(at::kByte));
        }
        {
            const c10::DeviceGuard guard(img.device());
            check_rc(fns_for(img.scalar_type(), proj.scalar_type())
                         .bwd_fused(gout.data_ptr(), img.data_ptr(), shapes.data_ptr<int64_t>(), proj.data_ptr(),
                                    ref.data_ptr(), want_value ? g_img.data_ptr() : nullptr, g_proj.data_ptr(),
                                    g_ref_part.data_ptr(), B, I, H, D, Q, L, P, (int)ref_dim, padding_mode,
                                    align_corners ? 1 : 0, level_cells, ws.defined() ? ws.data_ptr() : nullptr, ws_bytes,
                                    current_stream(img)),
                     "msda_bwd_fused");
        }
        return once_differentiable(grads, {g_img, at::Tensor(), ctx->needs_input_grad(2) ? g_proj : at::Tensor(),
                                           ctx->needs_input_grad(3) ? g_ref_part.sum(2) : at::Tensor(), at::Tensor(),
                                           at::Tensor(), at::Tensor()});
    }
};

at::Tensor msda(const at::Tensor &img, const at::Tensor &shapes, const at::Tensor &pts, const at::Tensor &att,
                int64_t padding_mode, bool align_corners, int64_t level_cells)
{
    return MSDAFunction::apply(img, shapes, pts, att, padding_mode, align_corners, level_cells);
}

at::Tensor msda_fused(const at::Tensor &img, const at::Tensor &shapes, const at::Tensor &proj, const at::Tensor &ref,
                      int64_t padding_mode, bool align_corners, int64_t level_cells)
{
    return MSDAFusedFunction::apply(img, shapes, proj, ref, padding_mode, align_corners, level_cells);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "C++ autograd glue over libmsda_hip.so (same C ABI as the ctypes route)";
    m.def("msda", &msda, "multi-scale deformable attention (forward; differentiable)", pybind11::arg("img"),
          pybind11::arg("shapes"), pybind11::arg("sampling_points"), pybind11::arg("attention_weights"),
          pybind11::arg("padding_mode"), pybind11::arg("align_corners"), pybind11::arg("level_cells") = 0);
    m.def("msda_fused", &msda_fused, "module core with the softmax / sampling-point prologue fused in (differentiable)",
          pybind11::arg("img"), pybind11::arg("shapes"), pybind11::arg("proj"), pybind11::arg("reference_points"),
          pybind11::arg("padding_mode"), pybind11::arg("align_corners"), pybind11::arg("level_cells") = 0);
    m.def("fused_lp_limit", [](int64_t D, int64_t elem_size) { return msda_fused_lp_limit(D, (int)elem_size); });
    m.def("abi_version", []() { return msda_abi_version(); });
}
