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

#include <algorithm>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/msda_hip.h"

namespace {

// (ABI 11: value_row_stride in front of the stream / the workspace)
using FwdFn = int (*)(const void *, const int64_t *, const void *, const void *, void *, int64_t, int64_t, int64_t,
                      int64_t, int64_t, int64_t, int64_t, int, int, int64_t, void *);
// (the level-size bound travels as an argument: include/msda_hip.h, max_level_cells)
using BwdFn = int (*)(const void *, const void *, const int64_t *, const void *, const void *, void *, void *, void *,
                      int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int, int64_t, int64_t, void *,
                      int64_t, void *);

using FwdFusedFn = int (*)(const void *, const int64_t *, const void *, const void *, void *, int64_t, int64_t, int64_t,
                           int64_t, int64_t, int64_t, int64_t, int, int, int, int64_t, void *);
using BwdFusedFn = int (*)(const void *, const void *, const int64_t *, const void *, const void *, void *, void *, void *,
                           int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int, int, int64_t, int64_t,
                           void *, int64_t, void *);

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

// The value pyramid as the kernels can address it, and its value_row_stride in bytes (0: dense).  A [B, I, H, D] view whose
// pixels sit a constant number of bytes apart with a pixel's H * D channels contiguous (functional.padded_value_rows) is read
// in place; any other layout is copied dense (functional._value_rows is the same rule).
std::pair<at::Tensor, int64_t> value_rows(const at::Tensor &img)
{
    if (img.is_contiguous()) return {img, 0};
    if (img.dim() == 4) {
        const int64_t B = img.size(0), I = img.size(1), H = img.size(2), D = img.size(3), es = (int64_t)img.element_size();
        const auto st = img.strides();
        if (D > 0 && H > 0 && I > 0 && st[3] == 1 && st[2] == D && st[1] >= H * D && (B == 1 || st[0] == I * st[1]) &&
            (st[1] * es) % 16 == 0 && I * st[1] * es < ((int64_t)1 << 31))
            return {img, st[1] * es};
    }
    return {img.contiguous(), 0};
}
// bytes from one batch element of `img` (as returned by value_rows) to the next
int64_t value_batch_bytes(const at::Tensor &img, int64_t vrow)
{
    return vrow > 0 ? img.size(1) * vrow : img.size(1) * img.size(2) * img.size(3) * (int64_t)img.element_size();
}

class MSDAFunction : public torch::autograd::Function<MSDAFunction> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &img_, const at::Tensor &shapes_,
                              const at::Tensor &pts_, const at::Tensor &att_, int64_t padding_mode, bool align_corners,
                              int64_t level_cells)
    {
        const auto [img, vrow] = value_rows(img_);
        const at::Tensor pts = pts_.contiguous(), att = att_.contiguous();
        const at::Tensor shapes = shapes_.to(at::kLong).contiguous();  // stays on the device
        const int64_t B = img.size(0), I = img.size(1), H = img.size(2), D = img.size(3);
        const int64_t Q = pts.size(1), L = pts.size(3), P = pts.size(4);
        at::Tensor out = at::empty({B, Q, H, D}, pts.options());
        const c10::DeviceGuard guard(img.device());
        check_rc(fns_for(img.scalar_type(), pts.scalar_type())
                     .fwd(img.data_ptr(), shapes.data_ptr<int64_t>(), pts.data_ptr(), att.data_ptr(), out.data_ptr(), B, I,
                          H, D, Q, L, P, (int)padding_mode, align_corners ? 1 : 0, vrow, current_stream(img)),
                 "msda_fwd");
        ctx->save_for_backward({img, shapes, pts, att});
        ctx->saved_data["padding_mode"] = padding_mode;
        ctx->saved_data["align_corners"] = align_corners;
        ctx->saved_data["level_cells"] = level_cells;
        ctx->saved_data["vrow"] = vrow;
        return out;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx,
                                                   torch::autograd::variable_list grads)
    {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor &img = saved[0], &shapes = saved[1], &pts = saved[2], &att = saved[3];
        const int64_t vrow = ctx->saved_data["vrow"].toInt();
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
            g_img = at::empty(img.sizes(), img.options());  // (dense, whatever the pyramid's row stride)
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
                              align_corners ? 1 : 0, level_cells, vrow, ws.defined() ? ws.data_ptr() : nullptr, ws_bytes,
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
        const auto [img, vrow] = value_rows(img_);
        const at::Tensor proj = proj_.contiguous(), ref = ref_.contiguous();
        const at::Tensor shapes = shapes_.to(at::kLong).contiguous();
        const int64_t B = img.size(0), I = img.size(1), H = img.size(2), D = img.size(3);
        const int64_t Q = proj.size(1), L = proj.size(3), P = proj.size(4);
        at::Tensor out = at::empty({B, Q, H, D}, proj.options());
        const c10::DeviceGuard guard(img.device());
        check_rc(fns_for(img.scalar_type(), proj.scalar_type())
                     .fwd_fused(img.data_ptr(), shapes.data_ptr<int64_t>(), proj.data_ptr(), ref.data_ptr(), out.data_ptr(),
                                B, I, H, D, Q, L, P, (int)ref.size(-1), (int)padding_mode, align_corners ? 1 : 0, vrow,
                                current_stream(img)),
                 "msda_fwd_fused");
        ctx->save_for_backward({img, shapes, proj, ref});
        ctx->saved_data["padding_mode"] = padding_mode;
        ctx->saved_data["align_corners"] = align_corners;
        ctx->saved_data["level_cells"] = level_cells;
        ctx->saved_data["vrow"] = vrow;
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
        const int64_t vrow = ctx->saved_data["vrow"].toInt();
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
            g_img = at::empty(img.sizes(), img.options());
            ws_bytes = msda_bwd_fused_workspace_bytes(B, I, H, D, Q, L, P, (int)proj.element_size(),
                                                      (int)img.element_size(), level_cells, 0);
            ws = at::empty({ws_bytes}, img.options().dtype(at::kByte));
        }
        {
            const c10::DeviceGuard guard(img.device());
            check_rc(fns_for(img.scalar_type(), proj.scalar_type())
                         .bwd_fused(gout.data_ptr(), img.data_ptr(), shapes.data_ptr<int64_t>(), proj.data_ptr(),
                                    ref.data_ptr(), want_value ? g_img.data_ptr() : nullptr, g_proj.data_ptr(),
                                    g_ref_part.data_ptr(), B, I, H, D, Q, L, P, (int)ref_dim, padding_mode,
                                    align_corners ? 1 : 0, level_cells, vrow, ws.defined() ? ws.data_ptr() : nullptr,
                                    ws_bytes, current_stream(img)),
                     "msda_bwd_fused");
        }
        return once_differentiable(grads, {g_img, at::Tensor(), ctx->needs_input_grad(2) ? g_proj : at::Tensor(),
                                           ctx->needs_input_grad(3) ? g_ref_part.sum(2) : at::Tensor(), at::Tensor(),
                                           at::Tensor(), at::Tensor()});
    }
};

// ------------------------------------------------------------------------------------------------------------------
// Row ranges of the flattened (b, q) row space — the launches of the row-sharded operator
// (msda_triton_amd/distributed.py; SURVEY 8e, reference independence argument kernels.py:18-21).  A rank's rows
// [r0, r1) meet a batch element in at most one piece: runs of whole batch elements are ONE launch (B = nb), a partial
// batch element its own (B = 1, value = img[b]).  Same C ABI, pointer arithmetic only.
// ------------------------------------------------------------------------------------------------------------------
struct RowDims {
    int64_t B, I, H, D, Q, L, P;
    int64_t es, vs;  // element sizes: sampling inputs / result, value pyramid
};

RowDims row_dims(const at::Tensor &img, const at::Tensor &pts_rows, const at::Tensor &att_rows, int64_t Q)
{
    // (pointers of another device — or of the host — must never reach the kernels)
    TORCH_CHECK_VALUE(img.is_cuda() && pts_rows.device() == img.device() && att_rows.device() == img.device(),
                      "expected all tensors on one gpu, got ", img.device(), ", ", pts_rows.device(), ", ", att_rows.device());
    TORCH_CHECK_VALUE(img.dim() == 4 && pts_rows.dim() == 5 && att_rows.dim() == 4 && pts_rows.size(4) == 2,
                      "expected img [B,I,H,D], sampling_points [rows,H,L,P,2], attention_weights [rows,H,L,P]");
    TORCH_CHECK_VALUE(pts_rows.size(1) == img.size(2) && att_rows.sizes() == pts_rows.sizes().slice(0, 4),
                      "inconsistent shapes between img, sampling_points and attention_weights rows");
    TORCH_CHECK_VALUE(pts_rows.scalar_type() == att_rows.scalar_type(), "sampling_points / attention_weights dtypes differ");
    TORCH_CHECK_VALUE(Q >= 0, "num_queries must be non-negative");
    return {img.size(0), img.size(1), img.size(2), img.size(3), Q, pts_rows.size(2), pts_rows.size(3),
            (int64_t)pts_rows.element_size(), (int64_t)img.element_size()};
}

// fn(b, nb, rows_before, n): piece of `n` rows starting `rows_before` rows behind `row0`, covering batch elements
// [b, b + nb) (nb > 1: whole batch elements)
template <typename F> void for_pieces(int64_t Q, int64_t row0, int64_t row1, F fn)
{
    int64_t r = row0;
    while (r < row1) {
        const int64_t b = r / Q, q0 = r % Q;
        if (q0 == 0 && row1 - r >= Q) {
            const int64_t nb = (row1 - r) / Q;
            fn(b, nb, r - row0, nb * Q);
            r += nb * Q;
        } else {
            const int64_t n = std::min(Q - q0, row1 - r);
            fn(b, (int64_t)1, r - row0, n);
            r += n;
        }
    }
}

// forward of rows [row0, row1) into full[row0:row1]; pts_rows / att_rows hold the rows from `in_row0` on
void rows_forward(const at::Tensor &img_, const at::Tensor &shapes, const at::Tensor &pts_rows, const at::Tensor &att_rows,
                  int64_t in_row0, const at::Tensor &full, int64_t row0, int64_t row1, int64_t Q, int64_t padding_mode,
                  bool align_corners)
{
    const RowDims d = row_dims(img_, pts_rows, att_rows, Q);
    const auto [img, vrow] = value_rows(img_);
    TORCH_CHECK_VALUE(pts_rows.is_contiguous() && att_rows.is_contiguous() && full.is_contiguous(),
                      "rows_forward takes contiguous tensors");
    TORCH_CHECK_VALUE(shapes.scalar_type() == at::kLong && shapes.is_contiguous() && shapes.device() == img.device() &&
                          full.device() == img.device(), "img_shapes: contiguous int64 on the tensors' device; result on it too");
    TORCH_CHECK_VALUE(0 <= in_row0 && in_row0 <= row0 && row0 <= row1 && row1 <= d.B * d.Q &&
                          row1 - in_row0 <= pts_rows.size(0) && full.numel() == d.B * d.Q * d.H * d.D &&
                          full.scalar_type() == pts_rows.scalar_type(),
                      "rows_forward: row range / buffer sizes do not match");
    if (row1 == row0) return;
    const Fns fns = fns_for(img.scalar_type(), pts_rows.scalar_type());
    const c10::DeviceGuard guard(img.device());
    void *stream = current_stream(img);
    const char *v = static_cast<const char *>(img.data_ptr());
    const char *p = static_cast<const char *>(pts_rows.data_ptr()), *a = static_cast<const char *>(att_rows.data_ptr());
    char *o = static_cast<char *>(full.data_ptr());
    const int64_t unit = d.H * d.L * d.P, vbatch = value_batch_bytes(img, vrow);
    for_pieces(d.Q, row0, row1, [&](int64_t b, int64_t nb, int64_t before, int64_t n) {
        const int64_t in_at = row0 + before - in_row0, out_at = row0 + before;
        check_rc(fns.fwd(v + b * vbatch, shapes.data_ptr<int64_t>(), p + in_at * unit * 2 * d.es,
                         a + in_at * unit * d.es, o + out_at * d.H * d.D * d.es, nb, d.I, d.H, d.D, n / nb, d.L, d.P,
                         (int)padding_mode, align_corners ? 1 : 0, vrow, stream),
                 "msda_fwd (row range)");
    });
}

// backward of rows [r0, r1): grad_rows / pts_rows / att_rows hold exactly those rows.  Returns (grad_img [B,I,H,D] with
// the batch elements the rows do not touch zeroed, grad_pts_rows, grad_att_rows); undefined tensors for what is not needed.
std::tuple<at::Tensor, at::Tensor, at::Tensor> rows_backward(const at::Tensor &grad_rows_, const at::Tensor &img_,
                                                             const at::Tensor &shapes, const at::Tensor &pts_rows,
                                                             const at::Tensor &att_rows, int64_t r0, int64_t r1, int64_t Q,
                                                             int64_t padding_mode, bool align_corners, bool need_img,
                                                             bool need_pts, bool need_att, int64_t level_cells)
{
    const RowDims d = row_dims(img_, pts_rows, att_rows, Q);
    const auto [img, vrow] = value_rows(img_);
    TORCH_CHECK_VALUE(pts_rows.is_contiguous() && att_rows.is_contiguous(), "rows_backward takes contiguous tensors");
    TORCH_CHECK_VALUE(shapes.scalar_type() == at::kLong && shapes.is_contiguous() && shapes.device() == img.device() &&
                          grad_rows_.device() == img.device(), "img_shapes: contiguous int64 on the tensors' device; grad_out on it too");
    TORCH_CHECK_VALUE(0 <= r0 && r0 <= r1 && r1 <= d.B * d.Q && pts_rows.size(0) == r1 - r0 &&
                          grad_rows_.numel() == (r1 - r0) * d.H * d.D,
                      "rows_backward: row range / buffer sizes do not match");
    at::Tensor grad_rows = grad_rows_.contiguous();
    if (grad_rows.scalar_type() != pts_rows.scalar_type()) grad_rows = grad_rows.to(pts_rows.scalar_type());
    const bool want_sample = need_pts || need_att;
    at::Tensor g_img, g_pts, g_att;
    if (need_img) g_img = at::empty(img.sizes(), img.options());  // (dense)
    if (want_sample) {
        g_pts = at::empty_like(pts_rows);
        g_att = at::empty_like(att_rows);
    }
    std::vector<bool> touched((size_t)d.B, false);
    if ((need_img || want_sample) && r1 > r0) {
        const Fns fns = fns_for(img.scalar_type(), pts_rows.scalar_type());
        const c10::DeviceGuard guard(img.device());
        void *stream = current_stream(img);
        const char *v = static_cast<const char *>(img.data_ptr());
        const char *p = static_cast<const char *>(pts_rows.data_ptr()), *a = static_cast<const char *>(att_rows.data_ptr());
        const char *go = static_cast<const char *>(grad_rows.data_ptr());
        char *gv = need_img ? static_cast<char *>(g_img.data_ptr()) : nullptr;
        char *gp = want_sample ? static_cast<char *>(g_pts.data_ptr()) : nullptr;
        char *ga = want_sample ? static_cast<char *>(g_att.data_ptr()) : nullptr;
        const int64_t unit = d.H * d.L * d.P, plane = d.I * d.H * d.D, vbatch = value_batch_bytes(img, vrow);
        const bool overlap_forced = msda_get_option("overlap") == 1;
        for_pieces(d.Q, r0, r1, [&](int64_t b, int64_t nb, int64_t before, int64_t n) {
            for (int64_t k = 0; k < nb; ++k) touched[(size_t)(b + k)] = true;
            void *pv = need_img ? gv + b * plane * d.vs : nullptr;
            void *pp = want_sample ? gp + before * unit * 2 * d.es : nullptr;
            void *pa = want_sample ? ga + before * unit * d.es : nullptr;
            at::Tensor ws;
            int64_t ws_bytes = 0;
            if (need_img) {
                const bool in_grads = want_sample && reinterpret_cast<uintptr_t>(pp) % 16 == 0 &&
                                      reinterpret_cast<uintptr_t>(pa) % 16 == 0 && reinterpret_cast<uintptr_t>(pv) % 16 == 0 &&
                                      !overlap_forced;
                ws_bytes = msda_bwd_workspace_bytes(nb, d.I, d.H, d.D, n / nb, d.L, d.P, (int)d.es, (int)d.vs, level_cells,
                                                    in_grads ? MSDA_WS_RECORDS_IN_GRADS : 0);
                ws = at::empty({ws_bytes}, img.options().dtype(at::kByte));
            }
            check_rc(fns.bwd(go + before * d.H * d.D * d.es, v + b * vbatch, shapes.data_ptr<int64_t>(),
                             p + before * unit * 2 * d.es, a + before * unit * d.es, pv, pp, pa, nb, d.I, d.H, d.D, n / nb,
                             d.L, d.P, (int)padding_mode, align_corners ? 1 : 0, level_cells, vrow,
                             ws.defined() ? ws.data_ptr() : nullptr, ws_bytes, stream),
                     "msda_bwd (row range)");
        });
    }
    if (need_img)
        for (int64_t b = 0; b < d.B; ++b)
            if (!touched[(size_t)b]) g_img.select(0, b).zero_();
    return {g_img, need_pts ? g_pts : at::Tensor(), need_att ? g_att : at::Tensor()};
}

int64_t chunk_begin(int64_t n, int64_t chunks, int64_t k)
{
    const int64_t cs = chunks > 0 ? (n + chunks - 1) / chunks : n;
    return std::min(n, k * cs);
}

// The row-sharded operator WITHOUT its exchange as one autograd node: a one-rank job, or one rank of a larger job
// played on one GPU (distributed.py `compute_only_as`).  Rows [r0, r1) are computed in `chunks` pieces straight into
// the full [B, Q, H, D] result; the other rows are left unwritten.
class MSDARowsFunction : public torch::autograd::Function<MSDARowsFunction> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &img_, const at::Tensor &shapes_,
                              const at::Tensor &pts_, const at::Tensor &att_, int64_t padding_mode, bool align_corners,
                              int64_t Q, int64_t r0, int64_t r1, int64_t chunks, int64_t level_cells)
    {
        const at::Tensor img = value_rows(img_).first, pts = pts_.contiguous(), att = att_.contiguous();
        const at::Tensor shapes = shapes_.to(at::kLong).contiguous();
        const int64_t B = img.size(0), H = img.size(2), D = img.size(3);
        at::Tensor full = at::empty({B, Q, H, D}, pts.options());
        chunks = std::max<int64_t>(1, chunks);
        for (int64_t k = 0; k < chunks; ++k) {
            const int64_t c0 = chunk_begin(r1 - r0, chunks, k), c1 = chunk_begin(r1 - r0, chunks, k + 1);
            rows_forward(img, shapes, pts, att, r0, full, r0 + c0, r0 + c1, Q, padding_mode, align_corners);
        }
        ctx->save_for_backward({img, shapes, pts, att});
        ctx->saved_data["padding_mode"] = padding_mode;
        ctx->saved_data["align_corners"] = align_corners;
        ctx->saved_data["level_cells"] = level_cells;
        ctx->saved_data["Q"] = Q;
        ctx->saved_data["r0"] = r0;
        ctx->saved_data["r1"] = r1;
        return full;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx,
                                                   torch::autograd::variable_list grads)
    {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor &img = saved[0], &shapes = saved[1], &pts = saved[2], &att = saved[3];
        const int64_t Q = ctx->saved_data["Q"].toInt(), r0 = ctx->saved_data["r0"].toInt(), r1 = ctx->saved_data["r1"].toInt();
        const int64_t H = img.size(2), D = img.size(3);
        const at::Tensor mine = grads[0].reshape({img.size(0) * Q, H, D}).slice(0, r0, r1);
        auto [g_img, g_pts, g_att] =
            rows_backward(mine, img, shapes, pts, att, r0, r1, Q, ctx->saved_data["padding_mode"].toInt(),
                          ctx->saved_data["align_corners"].toBool(), ctx->needs_input_grad(0), ctx->needs_input_grad(2),
                          ctx->needs_input_grad(3), ctx->saved_data["level_cells"].toInt());
        return once_differentiable(grads, {g_img, at::Tensor(), g_pts, g_att, at::Tensor(), at::Tensor(), at::Tensor(),
                                           at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()});
    }
};

at::Tensor msda_rows(const at::Tensor &img, const at::Tensor &shapes, const at::Tensor &pts_rows, const at::Tensor &att_rows,
                     int64_t padding_mode, bool align_corners, int64_t Q, int64_t r0, int64_t r1, int64_t chunks,
                     int64_t level_cells)
{
    const RowDims d = row_dims(img, pts_rows, att_rows, Q);
    TORCH_CHECK_VALUE(0 <= r0 && r0 <= r1 && r1 <= d.B * d.Q && pts_rows.size(0) == r1 - r0,
                      "msda_rows: rows [", r0, ", ", r1, ") against ", pts_rows.size(0), " input rows");
    return MSDARowsFunction::apply(img, shapes, pts_rows, att_rows, padding_mode, align_corners, Q, r0, r1, chunks,
                                   level_cells);
}

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
    m.def("msda_rows", &msda_rows,
          "rows [r0, r1) of the flattened (b, q) row space computed in `chunks` pieces into a full [B,Q,H,D] result "
          "(differentiable; the row-sharded operator without its exchange)",
          pybind11::arg("img"), pybind11::arg("shapes"), pybind11::arg("sampling_points_rows"),
          pybind11::arg("attention_weights_rows"), pybind11::arg("padding_mode"), pybind11::arg("align_corners"),
          pybind11::arg("num_queries"), pybind11::arg("r0"), pybind11::arg("r1"), pybind11::arg("chunks") = 1,
          pybind11::arg("level_cells") = 0);
    m.def("rows_forward", &rows_forward, "launches of a row range into its place in the full result (no autograd)");
    m.def("rows_backward", &rows_backward, "backward launches of a row range (no autograd)");
    m.def("fused_lp_limit", [](int64_t D, int64_t elem_size) { return msda_fused_lp_limit(D, (int)elem_size); });
    m.def("abi_version", []() { return msda_abi_version(); });
}
