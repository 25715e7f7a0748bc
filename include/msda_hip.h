/*
 * msda_hip.h — C ABI of libmsda_hip.so, the MI355X (gfx950) multi-scale deformable
 * attention kernels.  Plain pointers and sizes only: no torch / HIP types in the
 * signatures (the stream is passed as an opaque void* holding a hipStream_t).
 *
 * This is the drop-in boundary for the reference's L1 launcher pair (paths relative
 * to the reference checkout, rziga/msda-triton @ 2025-08-24):
 *
 *   msda_fwd_<dtype>  replaces  triton_multi_scale_deformable_attention_fwd
 *                                 src/msda_triton/kernels.py:351-379   (called from frontend.py:124-126)
 *   msda_bwd_<dtype>  replaces  triton_multi_scale_deformable_attention_bwd
 *                                 src/msda_triton/kernels.py:556-592   (called from frontend.py:139-141)
 *
 * Tensor conventions are the reference's (frontend.py:157-160, kernels.py:9-15); every
 * buffer is device memory, dense row-major, and is never modified unless it is an output:
 *
 *   value      [B, I, H, D]        packed pyramid, level l occupies pixels
 *                                  [start_l, start_l + h_l*w_l), row-major (y*w_l + x)
 *   shapes     [L, 2]  int64       (height, width) per level, on the device; the level
 *                                  start offsets are derived in-kernel (kernels.py:58-62)
 *   loc        [B, Q, H, L, P, 2]  (x, y) normalised to [0,1]
 *   attn       [B, Q, H, L, P]
 *   out        [B, Q, H, D]
 *   grad_out   [B, Q, H, D]
 *   grad_value / grad_loc / grad_attn   shaped like value / loc / attn
 *
 * <dtype> in {f32, f16, bf16, f64} is the storage type of every floating-point buffer
 * (one dtype per call, as in the reference).  Coordinates, bilinear weights and all
 * accumulation are float (double for f64); results are rounded once on store.
 *
 * padding_mode: MSDA_PADDING_BORDER | MSDA_PADDING_ZEROS   (frontend.py:150,161)
 * align_corners: 0 | 1                                      (frontend.py:151,162)
 *
 * Outputs are fully overwritten (no pre-zeroing required, no accumulation into them).
 * msda_bwd_*: grad_value may be NULL (that gradient is skipped), and grad_loc / grad_attn may
 * be NULL together (both skipped) — the autograd caller passes only what needs_input_grad asks.
 *
 * msda_fwd_fused_<dtype> is the forward of the reference's nn.Module core with its prologue fused in
 * (src/msda_triton/frontend.py:253-282): `proj` [B, Q, H, L, P, 3] is the raw query projection
 * (x offset, y offset, attention logit); `ref` [B, Q, ref_dim] the reference points, ref_dim 2 = (x, y),
 * 4 = (cx, cy, w, h).  The kernel takes the softmax over each (b, q, h)'s L*P logits and forms the
 * sampling points (ref + offset / img_shapes[l] — in the reference's (h, w) order — or
 * ref_xy + offset * ref_wh / (2 P)) in its prologue, so neither tensor is ever materialised.
 * Returns MSDA_ERR_UNSUPPORTED when L*P is too large for one pass (callers then use msda_fwd_<dtype>).
 *
 * msda_bwd_fused_<dtype> is its backward with the prologue's chain rule fused in (what autograd does for
 * frontend.py:253-282): from grad_out to grad_value [B, I, H, D] (may be NULL), grad_proj [B, Q, H, L, P, 3]
 * (offset gradients scaled back, softmax backward applied to the logits) and grad_ref_partial
 * [B, Q, H, ref_dim] — the per-head partial sums of the reference points' gradient, which the caller adds up
 * over H.  It needs msda_bwd_fused_workspace_bytes(...) of workspace when grad_value is wanted (the derived
 * sampling points / attention weights are parked there for the grad_value passes).  Same
 * MSDA_ERR_UNSUPPORTED rule as the fused forward (nothing is launched then).
 *
 * Backward workspace: grad_value is computed as a gather over a per-call inverted index (sample
 * records sorted by bilinear cell), which lives in caller-provided device memory so that the
 * library never allocates: pass `workspace` (256-byte aligned) of at least
 * msda_bwd_workspace_bytes(...) bytes; its contents are scratch and need no initialisation.
 * msda_bwd_workspace_bytes(...) is 0 for small problems (Grounding-DINO / Deformable-DETR decoder shapes): they take a
 * single-launch kernel that keeps its inverted index in LDS.  A larger problem without (enough) workspace is rejected
 * with MSDA_ERR_BAD_ARG when grad_value is wanted (grad_loc / grad_attn never need workspace).
 * Calls are asynchronous on `stream`; there is no host synchronisation and no allocation
 * inside, so a call sequence can be captured into a hipGraph.
 *
 * Return value: 0 on success; a negative MSDA_ERR_* for rejected arguments (nothing was
 * launched); a positive hipError_t if a launch failed.  msda_last_error() describes the
 * most recent failure on the calling thread.
 */
#ifndef MSDA_HIP_H
#define MSDA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSDA_ABI_VERSION 11

#if defined(__GNUC__)
#define MSDA_API __attribute__((visibility("default")))
#else
#define MSDA_API
#endif

#define MSDA_PADDING_BORDER 0
#define MSDA_PADDING_ZEROS 1

#define MSDA_MAX_LEVELS 32

#define MSDA_ERR_BAD_ARG (-1)      /* null pointer, negative size, unknown padding mode */
#define MSDA_ERR_TOO_MANY_LEVELS (-2) /* L > MSDA_MAX_LEVELS */
#define MSDA_ERR_TOO_LARGE (-3)    /* a per-batch-element extent does not fit the 32-bit plane offsets: I*H*D*sizeof, Q*H*D*sizeof or Q*H*L*P*2 >= 2^31, or I, Q >= 2^24 */
#define MSDA_ERR_MISALIGNED (-4)   /* a buffer is not aligned to its element size */
#define MSDA_ERR_UNSUPPORTED (-5)  /* valid arguments this entry point cannot serve (use the unfused call) */

/*
 * ONE forward and ONE backward entry point per storage type (since ABI 10; ABI 9 carried three generations of them).
 *
 * ABI 11 CHANGED THE ARGUMENT LISTS of these entry points and of msda_bwd_fused_workspace_bytes under their old names
 * (value_row_stride inserted in front of `workspace` / `stream`; a flags argument appended): a caller built against ABI
 * 10 still links and would pass its workspace pointer as a stride.  EVERY caller must check msda_abi_version() ==
 * MSDA_ABI_VERSION once after loading the library, before its first call (INTEGRATION.md shows it for ctypes and C).
 *
 * value_row_stride (ABI 11): bytes from one pixel's H rows of `value` to the next pixel's; 0 = dense (H * D * sizeof).
 * `value` is then addressed as value[b][i] at ((b * I + i) * value_row_stride) bytes, the head's row at h * D * sizeof
 * inside it; the last pixel needs only its H * D * sizeof bytes.  Must be a multiple of the element size (of 16 bytes for
 * the vector path: otherwise the scalar kernels run), >= H * D * sizeof, with I * value_row_stride < 2^31.  Why a caller
 * would pad: the gather kernels are bound by the vector L1, which picks one of its four tag RAMs from the low bits of a
 * row's 128-byte line index; rows exactly 1 024 bytes apart (H * D * sizeof = 1 KB: 8 heads x 32 channels x fp32) put
 * every row of one head on half of them, and that head's plane gathers ~20 % slower.  One extra 128-byte line per pixel
 * (1 152) makes every head cycle through all residues: forward -3.5 ... -8 %, sample gradients -10 % at B = 4 with 5 000
 * / 10 000 queries and at B = 8 with 900 (profiles/r06_row_stride_ab.txt; the whole training step -2.5 ... -5 %,
 * r06_padded_step_ab.txt; nothing at B = 4 with 1 000).  A caller that OWNS the layout — a module that writes the value projection itself — can do
 * that for free (GEMM output with a leading dimension of H * D + 32 floats); grad_value is always dense.  Results are
 * bit-identical to the dense layout.
 *
 * max_level_cells (backward): what the caller knows about the level sizes ON THE HOST.  `shapes` lives on the device, so
 * the library sizes the single-launch grad_value kernel's LDS cell table for the worst level `I` pixels can form,
 * (2 I + 2 L) cells — which rules that kernel out for the pyramids of real images (a 100 x 134 ... 13 x 17 pyramid:
 * 35.6 k cells by the bound, 13.6 k in its largest level) and sends decoder-sized calls on them to the sorted pipeline,
 * 1.4x slower there.  A caller that knows the level sizes (Hugging Face models carry `spatial_shapes_list`) passes the
 * bilinear cells of the largest level, max_l (h_l + 1) * (w_l + 1); 0 = unknown.  The workspace query takes the same
 * number.  A level larger than stated cannot be reported from the kernel: its grad_value rows come back NaN.
 */
#define MSDA_DECLARE(SUF)                                                                          \
    MSDA_API int msda_fwd_##SUF(const void *value, const int64_t *shapes, const void *loc,                  \
                       const void *attn, void *out, int64_t B, int64_t I, int64_t H, int64_t D,    \
                       int64_t Q, int64_t L, int64_t P, int padding_mode, int align_corners,       \
                       int64_t value_row_stride, void *stream);                                    \
    MSDA_API int msda_fwd_fused_##SUF(const void *value, const int64_t *shapes, const void *proj,           \
                       const void *ref, void *out, int64_t B, int64_t I, int64_t H, int64_t D,    \
                       int64_t Q, int64_t L, int64_t P, int ref_dim, int padding_mode,             \
                       int align_corners, int64_t value_row_stride, void *stream);                 \
    MSDA_API int msda_bwd_##SUF(const void *grad_out, const void *value, const int64_t *shapes,             \
                       const void *loc, const void *attn, void *grad_value, void *grad_loc,        \
                       void *grad_attn, int64_t B, int64_t I, int64_t H, int64_t D, int64_t Q,     \
                       int64_t L, int64_t P, int padding_mode, int align_corners,                  \
                       int64_t max_level_cells, int64_t value_row_stride, void *workspace,         \
                       int64_t workspace_bytes, void *stream);                                     \
    MSDA_API int msda_bwd_fused_##SUF(const void *grad_out, const void *value, const int64_t *shapes, \
                       const void *proj, const void *ref, void *grad_value, void *grad_proj,       \
                       void *grad_ref_partial, int64_t B, int64_t I, int64_t H, int64_t D,          \
                       int64_t Q, int64_t L, int64_t P, int ref_dim, int padding_mode,              \
                       int align_corners, int64_t max_level_cells, int64_t value_row_stride,        \
                       void *workspace, int64_t workspace_bytes, void *stream);

MSDA_DECLARE(f32)
MSDA_DECLARE(f16)
MSDA_DECLARE(bf16)
MSDA_DECLARE(f64)
/* Mixed storage: `value` and `grad_value` are bf16 / fp16, every other tensor (sampling points, attention weights,
 * projection, reference points, out, grad_out and the other gradients) is fp32.  For modules that keep the value
 * pyramid in 16 bits (the value projection written straight in the kernel's layout and dtype) without
 * giving up fp32 sampling coordinates; arithmetic is fp32 as everywhere.  Same signatures; workspace sizes are those
 * of elem_size 4, value_elem_size 2. */
MSDA_DECLARE(f32_vbf16)
MSDA_DECLARE(f32_vf16)
#undef MSDA_DECLARE
/* Module storage (round 5; fused entry points only): `value`, `proj`, `out` and their gradients (grad_value, grad_proj,
 * grad_out) are bf16 / fp16, the reference points `ref` and `grad_ref_partial` fp32, all arithmetic fp32 — what the
 * reference's nn.Module computes under torch.autocast (its core casts every input to fp32, frontend.py:111) without the
 * fp32 copies of the projection and of the result and their gradients: a 16-bit projection holds exactly what autocast's
 * GEMM produced, while reference points in 16 bits would put the samples up to a quarter pixel off.  Same signatures as
 * msda_fwd_fused_<dtype> / msda_bwd_fused_<dtype>; workspace: msda_bwd_fused_workspace_bytes(..., elem_size 4,
 * value_elem_size 2, ...). */
#define MSDA_DECLARE_FUSED_STORAGE(SUF)                                                                \
    MSDA_API int msda_fwd_fused_##SUF(const void *value, const int64_t *shapes, const void *proj,      \
                       const void *ref, void *out, int64_t B, int64_t I, int64_t H, int64_t D,        \
                       int64_t Q, int64_t L, int64_t P, int ref_dim, int padding_mode,                 \
                       int align_corners, int64_t value_row_stride, void *stream);                     \
    MSDA_API int msda_bwd_fused_##SUF(const void *grad_out, const void *value, const int64_t *shapes, \
                       const void *proj, const void *ref, void *grad_value, void *grad_proj,           \
                       void *grad_ref_partial, int64_t B, int64_t I, int64_t H, int64_t D,              \
                       int64_t Q, int64_t L, int64_t P, int ref_dim, int padding_mode,                  \
                       int align_corners, int64_t max_level_cells, int64_t value_row_stride,            \
                       void *workspace, int64_t workspace_bytes, void *stream);
MSDA_DECLARE_FUSED_STORAGE(f32_sbf16)
MSDA_DECLARE_FUSED_STORAGE(f32_sf16)
#undef MSDA_DECLARE_FUSED_STORAGE

/*
 * Bytes of device workspace msda_bwd_<dtype> wants for these sizes.  elem_size: of everything but `value`;
 * value_elem_size: of `value` / `grad_value` (0: the same — 2 next to elem_size 4 for the mixed-storage entry points);
 * max_level_cells: as for the call (0: unknown).
 * flags: MSDA_WS_RECORDS_IN_GRADS — the call will ALSO ask for grad_loc / grad_attn (all three gradient buffers
 * non-NULL and 16-byte aligned): the sorted sample records are dead once the gather has run, grad_loc / grad_attn are
 * written last and grad_value only by the finish kernel behind the gather, so the records of as many (batch, head)
 * planes as fit are kept in those buffers and the workspace shrinks (c2 @ 10k: 180 -> 98 MB).  A call with such a
 * workspace but without grad_loc / grad_attn (or misaligned buffers) is rejected (MSDA_ERR_BAD_ARG); a larger workspace
 * is fine.
 */
#define MSDA_WS_RECORDS_IN_GRADS 1
/*
 * MSDA_WS_PASSES(n), n = 2, 4, 8 ... (ABI 11): the size for n PASSES OVER THE BATCH.  The sorted pipeline's tables and partial
 * rows are per (batch, head) plane of the call; msda_bwd_<dtype> given a workspace that does not hold the whole batch but
 * does hold ceil(B / 2), ceil(B / 4) ... batch elements runs the pipeline once per such group in the same memory
 * (the workspace it is GIVEN decides: the fewest passes that fit).  c2 @ 10k fp32 with MSDA_WS_RECORDS_IN_GRADS:
 * 107 MB in one pass, 59 MB in two, 35 MB in four — the step's peak memory by the reference's recipe
 * (scripts/benchmark.py:158-172) 277 -> 229 -> 205 MB — for +16 % / +48 % of the step (0.307 -> 0.358 / 0.456 ms: the
 * pipeline's five kernels run at 68 % / 50 % of their rate on half / a quarter of the planes), which is why one pass is
 * the default.  grad_loc / grad_attn do not depend on the passes; grad_value is bitwise reproducible for a given number
 * of passes, and between two numbers of passes equal up to the rounding of the last bit (a group of fewer planes is cut
 * into more query slices, so a cell's records can sit in another order).
 */
#define MSDA_WS_PASSES(n) (((n) & 0xff) << 8)
MSDA_API int64_t msda_bwd_workspace_bytes(int64_t B, int64_t I, int64_t H, int64_t D, int64_t Q, int64_t L,
                                          int64_t P, int elem_size, int value_elem_size, int64_t max_level_cells,
                                          int flags);
/* ... and msda_bwd_fused_<dtype> (grad_value != NULL). */
MSDA_API int64_t msda_bwd_fused_workspace_bytes(int64_t B, int64_t I, int64_t H, int64_t D, int64_t Q, int64_t L,
                                                int64_t P, int elem_size, int value_elem_size,
                                                int64_t max_level_cells, int flags /* 0 | MSDA_WS_PASSES(n) */);

/*
 * 1 when msda_bwd_<dtype> can produce grad_value for these sizes, 0 when it would return MSDA_ERR_UNSUPPORTED (a plane
 * of 2^22 pixels or more, or a head dimension beyond the 32-bit slot offsets of the sorted pipeline, on a problem that
 * is also too large for the single-launch kernel).  Host arithmetic only: a caller asks at FORWARD time when the value
 * pyramid requires a gradient, instead of learning it from the backward in the middle of a training step.
 */
MSDA_API int msda_bwd_supported(int64_t B, int64_t I, int64_t H, int64_t D, int64_t Q, int64_t L, int64_t P,
                                int elem_size);

/* Largest L*P the fused entry points accept for head dimension D and this element size (beyond it they return
 * MSDA_ERR_UNSUPPORTED and the caller composes the prologue around msda_fwd_/msda_bwd_<dtype>). */
MSDA_API int64_t msda_fused_lp_limit(int64_t D, int elem_size);

/* ABI version of the loaded library (== MSDA_ABI_VERSION it was built with). */
MSDA_API int msda_abi_version(void);

/* Human-readable description of the last non-zero return on this thread ("" if none). */
MSDA_API const char *msda_last_error(void);

/*
 * Kernel-variant switches for A/B measurements (bench.py, profiling, the option-variant test runs).  Not needed for
 * normal use: the defaults are the fastest measured variants.  Unknown keys return MSDA_ERR_BAD_ARG.  Keys:
 *   "xcd_map"    1 (default): blockIdx -> (batch, head) mapping keeps each (b,h) plane on one XCD's L2
 *                0: plain linear mapping   2: as 1 with the planes of an XCD rotated through the heads
 *   "linear_slots" 320 (default): launches over 128-byte rows 1 KB apart (H * D * sizeof = 1024: the layout with a slow
 *                   head) with at least this many workgroups per (batch, head) plane use the plain
 *                   linear block order instead of the XCD-aware one of "xcd_map" 1 (they walk the planes one after another
 *                   anyway, and all XCDs then share every plane's work: c5 step 11.2 -> 10.4 ms)
 *   "lds_levels" 1 (default): problems large enough to amortise it run the forward / sample-gradient kernels with the
 *                   coarsest pyramid levels served from LDS (1024-thread workgroups, one per CU: c2 @ 10k forward
 *                   99 -> 75 us) — fp32 arithmetic over fp32 or 16-bit rows, and the 16-bit operators' forward over
 *                   64-byte rows; bit-identical results;  0: never;  2: wherever the variant exists (tests)
 *   "lds_planes" 0 (default): the LDS-level forward kernels (and the module's fused backward) serve TWO planes — the
 *                   neighbouring heads (b, 2k), (b, 2k + 1) — per workgroup where both planes' levels fit and the
 *                   pairs' workgroups come out no heavier than single planes' would; its waves take slices of whichever plane has more left
 *                   (c2 @ 10k forward 69.5 -> 64.3 us; bit-identical results);  1: never;  2: whenever H is even (tests)
 *   "unit_fwd"   1 (default): forwards of at most 12 288 (b, q, h) units take the one-wave-per-unit kernel (decoder
 *                   calls: cold-cache forward at Q = 100 12.8 -> 9.3 us);  0: never;  2: wherever it exists (tests)
 *   "unit_waves" 1 (default): that kernel serves one unit per wave;  2: two (one per half wave) where its rows allow it:
 *                   ~1 us faster at Q = 200-300 with the pyramid in HBM, 0.3-0.6 us slower with it cached
 *   "touch"      1 (default): a forward of few queries that will read most of the pyramid anyway (4 Q L P >= 2 I, at most
 *                   65 536 (b, q, h) units, one round of workgroups with at most 768 rows of the plane each, not the
 *                   one-wave-per-unit kernel) has its workgroups request one dword of every row of their plane behind
 *                   their first sampling points, so that the rows stream into the XCD's L2 while the points are on their
 *                   way: with the pyramid in HBM (nothing in L2 or the Infinity Cache) forward at B = 4, H = 8, Q = 900
 *                   26.6 -> 22.7 us, Q = 2000 35.8 -> 32.9; nothing measurable when the rows are cached (larger shares
 *                   per workgroup do cost then: hence the 768); the values are not used, results are bit-identical;
 *                   0: never;  2: every forward through msda_fwd_kernel (tests)
 *   "value_path" 0 (default): grad_value by the single-launch LDS kernel when a (plane, level) fits one workgroup
 *                   (small problems; no workspace needed), else by the sorted gather in the caller's workspace
 *                2: the sorted gather always   3: the single-launch kernel whenever it fits
 *   "small_ns"   0 (default): workgroups per (plane, level) of the single-launch kernel chosen by the plane count; 1..16
 *   "q_round"    0 (default): queries per round of the sorted pipeline chosen by size; n: that many (tests)
 *   "overlap"    -1 (default) / 0: the backward's kernels run one after the other on the caller's stream;  1: grad_loc /
 *                   grad_attn run on a forked side stream next to grad_value (fork / join with events inside the call,
 *                   graph-capturable; costs ~14 us of host time and ~19 us of latency; does not pay since round 4)
 *   "place_path" 0 (default) / 2: the level-major place pass;  1: the plane-major one;  3: the level-major pass with
 *                   256-thread workgroups everywhere (measurement only)
 *   "strict"     0 (default);  1: a backward whose grad_value would NOT be bitwise reproducible is refused with
 *                   MSDA_ERR_UNSUPPORTED instead of run — see "Reproducibility" below
 *   "records_in_grads" 1 (default): msda_bwd_<dtype> with all three gradients keeps sorted records in the grad_loc /
 *                   grad_attn buffers until the sample-gradient kernel overwrites them;  0: never
 *   "profile"    0 (default);  1: event pairs around every kernel launch, read with msda_profile_read (measurement only)
 *   "level_cells" 0 (default): unknown;  n: process-wide form of the max_level_cells argument (an argument wins)
 *   "ws_passes"  1 (default): passes over the batch the workspace queries size for when their flags carry no MSDA_WS_PASSES(n)
 *                   (n: smaller workspace, the sorted pipeline runs once per group of ceil(B / n) batch elements — callers that
 *                   simply allocate what the query returns follow it without a change; see MSDA_WS_PASSES for the price)
 * Builds with -DMSDA_DEV (development only; the shipped library rejects these keys) add the experiment knobs
 * "cell_slices", "gather_win", "wg_target", "lds_budget", "lds_stagger", "lds_over" and the ablation / phase-clock mask "debug":
 * see msda_triton_amd/csrc/msda_launch.hpp, msda_value_sorted.hpp and tools/phase_clock.py.
 *
 * Reproducibility.  out, grad_loc and grad_attn are bitwise reproducible by construction (no atomics, fixed summation
 * orders).  grad_value is bitwise reproducible for P <= 1024 points per level: the gather sums a cell's records in list
 * order, and the place pass of the sorted pipeline and the single-launch kernel both let their waves take the
 * list-cursor atomics in TURNS (acquire / release on the turn word), so the list order is the same in every run.  This
 * rests on OBSERVED, not architected, behaviour of gfx950: lanes of ONE wave instruction that hit the same LDS word
 * with ds_add_rtn_u32 are served in a fixed order.  It holds in every run of the test suite (three dedicated tests,
 * all storage types, option variants) and of the fuzzers, and it is what the tests pin; a future part may differ.
 * P > 1024 points per level (and "place_path" 1) take the plane-major place pass, whose order follows free-running
 * atomics: grad_value may then differ in the last bit from run to run — or, with "strict" 1, the call is refused.
 */
MSDA_API int msda_set_option(const char *key, int value);
/* Measurement only.  With msda_set_option("profile", 1) every kernel the library launches is bracketed by a HIP event
 * pair on its stream; msda_profile_read waits for the events recorded (by any thread) since the last read and
 * writes one line per kernel — "name launches total_microseconds\n" — into buf (NUL-terminated, at most cap bytes);
 * returns the characters written.  The read consumes the records.  bench.py's per-kernel figures come from here. */
MSDA_API int msda_profile_read(char *buf, int cap);
MSDA_API int msda_get_option(const char *key);
/* Measurement only (ABI 11): which variants the most recent launches took (process-wide; bench.py reads it to say how many
 * of the forward's rows came from LDS).  Keys: "fwd_variant" 0 = 256-thread kernel, 1 = LDS-served coarse levels, 2 = one
 * wave per unit;  "fwd_lds_level_bytes" LDS bytes per plane set aside for level rows (the kernel keeps the longest suffix
 * of the level list that fits);  "fwd_lds_planes";  "fwd_workgroups";  "sample_variant", "sample_lds_level_bytes" the same
 * for the sample-gradient kernel;  "value_path" 1 = single-launch kernel, 2 = sorted pipeline;  "value_passes" its passes
 * over the batch.  Unknown key: MSDA_ERR_BAD_ARG. */
MSDA_API int msda_last_launch_info(const char *key);

#ifdef __cplusplus
}
#endif
#endif /* MSDA_HIP_H */
