/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's multi-scale deformable attention
 * algorithm, instantiated once per REAL type by msda_oracle.c.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.
 *
 * Reference semantics restated here (paths relative to /root/reference):
 *   - level start offsets = exclusive cumsum of h*w      src/msda_triton/kernels.py:58-62
 *   - coordinate un-normalisation                        src/msda_triton/kernels.py:141-146
 *   - floor / +1 neighbours, unclamped dx, dy            src/msda_triton/kernels.py:150-153,235-237
 *   - "zeros" per-axis validity masks                    src/msda_triton/kernels.py:158-162,220-231
 *   - clamp of the four corner indices                   src/msda_triton/kernels.py:166-169
 *   - pixel addressing (start + y*w + x)*H*C + h*C + c   src/msda_triton/kernels.py:184-203
 *   - bilinear blend                                     src/msda_triton/kernels.py:239-244
 *   - attention-weighted (level, point) reduction        src/msda_triton/kernels.py:339
 *   - grad_attention_weights                             src/msda_triton/kernels.py:494
 *   - grad_sampling_points (scale W-1/H-1 or W/H)        src/msda_triton/kernels.py:510-524
 *   - grad_img scatter-add of the four corner tiles      src/msda_triton/kernels.py:543-553
 * which is the same function as the native fallback (per-level F.grid_sample +
 * weighted sum, src/msda_triton/frontend.py:15-68).
 *
 * Required macros: REAL (float|double), FN(name) (symbol suffixing), FLOOR.
 */

typedef struct {
    int64_t h, w, start;
} FN(level_t);

static int FN(load_levels)(const int64_t *shapes, int64_t L, FN(level_t) * lv, int64_t I)
{
    int64_t start = 0;
    for (int64_t l = 0; l < L; ++l) {
        lv[l].h = shapes[2 * l + 0];
        lv[l].w = shapes[2 * l + 1];
        lv[l].start = start;
        if (lv[l].h <= 0 || lv[l].w <= 0) return -2;
        start += lv[l].h * lv[l].w;
    }
    return start == I ? 0 : -3;
}

/* One bilinear tap set: indices (clamped), validity flags and fractional parts. */
typedef struct {
    int64_t i00, i01, i10, i11; /* pixel index inside the packed pyramid */
    int m00, m01, m10, m11;     /* corner contributes? (always 1 for border) */
    REAL dx, dy;
    int gx_on, gy_on; /* 0 where grid_sample's border clipping kills the location gradient */
} FN(taps_t);

static inline int64_t FN(clampi)(REAL v, int64_t hi)
{
    /* clamp in floating point first so far-OOB values cannot overflow (kernels.py:166-169) */
    if (!(v > (REAL)0)) return 0; /* also catches NaN */
    if (v > (REAL)hi) return hi;
    return (int64_t)v;
}

static inline void FN(make_taps)(REAL x, REAL y, const FN(level_t) * lv, int padding_zeros,
                                 int align_corners, FN(taps_t) * t)
{
    const REAL W = (REAL)lv->w, H = (REAL)lv->h;
    REAL px, py;
    if (align_corners) {
        px = x * (W - (REAL)1);
        py = y * (H - (REAL)1);
    } else {
        px = x * W - (REAL)0.5;
        py = y * H - (REAL)0.5;
    }
    const REAL x0 = FLOOR(px), y0 = FLOOR(py);
    const REAL x1 = x0 + (REAL)1, y1 = y0 + (REAL)1;
    int mx0 = 1, mx1 = 1, my0 = 1, my1 = 1;
    if (padding_zeros) {
        mx0 = ((REAL)0 <= x0) && (x0 <= W - (REAL)1);
        mx1 = ((REAL)0 <= x1) && (x1 <= W - (REAL)1);
        my0 = ((REAL)0 <= y0) && (y0 <= H - (REAL)1);
        my1 = ((REAL)0 <= y1) && (y1 <= H - (REAL)1);
    }
    const int64_t x0c = FN(clampi)(x0, lv->w - 1), x1c = FN(clampi)(x1, lv->w - 1);
    const int64_t y0c = FN(clampi)(y0, lv->h - 1), y1c = FN(clampi)(y1, lv->h - 1);
    t->i00 = lv->start + y0c * lv->w + x0c;
    t->i01 = lv->start + y0c * lv->w + x1c;
    t->i10 = lv->start + y1c * lv->w + x0c;
    t->i11 = lv->start + y1c * lv->w + x1c;
    t->m00 = my0 && mx0;
    t->m01 = my0 && mx1;
    t->m10 = my1 && mx0;
    t->m11 = my1 && mx1;
    t->dx = px - x0;
    t->dy = py - y0;
    /* "border" == grid_sample's clip_coordinates: a coordinate at or beyond the first/last pixel
     * centre is clipped and its gradient is zero (native fallback, frontend.py:53-56).  The clamped
     * corner formula already yields 0 everywhere except exactly at px == 0 / py == 0, where the
     * reference's Triton kernel (kernels.py:518-524) and its native fallback disagree; the parity
     * target is the native fallback, so that kink follows grid_sample. */
    t->gx_on = padding_zeros || (px > (REAL)0 && px < W - (REAL)1);
    t->gy_on = padding_zeros || (py > (REAL)0 && py < H - (REAL)1);
}

/*
 * out[b,q,h,:] = sum_{l,p} attn[b,q,h,l,p] * bilinear(value_l[b,:,h,:], loc[b,q,h,l,p,:])
 * Layouts (all contiguous, row-major):
 *   value [B,I,H,D]   shapes [L,2] (h,w) int64   loc [B,Q,H,L,P,2] (x,y)
 *   attn  [B,Q,H,L,P] out    [B,Q,H,D]
 */
int FN(msda_oracle_fwd)(const REAL *value, const int64_t *shapes, const REAL *loc,
                        const REAL *attn, REAL *out, int64_t B, int64_t I, int64_t H, int64_t D,
                        int64_t Q, int64_t L, int64_t P, int padding_zeros, int align_corners)
{
    if (L > MSDA_ORACLE_MAX_LEVELS) return -1;
    FN(level_t) lv[MSDA_ORACLE_MAX_LEVELS];
    int rc = FN(load_levels)(shapes, L, lv, I);
    if (rc) return rc;

#pragma omp parallel for schedule(static)
    for (int64_t bq = 0; bq < B * Q; ++bq) {
        const int64_t b = bq / Q;
        const REAL *vb = value + b * I * H * D;
        for (int64_t h = 0; h < H; ++h) {
            const int64_t u = bq * H + h;
            REAL *o = out + u * D;
            for (int64_t c = 0; c < D; ++c) o[c] = (REAL)0;
            for (int64_t l = 0; l < L; ++l) {
                for (int64_t p = 0; p < P; ++p) {
                    const int64_t s = (u * L + l) * P + p;
                    FN(taps_t) t;
                    FN(make_taps)(loc[2 * s], loc[2 * s + 1], &lv[l], padding_zeros, align_corners, &t);
                    const REAL a = attn[s];
                    const REAL w00 = ((REAL)1 - t.dy) * ((REAL)1 - t.dx), w01 = ((REAL)1 - t.dy) * t.dx;
                    const REAL w10 = t.dy * ((REAL)1 - t.dx), w11 = t.dy * t.dx;
                    const REAL *v00 = vb + (t.i00 * H + h) * D, *v01 = vb + (t.i01 * H + h) * D;
                    const REAL *v10 = vb + (t.i10 * H + h) * D, *v11 = vb + (t.i11 * H + h) * D;
                    for (int64_t c = 0; c < D; ++c) {
                        const REAL smp = (t.m00 ? v00[c] : (REAL)0) * w00 + (t.m01 ? v01[c] : (REAL)0) * w01 +
                                         (t.m10 ? v10[c] : (REAL)0) * w10 + (t.m11 ? v11[c] : (REAL)0) * w11;
                        o[c] += a * smp;
                    }
                }
            }
        }
    }
    return 0;
}

/*
 * Backward of the above.  grad_value [B,I,H,D] is zeroed here and then scatter-added;
 * grad_loc [B,Q,H,L,P,2] and grad_attn [B,Q,H,L,P] are written once per sample.
 * Parallel over (b,h): every (b,h) plane of grad_value has a single owner thread, so the
 * scatter needs no atomics and the result is run-to-run deterministic.
 */
int FN(msda_oracle_bwd)(const REAL *grad_out, const REAL *value, const int64_t *shapes,
                        const REAL *loc, const REAL *attn, REAL *grad_value, REAL *grad_loc,
                        REAL *grad_attn, int64_t B, int64_t I, int64_t H, int64_t D, int64_t Q,
                        int64_t L, int64_t P, int padding_zeros, int align_corners)
{
    if (L > MSDA_ORACLE_MAX_LEVELS) return -1;
    FN(level_t) lv[MSDA_ORACLE_MAX_LEVELS];
    int rc = FN(load_levels)(shapes, L, lv, I);
    if (rc) return rc;

    for (int64_t i = 0; i < B * I * H * D; ++i) grad_value[i] = (REAL)0;

#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t bh = 0; bh < B * H; ++bh) {
        const int64_t b = bh / H, h = bh % H;
        const REAL *vb = value + b * I * H * D;
        REAL *gvb = grad_value + b * I * H * D;
        for (int64_t q = 0; q < Q; ++q) {
            const int64_t u = (b * Q + q) * H + h;
            const REAL *go = grad_out + u * D;
            for (int64_t l = 0; l < L; ++l) {
                const REAL sx = align_corners ? (REAL)(lv[l].w - 1) : (REAL)lv[l].w;
                const REAL sy = align_corners ? (REAL)(lv[l].h - 1) : (REAL)lv[l].h;
                for (int64_t p = 0; p < P; ++p) {
                    const int64_t s = (u * L + l) * P + p;
                    FN(taps_t) t;
                    FN(make_taps)(loc[2 * s], loc[2 * s + 1], &lv[l], padding_zeros, align_corners, &t);
                    const REAL a = attn[s];
                    const REAL w00 = ((REAL)1 - t.dy) * ((REAL)1 - t.dx), w01 = ((REAL)1 - t.dy) * t.dx;
                    const REAL w10 = t.dy * ((REAL)1 - t.dx), w11 = t.dy * t.dx;
                    const int64_t o00 = (t.i00 * H + h) * D, o01 = (t.i01 * H + h) * D;
                    const int64_t o10 = (t.i10 * H + h) * D, o11 = (t.i11 * H + h) * D;
                    REAL ga = (REAL)0, gx = (REAL)0, gy = (REAL)0;
                    for (int64_t c = 0; c < D; ++c) {
                        const REAL v00 = t.m00 ? vb[o00 + c] : (REAL)0, v01 = t.m01 ? vb[o01 + c] : (REAL)0;
                        const REAL v10 = t.m10 ? vb[o10 + c] : (REAL)0, v11 = t.m11 ? vb[o11 + c] : (REAL)0;
                        const REAL g = go[c];
                        ga += g * (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11);
                        gx += g * (((REAL)1 - t.dy) * (v01 - v00) + t.dy * (v11 - v10));
                        gy += g * (((REAL)1 - t.dx) * (v10 - v00) + t.dx * (v11 - v01));
                        const REAL ag = a * g;
                        if (t.m00) gvb[o00 + c] += ag * w00;
                        if (t.m01) gvb[o01 + c] += ag * w01;
                        if (t.m10) gvb[o10 + c] += ag * w10;
                        if (t.m11) gvb[o11 + c] += ag * w11;
                    }
                    grad_attn[s] = ga;
                    grad_loc[2 * s + 0] = t.gx_on ? a * sx * gx : (REAL)0;
                    grad_loc[2 * s + 1] = t.gy_on ? a * sy * gy : (REAL)0;
                }
            }
        }
    }
    return 0;
}
