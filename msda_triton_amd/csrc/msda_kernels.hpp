// msda_kernels.hpp — the three gfx950 kernels and their host-side launch logic.
//
//   msda_fwd_kernel         out = sum_{l,p} attn * bilinear(value_l, loc)        (kernels.py:259-348)
//   msda_bwd_sample_kernel  grad_loc, grad_attn (private per sample, no atomics)  (kernels.py:494-537)
//   msda_bwd_value_kernel   grad_value, owner-computes tiles accumulated in LDS   (kernels.py:543-553)
//
// Work decomposition (all three): a workgroup owns ONE (batch, head) plane of `value` and a slice
// of the queries, so every row it gathers comes from one 2-D plane that the XCD-aware block map
// keeps resident in a single L2.  Inside the gather kernels a *unit* = one (b, q, h); a unit is
// served by G lanes (G * VEC >= D channels, VEC elements = one 16-byte load per lane), i.e.
// 64/G units per wavefront and 256/G per workgroup.
//
// Phase 1 (all 256 threads, one sample each): read (x, y, a), do the coordinate math ONCE per
//   sample, park {4 row offsets, 4 weights} in LDS.  Loads are coalesced along (l, p).
// Phase 2 (per unit, G lanes): walk the unit's samples, broadcast-read the parked record, issue
//   the four row gathers as 16-byte range-checked buffer loads, FMA into per-lane accumulators.
#pragma once

#include "msda_common.hpp"

namespace msda {

struct Params {
    const void *value;
    const int64_t *shapes;
    const void *loc;
    const void *attn;
    void *out;
    const void *grad_out;
    void *grad_value;
    void *grad_loc;
    void *grad_attn;
    int B, I, H, D, Q, L, P, LP;
    int nqc;       // query chunks per (b,h) plane (gather kernels)
    int sc;        // samples of a unit parked in LDS at a time (<= LP)
    int zeros, align, xcd_map;
    // grad_value kernel tiling
    int nchunks;   // channel chunks (D / CH)
    int nranges;   // pixel ranges
    int range_px;  // pixels per range
    // sorted (gather-formulated) grad_value path: caller-provided workspace, see msda_value_sorted.hpp
    int *ws_part;       // [pairs][nsplit][nc_cap]  per-slice cell counts, then each slice's first slot per cell
    int *ws_off;        // [pairs][nc_cap+1]  exclusive offsets of the cell lists
    int4 *ws_pixrec;    // [pairs][I][3]      per pixel: list starts, list lengths, (first item, chunks, -, -)
    int *ws_itemcnt;    // [pairs]            work items of the plane
    int4 *ws_items;     // [pairs][it_cap]    (pixel, chunk, chunks of the pixel, -)
    void *ws_entries;   // [pairs][Q*L*P]     Entry<acc>: sample records sorted by cell
    void *ws_scratch;   // [pairs][it_cap][D] acc-typed partial rows of multi-chunk pixels
    int nc_cap, it_cap;
    int nsplit;         // query slices per plane in the count / place passes
    int cell_cap;       // cells a count / place workgroup holds in LDS at a time
};

extern __shared__ __attribute__((aligned(16))) unsigned char msda_smem[];

template <typename A> struct alignas(16) Rec4 {
    A v[4];
};

// ==========================================================================================
// forward
// ==========================================================================================
template <typename T, int VEC, int G>
__global__ __launch_bounds__(kBlock) void msda_fwd_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    constexpr int NU = kBlock / G;

    int pair, qc;
    if (!decode_block(blockIdx.x, p.B * p.H, p.nqc, p.xcd_map, pair, qc)) return;
    const int b = pair / p.H, h = pair - b * p.H;
    const int q0 = qc * NU;

    LevelTab *tab = reinterpret_cast<LevelTab *>(msda_smem);
    const int scp = p.sc + 1;  // +1 record of padding: units land on different LDS banks
    uint4 *s_off = reinterpret_cast<uint4 *>(msda_smem + sizeof(LevelTab));
    Rec4<A> *s_wt = reinterpret_cast<Rec4<A> *>(s_off + NU * scp);

    load_level_table(tab, p.shapes, p.L);

    const int tid = threadIdx.x;
    const int unit = tid / G, j = tid % G;
    const int q = q0 + unit;
    const bool unit_ok = q < p.Q;

    const uint32_t row_bytes = (uint32_t)(p.H * p.D) * (uint32_t)sizeof(T);
    const T *plane = static_cast<const T *>(p.value) + (size_t)b * p.I * p.H * p.D + (size_t)h * p.D;
    const uint32_t plane_bytes = (uint32_t)(((size_t)p.I * p.H * p.D - (size_t)h * p.D) * sizeof(T));
    const rsrc_t rs = make_rsrc(plane, plane_bytes);

    const T *loc = static_cast<const T *>(p.loc);
    const T *attn = static_cast<const T *>(p.attn);
    const float inv_P = 1.0f / (float)p.P;

    const int nchan_chunks = (p.D + G * VEC - 1) / (G * VEC);
    for (int cc = 0; cc < nchan_chunks; ++cc) {
        const int c0 = (cc * G + j) * VEC;
        const bool lane_ok = unit_ok && (c0 < p.D);
        A acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = (A)0;

        for (int s0 = 0; s0 < p.LP; s0 += p.sc) {
            const int sc = min(p.sc, p.LP - s0);
            const float inv_sc = 1.0f / (float)sc;
            __syncthreads();  // level table ready / previous chunk's records consumed
            // ---- phase 1: one sample per thread ----
            for (int f = tid; f < NU * sc; f += kBlock) {
                const int fu = div_small(f, sc, inv_sc);
                const int sl = s0 + (f - fu * sc);
                const int fq = q0 + fu;
                if (fq < p.Q) {
                    const int l = div_small(sl, p.P, inv_P);
                    const size_t sidx = ((size_t)(b * (size_t)p.Q + fq) * p.H + h) * p.LP + sl;
                    const Pack<T, 2> xy = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
                    const A a = TR::to_acc(attn[sidx]);
                    Taps<A> t;
                    make_taps<A>(TR::to_acc(xy.v[0]), TR::to_acc(xy.v[1]), tab->h[l], tab->w[l], tab->start[l],
                                 p.zeros, p.align, row_bytes, t);
                    const A wy0 = (A)1 - t.dy, wx0 = (A)1 - t.dx;
                    Rec4<A> w;
                    w.v[0] = a * (wy0 * wx0);
                    w.v[1] = a * (wy0 * t.dx);
                    w.v[2] = a * (t.dy * wx0);
                    w.v[3] = a * (t.dy * t.dx);
                    const int slot = fu * scp + (sl - s0);
                    s_off[slot] = make_uint4(t.off[0], t.off[1], t.off[2], t.off[3]);
                    s_wt[slot] = w;
                }
            }
            __syncthreads();
            // ---- phase 2: gather + blend ----
            if (lane_ok) {
                const uint32_t lane_off = (uint32_t)c0 * (uint32_t)sizeof(T);
                const uint4 *uo = s_off + unit * scp;
                const Rec4<A> *uw = s_wt + unit * scp;
#pragma unroll 4
                for (int s = 0; s < sc; ++s) {
                    const uint4 o = uo[s];
                    const Rec4<A> w = uw[s];
                    A v0[VEC], v1[VEC], v2[VEC], v3[VEC];
                    load_row<T, VEC>(rs, o.x + lane_off, v0);
                    load_row<T, VEC>(rs, o.y + lane_off, v1);
                    load_row<T, VEC>(rs, o.z + lane_off, v2);
                    load_row<T, VEC>(rs, o.w + lane_off, v3);
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        acc[i] += w.v[0] * v0[i] + w.v[1] * v1[i] + w.v[2] * v2[i] + w.v[3] * v3[i];
                }
            }
        }
        if (lane_ok) {
            Pack<T, VEC> o;
#pragma unroll
            for (int i = 0; i < VEC; ++i) o.v[i] = TR::from_acc(acc[i]);
            T *dst = static_cast<T *>(p.out) + ((size_t)(b * (size_t)p.Q + q) * p.H + h) * p.D + c0;
            *reinterpret_cast<Pack<T, VEC> *>(dst) = o;
        }
    }
}

// ==========================================================================================
// backward, part 1: grad_loc and grad_attn.  Same decomposition as the forward; every sample's
// three results are reduced over the unit's G lanes with DPP moves and written exactly once.
// ==========================================================================================
template <typename T, int VEC, int G>
__global__ __launch_bounds__(kBlock) void msda_bwd_sample_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    constexpr int NU = kBlock / G;

    int pair, qc;
    if (!decode_block(blockIdx.x, p.B * p.H, p.nqc, p.xcd_map, pair, qc)) return;
    const int b = pair / p.H, h = pair - b * p.H;
    const int q0 = qc * NU;

    LevelTab *tab = reinterpret_cast<LevelTab *>(msda_smem);
    const int scp = p.sc + 1;
    uint4 *s_off = reinterpret_cast<uint4 *>(msda_smem + sizeof(LevelTab));
    // record in : {dx, dy, a*sx*gx_on, a*sy*gy_on};  record out (same slot): {gA, gX, gY, -}
    Rec4<A> *s_par = reinterpret_cast<Rec4<A> *>(s_off + NU * scp);

    load_level_table(tab, p.shapes, p.L);

    const int tid = threadIdx.x;
    const int unit = tid / G, j = tid % G;
    const int q = q0 + unit;
    const bool unit_ok = q < p.Q;

    const uint32_t row_bytes = (uint32_t)(p.H * p.D) * (uint32_t)sizeof(T);
    const T *plane = static_cast<const T *>(p.value) + (size_t)b * p.I * p.H * p.D + (size_t)h * p.D;
    const uint32_t plane_bytes = (uint32_t)(((size_t)p.I * p.H * p.D - (size_t)h * p.D) * sizeof(T));
    const rsrc_t rs = make_rsrc(plane, plane_bytes);

    const T *loc = static_cast<const T *>(p.loc);
    const T *attn = static_cast<const T *>(p.attn);
    const float inv_P = 1.0f / (float)p.P;
    const int nchan_chunks = (p.D + G * VEC - 1) / (G * VEC);

    for (int s0 = 0; s0 < p.LP; s0 += p.sc) {
        const int sc = min(p.sc, p.LP - s0);
        const float inv_sc = 1.0f / (float)sc;
        __syncthreads();
        // ---- phase 1 ----
        for (int f = tid; f < NU * sc; f += kBlock) {
            const int fu = div_small(f, sc, inv_sc);
            const int sl = s0 + (f - fu * sc);
            const int fq = q0 + fu;
            if (fq < p.Q) {
                const int l = div_small(sl, p.P, inv_P);
                const size_t sidx = ((size_t)(b * (size_t)p.Q + fq) * p.H + h) * p.LP + sl;
                const Pack<T, 2> xy = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
                const A a = TR::to_acc(attn[sidx]);
                const int lh = tab->h[l], lw = tab->w[l];
                Taps<A> t;
                make_taps<A>(TR::to_acc(xy.v[0]), TR::to_acc(xy.v[1]), lh, lw, tab->start[l], p.zeros, p.align,
                             row_bytes, t);
                const A sx = p.align ? (A)(lw - 1) : (A)lw;
                const A sy = p.align ? (A)(lh - 1) : (A)lh;
                Rec4<A> r;
                r.v[0] = t.dx;
                r.v[1] = t.dy;
                r.v[2] = t.gx_on ? a * sx : (A)0;
                r.v[3] = t.gy_on ? a * sy : (A)0;
                const int slot = fu * scp + (sl - s0);
                s_off[slot] = make_uint4(t.off[0], t.off[1], t.off[2], t.off[3]);
                s_par[slot] = r;
            }
        }
        __syncthreads();
        // ---- phase 2: four dot products with grad_out per sample, reduced over the unit ----
        if (unit_ok) {  // wave-divergent only at the ragged tail; idle lanes of a live unit still join the DPP sums
            const uint4 *uo = s_off + unit * scp;
            Rec4<A> *up = s_par + unit * scp;
            const T *go_row = static_cast<const T *>(p.grad_out) + ((size_t)(b * (size_t)p.Q + q) * p.H + h) * p.D;
            for (int s = 0; s < sc; ++s) {
                const uint4 o = uo[s];
                const Rec4<A> r = up[s];
                A d0 = 0, d1 = 0, d2 = 0, d3 = 0;
                for (int cc = 0; cc < nchan_chunks; ++cc) {
                    const int c0 = (cc * G + j) * VEC;
                    if (c0 < p.D) {
                        const uint32_t lane_off = (uint32_t)c0 * (uint32_t)sizeof(T);
                        const Pack<T, VEC> gp = *reinterpret_cast<const Pack<T, VEC> *>(go_row + c0);
                        A v0[VEC], v1[VEC], v2[VEC], v3[VEC];
                        load_row<T, VEC>(rs, o.x + lane_off, v0);
                        load_row<T, VEC>(rs, o.y + lane_off, v1);
                        load_row<T, VEC>(rs, o.z + lane_off, v2);
                        load_row<T, VEC>(rs, o.w + lane_off, v3);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            const A g = TR::to_acc(gp.v[i]);
                            d0 += g * v0[i];
                            d1 += g * v1[i];
                            d2 += g * v2[i];
                            d3 += g * v3[i];
                        }
                    }
                }
                const A dx = r.v[0], dy = r.v[1];
                const A wy0 = (A)1 - dy, wx0 = (A)1 - dx;
                A gA = (wy0 * wx0) * d0 + (wy0 * dx) * d1 + (dy * wx0) * d2 + (dy * dx) * d3;
                A gX = wy0 * (d1 - d0) + dy * (d3 - d2);
                A gY = wx0 * (d2 - d0) + dx * (d3 - d1);
                gA = group_sum<G>(gA);
                gX = group_sum<G>(gX);
                gY = group_sum<G>(gY);
                if (j == 0) {
                    Rec4<A> res;
                    res.v[0] = gA;
                    res.v[1] = r.v[2] * gX;
                    res.v[2] = r.v[3] * gY;
                    res.v[3] = (A)0;
                    up[s] = res;
                }
            }
        }
        __syncthreads();
        // ---- phase 3: coalesced write-out, one sample per thread ----
        for (int f = tid; f < NU * sc; f += kBlock) {
            const int fu = div_small(f, sc, inv_sc);
            const int sl = s0 + (f - fu * sc);
            const int fq = q0 + fu;
            if (fq < p.Q) {
                const size_t sidx = ((size_t)(b * (size_t)p.Q + fq) * p.H + h) * p.LP + sl;
                const Rec4<A> res = s_par[fu * scp + (sl - s0)];
                static_cast<T *>(p.grad_attn)[sidx] = TR::from_acc(res.v[0]);
                Pack<T, 2> g;
                g.v[0] = TR::from_acc(res.v[1]);
                g.v[1] = TR::from_acc(res.v[2]);
                *reinterpret_cast<Pack<T, 2> *>(static_cast<T *>(p.grad_loc) + 2 * sidx) = g;
            }
        }
    }
}

// ==========================================================================================
// backward, part 2: grad_value without global atomics.  A workgroup OWNS a tile of grad_value:
// one (b, h) plane x CH channels x a contiguous pixel range, held as accumulate-typed sums in
// LDS.  It streams every sample of its plane whose level intersects the range, adds the four
// corner contributions into LDS (ds_add_f32 / ds_add_f64) and finally stores the tile with plain
// stores; every element of grad_value is written exactly once, so no memset is needed either.
// ==========================================================================================
constexpr int kValueBlock = 1024;
using TileAcc = double;

template <typename T, int CH>
__global__ __launch_bounds__(kValueBlock) void msda_bwd_value_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;

    int pair, tile;
    if (!decode_block(blockIdx.x, p.B * p.H, p.nchunks * p.nranges, p.xcd_map, pair, tile)) return;
    const int b = pair / p.H, h = pair - b * p.H;
    const int range = tile / p.nchunks, chunk = tile - range * p.nchunks;
    const int p0 = range * p.range_px;
    const int p1 = min(p.I, p0 + p.range_px);
    const int npx = p1 - p0;
    if (npx <= 0) return;

    LevelTab *tab = reinterpret_cast<LevelTab *>(msda_smem);
    // Tile sums are double whatever the storage type: on gfx950 ds_add_f64 retires a wave-instruction
    // in ~21 cycles while ds_add_f32 takes ~193 (measured, tools/lds_atomic_bench.hip) — and the
    // wider sums make the scatter order irrelevant at fp32 output precision.  Layout [CH][npx]
    // (channel-major) spreads a wave's pixels over all LDS banks.
    TileAcc *s_acc = reinterpret_cast<TileAcc *>(msda_smem + sizeof(LevelTab));

    load_level_table(tab, p.shapes, p.L);
    const int tid = threadIdx.x;
    for (int i = tid; i < npx * CH; i += kValueBlock) s_acc[i] = (TileAcc)0;
    __syncthreads();

    // levels intersecting [p0, p1) form an interval [la, lb)
    int la = p.L, lb = 0;
    for (int l = 0; l < p.L; ++l) {
        const int ls = tab->start[l], le = ls + tab->h[l] * tab->w[l];
        if (le > p0 && ls < p1) {
            la = min(la, l);
            lb = max(lb, l + 1);
        }
    }
    const int nl = lb - la;
    if (nl > 0) {
        const int m = nl * p.P;  // samples of one unit that can touch this tile
        const float inv_P = 1.0f / (float)p.P;
        const T *loc = static_cast<const T *>(p.loc);
        const T *attn = static_cast<const T *>(p.attn);
        const T *gout = static_cast<const T *>(p.grad_out);
        const int c0 = chunk * CH;
        // (q, r) walk the flattened (query, sample-in-interval) space with stride kValueBlock
        int q = tid / m, r = tid - q * m;
        const int dq = kValueBlock / m, dr = kValueBlock - dq * m;
        for (; q < p.Q;) {
            const int li = div_small(r, p.P, inv_P);
            const int l = la + li;
            const size_t u = (size_t)(b * (size_t)p.Q + q) * p.H + h;
            const size_t sidx = u * p.LP + (size_t)la * p.P + r;
            const Pack<T, 2> xy = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
            const A a = TR::to_acc(attn[sidx]);
            const Pack<T, CH> gp = *reinterpret_cast<const Pack<T, CH> *>(gout + u * p.D + c0);
            Taps<A> t;
            make_taps<A>(TR::to_acc(xy.v[0]), TR::to_acc(xy.v[1]), tab->h[l], tab->w[l], tab->start[l], p.zeros,
                         p.align, /*row_bytes=*/1u, t);  // offsets == pixel indices here
            const A wy0 = (A)1 - t.dy, wx0 = (A)1 - t.dx;
            A w[4] = {a * (wy0 * wx0), a * (wy0 * t.dx), a * (t.dy * wx0), a * (t.dy * t.dx)};
            A g[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) g[c] = TR::to_acc(gp.v[c]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t rel = t.off[k] - (uint32_t)p0;  // masked / out-of-range wrap to huge values
                if (rel < (uint32_t)npx) {
#pragma unroll
                    for (int c = 0; c < CH; ++c) atomicAdd(&s_acc[c * npx + rel], (TileAcc)(w[k] * g[c]));
                }
            }
            q += dq;
            r += dr;
            if (r >= m) {
                r -= m;
                ++q;
            }
        }
    }
    __syncthreads();
    // tile write-out: CH contiguous channels per pixel
    T *gv = static_cast<T *>(p.grad_value) + (size_t)b * p.I * p.H * p.D + (size_t)h * p.D + chunk * CH;
    for (int i = tid; i < npx; i += kValueBlock) {
        Pack<T, CH> o;
#pragma unroll
        for (int c = 0; c < CH; ++c) o.v[c] = TR::from_acc((A)s_acc[c * npx + i]);
        *reinterpret_cast<Pack<T, CH> *>(gv + (size_t)(p0 + i) * p.H * p.D) = o;
    }
}

}  // namespace msda
