// msda_value_tile.hpp — grad_value last resort: owner-computes tiles in LDS, no workspace.
// Used when neither the single-launch kernel (msda_value_small.hpp: small problems) nor the sorted gather
// (msda_value_sorted.hpp: needs the caller's workspace, L <= 16) applies, or with msda_set_option("value_path", 1).
#pragma once

#include "msda_kernels.hpp"

namespace msda {

// ==========================================================================================
// backward, part 2: grad_value without global atomics.  A workgroup OWNS a tile of grad_value:
// one (b, h) plane x CH channels x a contiguous pixel range, held as accumulate-typed sums in
// LDS.  It streams every sample of its plane whose level intersects the range, adds the four
// corner contributions into LDS (ds_add_f32 / ds_add_f64) and finally stores the tile with plain
// stores; every element of grad_value is written exactly once, so no memset is needed either.
// ==========================================================================================
constexpr int kValueBlock = 1024;
using TileAcc = double;

template <typename T, int CH, typename TV = T>  // TV: storage type of grad_value (see msda_fwd_kernel)
__global__ __launch_bounds__(kValueBlock) void msda_bwd_value_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    using TVR = Traits<TV>;

    int pair, tile;
    if (!decode_block(p.grid3d, p.B * p.H, p.nchunks * p.nranges, p.xcd_map, pair, tile)) return;
    const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
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
    TV *gv = static_cast<TV *>(p.grad_value) + (size_t)b * p.I * p.H * p.D + (size_t)h * p.D + chunk * CH;
    for (int i = tid; i < npx; i += kValueBlock) {
        Pack<TV, CH> o;
#pragma unroll
        for (int c = 0; c < CH; ++c) o.v[c] = TVR::from_acc((A)s_acc[c * npx + i]);
        *reinterpret_cast<Pack<TV, CH> *>(gv + (size_t)(p0 + i) * p.H * p.D) = o;
    }
}

}  // namespace msda
