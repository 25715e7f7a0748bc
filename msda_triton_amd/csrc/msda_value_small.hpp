// msda_value_small.hpp — grad_value for SMALL problems in ONE launch, no workspace, no atomics on floating-point data.
//
// Grounding-DINO / Deformable-DETR decoder shapes (a few hundred queries per batch element) put only a few thousand
// samples on each (batch, head) plane and level; the sorted pipeline's five launches (msda_value_sorted.hpp) then
// cost more than the work.  Here one 1024-thread workgroup owns ONE (plane, level) and keeps everything in LDS:
//
//   1 count    every sample of the level -> its bilinear cell (msda_value_sorted.hpp: sample_cell); LDS histogram
//   2 scan     exclusive scan of the histogram in place: first record of every cell's list; the longest list
//   3 place    samples again -> records {q, dx, dy, a} at their cell list's cursor (LDS)
//   4 gather   PIXEL-major: a G-lane group owns a 2 x 2 block of pixels (or 1/S of a busy block's records), walks the
//              lists of the nine cells around it, loads each record's grad_out row (16 bytes per lane) and FMAs it with
//              the corner weights into the block's four accumulator rows; S > 1: the S partial row sets are summed with
//              shuffles.  Blocks are handed out heaviest first (the groups of a wave then need the same number of
//              batches).  A pixel's grad_value row is stored once — complete, so there are no partial rows in memory
//              and no finish pass.
//
// Every grad_value row of the level is written exactly once by plain stores (rows nobody samples: zeros).
// Replaces tl.atomic_add of the reference (kernels.py:543-553) for these shapes.
#pragma once

#include <type_traits>

#include "msda_value_sorted.hpp"

namespace msda {

constexpr int kSmallBlock = 1024;

template <typename A> struct alignas(16) SmallRec {
    uint32_t q;
    A dx, dy, a;
};

// dynamic LDS of the kernel for a level of at most `cells` cells and `samples` samples (host + device agree)
inline size_t small_lds_bytes(size_t cells, size_t samples, size_t acc_bytes, size_t vec)
{
    const size_t rec = acc_bytes == 8 ? 32 : 16;
    size_t o = sizeof(LevelTab);
    o = (o + 15) / 16 * 16 + (cells + 1) * 4;              // counters / cell list starts
    o = (o + 15) / 16 * 16 + samples * rec;                // records sorted by cell
    o = (o + 15) / 16 * 16 + (size_t)kSmallBlock * (4 + 4 * acc_bytes);  // converted (row offset, four weights) hand-off
    o = (o + 15) / 16 * 16 + (cells / 4 + 2) * 2;          // the level's 2 x 2-pixel blocks, ordered by their work
    (void)vec;
    return o + 64;
}

template <typename T, int VEC, int G, typename TV = T, typename TG = T>  // TV: storage type of grad_value, TG: of grad_out
__global__ __launch_bounds__(kSmallBlock) void msda_value_small_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    using TVR = Traits<TV>;
    constexpr int NG = kSmallBlock / G;  // lane groups per workgroup
    // row loads in flight per lane: 8 — 4 for rows of eight 16-bit channels per lane, whose widened copies (8 floats each) next
    // to the four accumulator rows do not fit the 128 VGPRs of a 1024-thread workgroup (6-26 spilled registers otherwise)
    constexpr int UB = G < 8 ? G : (VEC >= 8 ? 4 : 8);
    // p.small_ns workgroups per (plane, level): each builds the level's sorted records for itself (cheap) and takes
    // every small_ns-th 2 x 2-pixel block of the gather, so few planes still fill the chip
    int pair, slot;
    if (!decode_block(p.grid3d, p.B * p.H, p.L * p.small_ns, p.xcd_map, pair, slot)) return;
    const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
    const int tid = threadIdx.x;

    unsigned char *sm = msda_smem;
    LevelTab *tab = reinterpret_cast<LevelTab *>(sm);
    size_t o = (sizeof(LevelTab) + 15) / 16 * 16;
    int *s_off = reinterpret_cast<int *>(sm + o);  // [ncl + 1]
    // slot -> (level, share of the level's gather rounds)
    const int lvl = slot / p.small_ns, share = slot - lvl * p.small_ns, nshare = p.small_ns;
    // ---- samples of this (plane, level): thread t serves point t % P of the queries t / P + k * (threads / P) ----
    const size_t plane_s0 = ((size_t)b * p.Q * p.H + h) * p.LP;
    const T *loc = static_cast<const T *>(p.loc) + 2 * plane_s0;
    const T *attn = static_cast<const T *>(p.attn) + plane_s0;
    const int HLP = p.H * p.LP;
    const int dq = p.P <= kSmallBlock ? kSmallBlock / p.P : 1;
    const bool active = p.P <= kSmallBlock && tid < dq * p.P;
    const int pt = active ? tid % p.P : 0, q0 = active ? tid / p.P : 0;
    // This thread's samples, fetched ONCE and up front — before the level table is waited for, their addresses do not
    // need it — (both walks below reuse them): the kernel is a chain of latencies inside one workgroup, so every
    // global round trip saved counts.  Problems this kernel is chosen for have at most kPre samples per thread;
    // longer walks load on the spot.
    constexpr int kPre = 4;
    const bool pre = active && (p.Q + dq - 1) / dq <= kPre;
    Pack<T, 2> pre_xy[kPre];
    T pre_a[kPre];
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
        pre_xy[k].v[0] = pre_xy[k].v[1] = pre_a[k] = TR::from_acc((A)0);
        const int q = q0 + k * dq;
        if (pre && q < p.Q) {
            const int sidx = q * HLP + lvl * p.P + pt;
            pre_xy[k] = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
            pre_a[k] = attn[sidx];
        }
    }
    load_level_table(tab, p.shapes, p.L);
    __syncthreads();
    const int lw = tab->w[lvl], lh = tab->h[lvl], cw = lw + 1;
    const int ncl_true = (lh + 1) * cw;
    const int ncl = min(ncl_true, p.small_cells);  // (shapes that disagree with I: never index past the LDS table)
    const int npix = lw * lh;
    const int pstart = tab->start[lvl];
    if (p.small_hinted && ncl_true > p.small_cells) {
        // The caller promised smaller levels than this one (max_level_cells) and the table was sized on that
        // promise: there is no way to report it from here, so the level's rows come back as NaN instead of wrong.
        if (share == 0) {
            const A nan = (A)__builtin_nanf("");
            for (long long e = tid; e < (long long)npix * p.D; e += kSmallBlock) {
                const int px = (int)(e / p.D), c = (int)(e - (long long)px * p.D);
                if (pstart + px < p.I)
                    static_cast<TV *>(p.grad_value)[(((size_t)b * p.I + pstart + px) * p.H + h) * p.D + c] = TVR::from_acc(nan);
            }
        }
        return;  // (uniform in the workgroup)
    }
    o += ((size_t)p.small_cells + 1) * 4;
    o = (o + 15) / 16 * 16;
    SmallRec<A> *s_rec = reinterpret_cast<SmallRec<A> *>(sm + o);
    o += (size_t)p.Q * p.P * sizeof(SmallRec<A>);
    o = (o + 15) / 16 * 16;
    uint32_t *s_q = reinterpret_cast<uint32_t *>(sm + o);  // [kSmallBlock] row byte offsets
    CornerW<A> *s_w = reinterpret_cast<CornerW<A> *>(sm + o + (size_t)kSmallBlock * 4);  // [kSmallBlock] weights per block pixel
    o += (size_t)kSmallBlock * (4 + sizeof(CornerW<A>));
    o = (o + 15) / 16 * 16;
    uint16_t *s_order = reinterpret_cast<uint16_t *>(sm + o);  // [blocks of the level] heaviest first

    __shared__ int s_red[kSmallBlock / kWave];
    __shared__ int s_turn;

    for (int i = tid; i <= ncl; i += kSmallBlock) s_off[i] = 0;
    if (tid == 0) s_turn = 0;
    __syncthreads();

    auto walk = [&](auto &&visit) {  // visit(q, attention weight, in-level cell, dx, dy)
        auto one = [&](int q, const Pack<T, 2> &xy, T at) {
            int cell;
            uint32_t cellw;
            A dx, dy;
            if (sample_cell<A>(TR::to_acc(xy.v[0]), TR::to_acc(xy.v[1]), lh, lw, 0, 0, 0, p.zeros, p.align, cell, cellw, dx, dy) &&
                cell < ncl)
                visit(q, TR::to_acc(at), cell, dx, dy);
        };
        if (pre) {
#pragma unroll
            for (int k = 0; k < kPre; ++k)
                if (q0 + k * dq < p.Q) one(q0 + k * dq, pre_xy[k], pre_a[k]);
        } else if (active) {
            for (int q = q0; q < p.Q; q += dq) {
                const int sidx = q * HLP + lvl * p.P + pt;
                one(q, *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx), attn[sidx]);
            }
        } else if (p.P > kSmallBlock) {
            for (int q = 0; q < p.Q; ++q)
                for (int pp = tid; pp < p.P; pp += kSmallBlock) {
                    const int sidx = q * HLP + lvl * p.P + pp;
                    one(q, *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx), attn[sidx]);
                }
        }
    };
    // 1 count
    walk([&](int, A, int cell, A, A) { atomicAdd(&s_off[cell], 1); });
    __syncthreads();
    // 2 exclusive scan in place (each thread a contiguous run of cells)
    {
        const int per = (ncl + kSmallBlock - 1) / kSmallBlock;
        const int c_beg = min(ncl, tid * per), c_end = min(ncl, c_beg + per);
        int sum = 0;
        for (int c = c_beg; c < c_end; ++c) sum += s_off[c];
        const int lane = tid & (kWave - 1), wid = tid / kWave;
        int inc = sum;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int nn = __shfl_up(inc, d, kWave);
            if (lane >= d) inc += nn;
        }
        if (lane == kWave - 1) s_red[wid] = inc;
        __syncthreads();
        int base = inc - sum;
#pragma unroll
        for (int i = 0; i < kSmallBlock / kWave; ++i) {  // (fixed trip count: the 16 reads go out together)
            const int n = s_red[i];
            base += i < wid ? n : 0;
        }
        for (int c = c_beg; c < c_end; ++c) {
            const int n = s_off[c];
            s_off[c] = base;
            base += n;
        }
        if (tid == kSmallBlock - 1) s_off[ncl] = base;  // (the last thread's run ends at ncl, possibly empty)
        __syncthreads();
    }
    // 3 place: cursor = list start (restored afterwards by shifting: list c ends where list c + 1 began).
    // REPRODUCIBLE ORDER (as msda_value_place.hpp): the waves take their cursor atomics in turns — batch n of wave w
    // goes when the turn counter reads n * waves + w — so a list's order, and with it every sum of the gather below,
    // is the same in every run; the cells are computed before the turn, the records written after it.  Every wave
    // runs the same number of batches (inactive threads pass with nothing to place).  Cost: 16 hand-overs of ~140 ns
    // per batch, c4 48.1 -> 50.3 us, c1 22.5 -> 24.6 us (same-box A/B; spinning without s_sleep: the same).
    {
        constexpr int NW = kSmallBlock / kWave;
        const int wid = tid / kWave;
        auto batch = [&](int my, auto nconst, const int *qs, const Pack<T, 2> *xys, const T *ats) {
            constexpr int N = decltype(nconst)::value;
            int cell[N];
            A dx[N], dy[N];
            bool ok[N];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                uint32_t cellw;
                ok[k] = qs[k] >= 0 && sample_cell<A>(TR::to_acc(xys[k].v[0]), TR::to_acc(xys[k].v[1]), lh, lw, 0, 0, 0, p.zeros,
                                                     p.align, cell[k], cellw, dx[k], dy[k]) && cell[k] < ncl;
            }
            while (__hip_atomic_load(&s_turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != my) __builtin_amdgcn_s_sleep(1);
            int pos[N];
#pragma unroll
            for (int k = 0; k < N; ++k) pos[k] = ok[k] ? atomicAdd(&s_off[cell[k]], 1) : -1;
            // (acquire on the spin, release on the hand-over: the compiler may not move the cursor atomics across either —
                // ADVICE r04; the hardware runs a wave's DS operations in order anyway, so this costs one s_waitcnt lgkmcnt(0))
            if ((tid & (kWave - 1)) == 0) __hip_atomic_store(&s_turn, my + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                if (pos[k] < 0) continue;
                SmallRec<A> r;
                r.q = (uint32_t)qs[k];
                r.dx = dx[k];
                r.dy = dy[k];
                r.a = TR::to_acc(ats[k]);
                s_rec[pos[k]] = r;
            }
        };
        if (p.P <= kSmallBlock && (p.Q + dq - 1) / dq <= kPre) {  // (uniform) the prefetched samples: one batch
            int qs[kPre];
#pragma unroll
            for (int k = 0; k < kPre; ++k) qs[k] = pre && q0 + k * dq < p.Q ? q0 + k * dq : -1;
            batch(wid, std::integral_constant<int, kPre>{}, qs, pre_xy, pre_a);
        } else if (p.P <= kSmallBlock) {
            const int rounds = (p.Q + kPre * dq - 1) / (kPre * dq);
            for (int r = 0; r < rounds; ++r) {
                int qs[kPre];
                Pack<T, 2> xys[kPre];
                T ats[kPre];
#pragma unroll
                for (int k = 0; k < kPre; ++k) {
                    const int q = q0 + (r * kPre + k) * dq;
                    qs[k] = active && q < p.Q ? q : -1;
                    xys[k].v[0] = xys[k].v[1] = ats[k] = TR::from_acc((A)0);
                    if (qs[k] >= 0) {
                        const int sidx = q * HLP + lvl * p.P + pt;
                        xys[k] = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
                        ats[k] = attn[sidx];
                    }
                }
                batch(r * NW + wid, std::integral_constant<int, kPre>{}, qs, xys, ats);
            }
        } else {
            int it = 0;
            for (int q = 0; q < p.Q; ++q)
                for (int pp0 = 0; pp0 < p.P; pp0 += kSmallBlock, ++it) {
                    const int pp = pp0 + tid;
                    int qs[1] = {pp < p.P ? q : -1};
                    Pack<T, 2> xys[1];
                    T ats[1];
                    xys[0].v[0] = xys[0].v[1] = ats[0] = TR::from_acc((A)0);
                    if (qs[0] >= 0) {
                        const int sidx = q * HLP + lvl * p.P + pp;
                        xys[0] = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
                        ats[0] = attn[sidx];
                    }
                    batch(it * NW + wid, std::integral_constant<int, 1>{}, qs, xys, ats);
                }
        }
    }
    __syncthreads();
    // after placing, s_off[c] = END of list c (= start of list c + 1); list c = [c ? s_off[c-1] : 0, s_off[c])

    // ---- 4 gather: a lane group owns a 2 x 2 block of pixels (or 1/S of a busy block's records).  The block's
    // pixels are corners of the 3 x 3 cells around it, so each record's grad_out row is loaded ONCE per block (9 cell
    // lists for 4 pixels; a pixel alone would walk 4 lists) and FMAed into up to four accumulator rows. ----
    constexpr int SMAX = kWave / G;  // lane groups of one wave: their partial rows meet through shuffles
    const int unit = tid / G, j = tid % G;
    const int gbase = tid - j;
    const TG *gout = static_cast<const TG *>(p.grad_out) + ((size_t)b * p.Q * p.H + h) * p.D;
    const uint32_t q_stride = (uint32_t)(p.H * p.D) * (uint32_t)sizeof(TG);
    const rsrc_t rs_go = make_rsrc(gout, (uint32_t)(((size_t)p.Q * p.H * p.D - (size_t)h * p.D) * sizeof(TG)));
    const int nchan_chunks = (p.D + G * VEC - 1) / (G * VEC);
    const int nbx = (lw + 1) / 2, nby = (lh + 1) / 2;
    const int nblk = min(nbx * nby, p.small_cells / 4 + 2);  // (<= cells / 4; the cap only bites for shapes that disagree with I)
    // this workgroup's blocks: share, share + nshare, ...  (the order below is built with atomics, so it differs from
    // workgroup to workgroup: what a workgroup serves must not depend on it)
    const int nown = share < nblk ? (nblk - share + nshare - 1) / nshare : 0;
    // cumulative lengths of a block's nine lists: cell (2 bx - 1 + ci, 2 by - 1 + cj), list index ci + 3 cj
    const float inv_nbx = 1.0f / (float)max(nbx, 1);
    auto block_lists = [&](int blk, bool live, int(&lcum)[10]) {
        const int by = div_small(blk, max(nbx, 1), inv_nbx), bx = blk - by * nbx;  // (blk < 2^16)
        // cell (cx, cy) has id (cy + 1) * cw + cx + 1 and its list is [s_off[id - 1], s_off[id]): a row of three cells
        // needs four consecutive list ends.  All twelve are read unconditionally at clamped addresses (conditional reads
        // cost one LDS round trip each: 1.9 of a round's 6.4 thousand cycles on the 64 x 64 level of the c4 shape).
        int e[3][4];
#pragma unroll
        for (int cj = 0; cj < 3; ++cj) {
            const int id0 = (2 * by + cj) * cw + 2 * bx;  // id of the row's first cell (cx = 2 bx - 1)
#pragma unroll
            for (int t = 0; t < 4; ++t) e[cj][t] = s_off[min(max(id0 - 1 + t, 0), ncl)];
        }
        lcum[0] = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int ci = k % 3, cj = k / 3;
            const int cx = 2 * bx - 1 + ci, cy = 2 * by - 1 + cj;  // in [-1, lw - 1] x [-1, lh - 1] when valid
            const int cid = (cy + 1) * cw + (cx + 1);
            const bool ok = live && cx < lw && cy < lh && cid < ncl;
            lcum[k + 1] = lcum[k] + (ok ? e[cj][ci + 1] - (cid > 0 ? e[cj][ci] : 0) : 0);
        }
    };
    // The lane groups of a wave run as many batches as the busiest of them needs (the gather is bound by the
    // instructions it issues, not by the rows it waits for), and the records per block scatter widely (sparse level:
    // Poisson around 8 for batches of 8 — nearly every wave paid a second, almost empty batch).  So the blocks are
    // handed out in order of the batches they need, heaviest first: a counting sort over 16 classes.  Which block a
    // group serves changes; the order of the additions inside a pixel does not.
    // A busy block is split over S lane groups (S a power of two, the groups of one wave) so that no group walks more
    // than 64 records: S from the busiest BLOCK of the level (from the longest cell list times nine it was 2 for one
    // plane in 64 of the c4 shape — one cell with 8 samples — and that workgroup, with twice the rounds, set the
    // kernel's time: 78 against 49 thousand cycles).
    constexpr int kClasses = 16;
    __shared__ int s_cls[kClasses];
    __shared__ int s_maxtot;
    if (tid < kClasses) s_cls[tid] = 0;
    if (tid == 0) s_maxtot = 0;
    __syncthreads();
    {
        int mx = 0;
        for (int i = tid; i < nown; i += kSmallBlock) {
            int lcum[10];
            block_lists(share + i * nshare, true, lcum);
            mx = max(mx, lcum[9]);
        }
#pragma unroll
        for (int m = 1; m < kWave; m <<= 1) mx = max(mx, __shfl_xor(mx, m, kWave));
        if ((tid & (kWave - 1)) == 0 && mx > 0) atomicMax(&s_maxtot, mx);
    }
    __syncthreads();
    int lgS = 0;  // S = 1 << lgS
    while ((1 << lgS) < SMAX && ((s_maxtot + (1 << lgS) - 1) >> lgS) > 64) ++lgS;
    const int S = 1 << lgS;
    __shared__ int s_busy;  // blocks of this workgroup with at least one record: the gather's share of s_order
    {
        auto block_class = [&](int blk) {
            int lcum[10];
            block_lists(blk, true, lcum);
            const int per_group = (lcum[9] + S - 1) >> lgS;
            return kClasses - 1 - min(kClasses - 1, (per_group + G - 1) / G);  // 0: most batches
        };
        for (int i = tid; i < nown; i += kSmallBlock) atomicAdd(&s_cls[block_class(share + i * nshare)], 1);
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int c = 0; c < kClasses; ++c) {
                const int n = s_cls[c];
                s_cls[c] = run;
                run += n;
            }
            s_busy = s_cls[kClasses - 1];  // the last class: no batch at all, i.e. no record
        }
        __syncthreads();
        for (int i = tid; i < nown; i += kSmallBlock) {
            const int blk = share + i * nshare;
            s_order[atomicAdd(&s_cls[block_class(blk)], 1)] = (uint16_t)blk;
        }
        __syncthreads();
    }
    // Blocks nobody sampled (most of a fine level when the queries are few: a 100 x 134 level under 900 x 4 samples)
    // take no part in the gather rounds: their rows are zeros, stored here by all threads, 16 bytes at a time.
    const int nbusy = s_busy;
    {
        const int cpr = (p.D + VEC - 1) / VEC;  // stores per row
        const int total = (nown - nbusy) * 4 * cpr;  // (< 2^16 blocks, rows of at most a few hundred stores)
        Pack<TV, VEC> zero;
#pragma unroll
        for (int i = 0; i < VEC; ++i) zero.v[i] = TVR::from_acc((A)0);
        for (int e = tid; e < total; e += kSmallBlock) {
            const int row = (int)((unsigned)e / (unsigned)cpr), c0 = (e - row * cpr) * VEC;
            const int blk = (int)s_order[nbusy + (row >> 2)], k = row & 3;
            const int by = div_small(blk, max(nbx, 1), inv_nbx), bx = blk - by * nbx;
            const int px = 2 * bx + (k & 1), py = 2 * by + (k >> 1);
            if (px < lw && py < lh && pstart + py * lw + px < p.I && c0 < p.D)
                store_stream(static_cast<TV *>(p.grad_value) + (((size_t)b * p.I + pstart + py * lw + px) * p.H + h) * p.D + c0, zero);
        }
    }
    const int items = nbusy * S;
    const int rounds = (items + NG - 1) / NG;
    for (int r = 0; r < rounds; ++r) {
        const int item = r * NG + unit;
        const bool live = item < items;
        const int blk = live ? (int)s_order[item >> lgS] : 0, part = live ? item & (S - 1) : 0;
        const int by = div_small(blk, max(nbx, 1), inv_nbx), bx = blk - by * nbx;
        int lcum[10];
        block_lists(blk, live, lcum);
        const int ntot = lcum[9];
        const int mine = ntot > part ? (ntot - part + S - 1) >> lgS : 0;  // virtual positions part, part + S, ...
        for (int cc = 0; cc < nchan_chunks; ++cc) {
            const int c0 = (cc * G + j) * VEC;
            const bool lane_ok = c0 < p.D;
            const uint32_t lane_elem = (lane_ok ? (uint32_t)c0 : 0u) * (uint32_t)sizeof(TG);
            A acc[4][VEC];  // block pixel i + 2 j'
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[k][i] = (A)0;
            for (int v0 = 0; v0 < mine; v0 += G) {
                // convert: lane j takes this group's record v0 + j -> (row offset, one weight per block pixel)
                {
                    const int v = v0 + j;
                    const bool ok = v < mine;
                    const int vp = part + (v << lgS);
                    int k = 0;
#pragma unroll
                    for (int t = 1; t < 9; ++t) k += (vp >= lcum[t]) ? 1 : 0;
                    int cum = 0;
#pragma unroll
                    for (int t = 1; t < 9; ++t) cum = k == t ? lcum[t] : cum;
                    const int cj = k / 3, ci = k - 3 * cj;
                    const int cid = (2 * by + cj) * cw + 2 * bx + ci;  // the list's cell; its start is looked up again here
                    SmallRec<A> rec;                                   // (nine starts in registers made the kernel spill)
                    rec.q = 0;
                    rec.dx = rec.dy = rec.a = (A)0;
                    if (ok) rec = s_rec[(cid > 0 ? s_off[cid - 1] : 0) + vp - cum];
                    // weight of the record for block column i: the cell's x0 = 2 bx - 1 + ci, the pixel is 2 bx + i
                    const A wx0 = ci == 0 ? rec.dx : ci == 1 ? (A)1 - rec.dx : (A)0;
                    const A wx1 = ci == 1 ? rec.dx : ci == 2 ? (A)1 - rec.dx : (A)0;
                    const A wy0 = cj == 0 ? rec.dy : cj == 1 ? (A)1 - rec.dy : (A)0;
                    const A wy1 = cj == 1 ? rec.dy : cj == 2 ? (A)1 - rec.dy : (A)0;
                    const A ay0 = rec.a * wy0, ay1 = rec.a * wy1;
                    CornerW<A> w;
                    w.w[0] = ay0 * wx0;
                    w.w[1] = ay0 * wx1;
                    w.w[2] = ay1 * wx0;
                    w.w[3] = ay1 * wx1;
                    wave_lds_sync();  // the previous batch's hand-off has been read
                    s_q[tid] = ok ? mul24(rec.q, q_stride) : 0x80000000u;
                    s_w[tid] = w;
                    wave_lds_sync();
                }
#pragma unroll
                for (int jj = 0; jj < G; jj += UB) {
                    if (v0 + jj < mine) {  // uniform per group
                        Pack<TG, VEC> g[UB];
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) g[u].v[i] = Traits<TG>::from_acc((A)0);
                            if (v0 + jj + u < mine)  // (uniform inside the group) no record, no load: the workgroup's one CU
                                                     // is bound by its vector-memory path
                                g[u] = __builtin_bit_cast(Pack<TG, VEC>,
                                                          RawLoad<sizeof(TG) * VEC>::load(rs_go, s_q[gbase + jj + u] + lane_elem));
                        }
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            const CornerW<A> w = s_w[gbase + jj + u];
#pragma unroll
                            for (int k = 0; k < 4; ++k)
#pragma unroll
                                for (int i = 0; i < VEC; ++i) acc[k][i] = fma_t(w.w[k], Traits<TG>::to_acc(g[u].v[i]), acc[k][i]);
                        }
                    }
                }
            }
            // a split block's S partial row sets sit in S neighbouring lane groups of one wave: butterfly sum
            for (int m = G; m < G * S; m <<= 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[k][i] += __shfl_xor(acc[k][i], m, kWave);
            }
            if (live && part == 0 && lane_ok) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int px = 2 * bx + (k & 1), py = 2 * by + (k >> 1);
                    if (px < lw && py < lh) {
                        const int pix = py * lw + px;
                        Pack<TV, VEC> ov;
#pragma unroll
                        for (int i = 0; i < VEC; ++i) ov.v[i] = TVR::from_acc(acc[k][i]);
                        TV *dst = static_cast<TV *>(p.grad_value) + (((size_t)b * p.I + pstart + pix) * p.H + h) * p.D + c0;
                        if (pstart + pix < p.I) store_stream(dst, ov);  // (shapes that disagree with I: stay inside the plane)
                    }
                }
            }
        }
    }
    // pixels of `I` behind the last level (shapes that describe fewer than I pixels) belong to nobody: zeros
    if (lvl == p.L - 1 && share == 0) {
        const int tail0 = pstart + npix;
        const int row_elems = p.D;
        for (long long e = (long long)tail0 * row_elems + tid; e < (long long)p.I * row_elems; e += kSmallBlock) {
            const int px = (int)(e / row_elems), c = (int)(e - (long long)px * row_elems);
            static_cast<TV *>(p.grad_value)[(((size_t)b * p.I + px) * p.H + h) * p.D + c] = TVR::from_acc((A)0);
        }
    }
}

}  // namespace msda
