// msda_value_binned.hpp — grad_value as a two-level sort + gather: no atomics on floating-point data, no global atomics.
//
// The reference scatter-adds four corner tiles per sample with global atomics (kernels.py:543-553).  Here the
// scatter is inverted: samples are sorted by the bilinear *cell* they fall in, and pixels pull their rows.
//
//   A cell of level l is the unit square whose top-left corner is pixel (x0, y0), x0 in [-1, W-1], y0 in [-1, H-1];
//   it is addressed by c' = (x0 + 1, y0 + 1) in [0, W] x [0, H].  A sample lives in exactly one cell and touches only
//   that cell's (up to) four corner pixels; pixel (x, y) is touched by the cells c' in {x, x+1} x {y, y+1}.
//
//   The pyramid is cut into TILES of 8 x 8 pixels.  A tile's pixels draw on the 9 x 9 cells c' in
//   [8 tx, 8 tx + 8] x [8 ty, 8 ty + 8]: its own 8 x 8 cells plus the first column / row / corner of its right /
//   lower / lower-right neighbours.  So the cells of a tile are split into four CLASSES — interior, top row, corner,
//   left column — and (tile, class) is the coarse sort key ("bin"): a tile reads its own four bins and the edge bins
//   of three neighbours, and every grad_value row it owns comes out complete (no partial rows, no seams).
//
//   K1 bin     (plane, query slice): LDS histogram of the slice's samples over the plane's bins (ds_add_u32),
//              exclusive scan in LDS, offsets out (toff[plane][slice][bin]), then the samples again -> 16-byte records
//              {q, cell inside the tile, dx, dy, a} at the bin's cursor inside the SLICE'S OWN record range.  A
//              workgroup writes a few hundred densely packed runs, so the partial lines merge in L2 (the round-2
//              pipeline scattered every record into one global cell-sorted array: 2x write amplification).
//   K2 tile    (plane, tile, slice group): gathers the tile's runs (coalesced), 1024 records at a time: LDS counting
//              sort by cell (81 counters), records converted once to {grad_out row offset, four corner weights}.
//              CELL-OWNER accumulation: every lane group owns <= 3 of the 81 cells for the whole tile and keeps their
//              four corner rows in registers; it walks its cells' lists — ONE grad_out row load per record, 8 in flight
//              — with no flush, no carry, no hand-off.  At the end the corner rows meet in a 64-pixel LDS image in
//              four conflict-free phases (corner k of every cell hits a different pixel) and the tile's rows are
//              stored once, complete.
//              Dense (coarse) levels would put tens of thousands of records on one tile: there the tile is served by G
//              workgroups, each taking every G-th slice and leaving a partial tile; K3 sums those.
//   K3 reduce  pixels of split levels: sum of the G partial rows in a fixed order; pixels behind the last level: zeros.
//
// Every grad_value row is written exactly once by plain stores (no memset).  Padding semantics as in
// msda_value_sorted.hpp: "zeros" drops samples / corners outside the image, "border" clips the pixel coordinate first.
#pragma once

#include "msda_value_sorted.hpp"

#ifndef MSDA_K2_WAVES
#define MSDA_K2_WAVES 4
#endif

namespace msda {

constexpr int kTileShift = 3;
constexpr int kTile = 1 << kTileShift;                 // pixels per tile edge
constexpr int kTileCellsX = kTile + 1;                 // cells a tile draws on, per axis
constexpr int kTileCells = kTileCellsX * kTileCellsX;  // 81
constexpr int kBinBlock = 1024;                        // threads of K1
constexpr int kTileBlock = 256;                        // threads of K2 / K3
constexpr int kTileChunk = 1024;                       // records K2 sorts in LDS at a time
constexpr int kMaxSlices = 64;                         // query slices per plane (K2's run table: 4 runs per slice)
constexpr int kBinLdsInts = 36864;                     // bins a K1 workgroup keeps in LDS at a time (144 KiB)
constexpr int kSplitRecords = 6144;                    // a tile expected to hold more records than this is split over slice groups

// record of the binned path.  float accumulate type: 16 bytes, exact fractions; double: 32 bytes.
// qc: query (24 bits) | cell x inside the tile << 24 | cell y inside the tile << 28
template <typename A> struct BinRec;
template <> struct alignas(16) BinRec<float> {
    uint32_t qc;
    float dx, dy, a;
};
template <> struct alignas(16) BinRec<double> {
    uint32_t qc, pad;
    double dx, dy, a;
};

// per-level tile geometry, derived from the level table by every workgroup (the shapes live on the device)
struct TileTab {
    int nbx[kMaxLevels];    // bin-tile columns: w / 8 + 1 (cells c'x in [0, w])
    int ntx[kMaxLevels];    // pixel-tile columns: ceil(w / 8)
    int nty[kMaxLevels];
    int bin0[kMaxLevels];   // first bin of the level (4 bins per bin tile)
    int grp[kMaxLevels];    // slice groups per tile (a power of two; 1: tiles store final rows)
    int item0[kMaxLevels];  // first work item of the level; levels in REVERSE order (coarse levels first)
    int part0[kMaxLevels];  // first partial row of the level (grp > 1)
    int nbins, nitems, nparts, npix;
};

// qp: samples per level and plane in this round (queries x points).  Thread t < L fills level t.
__device__ __forceinline__ void load_tile_table(TileTab *tt, const LevelTab *tab, int L, long long qp, int nsplit, int part_cap)
{
    const int t = threadIdx.x;
    if (t < L) {
        int bin0 = 0, part0 = 0, npix = 0;
        int my_grp = 1, my_part0 = 0, my_bin0 = 0;
        for (int l = 0; l < L; ++l) {
            const int h = max(tab->h[l], 0), w = max(tab->w[l], 0);
            const long long px = (long long)h * w;
            // expected records of a tile = qp * 64 / px; split while it exceeds kSplitRecords per group (no division:
            // a 64-bit divide is several hundred instructions, and every workgroup of three kernels runs this)
            int grp = 1;
            while (grp < nsplit && qp * (kTile * kTile) > (long long)kSplitRecords * grp * px) grp <<= 1;
            // never past the workspace (the host sized it from I alone; only shapes that disagree with I get here)
            if (grp > 1 && (long long)part0 + (long long)grp * px > (long long)part_cap) grp = 1;
            if (l == t) {
                my_grp = grp;
                my_part0 = part0;
                my_bin0 = bin0;
            }
            bin0 += ((w >> kTileShift) + 1) * ((h >> kTileShift) + 1) * 4;
            if (grp > 1) part0 += grp * (int)px;
            npix += (int)px;
        }
        const int h = max(tab->h[t], 0), w = max(tab->w[t], 0);
        tt->nbx[t] = (w >> kTileShift) + 1;
        tt->ntx[t] = (w + kTile - 1) >> kTileShift;
        tt->nty[t] = (h + kTile - 1) >> kTileShift;
        tt->bin0[t] = my_bin0;
        tt->grp[t] = my_grp;
        tt->part0[t] = my_part0;
        if (t == 0) {
            tt->nbins = bin0;
            tt->nparts = part0;
            tt->npix = npix;
        }
    }
}
// the work items of the tile kernel, once every level's group count is in the table (a barrier after load_tile_table)
__device__ __forceinline__ void finish_tile_table(TileTab *tt, int L)
{
    const int t = threadIdx.x;
    if (t < L) {
        int item0 = 0;
        for (int l = L - 1; l > t; --l) item0 += tt->ntx[l] * tt->nty[l] * tt->grp[l];
        tt->item0[t] = item0;
        if (t == 0) tt->nitems = item0 + tt->ntx[0] * tt->nty[0] * tt->grp[0];
    }
}

// sample -> cell address c' = (x0 + 1, y0 + 1) and the fractional offsets.  false: the sample touches no pixel.
template <typename A>
__device__ __forceinline__ bool sample_cxy(A x, A y, int h, int w, bool zeros, bool align, int &cx, int &cy, A &dx, A &dy)
{
    const A W = (A)w, Hh = (A)h;
    A px, py;
    if (align) {
        px = x * (W - (A)1);
        py = y * (Hh - (A)1);
    } else {
        px = x * W - (A)0.5;
        py = y * Hh - (A)0.5;
    }
    A x0, y0;
    if (zeros) {
        x0 = floor_t(px);
        y0 = floor_t(py);
        if (!(x0 >= (A)-1 && x0 <= W - (A)1 && y0 >= (A)-1 && y0 <= Hh - (A)1)) return false;  // also NaN
    } else {
        px = fmin_t(fmax_t(px, (A)0), W - (A)1);
        py = fmin_t(fmax_t(py, (A)0), Hh - (A)1);
        x0 = floor_t(px);
        y0 = floor_t(py);
    }
    dx = px - x0;
    dy = py - y0;
    cx = (int)x0 + 1;
    cy = (int)y0 + 1;
    return true;
}

// ------------------------------------------------------------------------------------------
// K1: count + scan + place of one (plane, query slice), all in one workgroup.
// A thread keeps ONE (level, point) slot for its whole walk (the active threads are a multiple of L*P), so the
// level's constants sit in registers.
// ------------------------------------------------------------------------------------------
template <typename T, typename TV = T> __global__ __launch_bounds__(kBinBlock) void msda_bin_pass_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    int pair, slice;
    if (!decode_block(p.grid3d, p.B * p.H, p.nsplit, p.xcd_map, pair, slice)) return;
    const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
    const int nq = p.q_end - p.q_begin;
    const int qper = (nq + p.nsplit - 1) / p.nsplit;
    const int qa = min(p.q_end, p.q_begin + slice * qper), qb = min(p.q_end, qa + qper);

    LevelTab *tab = reinterpret_cast<LevelTab *>(msda_smem);
    TileTab *tt = reinterpret_cast<TileTab *>(msda_smem + sizeof(LevelTab));
    int *s_bin = reinterpret_cast<int *>(msda_smem + sizeof(LevelTab) + sizeof(TileTab));
    __shared__ int s_red[kBinBlock / kWave];
    __shared__ int s_base;
    load_level_table(tab, p.shapes, p.L);
    __syncthreads();
    load_tile_table(tt, tab, p.L, (long long)nq * p.P, p.nsplit, p.part_cap);
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const int nbins = min(tt->nbins, p.nbin_cap);
    const int cap = p.bin_lds_cap;
    // pixels of `I` behind the last level (shapes that describe fewer than I pixels) belong to nobody: zeros, once
    if (slice == 0 && p.finish_mode <= 1 && tt->npix < p.I) {
        const long long e0 = (long long)max(tt->npix, 0) * p.D, e1 = (long long)p.I * p.D;
        for (long long e = e0 + threadIdx.x; e < e1; e += kBinBlock) {
            const int px = (int)(e / p.D), c = (int)(e - (long long)px * p.D);
            static_cast<TV *>(p.grad_value)[(((size_t)b * p.I + px) * p.H + h) * p.D + c] = Traits<TV>::from_acc((A)0);
        }
    }
    int *toff = p.ws_toff + ((size_t)pair * p.nsplit + slice) * ((size_t)p.nbin_cap + 1);
    BinRec<A> *recs = static_cast<BinRec<A> *>(p.ws_entries) + (size_t)pair * p.ent_cap + (size_t)slice * qper * p.LP;
    const size_t plane_s0 = ((size_t)b * p.Q * p.H + h) * p.LP;
    const T *loc = static_cast<const T *>(p.loc) + 2 * plane_s0;
    const T *attn = static_cast<const T *>(p.attn) + plane_s0;
    const int HLP = p.H * p.LP;
    const int tid = threadIdx.x;
    const bool fixed = p.LP <= kBinBlock;
    const int dq = fixed ? kBinBlock / p.LP : 1;
    const int sl0 = fixed ? tid % p.LP : 0, tq = fixed ? tid / p.LP : 0;
    const bool active = fixed && tq < dq;

    // sample -> bin of the trip [b0, b0 + n) (or -1), its cell inside the tile and the fractions
    auto bin_of = [&](int lw, int lh, int nbx, int bin0, int b0, int n, A x, A y, uint32_t &cellbits, A &dx, A &dy) -> int {
        int cx, cy;
        if (!sample_cxy<A>(x, y, lh, lw, p.zeros, p.align, cx, cy, dx, dy)) return -1;
        const int lx = cx & (kTile - 1), ly = cy & (kTile - 1);
        const int cls = lx ? (ly ? 0 : 1) : (ly ? 3 : 2);  // interior, top row, corner, left column
        const int bin = bin0 + ((imul24(cy >> kTileShift, nbx) + (cx >> kTileShift)) << 2) + cls - b0;
        cellbits = ((uint32_t)lx << 24) | ((uint32_t)ly << 28);
        return (unsigned)bin < (unsigned)n ? bin : -1;
    };
    auto put = [&](int q, int pos, uint32_t cb, A dx, A dy, A at) {
        BinRec<A> r;
        r.qc = (uint32_t)q | cb;
        r.dx = dx;
        r.dy = dy;
        r.a = at;
        recs[pos] = r;  // plain store: the slice's runs are dense, the partial lines merge in L2
    };
    // exclusive scan of s_bin[0, n) in place, shifted by `base0`; returns (to every thread) base0 + the total
    auto scan_bins = [&](int n, int base0) -> int {
        const int per = (n + kBinBlock - 1) / kBinBlock;
        const int c_beg = min(n, tid * per), c_end = min(n, c_beg + per);
        int sum = 0;
        for (int c = c_beg; c < c_end; ++c) sum += s_bin[c];
        const int lane = tid & (kWave - 1), wid = tid / kWave;
        int inc = sum;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int nn = __shfl_up(inc, d, kWave);
            if (lane >= d) inc += nn;
        }
        if (lane == kWave - 1) s_red[wid] = inc;
        __syncthreads();
        int base = base0 + inc - sum, total = base0;
#pragma unroll
        for (int i = 0; i < kBinBlock / kWave; ++i) {
            const int r = s_red[i];
            if (i < wid) base += r;
            total += r;
        }
        for (int c = c_beg; c < c_end; ++c) {
            const int v = s_bin[c];
            s_bin[c] = base;
            base += v;
        }
        __syncthreads();
        return total;
    };

    // ---- fast path: the slice's samples fit the registers of the workgroup (kCache per thread) and its bins fit LDS:
    // every (x, y, a) is read ONCE, all loads in flight together; count, scan and place run from registers ----
    constexpr int kCache = sizeof(A) == 8 ? 8 : 16;
    if (p.bin_cached && fixed && nbins <= cap) {
        for (int i = tid; i < nbins; i += kBinBlock) s_bin[i] = 0;
        Pack<T, 2> xy[kCache];
        T at[kCache];
        const int l = sl0 / p.P;
        const int lw = tab->w[l], lh = tab->h[l], nbx = tt->nbx[l], bin0 = tt->bin0[l];
        const int q0 = qa + tq;
        const int sidx0 = q0 * HLP + sl0, d_sidx = dq * HLP;
#pragma unroll
        for (int k = 0; k < kCache; ++k) {
            xy[k].v[0] = xy[k].v[1] = at[k] = TR::from_acc((A)0);
            if (active && q0 + k * dq < qb) {
                xy[k] = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * (sidx0 + k * d_sidx));
                at[k] = attn[sidx0 + k * d_sidx];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kCache; ++k) {
            if (active && q0 + k * dq < qb) {
                uint32_t cb;
                A dx, dy;
                const int bin = bin_of(lw, lh, nbx, bin0, 0, nbins, TR::to_acc(xy[k].v[0]), TR::to_acc(xy[k].v[1]), cb, dx, dy);
                if (bin >= 0 && !(p.debug & 128)) atomicAdd(&s_bin[bin], 1);
            }
        }
        __syncthreads();
        const int total = scan_bins(nbins, 0);
        for (int i = tid; i < nbins; i += kBinBlock) toff[i] = s_bin[i];
        if (tid == 0) toff[nbins] = total;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kCache; ++k) {
            if (active && q0 + k * dq < qb) {
                uint32_t cb;
                A dx, dy;
                const int bin = bin_of(lw, lh, nbx, bin0, 0, nbins, TR::to_acc(xy[k].v[0]), TR::to_acc(xy[k].v[1]), cb, dx, dy);
                if (bin >= 0) {
                    const int pos = (p.debug & 32) ? min(s_bin[bin] + (tid & 7), qper * p.LP - 1) : atomicAdd(&s_bin[bin], 1);
                    if (!(p.debug & 64)) put(q0 + k * dq, pos, cb, dx, dy, TR::to_acc(at[k]));
                }
            }
        }
        return;
    }

    // ---- general path: two walks over the samples per trip ----
    for (int b0 = 0; b0 == 0 || b0 < nbins; b0 += cap) {  // one trip unless the plane has more bins than fit in LDS
        const int n = max(0, min(cap, nbins - b0));
        for (int i = tid; i < n; i += kBinBlock) s_bin[i] = 0;
        __syncthreads();
        // ---- count ----
        if (active) {
            const int l = sl0 / p.P;
            const int lw = tab->w[l], lh = tab->h[l], nbx = tt->nbx[l], bin0 = tt->bin0[l];
            int q = qa + tq;
            int sidx = q * HLP + sl0;
            const int d_sidx = dq * HLP;
            // four samples in flight per thread (static ring): the walk is latency-bound, nothing is stored here
            constexpr int RING = 4;
            Pack<T, 2> ring[RING];
#pragma unroll
            for (int k = 0; k < RING; ++k) {
                ring[k].v[0] = ring[k].v[1] = TR::from_acc((A)0);
                if (q + k * dq < qb) ring[k] = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * (sidx + k * d_sidx));
            }
            while (q < qb) {
#pragma unroll
                for (int k = 0; k < RING; ++k) {
                    const Pack<T, 2> xy = ring[k];
                    const int qk = q + k * dq;
                    if (qk + RING * dq < qb)
                        ring[k] = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * (sidx + (k + RING) * d_sidx));
                    uint32_t cb;
                    A dx, dy;
                    if (qk < qb) {
                        const int bin = bin_of(lw, lh, nbx, bin0, b0, n, TR::to_acc(xy.v[0]), TR::to_acc(xy.v[1]), cb, dx, dy);
                        if (bin >= 0) atomicAdd(&s_bin[bin], 1);
                    }
                }
                q += RING * dq;
                sidx += RING * d_sidx;
            }
        } else if (!fixed) {
            for (int q = qa; q < qb; ++q)
                for (int sl = tid; sl < p.LP; sl += kBinBlock) {
                    const int l = sl / p.P;
                    const int sidx = q * HLP + sl;
                    const Pack<T, 2> xy = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
                    uint32_t cb;
                    A dx, dy;
                    const int bin = bin_of(tab->w[l], tab->h[l], tt->nbx[l], tt->bin0[l], b0, n, TR::to_acc(xy.v[0]),
                                           TR::to_acc(xy.v[1]), cb, dx, dy);
                    if (bin >= 0) atomicAdd(&s_bin[bin], 1);
                }
        }
        __syncthreads();
        // ---- exclusive scan in place; offsets out ----
        const int total = scan_bins(n, s_base);
        for (int i = tid; i < n; i += kBinBlock) toff[b0 + i] = s_bin[i];
        // ---- place ----
        if (active) {
            const int l = sl0 / p.P;
            const int lw = tab->w[l], lh = tab->h[l], nbx = tt->nbx[l], bin0 = tt->bin0[l];
            int q = qa + tq;
            int sidx = q * HLP + sl0;
            const int d_sidx = dq * HLP;
            // four samples in flight per thread here too
            constexpr int RING = 4;
            Pack<T, 2> ring[RING];
            T ring_a[RING];
#pragma unroll
            for (int k = 0; k < RING; ++k) {
                ring[k].v[0] = ring[k].v[1] = ring_a[k] = TR::from_acc((A)0);
                if (q + k * dq < qb) {
                    ring[k] = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * (sidx + k * d_sidx));
                    ring_a[k] = attn[sidx + k * d_sidx];
                }
            }
            while (q < qb) {
#pragma unroll
                for (int k = 0; k < RING; ++k) {
                    const Pack<T, 2> xy = ring[k];
                    const T at = ring_a[k];
                    const int qk = q + k * dq;
                    if (qk + RING * dq < qb) {
                        ring[k] = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * (sidx + (k + RING) * d_sidx));
                        ring_a[k] = attn[sidx + (k + RING) * d_sidx];
                    }
                    uint32_t cb;
                    A dx, dy;
                    if (qk < qb) {
                        const int bin = bin_of(lw, lh, nbx, bin0, b0, n, TR::to_acc(xy.v[0]), TR::to_acc(xy.v[1]), cb, dx, dy);
                        if (bin >= 0) put(qk, atomicAdd(&s_bin[bin], 1), cb, dx, dy, TR::to_acc(at));
                    }
                }
                q += RING * dq;
                sidx += RING * d_sidx;
            }
        } else if (!fixed) {
            for (int q = qa; q < qb; ++q)
                for (int sl = tid; sl < p.LP; sl += kBinBlock) {
                    const int l = sl / p.P;
                    const int sidx = q * HLP + sl;
                    const Pack<T, 2> xy = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
                    uint32_t cb;
                    A dx, dy;
                    const int bin = bin_of(tab->w[l], tab->h[l], tt->nbx[l], tt->bin0[l], b0, n, TR::to_acc(xy.v[0]),
                                           TR::to_acc(xy.v[1]), cb, dx, dy);
                    if (bin >= 0) put(q, atomicAdd(&s_bin[bin], 1), cb, dx, dy, TR::to_acc(attn[sidx]));
                }
        }
        __syncthreads();
        if (tid == 0) s_base = total;
        __syncthreads();
    }
    if (tid == 0) toff[nbins] = s_base;
}

// ------------------------------------------------------------------------------------------
// one finished grad_value row piece -> memory (rounds over the queries: running sums in the accumulate type)
// ------------------------------------------------------------------------------------------
template <typename T, int VEC, typename TV>
__device__ __forceinline__ void emit_value_row(const Params &p, int pair, int b, int h, int pix, int c0,
                                               typename Traits<T>::acc (&acc)[VEC])
{
    using A = typename Traits<T>::acc;
    using TVR = Traits<TV>;
    if (p.finish_mode != 0) {
        A *run = static_cast<A *>(p.ws_accum) + ((size_t)pair * p.I + pix) * p.D + c0;
        if (p.finish_mode != 1) {
            const Pack<A, VEC> prev = *reinterpret_cast<const Pack<A, VEC> *>(run);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] += prev.v[v];
        }
        if (p.finish_mode != 3) {
            Pack<A, VEC> keep;
#pragma unroll
            for (int v = 0; v < VEC; ++v) keep.v[v] = acc[v];
            *reinterpret_cast<Pack<A, VEC> *>(run) = keep;
            return;
        }
    }
    Pack<TV, VEC> o;
#pragma unroll
    for (int v = 0; v < VEC; ++v) o.v[v] = TVR::from_acc(acc[v]);
    TV *dst = static_cast<TV *>(p.grad_value) + (((size_t)b * p.I + pix) * p.H + h) * p.D + c0;
    store_stream(dst, o);
}

// ------------------------------------------------------------------------------------------
// K2: persistent workgroups of 512 threads = 64 lane groups of G lanes (VEC channels per lane).  Lane group u owns the
// INTERIOR cell u of the tile in hand — its four corner rows stay in registers for the whole item — and lane groups
// 0..16 also serve the 17 HALO cells (first column / row / corner of the neighbours), whose rows are summed through a
// small LDS buffer chunk by chunk.  So a thread carries 4 x VEC accumulators, two workgroups fit a CU and their phases
// overlap.  A workgroup belongs to one XCD (blockIdx & 7 under round-robin dispatch — speed only) and walks that XCD's
// planes one after another, so a plane's grad_out rows are pulled into exactly one L2.  The chain  offsets -> run table
// -> records -> sort -> gather  is four dependent memory round trips per item, so it is software-pipelined across
// chunks and items: while a chunk is gathered, the next records (of this item, or the first of the next) are in flight.
//
// Per chunk of 1024 records: histogram over the 81 cells with RETURNING LDS atomics (the rank of a record inside its
// cell) | barrier | every wave scans the 81 counts for itself (DPP, no serial section), records go to base + rank
// (no second atomic), lists padded to multiples of 4 with masked records | barrier | gather.
// ------------------------------------------------------------------------------------------
constexpr int kGatherBlock = 512;                      // threads of K2
constexpr int kGatherChunk = 2 * kGatherBlock;         // records sorted in LDS at a time
constexpr int kHaloCells = 2 * kTile + 1;              // 17
constexpr int kHaloParts = 3;                          // lane groups sharing a halo cell's list (3 * 17 = 51 of the 64)

// inclusive scan over the 64 lanes of a wave (DPP: four shifts inside the rows of 16, two row broadcasts)
__device__ __forceinline__ int wave_scan_incl(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
    return v;
}

template <typename T, int VEC, int G, typename TV = T, bool STAMPS = false>
__global__ __launch_bounds__(kGatherBlock) __attribute__((amdgpu_waves_per_eu(MSDA_K2_WAVES))) void msda_tile_gather_kernel(const Params p)
{
    // dev-only phase clock (STAMPS): s_memtime deltas per phase, summed per workgroup, written to p.dbg_out
    uint64_t st_prev = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (STAMPS) st_prev = __builtin_amdgcn_s_memtime();
    auto stamp = [&](int k) {
        if constexpr (STAMPS) {
            const uint64_t now = __builtin_amdgcn_s_memtime();
            st_acc[k] += now - st_prev;
            st_prev = now;
        }
    };
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    constexpr int NG = kGatherBlock / G;               // lane groups
    constexpr int NW = kGatherBlock / kWave;           // waves
    constexpr int UB = 8;                              // row loads in flight per lane
    constexpr int PADTO = 4;                           // lists are padded to a multiple of this (16-byte reads of the offsets)
    constexpr int RPT = kGatherChunk / kGatherBlock;   // records a thread sorts per chunk
    constexpr int RS = G * VEC;                        // accumulators per row piece
    constexpr int kRuns = 4 * kMaxSlices;              // run table capacity (a power of two)
    constexpr int kSorted = kGatherChunk + kTileCells * (PADTO - 1) + UB;  // sorted records + per-cell padding
    static_assert(NG == kTile * kTile, "one lane group per interior cell and per pixel");
    static_assert(kRuns <= kGatherBlock, "one thread per run");
    static_assert(kTile * kTile * RS * sizeof(A) <= kSorted * sizeof(CornerW<A>), "pixel image fits the weight array");

    const int tid = threadIdx.x;
    const int wave = tid / kWave, lane = tid % kWave, j = tid % G;
    const int unit = tid / G;  // lane group: interior cell (unit & 7, unit >> 3), pixel unit, halo cell `unit` if < 17
    // cell ids in the 9 x 9 numbering
    const int my_cell = (unit >> kTileShift) * kTileCellsX + (unit & (kTile - 1));
    // halo duty: lane group u < 51 walks part u / 17 of the list of halo cell u % 17
    const int hcell = unit % kHaloCells, hpart = unit / kHaloCells;
    const bool halo_duty = unit < kHaloCells * kHaloParts;
    const int my_halo = hcell < kTile ? hcell * kTileCellsX + kTile                  // right column, rows 0..7
                        : hcell < 2 * kTile ? kTile * kTileCellsX + (hcell - kTile)  // bottom row, columns 0..7
                                            : kTileCells - 1;                        // the corner

    __shared__ LevelTab tab;
    __shared__ TileTab tt;
    __shared__ int s_cnt[2][kTileCells + 3];     // histogram, double-buffered over the chunks
    __shared__ int s_run_start[2][kRuns];        // first record of a run (index into the plane's record range)
    __shared__ int s_run_pref[2][kRuns + 1];     // records before the run; behind the last run: INT_MAX
    __shared__ int s_total[2];
    __shared__ int s_red[NW];
    __shared__ __attribute__((aligned(16))) uint32_t s_q[kSorted];    // grad_out row byte offset (padding: masked)
    __shared__ __attribute__((aligned(16))) CornerW<A> s_w[kSorted];  // corner weights (padding: zeros)
    __shared__ __attribute__((aligned(16))) A s_halo[kHaloParts * kHaloCells * 4 * RS];  // corner rows of the halo cells, per part

    load_level_table(&tab, p.shapes, p.L);
    for (int i = tid; i < 2 * (kTileCells + 3); i += kGatherBlock) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const int nq = p.q_end - p.q_begin;
    load_tile_table(&tt, &tab, p.L, (long long)nq * p.P, p.nsplit, p.part_cap);
    __syncthreads();
    finish_tile_table(&tt, p.L);
    __syncthreads();
    if (p.debug & 16) return;
    stamp(0);  // tables
    const int nitems = __builtin_amdgcn_readfirstlane(tt.nitems);
    const int nbins = __builtin_amdgcn_readfirstlane(min(tt.nbins, p.nbin_cap));
    const int qper = (nq + p.nsplit - 1) / p.nsplit;
    const uint32_t q_stride = (uint32_t)(p.H * p.D) * (uint32_t)sizeof(T);

    // ---- this workgroup's sequence of work: the planes of its XCD one after another, inside a plane the items (and
    // channel chunks) seq, seq + W, ... ----
    const int npairs = p.B * p.H;
    const int nx = p.xcd_map ? 8 : 1;
    const int xcd = (int)blockIdx.x % nx, wj = (int)blockIdx.x / nx, W = (int)gridDim.x / nx;
    const int npl = xcd < npairs ? (npairs - xcd + nx - 1) / nx : 0;   // planes xcd, xcd + nx, ...
    const int per_plane = nitems * p.ncc;
    const long long nseq = (long long)npl * per_plane;

    // work item -> (plane, level, tile, slice group, channel chunk)
    struct Item {
        int pair, cc, l, grp, g, tx, ty, nruns;
    };
    auto decode_item = [&](long long seq) -> Item {
        Item I;
        const int m = (int)(seq / per_plane), r = (int)(seq - (long long)m * per_plane);
        I.pair = xcd + m * nx;
        const int it = r / p.ncc;
        I.cc = r - it * p.ncc;
        int l = p.L - 1;
        while (l > 0 && it >= tt.item0[l] + tt.ntx[l] * tt.nty[l] * tt.grp[l]) --l;  // levels in reverse order
        I.l = l;
        I.grp = tt.grp[l];
        const int rem = it - tt.item0[l];
        I.g = rem & (I.grp - 1);
        const int tile = rem >> (__ffs(I.grp) - 1);
        const int ntx = max(tt.ntx[l], 1);
        I.ty = tile / ntx;
        I.tx = tile - I.ty * ntx;
        I.nruns = ((p.nsplit - I.g + I.grp - 1) / I.grp) * 4;  // slices g, g + grp, ...: four runs each
        // (everything here is uniform, but it comes out of LDS: tell the compiler, or it all lives in vector registers)
        I.pair = __builtin_amdgcn_readfirstlane(I.pair);
        I.cc = __builtin_amdgcn_readfirstlane(I.cc);
        I.l = __builtin_amdgcn_readfirstlane(I.l);
        I.grp = __builtin_amdgcn_readfirstlane(I.grp);
        I.g = __builtin_amdgcn_readfirstlane(I.g);
        I.tx = __builtin_amdgcn_readfirstlane(I.tx);
        I.ty = __builtin_amdgcn_readfirstlane(I.ty);
        I.nruns = __builtin_amdgcn_readfirstlane(I.nruns);
        return I;
    };
    // an item's runs, one per thread: per slice the own tile (4 bins), the right neighbour's corner + left column, the
    // lower neighbour's top row + corner, the lower-right neighbour's corner.  Two offsets per run (global loads).
    auto issue_offsets = [&](const Item &I, int &a0, int &a1) {
        a0 = a1 = 0;
        if (tid < I.nruns) {
            const int l = I.l;
            const int nbx = __builtin_amdgcn_readfirstlane(tt.nbx[l]);
            const int nby = __builtin_amdgcn_readfirstlane((max(tab.h[l], 0) >> kTileShift) + 1);
            const int s = I.g + (tid >> 2) * I.grp, kind = tid & 3;
            const int bt = I.ty * nbx + I.tx;
            int lo = 0, hi = 0;
            if (kind == 0) {
                lo = 4 * bt;
                hi = lo + 4;
            } else if (kind == 1 && I.tx + 1 < nbx) {
                lo = 4 * (bt + 1) + 2;
                hi = lo + 2;
            } else if (kind == 2 && I.ty + 1 < nby) {
                lo = 4 * (bt + nbx) + 1;
                hi = lo + 2;
            } else if (kind == 3 && I.tx + 1 < nbx && I.ty + 1 < nby) {
                lo = 4 * (bt + nbx + 1) + 2;
                hi = lo + 1;
            }
            const int bin0 = __builtin_amdgcn_readfirstlane(tt.bin0[l]);
            lo = min(bin0 + lo, nbins);
            hi = min(bin0 + hi, nbins);
            const int *to = p.ws_toff + ((size_t)I.pair * p.nsplit + s) * ((size_t)p.nbin_cap + 1);
            a0 = to[lo];
            a1 = to[hi];
        }
    };
    // offsets -> run table `buf` (exclusive scan of the run lengths over the workgroup; two barriers)
    auto build_table = [&](int buf, const Item &I, int a0, int a1) {
        const int len = tid < I.nruns ? max(a1 - a0, 0) : 0;
        const int inc = wave_scan_incl(len);
        if (lane == kWave - 1) s_red[wave] = inc;
        __syncthreads();
        int pre = inc - len;
        for (int i = 0; i < wave; ++i) pre += s_red[i];
        if (tid < kRuns) {
            const int s = I.g + (tid >> 2) * I.grp;
            s_run_start[buf][tid] = s * qper * p.LP + a0;
            s_run_pref[buf][tid] = tid < I.nruns ? pre : 0x7fffffff;
            if (tid == I.nruns - 1) s_total[buf] = pre + len;
            if (tid == kRuns - 1) s_run_pref[buf][kRuns] = 0x7fffffff;
        }
        __syncthreads();
    };
    // records ch0 + tid + k * block of the item -> registers (kind: which neighbour a record came from; -1: none).
    // Branch-free search of the run table, the thread's records side by side.
    auto fetch_chunk = [&](int buf, const Item &I, int ch0, int n, BinRec<A>(&rec)[RPT], int(&lc)[RPT]) {
        const BinRec<A> *entries = static_cast<const BinRec<A> *>(p.ws_entries) + (size_t)I.pair * p.ent_cap;
        int lo[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) lo[k] = 0;
        int top = kRuns / 2;
        while (top >= I.nruns && top > 1) top >>= 1;  // (uniform) largest power of two below the run count
        for (int half = top; half >= 1; half >>= 1) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int i = ch0 + tid + k * kGatherBlock;
                if (s_run_pref[buf][lo[k] + half] <= i) lo[k] += half;
            }
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int i = ch0 + tid + k * kGatherBlock;
            lc[k] = -1;
            if (i < n) {
                lc[k] = lo[k] & 3;
                rec[k] = entries[s_run_start[buf][lo[k]] + (i - s_run_pref[buf][lo[k]])];
            }
        }
    };

    // The loop below serves item `cur` and, in the middle of cur's last chunk, sets up the next one (offsets requested
    // at the start of that chunk, run table + first records behind its place step).  It is entered with a DUMMY item
    // without records, whose only effect is that set-up for the workgroup's first real item: one copy of every helper.
    long long seq = (long long)wj - W;
    if (wj >= nseq) return;
    bool cur_valid = false;
    Item cur = decode_item(wj), nxt = cur;
    bool have_next = true;
    int a0 = 0, a1 = 0;
    issue_offsets(nxt, a0, a1);  // the first real item's
    int buf = 0, chunk = 0;
    int n = 0;
    BinRec<A> rec[RPT];
    int lc[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) lc[k] = -1;
    stamp(1);

    while (true) {
        // per item: plane bases
        const int b = (int)fast_div((uint32_t)cur.pair, p.div_h), h = cur.pair - b * p.H;
        const T *gout = static_cast<const T *>(p.grad_out) + ((size_t)b * p.Q * p.H + h) * p.D;  // uniform base
        const rsrc_t rs_go = make_rsrc(gout, (uint32_t)(((size_t)p.Q * p.H * p.D - (size_t)h * p.D) * sizeof(T)));
        const int c0 = (cur.cc * G + j) * VEC;
        const bool lane_ok = c0 < p.D;
        const uint32_t lane_elem = lane_ok ? (uint32_t)c0 * (uint32_t)sizeof(T) : kMaskedOffset;
        A acc[4][VEC];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[c][v] = (A)0;
        if (halo_duty) {  // this item's halo sums start at zero (the group that adds to them clears them)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                Pack<A, VEC> z;
#pragma unroll
                for (int v = 0; v < VEC; ++v) z.v[v] = (A)0;
                *reinterpret_cast<Pack<A, VEC> *>(&s_halo[(unit * 4 + c) * RS + j * VEC]) = z;
            }
        }
        int n_next = 0;
        for (int ch0 = 0;; ch0 += kGatherChunk, ++chunk) {
            const bool last = ch0 + kGatherChunk >= n;
            int *cnt = s_cnt[chunk & 1];
            // ---- histogram over the 81 cells; the atomic returns the record's rank inside its cell ----
            int rank[RPT];
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                rank[k] = 0;
                if (lc[k] >= 0) {
                    const int kind = lc[k];
                    const int lx = (kind & 1) ? kTile : (int)((rec[k].qc >> 24) & 15u);
                    const int ly = (kind & 2) ? kTile : (int)(rec[k].qc >> 28);
                    lc[k] = min(ly, kTile) * kTileCellsX + min(lx, kTile);
                    rank[k] = atomicAdd(&cnt[lc[k]], 1);
                }
            }
            __syncthreads();
            stamp(2);  // records arrive + histogram
            // ---- every wave scans the padded list lengths for itself: lane L holds cells L and 64 + L ----
            const int v0 = cnt[lane], v1 = lane < kTileCells - kWave ? cnt[kWave + lane] : 0;
            const int p0 = (v0 + PADTO - 1) & ~(PADTO - 1), p1 = (v1 + PADTO - 1) & ~(PADTO - 1);
            const int i0 = wave_scan_incl(p0), i1 = wave_scan_incl(p1);
            const int tot0 = __builtin_amdgcn_readlane(i0, kWave - 1);
            const int beg0 = i0 - p0, beg1 = tot0 + i1 - p1;  // padded list starts of cells L and 64 + L
            auto list_of = [&](int c, int &beg, int &lenp) {   // any cell's list, from the lanes that hold it
                const int src = c & (kWave - 1);
                const int ba = __shfl(beg0, src, kWave), bb = __shfl(beg1, src, kWave);
                const int la = __shfl(p0, src, kWave), lb = __shfl(p1, src, kWave);
                beg = c < kWave ? ba : bb;
                lenp = c < kWave ? la : lb;
            };
            // the padding behind the lists: masked records (wave L % 8 serves cell L, wave (64 + L) % 8 cell 64 + L)
            {
                CornerW<A> z;
                z.w[0] = z.w[1] = z.w[2] = z.w[3] = (A)0;
                if ((lane & (NW - 1)) == wave)
                    for (int e = v0; e < p0; ++e) {
                        s_q[beg0 + e] = kMaskedOffset;
                        s_w[beg0 + e] = z;
                    }
                if (lane < kTileCells - kWave && ((kWave + lane) & (NW - 1)) == wave)
                    for (int e = v1; e < p1; ++e) {
                        s_q[beg1 + e] = kMaskedOffset;
                        s_w[beg1 + e] = z;
                    }
                if (wave == 1) {  // and the other histogram for the next chunk
                    int *other = s_cnt[(chunk + 1) & 1];
                    other[lane] = 0;
                    if (lane < kTileCells - kWave) other[kWave + lane] = 0;
                }
            }
            // ---- place: converted once per record, at list start + rank ----
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int c = max(lc[k], 0);
                int cb, cl;
                list_of(c, cb, cl);
                if (lc[k] >= 0) {
                    const int pos = cb + rank[k];
                    const A a = rec[k].a, dx = rec[k].dx, dy = rec[k].dy;
                    const A ax1 = a * dx, ax0 = a - ax1;
                    CornerW<A> cw;
                    cw.w[3] = ax1 * dy;
                    cw.w[2] = ax0 * dy;
                    cw.w[1] = ax1 - cw.w[3];
                    cw.w[0] = ax0 - cw.w[2];
                    s_q[pos] = mul24(rec[k].qc & 0xFFFFFFu, q_stride);  // both < 2^24 (host check)
                    s_w[pos] = cw;
                }
            }
            // the lists this lane group will walk
            int mbeg, mlen, hbeg, hlen;
            list_of(my_cell, mbeg, mlen);
            list_of(my_halo, hbeg, hlen);
            {   // this group's part of the halo list, in batches of PADTO records
                const int nb = hlen / PADTO;
                const int b_lo = nb * hpart / kHaloParts, b_hi = nb * (hpart + 1) / kHaloParts;
                hbeg += b_lo * PADTO;
                hlen = halo_duty ? (b_hi - b_lo) * PADTO : 0;
            }
            stamp(3);  // scan + place
            // ---- what travels while this chunk is gathered: the item's next chunk, or (behind its last chunk) the
            // next item's first records (its run table is built here from the offsets requested above) ----
            {
                const bool nx2 = last && have_next;
                if (nx2) {
                    build_table(buf ^ 1, nxt, a0, a1);
                    n_next = __builtin_amdgcn_readfirstlane(s_total[buf ^ 1]);
                }
                if (!last || nx2)
                    fetch_chunk(nx2 ? buf ^ 1 : buf, nx2 ? nxt : cur, nx2 ? 0 : ch0 + kGatherChunk, nx2 ? n_next : n, rec, lc);
                else {
#pragma unroll
                    for (int k = 0; k < RPT; ++k) lc[k] = -1;
                }
                // the offsets of the item after the next one travel for a whole item
                if (nx2 && seq + 2 * (long long)W < nseq) issue_offsets(decode_item(seq + 2 * (long long)W), a0, a1);
            }
            __syncthreads();
            stamp(4);  // issue of what travels
            // ---- gather: one (padded) list, UB rows in flight; FMA into a[4][VEC] ----
            auto walk = [&](int beg, int lenp, A(&a4)[4][VEC]) {
                for (int i = 0; i < lenp; i += UB) {
                    // lists are padded to multiples of 4 with masked records; the second half of a batch is only taken
                    // when it still belongs to this list
                    const uint4 qa = *reinterpret_cast<const uint4 *>(&s_q[beg + i]);
                    const bool two = i + 4 < lenp;  // uniform inside the group
                    uint4 qb = make_uint4(kMaskedOffset, kMaskedOffset, kMaskedOffset, kMaskedOffset);
                    if (two) qb = *reinterpret_cast<const uint4 *>(&s_q[beg + i + 4]);
                    const uint32_t qo[UB] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
                    Pack<T, VEC> gr[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u)
                        gr[u] = __builtin_bit_cast(Pack<T, VEC>, RawLoad<sizeof(T) * VEC>::load(rs_go, qo[u] + lane_elem));
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        if (u < 4 || two) {
                            const CornerW<A> w = s_w[beg + i + u];
#pragma unroll
                            for (int cn = 0; cn < 4; ++cn)
#pragma unroll
                                for (int vv = 0; vv < VEC; ++vv)
                                    a4[cn][vv] = fma_t(w.w[cn], TR::to_acc(gr[u].v[vv]), a4[cn][vv]);
                        }
                    }
                }
            };
            if (!(p.debug & 2)) {
                walk(mbeg, mlen, acc);
                if (hlen > 0) {  // a halo cell's rows of this chunk: into the LDS sums (this group's alone)
                    A ha[4][VEC];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) ha[c][v] = (A)0;
                    walk(hbeg, hlen, ha);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        Pack<A, VEC> *row = reinterpret_cast<Pack<A, VEC> *>(&s_halo[(unit * 4 + c) * RS + j * VEC]);
                        Pack<A, VEC> o = *row;
#pragma unroll
                        for (int v = 0; v < VEC; ++v) o.v[v] += ha[c][v];
                        *row = o;
                    }
                }
            }
            stamp(5);  // gather
            if (last) {
                ++chunk;
                break;
            }
            // (no barrier here: the next chunk's histogram uses the other counter array, and nothing of this chunk
            //  is overwritten before the barrier behind that histogram)
        }

        if (cur_valid) {  // (uniform)
            // ---- the corner rows meet in the tile's pixel image: corner k of every cell is a different pixel, so each
            // of the four phases is conflict-free.  Corner 3 (cell (lx, ly) -> pixel (lx, ly)) covers all 64 pixels: it
            // goes first and stores, the others add. ----
            __syncthreads();  // every gather has read its weights: the array becomes the image
            A *img = reinterpret_cast<A *>(&s_w[0]);
            {
                const int cxl = unit & (kTile - 1), cyl = unit >> kTileShift;           // interior cell
                const int hyl = my_halo / kTileCellsX, hxl = my_halo - hyl * kTileCellsX;  // halo cell
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) {
                    const int cn = 3 - ph;
                    {
                        const int ux = cxl - 1 + (cn & 1), uy = cyl - 1 + (cn >> 1);
                        if ((unsigned)ux < (unsigned)kTile && (unsigned)uy < (unsigned)kTile) {
                            Pack<A, VEC> *row = reinterpret_cast<Pack<A, VEC> *>(img + (uy * kTile + ux) * RS + j * VEC);
                            Pack<A, VEC> o;
                            if (ph == 0) {
#pragma unroll
                                for (int vv = 0; vv < VEC; ++vv) o.v[vv] = acc[cn][vv];
                            } else {
                                o = *row;
#pragma unroll
                                for (int vv = 0; vv < VEC; ++vv) o.v[vv] += acc[cn][vv];
                            }
                            *row = o;
                        }
                    }
                    if (ph > 0 && unit < kHaloCells) {  // (a halo cell has no corner 3 inside the tile)
                        const int ux = hxl - 1 + (cn & 1), uy = hyl - 1 + (cn >> 1);
                        if ((unsigned)ux < (unsigned)kTile && (unsigned)uy < (unsigned)kTile) {
                            Pack<A, VEC> *row = reinterpret_cast<Pack<A, VEC> *>(img + (uy * kTile + ux) * RS + j * VEC);
                            Pack<A, VEC> o = *row;
#pragma unroll
                            for (int pt = 0; pt < kHaloParts; ++pt) {  // fixed order
                                const Pack<A, VEC> hv = *reinterpret_cast<const Pack<A, VEC> *>(
                                    &s_halo[((pt * kHaloCells + unit) * 4 + cn) * RS + j * VEC]);
#pragma unroll
                                for (int vv = 0; vv < VEC; ++vv) o.v[vv] += hv.v[vv];
                            }
                            *row = o;
                        }
                    }
                    __syncthreads();
                }
            }
            stamp(6);  // assembly
            // ---- rows out: final (grad_value) or this group's partial tile; lane group = pixel ----
            const int l = cur.l;
            const int lw = __builtin_amdgcn_readfirstlane(tab.w[l]), lh = __builtin_amdgcn_readfirstlane(tab.h[l]);
            {
                const int pi = unit;
                const int ux = pi & (kTile - 1), uy = pi >> kTileShift;
                const int x = cur.tx * kTile + ux, y = cur.ty * kTile + uy;
                if (x < lw && y < lh && lane_ok) {
                    const int rel = y * lw + x;
                    const int pix = tab.start[l] + rel;
                    if (pix < p.I) {
                        const Pack<A, VEC> r = *reinterpret_cast<const Pack<A, VEC> *>(img + pi * RS + j * VEC);
                        if (cur.grp == 1) {
                            A o[VEC];
#pragma unroll
                            for (int vv = 0; vv < VEC; ++vv) o[vv] = r.v[vv];
                            emit_value_row<T, VEC, TV>(p, cur.pair, b, h, pix, c0, o);
                        } else {
                            A *dst = static_cast<A *>(p.ws_ptile) +
                                     ((size_t)cur.pair * p.part_cap + tt.part0[l] + (size_t)cur.g * ((size_t)lw * lh) + rel) * p.D + c0;
                            store_stream(dst, r);
                        }
                    }
                }
            }
            stamp(7);  // rows out
        }
        if (!have_next) break;
        // (the image is not overwritten before the barrier behind the next histogram)
        seq += W;
        cur = nxt;
        cur_valid = true;
        buf ^= 1;
        n = n_next;
        have_next = seq + W < nseq;
        if (have_next) nxt = decode_item(seq + W);
    }
    if constexpr (STAMPS) {
        if (tid == 0 && p.dbg_out != nullptr && blockIdx.x < 1024)
            for (int k = 0; k < 8; ++k) p.dbg_out[blockIdx.x * 8 + k] = st_acc[k];
    }
}

// ------------------------------------------------------------------------------------------
// K3: pixels of split levels (sum of the slice groups' partial rows, fixed order) and pixels behind the last level
// (zeros).  64 pixels per workgroup, G lanes per row piece.
// ------------------------------------------------------------------------------------------
template <typename T, int VEC, int G, typename TV = T>
__global__ __launch_bounds__(kTileBlock) void msda_tile_reduce_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    constexpr int NG = kTileBlock / G;
    constexpr int kPix = 64;
    // slot -> (level, chunk of 64 pixels): a split level has fewer than qp * 64 / kSplitRecords pixels (load_tile_table),
    // so p.reduce_chunks chunks per level cover it; unsplit levels leave at once
    int pair, slot;
    if (!decode_block(p.grid3d, p.B * p.H, p.L * p.reduce_chunks, p.xcd_map, pair, slot)) return;
    __shared__ LevelTab tab;
    __shared__ TileTab tt;
    load_level_table(&tab, p.shapes, p.L);
    __syncthreads();
    load_tile_table(&tt, &tab, p.L, (long long)(p.q_end - p.q_begin) * p.P, p.nsplit, p.part_cap);
    __syncthreads();
    const int l = slot / p.reduce_chunks, chunk = slot - l * p.reduce_chunks;
    const int grp = tt.grp[l];
    if (grp <= 1) return;  // (uniform) the tile kernel stored this level's final rows
    const int lw = max(tab.w[l], 0), lh = max(tab.h[l], 0);
    const int npx = lw * lh;
    const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
    const int tid = threadIdx.x;
    const int unit = tid / G, j = tid % G;
    const int nchan_chunks = (p.D + G * VEC - 1) / (G * VEC);
    const size_t gstep = (size_t)npx * p.D;
    for (int pi = unit; pi < kPix; pi += NG) {
        const int rel = chunk * kPix + pi;
        const int pix = tab.start[l] + rel;
        if (rel >= npx || pix >= p.I) break;
        for (int ccx = 0; ccx < nchan_chunks; ++ccx) {
            const int c0 = (ccx * G + j) * VEC;
            if (c0 >= p.D) continue;
            A acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = (A)0;
            const A *src = static_cast<const A *>(p.ws_ptile) + ((size_t)pair * p.part_cap + tt.part0[l] + rel) * p.D + c0;
            for (int g0 = 0; g0 < grp; g0 += 8) {  // eight partial rows in flight; summed in a fixed order
                Pack<A, VEC> r[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) r[u].v[v] = (A)0;
                    if (g0 + u < grp) r[u] = *reinterpret_cast<const Pack<A, VEC> *>(src + (size_t)(g0 + u) * gstep);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] += r[u].v[v];
            }
            emit_value_row<T, VEC, TV>(p, pair, b, h, pix, c0, acc);
        }
    }
}

// ------------------------------------------------------------------------------------------
// workspace layout (host + device agree through these helpers)
// ------------------------------------------------------------------------------------------
struct BinnedWsLayout {
    int nbin_cap, nsplit, part_cap, tile_slots;
    int q_round, rounds;
    size_t off_toff, off_entries, off_part, off_accum, off_dbg, total;
};

int option_q_round();
int option_cell_slices();

inline BinnedWsLayout binned_ws_layout(int64_t B, int64_t I, int64_t H, int64_t D, int64_t Q, int64_t L, int64_t P,
                                       size_t acc_bytes, bool need_accum)
{
    BinnedWsLayout w;
    const size_t pairs = (size_t)(B * H);
    const size_t entry_bytes = acc_bytes == 8 ? 32 : 16;
    // rounds over the queries bound the record buffer (~1 GiB at a time)
    int64_t q_round = Q;
    {
        const int64_t per_query = (int64_t)pairs * L * P * (int64_t)entry_bytes;
        const int64_t budget = (int64_t)1 << 30;
        if (per_query > 0 && Q * per_query > budget + budget / 4) {
            const int64_t rounds = (Q * per_query + budget - 1) / budget;
            q_round = (Q + rounds - 1) / rounds;
        }
    }
    if (option_q_round() > 0) q_round = option_q_round();
    if (q_round < 1) q_round = 1;
    if (q_round > Q) q_round = Q > 0 ? Q : 1;
    w.q_round = (int)q_round;
    w.rounds = (int)((Q + q_round - 1) / q_round);
    if (w.rounds < 1) w.rounds = 1;
    const int64_t samples = q_round * L * P;  // per plane and round
    // query slices per plane (a power of two): two 1024-thread workgroups per CU, ~8-16k samples each
    int64_t want = pairs ? (int64_t)((512 + pairs - 1) / pairs) : 1;
    if (samples / 16384 > want) want = samples / 16384;
    if (want > samples / 1024) want = samples / 1024;
    if (option_cell_slices() > 0) want = option_cell_slices();
    int ns = 1;
    while (ns < want && ns < kMaxSlices) ns <<= 1;
    while (ns > 1 && ns > q_round) ns >>= 1;
    w.nsplit = ns;
    // bins: 4 per bin tile, sum over the levels of (w/8 + 1)(h/8 + 1) <= I/64 + (I + L)/8 + L  (w + h <= w h + 1)
    w.nbin_cap = (int)(4 * (I / 64 + (I + L) / 8 + L) + 4);
    // partial rows of split levels: a level is split only while its tiles expect more than kSplitRecords records
    // (so it has fewer than Q P 64 / kSplitRecords pixels) and into fewer than 2 x expected / kSplitRecords groups
    w.part_cap = (int)(3 * L * (q_round * P * 64 / kSplitRecords + 1) + 64);
    // workgroup slots of the tile kernel per plane: about one per work item (tiles + the groups of split tiles); a
    // workgroup loops when the shapes hold more items than this estimate
    int64_t slots = I / 64 + 3 * L + 2 * samples / kSplitRecords + 1;
    if (slots > 32768) slots = 32768;
    w.tile_slots = (int)slots;
    size_t o = 0;
    w.off_toff = o;    o = align_up(o + pairs * (size_t)w.nsplit * ((size_t)w.nbin_cap + 1) * 4, 256);
    w.off_entries = o; o = align_up(o + pairs * (size_t)w.nsplit * (size_t)((q_round + w.nsplit - 1) / w.nsplit) * (size_t)(L * P) * entry_bytes, 256);
    w.off_part = o;    o = align_up(o + pairs * (size_t)w.part_cap * (size_t)D * acc_bytes, 256);
    w.off_accum = o;   if (w.rounds > 1 && need_accum) o = align_up(o + pairs * (size_t)I * (size_t)D * acc_bytes, 256);
    w.off_dbg = o;     o = align_up(o + 65536, 256);  // dev-only phase clock (the last 64 KiB)
    w.total = o;
    return w;
}

}  // namespace msda
