// msda_value_sorted.hpp — grad_value as a GATHER: no atomics on floating-point data, no global atomics.
//
// The reference scatter-adds four corner tiles per sample with global atomics (kernels.py:543-553).
// Here the scatter is inverted once per call:
//
//   a bilinear *cell* of level l is the unit square whose top-left corner is pixel (x0, y0),
//   x0 in [-1, W-1], y0 in [-1, H-1]  ->  (W+1)*(H+1) cells per level, cell id = cstart_l + (y0+1)*(W+1) + (x0+1).
//   A sample lives in exactly one cell and touches only that cell's (up to) four corner pixels.
//
//   K1 count    (plane, query slice j): histogram of the slice's samples over the plane's cells, kept in LDS
//               (ds_add_u32), written out as part[plane][j][cell]; per block of 256 cells the slice's total
//   K2 scan     per block of 256 cells, ONE launch: records per cell (sum over the slices; part[j][cell] becomes
//               slice j's first slot inside the cell's list), base = the block totals before it, exclusive scan:
//               off[cell] = first record of the cell's list
//   K3 place    (plane, j): cursor[cell] = off[cell] + part[j][cell] in LDS; every sample ->
//               entries[cursor[cell]++] = {q, cell, a, dx, dy} (16 bytes, packed)
//   K4 gather   the sorted record array is cut into WINDOWS of 64 records; one G-lane group walks one window, G
//               records at a time: each lane fetches one record and turns it into (grad_out row offset, four corner
//               weights, segment flags); the group issues the G grad_out row loads back to back — ONE load per
//               sample — and FMAs each row into four corner accumulators.  A *segment* is a run of records of one
//               cell inside one window.  When a segment ends its corner rows go to the pixels' slots
//               scratch[pixel][corner] (streaming stores); if the next cell is the x-neighbour, the right-hand
//               corners are carried over in registers instead (they are the neighbour's left-hand corners).
//               Segments that continue a cell from the previous window are summed through LDS into the segment
//               they continue (same workgroup) or leave as one "continuation" row set per workgroup.
//   K5 finish   per pixel: its four slots (written by the cells (x,y), (x-1,y), (x,y-1), (x-1,y-1)) are 4 contiguous
//               rows; which of them were written follows from off[] alone; plus the continuation row sets of cells
//               longer than a workgroup's windows.  Sum in a fixed order, store the grad_value row.
//
// Every grad_value row is written exactly once by plain stores (no memset); the work per lane group is 64
// records whatever the sampling distribution.  Padding semantics: "zeros" drops samples/corners outside the
// image; "border" clips the pixel coordinate to [0, size-1] first (grid_sample), which puts the whole weight on
// the edge pixel exactly as the reference's clamped corners do.
#pragma once

#include "msda_kernels.hpp"

namespace msda {

constexpr int kWinMax = 64;           // records per gather window: at most (chosen per call, sorted_ws_layout)
constexpr int kWinMin = 16;           // ... and at least
constexpr int kGatherMinBlock = 256;  // threads per gather workgroup (sizes the continuation rows)
constexpr int kCellBlock = 1024;      // threads of K1 / K3
constexpr int kScanCells = 256;       // cells per K2 workgroup (= its thread count)
constexpr int kCellLdsInts = 36864;   // cells a K1 / K3 workgroup keeps in LDS at a time (144 KiB)
// record cell word: where the cell's rows go, decoded once by the place pass (the gather needs no division):
//   bits 0-22  index of the cell's corner-00 pixel (x0, y0) inside the plane + kPixBias (virtual — possibly
//              negative before the bias — when x0 or y0 is -1);  23-27 level;  28-31 which corners are pixels of the image
// Two records belong to the same cell iff their words are equal (cells that alias one virtual index differ in the
// validity bits).
constexpr uint32_t kPixBias = 1u << 22;
constexpr uint32_t kPixMask = 0x7FFFFFu;  // 23 bits: biased index < 2^22 + I, I < 2^22 (host check)
constexpr int kLevelShift = 23;
constexpr int kSortedMaxLevels = 32;  // 5 level bits = MSDA_MAX_LEVELS (round 3: 4 bits; L = 17..32 had no grad_value route beyond the single-launch kernel)

// ------------------------------------------------------------------------------------------
// sorted sample records.  float accumulate type: 16 bytes, the fractional offsets carried with 20 fractional bits
// (the fp32 pixel coordinate they come from has fewer at every level wider than 8 px); double: 32 bytes, exact.
// ------------------------------------------------------------------------------------------
template <typename A> struct Entry;
template <> struct alignas(16) Entry<float> {
    uint32_t w0, w1, w2, w3;  // q | dxq[7:0] << 24,  cell word (kPixBias),  bits(a),  dxq[19:8] | dyq << 12
    static __device__ __forceinline__ Entry pack(uint32_t q, uint32_t cellflag, float a, float dx, float dy)
    {
        const uint32_t dxq = min((uint32_t)(dx * 1048576.0f + 0.5f), 0xFFFFFu);
        const uint32_t dyq = min((uint32_t)(dy * 1048576.0f + 0.5f), 0xFFFFFu);
        Entry e;
        e.w0 = q | (dxq << 24);
        e.w1 = cellflag;
        e.w2 = __builtin_bit_cast(uint32_t, a);
        e.w3 = (dxq >> 8) | (dyq << 12);
        return e;
    }
    __device__ __forceinline__ uint32_t q() const { return w0 & 0xFFFFFFu; }
    __device__ __forceinline__ uint32_t cellflag() const { return w1; }
    __device__ __forceinline__ float a() const { return __builtin_bit_cast(float, w2); }
    __device__ __forceinline__ float dx() const { return (float)((w0 >> 24) | ((w3 & 0xFFFu) << 8)) * (1.0f / 1048576.0f); }
    __device__ __forceinline__ float dy() const { return (float)(w3 >> 12) * (1.0f / 1048576.0f); }
};
template <> struct alignas(16) Entry<double> {
    uint32_t qq, cf;
    double aa, ddx, ddy;
    static __device__ __forceinline__ Entry pack(uint32_t q, uint32_t cellflag, double a, double dx, double dy)
    {
        Entry e;
        e.qq = q;
        e.cf = cellflag;
        e.aa = a;
        e.ddx = dx;
        e.ddy = dy;
        return e;
    }
    __device__ __forceinline__ uint32_t q() const { return qq; }
    __device__ __forceinline__ uint32_t cellflag() const { return cf; }
    __device__ __forceinline__ double a() const { return aa; }
    __device__ __forceinline__ double dx() const { return ddx; }
    __device__ __forceinline__ double dy() const { return ddy; }
};

// the record array of plane `pair` (uniform): the caller's gradient buffers first (grad_loc, grad_attn, grad_value),
// then the workspace
template <typename A> __device__ __forceinline__ Entry<A> *plane_entries(const Params &p, int pair)
{
    if (pair < p.ent_n0) return static_cast<Entry<A> *>(p.ent_alt0) + (size_t)pair * p.ent_cap;
    pair -= p.ent_n0;
    if (pair < p.ent_n1) return static_cast<Entry<A> *>(p.ent_alt1) + (size_t)pair * p.ent_cap;
    pair -= p.ent_n1;
    if (pair < p.ent_n2) return static_cast<Entry<A> *>(p.ent_alt2) + (size_t)pair * p.ent_cap;
    return static_cast<Entry<A> *>(p.ws_entries) + (size_t)(pair - p.ent_n2) * p.ent_cap;
}

__device__ __forceinline__ int plane_cells(const LevelTab &tab, int L)
{
    return tab.cstart[L - 1] + (tab.h[L - 1] + 1) * (tab.w[L - 1] + 1);
}

// sample -> (cell id inside the plane, record cell word, fractional offsets).  false: the sample touches no pixel.
template <typename A>
__device__ __forceinline__ bool sample_cell(A x, A y, int h, int w, int cstart, int start, int level, bool zeros,
                                            bool align, int &cell, uint32_t &cellw, A &dx, A &dy)
{
    const A W = (A)w, Hh = (A)h;
    A px, py;
    if (align) {
        px = x * (W - (A)1);
        py = y * (Hh - (A)1);
    } else {
        // An EXPLICIT fused multiply-add: every copy of this function — the count pass, the place pass before and
        // after its turn, the single-launch kernel's two walks — must put a sample into the same cell, and under
        // -ffp-contract=fast the compiler fuses `x * W - 0.5` in one inlined copy and not in another (a packed multiply
        // + add before the turn, v_pk_fma behind it): a coordinate within an ulp of a cell boundary then landed in one
        // cell's list carrying the other cell's word (found by tools/fuzz_parity.py, seed 1000117).
        px = fma_t(x, W, (A)-0.5);
        py = fma_t(y, Hh, (A)-0.5);
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
    const int ix = (int)x0, iy = (int)y0;
    cell = cstart + (int)mul24((uint32_t)(iy + 1), (uint32_t)(w + 1)) + (ix + 1);
    const bool xv0 = ix >= 0, xv1 = ix + 1 < w, yv0 = iy >= 0, yv1 = iy + 1 < h;
    const uint32_t valid = (uint32_t)(xv0 && yv0) | ((uint32_t)(xv1 && yv0) << 1) | ((uint32_t)(xv0 && yv1) << 2) |
                           ((uint32_t)(xv1 && yv1) << 3);
    // start + iy * w + ix, biased: (iy + 1) * w is non-negative, so the 24-bit multiply applies
    const uint32_t pixq = (uint32_t)((int)kPixBias + start + (int)mul24((uint32_t)(iy + 1), (uint32_t)w) - w + ix);
    cellw = (pixq & kPixMask) | ((uint32_t)level << kLevelShift) | (valid << 28);
    return true;
}

// ------------------------------------------------------------------------------------------
// K1 / K3: one pass over the samples of a (plane, query slice).  PLACE=false counts, true places.
// A thread keeps ONE (level, point) slot for its whole walk (the active threads are a multiple of L*P), so the
// level's constants sit in registers and the loop body is the coordinate math, one LDS atomic and (K3) one store.
// ------------------------------------------------------------------------------------------
template <typename T, bool PLACE>
__global__ __launch_bounds__(kCellBlock) void msda_cell_pass_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    int pair, slice;
    if (!decode_block(p.grid3d, p.B * p.H, p.nsplit, p.xcd_map, pair, slice)) return;
    const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
    const int qper = (p.q_end - p.q_begin + p.nsplit - 1) / p.nsplit;  // this round's queries, cut into the slices
    const int qa = min(p.q_end, p.q_begin + slice * qper), qb = min(p.q_end, qa + qper);

    LevelTab *tab = reinterpret_cast<LevelTab *>(msda_smem);
    int *s_cell = reinterpret_cast<int *>(msda_smem + sizeof(LevelTab));
    load_level_table(tab, p.shapes, p.L);
    __syncthreads();
    // the workspace was sized on the host from I alone (nc_cap = 2 I + 2 L); shapes that disagree with I (or
    // zero-sized levels) must not push the cell tables past it: cells beyond the capacity are dropped
    const int ncells = min(plane_cells(*tab, p.L), p.nc_cap);
    const int cap = p.cell_cap;
    if constexpr (!PLACE)
        if (threadIdx.x == 0) p.ws_meta[0] = ncells;  // (every workgroup writes the same value) for K2

    int *part = p.ws_part + ((size_t)pair * p.nsplit + slice) * p.nc_cap;  // [cell] of this slice
    int *blocktot = p.ws_blocktot + ((size_t)pair * p.nsplit + slice) * p.nblk_cap;
    const int *off = p.ws_off + (size_t)pair * (p.nc_cap + 1);
    Entry<A> *entries = plane_entries<A>(p, pair);
    // per-plane bases (64-bit, uniform) + 32-bit per-sample offsets (the host checks Q*H*L*P*2 < 2^31)
    const size_t plane_s0 = ((size_t)b * p.Q * p.H + h) * p.LP;
    const T *loc = static_cast<const T *>(p.loc) + 2 * plane_s0;
    const T *attn = static_cast<const T *>(p.attn) + plane_s0;
    const int HLP = p.H * p.LP;
    const int tid = threadIdx.x;

    // fixed slot per thread: with LPt (level, point) slots in play, threads [0, dq * LPt) are active; thread t serves
    // slot t % LPt of the queries qa + t / LPt + k * dq.  (More slots than threads: each thread strides over the
    // slots of one query at a time.)
    const bool fixed = p.LP <= kCellBlock;

    for (int c0 = 0; c0 < ncells; c0 += cap) {  // one trip unless the plane has more cells than fit in LDS
        const int n = min(cap, ncells - c0);
        // A trip only walks the samples of the levels whose cells it holds (a sample's level is known from its
        // index): a plane with several trips reads its samples about once in total instead of once per trip, and
        // all threads share the trip's levels (c5, 88k cells, three trips: levels 0 | 0-1 | 1-4).
        int l_lo = 0, l_hi = p.L - 1;
        if (fixed && ncells > cap) {
            while (l_lo < p.L - 1 && tab->cstart[l_lo + 1] <= c0) ++l_lo;           // last level starting at or before c0
            while (l_hi > l_lo && tab->cstart[l_hi] >= c0 + n) --l_hi;               // ... and before the chunk's end
        }
        const int LPt = fixed ? (l_hi - l_lo + 1) * p.P : p.LP;
        const int dq = fixed ? kCellBlock / LPt : 1;
        const int sl0 = fixed ? l_lo * p.P + tid % LPt : 0, tq = fixed ? tid / LPt : 0;
        const bool active = fixed && tq < dq;
        for (int i = tid; i < n; i += kCellBlock) {
            int v = 0;
            if constexpr (PLACE)  // this slice's first slot in every cell list
                v = off[c0 + i] + part[c0 + i];
            s_cell[i] = v;
        }
        __syncthreads();
        auto visit = [&](int q, int cell, uint32_t cellw, A dx, A dy, A at) {
            const unsigned rel = (unsigned)(cell - c0);
            if (rel < (unsigned)n) {
                if constexpr (!PLACE) {
                    atomicAdd(&s_cell[rel], 1);
                } else {
                    const int pos = atomicAdd(&s_cell[rel], 1);
                    // (plain store: these partial lines must merge in L2 — streaming stores: 57 -> 202 us)
                    entries[pos] = Entry<A>::pack((uint32_t)q, cellw, at, dx, dy);
                }
            }
        };
        if (active) {
            const int l = sl0 / p.P;
            const int lw = tab->w[l], lh = tab->h[l], cs = tab->cstart[l], ps = tab->start[l];
            int q = qa + tq;
            int sidx = q * HLP + sl0;
            const int d_sidx = dq * HLP;
            if constexpr (!PLACE) {
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
                        int cell;
                        uint32_t cellw;
                        A dx, dy;
                        if (qk < qb && sample_cell<A>(TR::to_acc(xy.v[0]), TR::to_acc(xy.v[1]), lh, lw, cs, ps, l, p.zeros,
                                                      p.align, cell, cellw, dx, dy))
                            visit(qk, cell, cellw, dx, dy, (A)0);
                    }
                    q += RING * dq;
                    sidx += RING * d_sidx;
                }
            } else {
            // software-pipelined walk: the next sample's (x, y, a) are requested before this sample's record is
            // stored, so the wait for them never has to drain the scattered store behind it (one vmcnt queue)
            Pack<T, 2> xy, xy_n;
            T at = TR::from_acc((A)0), at_n = at;
            xy.v[0] = xy.v[1] = at;
            xy_n = xy;
            if (q < qb) {
                xy = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
                at = attn[sidx];
            }
            while (q < qb) {
                const int qn = q + dq;
                sidx += d_sidx;
                if (qn < qb) {
                    xy_n = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
                    at_n = attn[sidx];
                }
                int cell;
                uint32_t cellw;
                A dx, dy;
                if (sample_cell<A>(TR::to_acc(xy.v[0]), TR::to_acc(xy.v[1]), lh, lw, cs, ps, l, p.zeros, p.align, cell, cellw,
                                   dx, dy))
                    visit(q, cell, cellw, dx, dy, TR::to_acc(at));
                q = qn;
                xy = xy_n;
                at = at_n;
            }
            }
        } else if (!fixed) {
            for (int q = qa; q < qb; ++q) {
                for (int sl = tid; sl < p.LP; sl += kCellBlock) {
                    const int l = sl / p.P;
                    const int sidx = q * HLP + sl;
                    const Pack<T, 2> xy = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
                    int cell;
                    uint32_t cellw;
                    A dx, dy;
                    if (sample_cell<A>(TR::to_acc(xy.v[0]), TR::to_acc(xy.v[1]), tab->h[l], tab->w[l], tab->cstart[l],
                                       tab->start[l], l, p.zeros, p.align, cell, cellw, dx, dy))
                        visit(q, cell, cellw, dx, dy, PLACE ? TR::to_acc(attn[sidx]) : (A)0);
                }
            }
        }
        __syncthreads();
        if constexpr (!PLACE) {
            // counts out; and per block of kScanCells cells this slice's total (K2 derives every block's base from
            // them): wave sums, added into a small LDS table behind the cell table
            int *s_blk = s_cell + cap;
            for (int i = tid; i < p.nblk_cap; i += kCellBlock)
                if (c0 == 0) s_blk[i] = 0;
            __syncthreads();
            for (int i0 = 0; i0 < n; i0 += kCellBlock) {
                const int i = i0 + tid;
                int v = i < n ? s_cell[i] : 0;
                if (i < n) part[c0 + i] = v;
#pragma unroll
                for (int m = 1; m < kWave; m <<= 1) v += __shfl_xor(v, m, kWave);
                if ((tid & (kWave - 1)) == 0 && i < n && v != 0) atomicAdd(&s_blk[(c0 + i) / kScanCells], v);
            }
            __syncthreads();
            if (c0 + cap >= ncells)  // last trip: every block of cells has its total
                for (int i = tid; i < p.nblk_cap; i += kCellBlock) blocktot[i] = s_blk[i];
        }
    }
}

// ------------------------------------------------------------------------------------------
// K2: cell lists' first records, one launch.  The base of a block of kScanCells cells is the sum of the per-slice
// block totals K1 left (a few hundred loads), so no pass over the preceding cells and no look-back is needed.
// ------------------------------------------------------------------------------------------
template <typename Tag> __global__ __launch_bounds__(kScanCells) void msda_cell_scan_kernel(const Params p)
{
    const int pair = blockIdx.x / p.nblk_cap, blk = blockIdx.x - pair * p.nblk_cap;
    const int nc = p.ws_meta[0];  // real cell count of a plane, left by the count pass
    if (blk * kScanCells >= nc) return;  // block-uniform
    __shared__ int s_red[kScanCells / kWave];
    __shared__ int s_wave[kScanCells / kWave];
    const int t = threadIdx.x;
    const int lane = t & (kWave - 1), wid = t / kWave;
    // base: records in the blocks before this one (all slices)
    int acc = 0;
    for (int i = t; i < p.nsplit * blk; i += kScanCells) {
        const int jj = i / blk, bb = i - jj * blk;
        acc += p.ws_blocktot[((size_t)pair * p.nsplit + jj) * p.nblk_cap + bb];
    }
#pragma unroll
    for (int m = 1; m < kWave; m <<= 1) acc += __shfl_xor(acc, m, kWave);
    if (lane == 0) s_red[wid] = acc;
    const int c = blk * kScanCells + t;
    int tot = 0;
    if (c < nc) {
        int *part = p.ws_part + (size_t)pair * p.nsplit * p.nc_cap + c;
        int j = 0;
        for (; j + 4 <= p.nsplit; j += 4) {  // four independent loads in flight
            const int n0 = part[(size_t)(j + 0) * p.nc_cap], n1 = part[(size_t)(j + 1) * p.nc_cap];
            const int n2 = part[(size_t)(j + 2) * p.nc_cap], n3 = part[(size_t)(j + 3) * p.nc_cap];
            part[(size_t)(j + 0) * p.nc_cap] = tot;
            part[(size_t)(j + 1) * p.nc_cap] = tot + n0;
            part[(size_t)(j + 2) * p.nc_cap] = tot + n0 + n1;
            part[(size_t)(j + 3) * p.nc_cap] = tot + n0 + n1 + n2;
            tot += n0 + n1 + n2 + n3;
        }
        for (; j < p.nsplit; ++j) {
            const int n = part[(size_t)j * p.nc_cap];
            part[(size_t)j * p.nc_cap] = tot;
            tot += n;
        }
    }
    // exclusive scan of tot over the block
    int inc = tot;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int nn = __shfl_up(inc, d, kWave);
        if (lane >= d) inc += nn;
    }
    if (lane == kWave - 1) s_wave[wid] = inc;
    __syncthreads();
    int beg = inc - tot;
#pragma unroll
    for (int i = 0; i < kScanCells / kWave; ++i) {
        beg += s_red[i];
        if (i < wid) beg += s_wave[i];
    }
    if (c < nc) {
        int *off = p.ws_off + (size_t)pair * (p.nc_cap + 1);
        off[c] = beg;
        if (c == nc - 1) {  // the plane's total behind the last cell
            off[nc] = beg + tot;
            p.ws_total[pair] = beg + tot;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K4: gather.  One G-lane group per window of p.win sorted records, VEC channels per lane.
// ------------------------------------------------------------------------------------------
template <typename A> struct alignas(16) CornerW {
    A w[4];  // a * {(1-dx)(1-dy), dx(1-dy), (1-dx)dy, dx dy}: corners 00, 01, 10, 11
};

// streaming (nontemporal) range-checked store of VEC accumulators at a 32-bit byte offset
template <typename A, int VEC> __device__ __forceinline__ void store_acc_pack(rsrc_t r, uint32_t off, const Pack<A, VEC> &v)
{
    constexpr int BYTES = (int)sizeof(A) * VEC;
    constexpr int kNt = 2;  // aux: nt
    if constexpr (BYTES == 16) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(RawLoad<16>::type, v), r, off, 0, kNt);
    } else if constexpr (BYTES == 8) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(RawLoad<8>::type, v), r, off, 0, kNt);
    } else if constexpr (BYTES == 4) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, off, 0, kNt);
    } else {
        static_assert(BYTES % 16 == 0, "accumulator pack must be 4, 8 or a multiple of 16 bytes");
        struct Pieces {
            RawLoad<16>::type p[BYTES / 16];
        };
        const Pieces ps = __builtin_bit_cast(Pieces, v);
#pragma unroll
        for (int i = 0; i < BYTES / 16; ++i) __builtin_amdgcn_raw_buffer_store_b128(ps.p[i], r, off + 16u * i, 0, kNt);
    }
}

// N consecutive 32-bit LDS words (N = 4 or 8, 16-byte aligned) with 16-byte reads
template <int N> __device__ __forceinline__ void lds_read_u32s(const uint32_t *src, uint32_t (&dst)[N])
{
    static_assert(N == 4 || N == 8, "4 or 8 words");
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + 4 * i);
        dst[4 * i] = v.x;
        dst[4 * i + 1] = v.y;
        dst[4 * i + 2] = v.z;
        dst[4 * i + 3] = v.w;
    }
}

constexpr uint32_t kSegStart = 1u;  // first record of a segment (a new cell, or the window's first record)
constexpr uint32_t kSegCarry = 2u;  // ... and the cell is the right-hand x-neighbour of the previous record's cell

// Occupancy: four workgroups per CU (<= 128 VGPRs, <= 40 KB of LDS).  The compiler left the 16-bit variants at 136
// VGPRs; asked for four waves per SIMD, 4-lane groups (D = 32 in bf16 / fp16) take 118 without a spill — c3's
// grad_value group 175.5 -> 171.5 us, same-box A/B — while 8-lane groups and wider spill (c5: no gain), so those keep
// the compiler's choice.
// TG: storage type of grad_out (the module's 16-bit storage next to fp32 points: msda_bwd_fused_f32_sbf16 / _sf16)
template <typename T, int VEC, int G, int GB, typename TG = T>
__global__ __launch_bounds__(GB, (VEC == 8 && G > 4) ? 1 : 4) void msda_value_gather_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR [[maybe_unused]] = Traits<T>;
    constexpr int kGatherItemBlock = GB;  // threads per workgroup
    constexpr int NU = kGatherItemBlock / G;
    constexpr int UB = G < 8 ? G : 8;  // row loads in flight per lane
    const int slots = (p.win_cap + NU - 1) / NU;
    int pair, slot;
    if (!decode_block(p.grid3d, p.B * p.H, slots, p.xcd_map, pair, slot)) return;
    const int N = p.ws_total[pair];  // records of the plane
    if ((long long)slot * NU * p.win >= N) return;  // block-uniform
    const int tid = threadIdx.x;
    const int unit = tid / G, j = tid % G;
    const int win = slot * NU + unit;
    const int r0 = win * p.win;
    const int count = max(0, min(p.win, N - r0));  // idle groups (count == 0) still take part in the block barriers
    const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
    const int gbase = tid - j;

    __shared__ LevelTab tab;
    __shared__ uint32_t s_wstep[kSortedMaxLevels];  // bytes between the slots of vertically adjacent pixels, per level
    __shared__ CornerW<A> s_w[kGatherItemBlock];
    __shared__ __attribute__((aligned(32))) uint32_t s_q[kGatherItemBlock];     // grad_out row byte offset (0x80000000: past the end, reads 0)
    __shared__ __attribute__((aligned(32))) uint32_t s_flag[kGatherItemBlock];  // kSegStart | kSegCarry
    __shared__ uint32_t s_cellw[kGatherItemBlock];  // cell word of the record
    __shared__ __attribute__((aligned(32))) A s_rows[kGatherItemBlock * 4 * VEC];  // [unit][corner][G * VEC]: parked continuation rows
    __shared__ unsigned char s_cont[NU];   // the unit's first segment continues the previous window's cell
    __shared__ unsigned char s_pure[NU];   // ... and is the unit's only segment (bytes: with 4-lane groups the kernel's
                                           // LDS is 40 KB, and four workgroups must fit a CU's 160 KB)

    const Entry<A> *entries = plane_entries<A>(p, pair);
    const TG *gout = static_cast<const TG *>(p.grad_out) + ((size_t)b * p.Q * p.H + h) * p.D;  // uniform base
    const uint32_t q_stride = (uint32_t)(p.H * p.D) * (uint32_t)sizeof(TG);  // bytes; Q*H*D*sizeof < 2^31 is checked on the host
    // grad_out rows of this plane through a buffer descriptor: scalar base + 32-bit byte offset per lane
    const rsrc_t rs_go = make_rsrc(gout, (uint32_t)(((size_t)p.Q * p.H * p.D - (size_t)h * p.D) * sizeof(TG)));
    const int nchan_chunks = (p.D + G * VEC - 1) / (G * VEC);

    // first requests: the window's first batch of records and the record in front of the window (idle groups: the
    // plane's record 0, unused)
    const int first_idx = count > 0 ? r0 + min(j, count - 1) : 0;
    const Entry<A> e_first = entries[first_idx];
    const uint32_t prev_cellw = (win > 0 && count > 0) ? entries[r0 - 1].cellflag() : 0xFFFFFFFFu;
    load_level_table(&tab, p.shapes, p.L);
    __syncthreads();
    const uint32_t rowstep = 4u * (uint32_t)p.D * (uint32_t)sizeof(A);  // bytes per pixel: four slots of D accumulators
    if (tid < p.L) s_wstep[tid] = (uint32_t)tab.w[tid] * rowstep;
    __syncthreads();
    // the window's first segment continues the cell of the record in front of it (decided by lane 0 of the group:
    // it holds record r0)
    const bool cont_here = count > 0 && win > 0 && j == 0 && prev_cellw == e_first.cellflag();
    const bool first_cont = __shfl(cont_here ? 1 : 0, gbase & (kWave - 1), kWave) != 0;

    // the plane's slots [pixel][corner][D] through a buffer descriptor: 32-bit byte offsets, and a corner outside the
    // image gets an out-of-range offset, so the hardware drops its store (no branch)
    const size_t plane_slots = (size_t)p.I * rowstep;
    const rsrc_t rs_sc = make_rsrc(static_cast<unsigned char *>(p.ws_scratch) + (size_t)pair * plane_slots, (uint32_t)plane_slots);
    const uint32_t bias_off = kPixBias * rowstep;  // (mod 2^32, like the products it is subtracted from)
    A *controw = static_cast<A *>(p.ws_cont) + ((size_t)pair * p.cont_cap + slot) * 4 * p.D;

    for (int cc = 0; cc < nchan_chunks; ++cc) {
        const int c0 = (cc * G + j) * VEC;
        const bool lane_ok = c0 < p.D;
        const uint32_t lane_elem = (lane_ok ? (uint32_t)c0 : 0u) * (uint32_t)sizeof(TG);
        A acc[4][VEC];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[k][i] = (A)0;

        // state of the segment being accumulated (uniform inside the group)
        int nseg = 0;              // segments started so far in this window
        bool seg_cont = false;     // the current segment is the window's first AND continues the previous window's cell
        uint32_t cur_cellw = 0;    // cell word of the current segment
        uint32_t last_cellw = prev_cellw;  // cell word of the record before the current batch

        // one segment's rows -> the pixels' slots.  skip_right: the right-hand corners were carried over.
        auto flush = [&](bool skip_right) {
            if (!lane_ok) return;
            const uint32_t base = mul24(cur_cellw & kPixMask, rowstep) - bias_off + (uint32_t)c0 * (uint32_t)sizeof(A);
            const uint32_t wstep = s_wstep[(cur_cellw >> kLevelShift) & 31u];
            const uint32_t dstep = (uint32_t)p.D * (uint32_t)sizeof(A);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool on = ((cur_cellw >> (28 + k)) & 1u) && !(skip_right && (k & 1));
                const uint32_t o = on ? base + ((k & 1) ? rowstep : 0u) + ((k & 2) ? wstep : 0u) + (uint32_t)k * dstep : 0x80000000u;
                Pack<A, VEC> v;
#pragma unroll
                for (int i = 0; i < VEC; ++i) v.v[i] = acc[k][i];
                // written once, read once by the finish kernel: streaming, so it does not displace grad_out rows in L2
                store_acc_pack<A, VEC>(rs_sc, o, v);
            }
        };
        auto park = [&]() {  // a continuation segment's rows -> LDS, for the segment it continues
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                Pack<A, VEC> o;
#pragma unroll
                for (int v = 0; v < VEC; ++v) o.v[v] = acc[k][v];
                *reinterpret_cast<Pack<A, VEC> *>(&s_rows[((unit * 4 + k) * G + j) * VEC]) = o;
            }
        };
        auto to_cont = [&]() {  // ... or, for the workgroup's first window, global memory (one row set per workgroup)
            if (!lane_ok) return;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                Pack<A, VEC> o;
#pragma unroll
                for (int v = 0; v < VEC; ++v) o.v[v] = acc[k][v];
                store_stream(controw + (size_t)k * p.D + c0, o);
            }
        };

        // one batch: lane j converts record v0 + j, the group gathers and accumulates G rows
        auto batch = [&](const Entry<A> &e_cur, int v0) {
            // ---- convert ----
            {
                const int v = v0 + j;
                const bool ok = v < count;
                const A a = ok ? e_cur.a() : (A)0;
                const A dx = e_cur.dx(), dy = e_cur.dy();
                const A ax1 = a * dx, ax0 = a - ax1;
                CornerW<A> cw;
                cw.w[3] = ax1 * dy;
                cw.w[2] = ax0 * dy;
                cw.w[1] = ax1 - cw.w[3];
                cw.w[0] = ax0 - cw.w[2];
                const uint32_t cellw = ok ? e_cur.cellflag() : 0xFFFFFFFEu;
                // the previous record's cell: the lane to the left, or (lane 0) the record before this batch
                // (wave_shr:1 — a DPP move, no LDS round trip; lane 0 of the wave keeps its own value and is a group's lane 0)
                uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp((int)cellw, (int)cellw, 0x138, 0xF, 0xF, false);
                if (j == 0) left = last_cellw;
                // a segment starts at the window's first record and wherever the cell changes; it takes over the ended
                // cell's right-hand corners when it is the right-hand x-neighbour in the same row: next pixel index,
                // same level, and the shared column's pixels exist on either side (left cell: corners 01 / 11, this
                // cell: 00 / 10)
                const bool start = ok && (v == 0 || cellw != left);
                const bool carry = start && v != 0 && (cellw & 0x0FFFFFFFu) == (left & 0x0FFFFFFFu) + 1u &&
                                   (left & 0xA0000000u) != 0 && (cellw & 0x50000000u) != 0;
                const uint32_t flag = (start ? kSegStart : 0u) | (carry ? kSegCarry : 0u);
                wave_lds_sync();  // the previous batch's hand-off has been read
                s_q[tid] = ok ? mul24(e_cur.q(), q_stride) : 0x80000000u;  // both < 2^24 (host check)
                s_w[tid] = cw;
                s_flag[tid] = flag;
                s_cellw[tid] = cellw;
                wave_lds_sync();
                last_cellw = s_cellw[gbase + G - 1];
            }
            // ---- gather + accumulate ----
#pragma unroll
            for (int jj = 0; jj < G; jj += UB) {  // UB row loads issued back to back, then consumed
                if (v0 + jj < count) {            // uniform per group; G == UB: always true
                    Pack<TG, VEC> g[UB];
                    uint32_t qs[UB], fl[UB];
                    lds_read_u32s<UB>(&s_q[gbase + jj], qs);   // one or two 16-byte LDS reads each
                    lds_read_u32s<UB>(&s_flag[gbase + jj], fl);
#pragma unroll
                    for (int u = 0; u < UB; ++u)
                        g[u] = __builtin_bit_cast(Pack<TG, VEC>,
                                                  RawLoad<sizeof(TG) * VEC>::load(rs_go, qs[u] + lane_elem));
                    __builtin_amdgcn_sched_barrier(0);  // keep the UB loads together: hipcc otherwise serialises some
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const uint32_t flag = fl[u];
                        if (flag & kSegStart) {  // uniform inside the group, divergent across the wave's groups
                            bool carry = false;
                            if (nseg > 0) {
                                carry = (flag & kSegCarry) != 0 && !seg_cont;
                                if (seg_cont) {
                                    if (unit == 0)
                                        to_cont();
                                    else
                                        park();
                                } else {
                                    flush(carry);
                                }
                            }
                            if (carry) {  // the ended cell's right-hand corners are this cell's left-hand ones
#pragma unroll
                                for (int v = 0; v < VEC; ++v) {
                                    acc[0][v] = acc[1][v];
                                    acc[2][v] = acc[3][v];
                                    acc[1][v] = (A)0;
                                    acc[3][v] = (A)0;
                                }
                            } else {
#pragma unroll
                                for (int k = 0; k < 4; ++k)
#pragma unroll
                                    for (int v = 0; v < VEC; ++v) acc[k][v] = (A)0;
                            }
                            seg_cont = nseg == 0 && first_cont;
                            ++nseg;
                            cur_cellw = s_cellw[gbase + jj + u];
                        }
                        const CornerW<A> w = s_w[gbase + jj + u];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int v = 0; v < VEC; ++v) acc[k][v] = fma_t(w.w[k], Traits<TG>::to_acc(g[u].v[v]), acc[k][v]);
                    }
                }
            }
        };
        Entry<A> e_cur = cc == 0 ? e_first : entries[first_idx];
        for (int v0 = 0; v0 < count; v0 += G) {
            // the next batch's record (clamped: always a valid address) is in flight while this batch is consumed
            const Entry<A> e_next = entries[r0 + min(v0 + G + j, count - 1)];
            batch(e_cur, v0);
            e_cur = e_next;
        }
        // ---- the window's open last segment.  A continuation that fills its whole window ("pure") is handed to the
        // segment it continues; a segment that started at a cell start collects the continuations behind it. ----
        const bool pure = count > 0 && nseg == 1 && seg_cont;
        if (j == 0) {
            s_cont[unit] = (count > 0 && first_cont && unit > 0) ? 1 : 0;
            s_pure[unit] = pure ? 1 : 0;
        }
        if (pure && unit > 0) park();
        __syncthreads();
        if (count > 0 && !(pure && unit > 0)) {
            for (int u = unit + 1; u < NU && s_cont[u]; ++u) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const Pack<A, VEC> r = *reinterpret_cast<const Pack<A, VEC> *>(&s_rows[((u * 4 + k) * G + j) * VEC]);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[k][v] += r.v[v];
                }
                if (!s_pure[u]) break;  // that window goes on with other cells
            }
            if (pure)
                to_cont();
            else
                flush(false);
        }
        if (cc + 1 < nchan_chunks) __syncthreads();  // the parked rows have been consumed before the next chunk parks
    }
}

// ------------------------------------------------------------------------------------------
// K5: per pixel.  Slot k of a pixel was written by: 0 cell (x, y) [its corner 00], 1 cell (x-1, y) [01],
// 2 cell (x, y-1) [10], 3 cell (x-1, y-1) [11] — iff that cell has records, and, for the odd slots, unless the
// cell's right-hand corners were carried into its x-neighbour (both start in the same window).  Cells longer than a
// gather workgroup's windows add one continuation row set per further workgroup.
// ------------------------------------------------------------------------------------------
template <typename A, int VEC> __device__ __forceinline__ Pack<A, VEC> load_acc_pack(rsrc_t r, uint32_t off)
{
    constexpr int BYTES = (int)sizeof(A) * VEC;
    if constexpr (BYTES <= 16) {
        return __builtin_bit_cast(Pack<A, VEC>, RawLoad<BYTES>::load(r, off));
    } else {
        constexpr int N = BYTES / 16, SUB = VEC / N;
        Pack<A, VEC> o;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const Pack<A, SUB> part = __builtin_bit_cast(Pack<A, SUB>, RawLoad<16>::load(r, off + 16u * i));
#pragma unroll
            for (int v = 0; v < SUB; ++v) o.v[i * SUB + v] = part.v[v];
        }
        return o;
    }
}

// Pixels per finish workgroup.  Measured (finish kernel, us): 1.1 M pixels with ~0.8 samples per pixel (a 100 x 134 ..
// 13 x 17 pyramid, 64 planes, 900 queries: Deformable-DETR / Grounding-DINO decoder at COCO size) 32: 81, 64: 62, 128:
// 64; c2-10k (174 k pixels) 32: 20.3, 64: 21.4, 128: 24.6 — few pixels want many small workgroups.
inline int finish_pixels(long long planes, long long I) { return planes * I >= 400000 ? 64 : 32; }

// K5, two phases per workgroup of kFinishPixels consecutive pixels.
// Phase 1, ONE LANE PER PIXEL (the first kFinishPixels threads): pixel -> (level, x, y), the six list starts around
// it, which of its four slots were written, whether continuation rows exist.  That is ~250 vector + ~140 scalar
// instructions; with one G-lane group per pixel doing it all (the previous kernel) a wave paid them for 8 pixels, and
// they were 40 of the kernel's 88 us on a 1.1 M-pixel problem (ablation: no memory operation at all in the body still
// took 56 us against 16 for the launch + level table).
// Phase 2, G lanes per pixel row: the loads of up to four rows' slots are issued together (only written slots; a load
// no lane of the wave needs is not issued), fixed-order sums, one store per row.
// TV: storage type of grad_value (see msda_fwd_kernel); kFinishPixels: pixels per workgroup (finish_pixels)
template <typename T, int VEC, int G, int GB, typename TV = T, int kFinishPixels = 32>
__global__ __launch_bounds__(kBlock) void msda_value_finish_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<TV>;
    constexpr int NU = kBlock / G;
    constexpr int NUG = GB / G;  // windows per gather workgroup
    constexpr int ROUNDS = (kFinishPixels + NU - 1) / NU;  // pixel rows per lane group
    constexpr int RB = ROUNDS < 4 ? ROUNDS : 4;            // ... whose slot loads are in flight together
    const int slots = (p.I + kFinishPixels - 1) / kFinishPixels;
    int pair, slot;
    if (!decode_block(p.grid3d, p.B * p.H, slots, p.xcd_map, pair, slot)) return;
    __shared__ LevelTab tab;
    __shared__ int s_rng[kFinishPixels][6];     // list starts t0 t1 t2 (cell row y-1), u0 u1 u2 (cell row y)
    __shared__ uint32_t s_flags[kFinishPixels];  // bits 0-3: slot k was written; bit 4: continuation rows to add
    load_level_table(&tab, p.shapes, p.L);
    __syncthreads();
    const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
    const int tid = threadIdx.x;
    const int pix_base = slot * kFinishPixels;
    const FastDiv dw = p.div_win;  // record index -> gather window (the window size is chosen per call)
    auto window = [dw](int r) { return fast_div((uint32_t)r, dw); };
    auto carried = [&](int a0, int a1, int a2) { return a1 > a0 && a2 > a1 && window(a0) == window(a1); };
    // continuation row sets: gather-workgroup boundaries strictly inside a cell's window range
    auto nconts = [&](int beg, int end) {
        return end > beg ? (int)(window(end - 1) / (uint32_t)NUG) - (int)(window(beg) / (uint32_t)NUG) : 0;
    };
    // ---- phase 1: record ranges of the cells (x-1,y-1), (x,y-1) | (x-1,y), (x,y) of the thread's pixel: consecutive
    // cell ids, consecutive ranges.  A pixel the shapes tensor does not describe (sum h*w != I), or one whose cells
    // were dropped for lack of workspace, has no cells: all ranges empty, the row is stored as zeros. ----
    if (tid < kFinishPixels) {
        const int pix = pix_base + tid;
        struct Three {
            int v[3];
        };
        Three ta{{0, 0, 0}}, ua{{0, 0, 0}};
        if (pix < p.I) {
            const int *off = p.ws_off + (size_t)pair * (p.nc_cap + 1);
            const int ncells = min(plane_cells(tab, p.L), p.nc_cap);
            int l = 0;
            while (l < p.L - 1 && pix >= tab.start[l + 1]) ++l;
            const int rel = pix - tab.start[l], w = tab.w[l], cw = w + 1;
            const int y = rel / max(w, 1), x = rel - y * w;
            const int c11 = tab.cstart[l] + y * cw + x;  // cell (x-1, y-1)
            if (w > 0 && y < tab.h[l] && c11 + cw + 2 <= ncells) {
                __builtin_memcpy(&ta, off + c11, sizeof(Three));  // three consecutive list starts: one 12-byte load
                __builtin_memcpy(&ua, off + c11 + cw, sizeof(Three));
            }
        }
        const int t0 = ta.v[0], t1 = ta.v[1], t2 = ta.v[2], u0 = ua.v[0], u1 = ua.v[1], u2 = ua.v[2];
        uint32_t flags = (u2 > u1 ? 1u : 0u) | ((u1 > u0 && !carried(u0, u1, u2)) ? 2u : 0u) | (t2 > t1 ? 4u : 0u) |
                         ((t1 > t0 && !carried(t0, t1, t2)) ? 8u : 0u);
        if (nconts(u1, u2) + nconts(u0, u1) + nconts(t1, t2) + nconts(t0, t1) != 0) flags |= 16u;
        s_flags[tid] = flags;
        s_rng[tid][0] = t0, s_rng[tid][1] = t1, s_rng[tid][2] = t2;
        s_rng[tid][3] = u0, s_rng[tid][4] = u1, s_rng[tid][5] = u2;
    }
    __syncthreads();
    // ---- phase 2 ----
    const int unit = tid / G, j = tid % G;
    const A *cont = static_cast<const A *>(p.ws_cont) + (size_t)pair * p.cont_cap * 4 * p.D;
    const size_t plane_slots = (size_t)p.I * 4 * p.D * sizeof(A);  // < 2^31 (host check)
    const rsrc_t rs_sc = make_rsrc(static_cast<const unsigned char *>(p.ws_scratch) + (size_t)pair * plane_slots, (uint32_t)plane_slots);
    const int nchan_chunks = (p.D + G * VEC - 1) / (G * VEC);
    for (int cc = 0; cc < nchan_chunks; ++cc) {
        const int c0 = (cc * G + j) * VEC;
        if (c0 >= p.D) continue;
        for (int r0 = 0; r0 < ROUNDS; r0 += RB) {
            if (pix_base + r0 * NU >= p.I) break;  // block-uniform: nothing left
            // the rows' four slots each: independent range-checked loads through a buffer descriptor; a slot nobody
            // wrote gets an out-of-range offset and reads 0 without touching memory, and an instruction whose lanes
            // are all masked is not issued at all (it would still occupy the vector-memory path)
            uint32_t flags[RB];
            Pack<A, VEC> rr[RB][4];
#pragma unroll
            for (int t = 0; t < RB; ++t) {
                const int pt = (r0 + t) * NU + unit, pix = pix_base + pt;
                const bool live = pt < kFinishPixels && pix < p.I;  // (64 lane groups of 4 lanes, 32 pixels: half the groups idle)
                flags[t] = live ? s_flags[pt] | 32u : 0u;  // bit 5: the row exists
                const uint32_t base = ((uint32_t)(live ? pix : 0) * 4u * (uint32_t)p.D + (uint32_t)c0) * (uint32_t)sizeof(A);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) rr[t][k].v[v] = (A)0;
                    const bool on = ((flags[t] >> k) & 1u) != 0;
                    if (__builtin_amdgcn_ballot_w64(on) != 0)
                        rr[t][k] = load_acc_pack<A, VEC>(rs_sc, on ? base + (uint32_t)k * (uint32_t)p.D * (uint32_t)sizeof(A) : 0x80000000u);
                }
            }
#pragma unroll
            for (int t = 0; t < RB; ++t) {
                if (!(flags[t] & 32u)) continue;
                const int pt = (r0 + t) * NU + unit, pix = pix_base + pt;
                A acc[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = ((rr[t][0].v[v] + rr[t][1].v[v]) + rr[t][2].v[v]) + rr[t][3].v[v];
                if (flags[t] & 16u) {
                    auto add = [&](int beg, int end, int corner) {
                        if (end <= beg) return;
                        const int g1 = (int)(window(end - 1) / (uint32_t)NUG);
                        for (int g = (int)(window(beg) / (uint32_t)NUG) + 1; g <= g1; ++g) {
                            const Pack<A, VEC> cr =
                                *reinterpret_cast<const Pack<A, VEC> *>(cont + ((size_t)g * 4 + corner) * p.D + c0);
#pragma unroll
                            for (int v = 0; v < VEC; ++v) acc[v] += cr.v[v];
                        }
                    };
                    const int t0 = s_rng[pt][0], t1 = s_rng[pt][1], t2 = s_rng[pt][2];
                    const int u0 = s_rng[pt][3], u1 = s_rng[pt][4], u2 = s_rng[pt][5];
                    add(u1, u2, 0);
                    add(u0, u1, 1);
                    add(t1, t2, 2);
                    add(t0, t1, 3);
                }
                // several rounds over the queries: running sums in the accumulate type between them
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
                        continue;
                    }
                }
                Pack<TV, VEC> o;
#pragma unroll
                for (int v = 0; v < VEC; ++v) o.v[v] = TR::from_acc(acc[v]);
                TV *dst = static_cast<TV *>(p.grad_value) + (((size_t)b * p.I + pix) * p.H + h) * p.D + c0;
                store_stream(dst, o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// workspace layout (host + device agree through these helpers)
// ------------------------------------------------------------------------------------------
struct SortedWsLayout {
    int nc_cap, nblk_cap, win, win_cap, cont_cap, nsplit;
    int q_round, rounds;  // queries per round and rounds over the queries (1: everything at once)
    int ent_n0, ent_n1, ent_n2;   // planes whose records live in the caller's grad_loc / grad_attn / grad_value buffer
    size_t off_part, off_blocktot, off_off, off_total, off_meta, off_entries, off_scratch, off_cont, off_accum, total;
};

int option_q_round();  // queries per round of the sorted path (0: automatic), msda_api.hip

int option_cell_slices();  // 0: automatic (msda_api.hip)
int option_gather_win();   // records per gather window (0: automatic)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// lanes per row the gather kernel will use (the same choice as dispatch_value_gather_group)
inline int gather_group_lanes(int64_t D, size_t elem_bytes, bool vec)
{
    const int vecf = (int)(16 / elem_bytes);
    const int64_t lanes = vec && (D % vecf) == 0 ? D / vecf : D;
    return lanes <= 4 ? 4 : lanes <= 8 ? 8 : lanes <= 16 ? 16 : lanes <= 32 ? 32 : 64;
}

// `vec`: size for the 16-byte vector path (aligned grad_out / grad_value, D a multiple of 16 bytes of elements);
// the scalar path has fewer windows per workgroup and needs more continuation rows (msda_bwd_workspace_bytes reports
// the larger of the two layouts)
// records_in_grads: the caller's grad_loc / grad_attn buffers (of elem_bytes elements) and — single-round problems —
// its grad_value buffer (of value_elem_bytes elements; 0: elem_bytes) hold the records of as many planes as fit them
// (MSDA_WS_RECORDS_IN_GRADS); ent_n0 / ent_n1 / ent_n2 say how many
inline SortedWsLayout sorted_ws_layout(int64_t B, int64_t I, int64_t H, int64_t D, int64_t Q, int64_t L, int64_t P,
                                       size_t acc_bytes, size_t elem_bytes, bool vec = true, bool records_in_grads = false,
                                       size_t value_elem_bytes = 0)
{
    SortedWsLayout w;
    const size_t pairs = (size_t)(B * H);
    // Rounds over the queries bound the record buffer: at most ~1 GiB of sorted records at a time (c5: 2 rounds of
    // 50 000 queries, workspace 2.9 -> 2.0 GB; everything up to c2 / c3 size: one round).  Measured at c5 (grad_value,
    // us): 1 round 5529, 2 rounds 5425, 3 rounds 5518, 4 rounds 5601, 7 rounds 6720 — the place pass gains from the
    // smaller scatter range what the repeated finish pass and fixed costs take; keeping a plane's grad_out rows in L2
    // this way (rounds of 16k queries) did not speed the gather up.
    int64_t q_round = Q;
    {
        const int64_t per_query = (int64_t)pairs * L * P * (acc_bytes == 8 ? 32 : 16);
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
    const size_t samples = (size_t)(q_round * L * P);  // per plane and round
    w.nc_cap = (int)(2 * I + 2 * L);             // (w+1)(h+1) <= 2wh + 2 per level
    w.nblk_cap = (w.nc_cap + kScanCells - 1) / kScanCells;
    const int gl = gather_group_lanes(D, elem_bytes, vec);
    const int nug = gl >= kGatherMinBlock ? 1 : kGatherMinBlock / gl;
    // records per gather window: 64, fewer when that would leave the chip short of workgroups (1024 run at a time; aim
    // for two rounds of them).  Measured (grad_value, us; 64 / 32 / 16 records): c2 @ 5k 116 / 110 / -, c2 @ 2k 128 /
    // 125 / 121, a 17 821-pixel pyramid with 900 queries and 16 planes: gather alone 23.2 / 15.3 / 12.0.  At c2-10k
    // 40..80 records are all the same (64.9-65.8) and more only lengthens the tail (96: 73, 128: 81, 256: 102).
    {
        long long win = (long long)(pairs * samples) / (2048LL * nug) / 8 * 8;
        win = win < kWinMin ? kWinMin : win > kWinMax ? kWinMax : win;
        if (option_gather_win() >= 8) win = option_gather_win();
        w.win = (int)win;
    }
    w.win_cap = (int)((samples + w.win - 1) / w.win);
    w.cont_cap = (w.win_cap + nug - 1) / nug + 1;
    // query slices per plane: enough workgroups to fill the chip, at least ~2k samples each
    int64_t ns = pairs ? (int64_t)((256 + pairs - 1) / pairs) : 1;
    const int64_t by_work = (int64_t)((samples + 2047) / 2048);
    if (ns > by_work) ns = by_work;
    // ... and no more slices than the per-slice cell tables are worth: every slice writes, scans and reads back a
    // table of ~I counters, so with few samples per pixel the tables outweigh the samples.  About 36 bytes of sample
    // traffic against 16 bytes of table traffic per cell and slice; measured on a 256..32 px pyramid, 8 planes
    // (grad_value us): Q=20 000: 32 slices 288, 16 slices 255, 8 slices 274; Q=5 000: 32 slices 196, 8 slices 172,
    // 4 slices 183.
    const int64_t by_tables = I > 0 ? (int64_t)((9 * samples) / (2 * (size_t)I)) : ns;
    if (ns > by_tables) ns = by_tables;
    if (option_cell_slices() > 0) ns = option_cell_slices();
    if (ns > 64) ns = 64;
    if (ns > q_round) ns = q_round;
    if (ns < 1) ns = 1;
    w.nsplit = (int)ns;
    const size_t entry_bytes = acc_bytes == 8 ? 32 : 16;
    size_t o = 0;
    w.off_part = o;     o = align_up(o + pairs * w.nsplit * (size_t)w.nc_cap * 4, 256);
    w.off_blocktot = o; o = align_up(o + pairs * w.nsplit * (size_t)w.nblk_cap * 4, 256);
    w.off_off = o;      o = align_up(o + pairs * ((size_t)w.nc_cap + 1) * 4, 256);
    w.off_total = o;    o = align_up(o + pairs * 4, 256);
    w.off_meta = o;     o = align_up(o + 256, 256);
    w.ent_n0 = w.ent_n1 = w.ent_n2 = 0;
    if (records_in_grads && samples > 0) {
        const size_t plane_bytes = samples * entry_bytes;
        const size_t loc_bytes = (size_t)(B * Q * H * L * P) * 2 * elem_bytes, attn_bytes = loc_bytes / 2;
        const size_t value_bytes = (size_t)(B * I * H * D) * (value_elem_bytes ? value_elem_bytes : elem_bytes);
        size_t n0 = loc_bytes / plane_bytes, n1 = attn_bytes / plane_bytes, n2 = w.rounds == 1 ? value_bytes / plane_bytes : 0;
        if (n0 > pairs) n0 = pairs;
        if (n1 > pairs - n0) n1 = pairs - n0;
        if (n2 > pairs - n0 - n1) n2 = pairs - n0 - n1;
        w.ent_n0 = (int)n0;
        w.ent_n1 = (int)n1;
        w.ent_n2 = (int)n2;
    }
    w.off_entries = o;  o = align_up(o + (pairs - w.ent_n0 - w.ent_n1 - w.ent_n2) * samples * entry_bytes, 256);
    w.off_scratch = o;  o = align_up(o + pairs * (size_t)I * 4 * (size_t)D * acc_bytes, 256);
    w.off_cont = o;     o = align_up(o + pairs * (size_t)w.cont_cap * 4 * (size_t)D * acc_bytes, 256);
    w.off_accum = o;    if (w.rounds > 1) o = align_up(o + pairs * (size_t)I * (size_t)D * acc_bytes, 256);
    w.total = o;
    return w;
}

}  // namespace msda
