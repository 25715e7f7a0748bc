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
//               (ds_add_u32), written out as part[plane][j][cell]
//   K2a total   per cell: records in the cell (sum over the slices), part[j][cell] -> slice j's first slot inside
//               the cell's list; per block of 256 cells: (records, work items)
//   K2b scan    per block of 256 cells: base = sums of the preceding blocks, exclusive scan inside the block:
//               off[cell] = first record, cellitem[cell] = first work item; a *work item* is a window of <= kChunk
//               records of ONE cell (first record, count, continuation / last flags, cell)
//   K3 place    (plane, j): cursor[cell] = off[cell] + part[j][cell] in LDS; every sample ->
//               entries[cursor[cell]++] = {q, dx, dy, a}
//   K4 gather   one G-lane group per work item, G records at a time: each lane fetches one record and turns it
//               into (grad_out row offset, four corner weights); the group issues the grad_out row loads (16 bytes
//               per lane, ONE load per sample) back to back and FMAs each row into four corner accumulators.
//               Items of one cell inside a workgroup are summed through LDS, x-neighbouring one-item cells share
//               rows; the leaders store 4 (or 2) partial rows scratch[item][corner]
//   K5 finish   per pixel: sum the partial rows of its four incident cells (corner 00 of cell (x, y), 01 of
//               (x-1, y), 10 of (x, y-1), 11 of (x-1, y-1)) in a fixed order, store the grad_value row
//
// Every grad_value row is written exactly once by plain stores (no memset); hot pixels of coarse
// levels are split into kChunk-entry work items so the load stays balanced whatever the sampling
// distribution.  Padding semantics: "zeros" drops samples/corners outside the image; "border" clips
// the pixel coordinate to [0, size-1] first (grid_sample), which puts the whole weight on the edge
// pixel exactly as the reference's clamped corners do.
#pragma once

#include "msda_kernels.hpp"

namespace msda {

constexpr int kChunk = 64;           // entries per work item
constexpr int kGatherItemBlock = 256; // threads per workgroup of the gather kernel (no block barriers)
constexpr int kContFlag = 1 << 30;   // work-item record: this window continues the previous item's cell
constexpr int kLastFlag = 1 << 29;   // work-item record: last window of its cell (kLastFlag without kContFlag: the
                                     // cell is ONE item, and may share rows with a neighbouring one-item cell)
constexpr int kCountMask = kLastFlag - 1;
constexpr int kBigChunks = 16;       // cells with more work items than this have their records written by the whole block
constexpr int kBigCells = 64;        // ... at most this many per plane (the rest falls back to the owning thread)
constexpr int kCellBlock = 1024;     // threads of K1 / K2b / K3
constexpr int kCellLdsInts = 36864;  // cells a workgroup keeps in LDS at a time (144 KiB)

template <typename A> struct alignas(16) Entry {
    uint32_t q;
    A dx, dy, a;
};

// sample -> (cell id inside the plane, fractional offsets).  false: the sample touches no pixel.
template <typename A>
__device__ __forceinline__ bool sample_cell(A x, A y, int h, int w, int cstart, bool zeros, bool align, int &cell,
                                            A &dx, A &dy)
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
    cell = cstart + (int)mul24((uint32_t)((int)y0 + 1), (uint32_t)(w + 1)) + ((int)x0 + 1);
    return true;
}

__device__ __forceinline__ int plane_cells(const LevelTab &tab, int L)
{
    return tab.cstart[L - 1] + (tab.h[L - 1] + 1) * (tab.w[L - 1] + 1);
}

// ------------------------------------------------------------------------------------------
// K1 / K3: one pass over the samples of a (plane, query slice).  PLACE=false counts, true places.
// ------------------------------------------------------------------------------------------
template <typename T, bool PLACE>
__global__ __launch_bounds__(kCellBlock) void msda_cell_pass_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    int pair, slice;
    if (!decode_block(p.grid3d, p.B * p.H, p.nsplit, p.xcd_map, pair, slice)) return;
    const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
    const int qper = (p.Q + p.nsplit - 1) / p.nsplit;
    const int qa = min(p.Q, slice * qper), qb = min(p.Q, qa + qper);

    LevelTab *tab = reinterpret_cast<LevelTab *>(msda_smem);
    int *s_cell = reinterpret_cast<int *>(msda_smem + sizeof(LevelTab));
    load_level_table(tab, p.shapes, p.L);
    __syncthreads();
    // the workspace was sized on the host from I alone (nc_cap = 2 I + 2 L); shapes that disagree with I (or
    // zero-sized levels) must not push the cell tables past it: cells beyond the capacity are dropped
    const int ncells = min(plane_cells(*tab, p.L), p.nc_cap);
    const int cap = p.cell_cap;
    if constexpr (!PLACE)
        if (threadIdx.x == 0) p.ws_meta[0] = ncells;  // (every workgroup writes the same value) for the scan kernels

    int *part = p.ws_part + (size_t)pair * p.nsplit * p.nc_cap;  // [slice][cell]
    const int *off = p.ws_off + (size_t)pair * (p.nc_cap + 1);
    Entry<A> *entries = static_cast<Entry<A> *>(p.ws_entries) + (size_t)pair * p.Q * p.LP;
    // per-plane bases (64-bit, uniform) + 32-bit per-sample offsets (the host checks Q*H*L*P*2 < 2^31)
    const size_t plane_s0 = ((size_t)b * p.Q * p.H + h) * p.LP;
    const T *loc = static_cast<const T *>(p.loc) + 2 * plane_s0;
    const T *attn = static_cast<const T *>(p.attn) + plane_s0;
    const int HLP = p.H * p.LP;
    const float inv_P = 1.0f / (float)p.P;
    const int tid = threadIdx.x;
    const int dq = kCellBlock / p.LP, dr = kCellBlock - dq * p.LP;

    for (int c0 = 0; c0 < ncells; c0 += cap) {  // one trip unless the plane has more cells than fit in LDS
        const int n = min(cap, ncells - c0);
        for (int i = tid; i < n; i += kCellBlock) {
            int v = 0;
            if constexpr (PLACE)  // this slice's first slot in every cell list
                v = off[c0 + i] + part[(size_t)slice * p.nc_cap + c0 + i];
            s_cell[i] = v;
        }
        __syncthreads();
        // software-pipelined walk: the next sample's (x, y, a) are requested before this sample's entry is
        // stored, so the wait for them never has to drain the scattered store behind it (one vmcnt queue)
        int q = qa + tid / p.LP, sl = tid % p.LP;
        int sidx = q * HLP + sl;  // sample offset inside the plane, advanced incrementally
        const int d_sidx = dq * HLP + dr;
        Pack<T, 2> xy, xy_n;
        T at = TR::from_acc((A)0), at_n = at;
        xy.v[0] = xy.v[1] = at;
        xy_n = xy;
        if (q < qb) {
            xy = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
            if constexpr (PLACE) at = attn[sidx];
        }
        while (q < qb) {
            int qn = q + dq, sn = sl + dr;
            sidx += d_sidx;
            if (sn >= p.LP) {
                sn -= p.LP;
                ++qn;
                sidx += HLP - p.LP;
            }
            if (qn < qb) {
                xy_n = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
                if constexpr (PLACE) at_n = attn[sidx];
            }
            const int l = div_small(sl, p.P, inv_P);
            int cell;
            A dx, dy;
            if (sample_cell<A>(TR::to_acc(xy.v[0]), TR::to_acc(xy.v[1]), tab->h[l], tab->w[l], tab->cstart[l], p.zeros,
                               p.align, cell, dx, dy)) {
                const unsigned rel = (unsigned)(cell - c0);
                if (rel < (unsigned)n) {
                    if constexpr (!PLACE) {
                        atomicAdd(&s_cell[rel], 1);
                    } else {
                        const int pos = atomicAdd(&s_cell[rel], 1);
                        Entry<A> e;
                        e.q = (uint32_t)q;
                        e.dx = dx;
                        e.dy = dy;
                        e.a = TR::to_acc(at);
                        entries[pos] = e;  // (plain store: these partial lines must merge in L2 — streaming stores: 57 -> 202 us)
                    }
                }
            }
            q = qn;
            sl = sn;
            xy = xy_n;
            at = at_n;
        }
        __syncthreads();
        if constexpr (!PLACE) {
            for (int i = tid; i < n; i += kCellBlock) part[(size_t)slice * p.nc_cap + c0 + i] = s_cell[i];
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// K2a / K2b: the per-plane scans, as a three-phase parallel scan over blocks of kBlock cells (a plane's cell
// table is a few thousand to tens of thousands of entries: one workgroup per plane left 240 CUs idle).
//   K2a  per cell: off[cell] = records in the cell (sum over the query slices) and part[j][cell] becomes slice j's
//        first slot relative to the start of the cell's list; per block: (records, work items) -> blocksum
//   K2b  per block: base = sum of the preceding blocks' sums (a few dozen loads), exclusive scan inside the block:
//        off[cell] = first record, cellitem[cell] = first work item, work-item records of the block's cells
// ------------------------------------------------------------------------------------------
// exclusive scan of one int over the kCellBlock threads of a workgroup (s_wave: kCellBlock / kWave ints of LDS)
__device__ __forceinline__ int block_exclusive_scan(int v, int *s_wave, int &total)
{
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    int inc = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int n = __shfl_up(inc, d, kWave);
        if (lane >= d) inc += n;
    }
    __syncthreads();  // s_wave may still be in use by a previous scan
    if (lane == kWave - 1) s_wave[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kCellBlock / kWave; ++i) {
        const int s = s_wave[i];
        if (i < wid) base += s;
        tot += s;
    }
    total = tot;
    return base + inc - v;
}

// exclusive scan of two ints over the kBlock threads of a workgroup (s: 2 * kBlock / kWave ints of LDS)
__device__ __forceinline__ void block_scan2(int &a, int &b, int *s, int &tot_a, int &tot_b)
{
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    int ia = a, ib = b;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int na = __shfl_up(ia, d, kWave), nb = __shfl_up(ib, d, kWave);
        if (lane >= d) {
            ia += na;
            ib += nb;
        }
    }
    __syncthreads();  // s may still be in use by a previous scan
    if (lane == kWave - 1) {
        s[2 * wid] = ia;
        s[2 * wid + 1] = ib;
    }
    __syncthreads();
    int base_a = 0, base_b = 0;
    tot_a = tot_b = 0;
#pragma unroll
    for (int i = 0; i < kBlock / kWave; ++i) {
        const int sa = s[2 * i], sb = s[2 * i + 1];
        if (i < wid) {
            base_a += sa;
            base_b += sb;
        }
        tot_a += sa;
        tot_b += sb;
    }
    a = base_a + ia - a;
    b = base_b + ib - b;
}

template <typename Tag> __global__ __launch_bounds__(kBlock) void msda_cell_total_kernel(const Params p)
{
    const int per_plane = (p.nc_cap + kBlock - 1) / kBlock;
    const int pair = blockIdx.x / per_plane, blk = blockIdx.x - pair * per_plane;
    const int nc = p.ws_meta[0];  // real cell count of a plane, left by the count pass
    if (blk * kBlock >= nc) return;  // block-uniform
    __shared__ int s_scan[2 * kBlock / kWave];
    const int c = blk * kBlock + threadIdx.x;
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
        p.ws_off[(size_t)pair * (p.nc_cap + 1) + c] = tot;
    }
    int ex_n = tot, ex_c = (tot + kChunk - 1) / kChunk, sum_n, sum_c;
    block_scan2(ex_n, ex_c, s_scan, sum_n, sum_c);
    if (threadIdx.x == 0) p.ws_blocksum[(size_t)pair * per_plane + blk] = make_int2(sum_n, sum_c);
}

template <typename Tag> __global__ __launch_bounds__(kBlock) void msda_cell_scan_kernel(const Params p)
{
    const int per_plane = (p.nc_cap + kBlock - 1) / kBlock;
    const int pair = blockIdx.x / per_plane, blk = blockIdx.x - pair * per_plane;
    const int nc = p.ws_meta[0];
    if (blk * kBlock >= nc) return;  // block-uniform
    __shared__ int s_scan[2 * kBlock / kWave];
    __shared__ int s_big[1 + 4 * kBigCells];  // [0] = count, then (first item, first record, records, cell) of the cells whose
                                              // work-item records the whole block writes
    const int t = threadIdx.x;
    if (t == 0) s_big[0] = 0;
    // base of this block: the sums of the blocks before it
    int base_n = 0, base_c = 0;
    {
        int an = 0, ac = 0;
        for (int i = t; i < blk; i += kBlock) {
            const int2 v = p.ws_blocksum[(size_t)pair * per_plane + i];
            an += v.x;
            ac += v.y;
        }
        int tn, tc;
        block_scan2(an, ac, s_scan, tn, tc);  // (only the totals are used)
        base_n = tn;
        base_c = tc;
    }
    int *off = p.ws_off + (size_t)pair * (p.nc_cap + 1);
    int *cellitem = p.ws_cellitem + (size_t)pair * (p.nc_cap + 1);
    int4 *items = p.ws_items + (size_t)pair * p.it_cap;
    const int c = blk * kBlock + t;
    const int n = c < nc ? off[c] : 0;
    const int chunks = (n + kChunk - 1) / kChunk;
    int ex_n = n, ex_c = chunks, sum_n, sum_c;
    block_scan2(ex_n, ex_c, s_scan, sum_n, sum_c);  // (its barriers also publish s_big[0] = 0)
    const int beg = base_n + ex_n, first = base_c + ex_c;
    if (c < nc) {
        off[c] = beg;
        cellitem[c] = first;
        int slot = kBigCells;
        if (chunks > kBigChunks) slot = atomicAdd(&s_big[0], 1);
        if (chunks <= kBigChunks || slot >= kBigCells) {
            for (int k = 0; k < chunks; ++k)
                items[first + k] = make_int4(beg + k * kChunk,
                                             min(kChunk, n - k * kChunk) | (k ? kContFlag : 0) | (k == chunks - 1 ? kLastFlag : 0),
                                             c, 0);
        } else {
            s_big[1 + 4 * slot] = first;
            s_big[2 + 4 * slot] = beg;
            s_big[3 + 4 * slot] = n;
            s_big[4 + 4 * slot] = c;
        }
        if (c == nc - 1) {  // the plane's totals behind the last cell
            off[nc] = beg + n;
            cellitem[nc] = first + chunks;
            p.ws_itemcnt[pair] = first + chunks;
        }
    }
    __syncthreads();
    const int nbig = min(s_big[0], kBigCells);
    for (int i = 0; i < nbig; ++i) {  // hot cells (coarse levels, clustered samples): records written by all threads
        const int bfirst = s_big[1 + 4 * i], bbeg = s_big[2 + 4 * i], bn = s_big[3 + 4 * i], bcell = s_big[4 + 4 * i];
        const int bchunks = (bn + kChunk - 1) / kChunk;
        for (int k = t; k < bchunks; k += kBlock)
            items[bfirst + k] = make_int4(bbeg + k * kChunk,
                                          min(kChunk, bn - k * kChunk) | (k ? kContFlag : 0) | (k == bchunks - 1 ? kLastFlag : 0),
                                          bcell, 0);
    }
}

// ------------------------------------------------------------------------------------------
// K4: gather.  One G-lane group per work item (a window of <= kChunk records of ONE cell), VEC channels per lane.
// A sample's grad_out row is loaded once and blended into the cell's four corner rows; the four partial rows
// go to scratch[item][corner].
// ------------------------------------------------------------------------------------------
template <typename A> struct alignas(16) CornerW {
    A w[4];  // a * {(1-dx)(1-dy), dx(1-dy), (1-dx)dy, dx dy}: corners 00, 01, 10, 11
};

template <typename T, int VEC, int G>
__global__ __launch_bounds__(kGatherItemBlock) void msda_value_gather_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    constexpr int NU = kGatherItemBlock / G;
    constexpr int UB = G < 8 ? G : 8;  // row loads in flight per lane
    const int slots = (p.it_cap + NU - 1) / NU;
    int pair, slot;
    if (!decode_block(p.grid3d, p.B * p.H, slots, p.xcd_map, pair, slot)) return;
    const int nitems = p.ws_itemcnt[pair];
    if (slot * NU >= nitems) return;  // block-uniform
    const int tid = threadIdx.x;
    const int unit = tid / G, j = tid % G;
    const int item = slot * NU + unit;
    const bool valid = item < nitems;  // idle groups still take part in the block barriers below
    const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
    const int gbase = tid - j;

    __shared__ CornerW<A> s_w[kGatherItemBlock];
    __shared__ uint32_t s_q[kGatherItemBlock];
    __shared__ __attribute__((aligned(32))) A s_rows[kGatherItemBlock * 4 * VEC];  // [unit][corner][G * VEC]: partial rows of continuation items
    __shared__ int s_cont[NU];

    const Entry<A> *entries = static_cast<const Entry<A> *>(p.ws_entries) + (size_t)pair * p.Q * p.LP;
    const T *gout = static_cast<const T *>(p.grad_out) + ((size_t)b * p.Q * p.H + h) * p.D;  // uniform base
    const uint32_t q_stride = (uint32_t)(p.H * p.D) * (uint32_t)sizeof(T);  // bytes; Q*H*D*sizeof < 2^31 is checked on the host
    // grad_out rows of this plane through a buffer descriptor: scalar base + 32-bit byte offset per lane
    const rsrc_t rs_go = make_rsrc(gout, (uint32_t)(((size_t)p.Q * p.H * p.D - (size_t)h * p.D) * sizeof(T)));
    const int nchan_chunks = (p.D + G * VEC - 1) / (G * VEC);

    int start = 0, count = 0;  // valid items: 1 <= count <= kChunk
    bool follower = false;     // this item continues the cell of the previous item of the same workgroup
    // One-item cells that are neighbours in x AND in this workgroup share rows: the right one takes over the left
    // one's right-hand corners (its corner 01 is the neighbour's 00, its 11 the neighbour's 10), so a cell then
    // leaves two rows instead of four.  The finish kernel derives the same predicate from the item indices.
    bool give_right = false, take_left = false;
    if (valid) {
        const int4 *recs = p.ws_items + (size_t)pair * p.it_cap;
        const int4 rec = recs[item];
        const int4 rec_l = recs[max(item - 1, 0)], rec_r = recs[min(item + 1, nitems - 1)];  // same round trip
        start = rec.x;
        count = rec.y & kCountMask;
        follower = (rec.y & kContFlag) != 0 && unit != 0;
        auto single = [](const int4 &r) { return (r.y & (kContFlag | kLastFlag)) == kLastFlag; };
        give_right = unit + 1 < NU && item + 1 < nitems && single(rec) && single(rec_r) && rec_r.z == rec.z + 1;
        take_left = unit > 0 && single(rec_l) && single(rec) && rec.z == rec_l.z + 1;
    }

    // record v of the window -> (byte offset of the query's grad_out row inside the plane, four corner weights).
    // Positions past the end get weight 0 and an out-of-range offset (the buffer load then returns 0 without
    // touching memory), so the unrolled batches need no tail loop.
    auto convert = [&](const Entry<A> &e, int v, uint32_t &q, CornerW<A> &cw) {
        const bool ok = v < count;
        const A a = ok ? e.a : (A)0;
        const A ax1 = a * e.dx, ax0 = a - ax1;
        q = ok ? e.q * q_stride : 0x80000000u;
        cw.w[3] = ax1 * e.dy;
        cw.w[2] = ax0 * e.dy;
        cw.w[1] = ax1 - cw.w[3];
        cw.w[0] = ax0 - cw.w[2];
    };

    A *scratch = static_cast<A *>(p.ws_scratch) + ((size_t)pair * p.it_cap + item) * 4 * p.D;
    for (int cc = 0; cc < nchan_chunks; ++cc) {
        const int c0 = (cc * G + j) * VEC;
        const bool lane_ok = c0 < p.D;
        const uint32_t lane_elem = (lane_ok ? (uint32_t)c0 : 0u) * (uint32_t)sizeof(T);
        A acc[4][VEC];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[k][i] = (A)0;

        Entry<A> e_cur = entries[start + max(min(j, count - 1), 0)];  // (idle groups: count == 0, record 0 of the plane, unused)
        for (int v0 = 0; v0 < count; v0 += G) {
            uint32_t cur_q;
            CornerW<A> cur_w;
            convert(e_cur, v0 + j, cur_q, cur_w);
            wave_lds_sync();  // the previous batch's records have been read
            s_q[tid] = cur_q;
            s_w[tid] = cur_w;
            // the next batch's record (clamped: always a valid address) is in flight while this batch is consumed
            e_cur = entries[start + min(v0 + G + j, count - 1)];
            wave_lds_sync();
            const int cnt = min(G, count - v0);
#pragma unroll
            for (int jj = 0; jj < G; jj += UB) {  // UB row loads issued back to back, then consumed
                if (jj < cnt) {                   // uniform per group; G == UB: always true
                    Pack<T, VEC> g[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u)
                        g[u] = __builtin_bit_cast(Pack<T, VEC>,
                                                  RawLoad<sizeof(T) * VEC>::load(rs_go, s_q[gbase + jj + u] + lane_elem));
                    __builtin_amdgcn_sched_barrier(0);  // keep the UB loads together: hipcc otherwise serialises some
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const CornerW<A> w = s_w[gbase + jj + u];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int v = 0; v < VEC; ++v) acc[k][v] = fma_t(w.w[k], TR::to_acc(g[u].v[v]), acc[k][v]);
                    }
                }
            }
        }
        // Items of one cell that sit in the same workgroup are summed here (in item order) and leave as ONE set of
        // four rows, stored at the first of them: hot cells (coarse levels, clustered samples) then cost the finish
        // kernel one row per NU items, not one per item.
        if (cc > 0) __syncthreads();  // the previous channel chunk's rows have been consumed
        if (j == 0) s_cont[unit] = follower ? 1 : 0;
        if (follower || give_right) {  // (a one-item cell is never a follower: the two cases are disjoint)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (follower || (k & 1)) {
                    Pack<A, VEC> o;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o.v[v] = acc[k][v];
                    *reinterpret_cast<Pack<A, VEC> *>(&s_rows[((unit * 4 + k) * G + j) * VEC]) = o;
                }
            }
        }
        __syncthreads();
        if (valid && !follower) {
            for (int u = unit + 1; u < NU && s_cont[u]; ++u) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const Pack<A, VEC> r = *reinterpret_cast<const Pack<A, VEC> *>(&s_rows[((u * 4 + k) * G + j) * VEC]);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[k][v] += r.v[v];
                }
            }
            if (take_left) {  // the left neighbour's corners 01 / 11 are this cell's 00 / 10
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    const Pack<A, VEC> r =
                        *reinterpret_cast<const Pack<A, VEC> *>(&s_rows[(((unit - 1) * 4 + k + 1) * G + j) * VEC]);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[k][v] += r.v[v];
                }
            }
            if (lane_ok) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if ((k & 1) && give_right) continue;  // taken over by the right neighbour
                    Pack<A, VEC> o;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o.v[v] = acc[k][v];
                    // written once, read once by the finish kernel: keep it from displacing grad_out rows in L2
                    store_stream(scratch + (size_t)k * p.D + c0, o);
                }
            }
        }
    }
}

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

// ------------------------------------------------------------------------------------------
// K5: per pixel: sum the partial rows of its four incident cells (the pixel is corner 00 of cell (x, y), 01 of
// (x-1, y), 10 of (x, y-1), 11 of (x-1, y-1)), in cell and chunk order; every grad_value row is stored here.
// ------------------------------------------------------------------------------------------
template <typename T, int VEC, int G>
__global__ __launch_bounds__(kBlock) void msda_value_finish_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    constexpr int NU = kBlock / G;
    const int slots = (p.I + NU - 1) / NU;
    int pair, slot;
    if (!decode_block(p.grid3d, p.B * p.H, slots, p.xcd_map, pair, slot)) return;
    __shared__ LevelTab tab;
    load_level_table(&tab, p.shapes, p.L);
    __syncthreads();
    const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
    const int unit = threadIdx.x / G, j = threadIdx.x % G;
    const int pix = slot * NU + unit;
    if (pix >= p.I) return;
    int l = 0;
    while (l < p.L - 1 && pix >= tab.start[l + 1]) ++l;
    const int rel = pix - tab.start[l], w = tab.w[l], cw = w + 1;
    const int y = rel / w, x = rel - y * w;
    const int c11 = tab.cstart[l] + y * cw + x;  // cell (x-1, y-1)
    const int *cellitem = p.ws_cellitem + (size_t)pair * (p.nc_cap + 1);
    // item ranges of the cells (x-1,y-1), (x,y-1) | (x-1,y), (x,y): consecutive cell ids, consecutive item ranges.
    // A pixel the shapes tensor does not describe (sum h*w != I), or one whose cells were dropped for lack of
    // workspace, has no cells: all ranges empty, the row is stored as zeros.
    const bool in_tab = y < tab.h[l] && c11 + cw + 2 <= min(plane_cells(tab, p.L), p.nc_cap);
    const int t0 = in_tab ? cellitem[c11] : 0, t1 = in_tab ? cellitem[c11 + 1] : 0, t2 = in_tab ? cellitem[c11 + 2] : 0;
    const int u0 = in_tab ? cellitem[c11 + cw] : 0, u1 = in_tab ? cellitem[c11 + cw + 1] : 0,
              u2 = in_tab ? cellitem[c11 + cw + 2] : 0;
    const A *src = static_cast<const A *>(p.ws_scratch) + (size_t)pair * p.it_cap * 4 * p.D;
    // the plane's partial rows through a buffer descriptor: an offset of 0x80000000 is out of range and reads 0,
    // which is how an empty cell contributes nothing without a branch
    const size_t plane_bytes = (size_t)p.it_cap * 4 * p.D * sizeof(A);
    const bool use_rsrc = plane_bytes < ((size_t)1 << 31);
    const rsrc_t rs = make_rsrc(src, (uint32_t)(use_rsrc ? plane_bytes : 0));
    const uint32_t row_bytes = (uint32_t)p.D * (uint32_t)sizeof(A);
    const bool simple = use_rsrc && (t1 - t0) <= 1 && (t2 - t1) <= 1 && (u1 - u0) <= 1 && (u2 - u1) <= 1;
    // Two one-item cells that are neighbours in x and sit in the same gather workgroup share rows (see the gather
    // kernel): the left cell's corners 01 / 11 are already inside the right cell's rows 00 / 10 and were not stored.
    constexpr int NUG = kGatherItemBlock / G;
    const bool comb_top = (u1 - u0) == 1 && (u2 - u1) == 1 && (u0 % NUG) != NUG - 1;  // cells (x-1, y) and (x, y)
    const bool comb_bot = (t1 - t0) == 1 && (t2 - t1) == 1 && (t0 % NUG) != NUG - 1;  // cells (x-1, y-1) and (x, y-1)
    const int nchan_chunks = (p.D + G * VEC - 1) / (G * VEC);
    for (int cc = 0; cc < nchan_chunks; ++cc) {
        const int c0 = (cc * G + j) * VEC;
        if (c0 >= p.D) continue;
        A acc[VEC];
        if (simple) {  // at most one work item per cell (fine levels): four independent loads
            const uint32_t lane_off = (uint32_t)c0 * (uint32_t)sizeof(A);
            const uint32_t o0 = u2 > u1 ? ((uint32_t)u1 * 4 + 0) * row_bytes + lane_off : 0x80000000u;
            const uint32_t o1 = (u1 > u0 && !comb_top) ? ((uint32_t)u0 * 4 + 1) * row_bytes + lane_off : 0x80000000u;
            const uint32_t o2 = t2 > t1 ? ((uint32_t)t1 * 4 + 2) * row_bytes + lane_off : 0x80000000u;
            const uint32_t o3 = (t1 > t0 && !comb_bot) ? ((uint32_t)t0 * 4 + 3) * row_bytes + lane_off : 0x80000000u;
            const Pack<A, VEC> r0 = load_acc_pack<A, VEC>(rs, o0);
            const Pack<A, VEC> r1 = load_acc_pack<A, VEC>(rs, o1);
            const Pack<A, VEC> r2 = load_acc_pack<A, VEC>(rs, o2);
            const Pack<A, VEC> r3 = load_acc_pack<A, VEC>(rs, o3);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = ((r0.v[v] + r1.v[v]) + r2.v[v]) + r3.v[v];
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = (A)0;
            // rows of a cell: its first item, then the first item of every further gather workgroup it extends into
            auto add = [&](int first, int last, int corner) {
                for (int it = first; it < last; it = (it / NUG + 1) * NUG) {
                    const Pack<A, VEC> r =
                        *reinterpret_cast<const Pack<A, VEC> *>(src + ((size_t)it * 4 + corner) * p.D + c0);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] += r.v[v];
                }
            };
            add(u1, u2, 0);
            if (!comb_top) add(u0, u1, 1);
            add(t1, t2, 2);
            if (!comb_bot) add(t0, t1, 3);
        }
        Pack<T, VEC> o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o.v[v] = TR::from_acc(acc[v]);
        T *dst = static_cast<T *>(p.grad_value) + (((size_t)b * p.I + pix) * p.H + h) * p.D + c0;
        store_stream(dst, o);
    }
}

// ------------------------------------------------------------------------------------------
// workspace layout (host + device agree through these helpers)
// ------------------------------------------------------------------------------------------
struct SortedWsLayout {
    int nc_cap, it_cap, nsplit;
    size_t off_part, off_off, off_cellitem, off_itemcnt, off_meta, off_blocksum, off_items, off_entries, off_scratch, total;
};

int option_cell_slices();  // 0: automatic (msda_api.hip)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline SortedWsLayout sorted_ws_layout(int64_t B, int64_t I, int64_t H, int64_t D, int64_t Q, int64_t L, int64_t P,
                                       size_t acc_bytes)
{
    SortedWsLayout w;
    const size_t pairs = (size_t)(B * H);
    const size_t samples = (size_t)(Q * L * P);  // per plane
    w.nc_cap = (int)(2 * I + 2 * L);             // (w+1)(h+1) <= 2wh + 2 per level
    // work items: sum over cells of ceil(n / kChunk) <= cells + samples / kChunk, and never more than the samples
    size_t items = (size_t)w.nc_cap + samples / kChunk + 1;
    if (items > samples) items = samples > 0 ? samples : 1;
    w.it_cap = (int)items;
    // query slices per plane: enough workgroups to fill the chip, at least ~2k samples each
    int64_t ns = pairs ? (int64_t)((256 + pairs - 1) / pairs) : 1;
    const int64_t by_work = (int64_t)((samples + 2047) / 2048);
    if (ns > by_work) ns = by_work;
    if (option_cell_slices() > 0) ns = option_cell_slices();
    if (ns > 64) ns = 64;
    if (ns > Q) ns = Q;
    if (ns < 1) ns = 1;
    w.nsplit = (int)ns;
    const size_t entry_bytes = acc_bytes == 8 ? 32 : 16;
    size_t o = 0;
    w.off_part = o;     o = align_up(o + pairs * w.nsplit * (size_t)w.nc_cap * 4, 256);
    w.off_off = o;      o = align_up(o + pairs * ((size_t)w.nc_cap + 1) * 4, 256);
    w.off_cellitem = o; o = align_up(o + pairs * ((size_t)w.nc_cap + 1) * 4, 256);
    w.off_itemcnt = o;  o = align_up(o + pairs * 4, 256);
    w.off_meta = o;     o = align_up(o + 256, 256);
    w.off_blocksum = o; o = align_up(o + pairs * (((size_t)w.nc_cap + kBlock - 1) / kBlock) * 8, 256);
    w.off_items = o;    o = align_up(o + pairs * (size_t)w.it_cap * 16, 256);
    w.off_entries = o;  o = align_up(o + pairs * samples * entry_bytes, 256);
    w.off_scratch = o;  o = align_up(o + pairs * (size_t)w.it_cap * 4 * (size_t)D * acc_bytes, 256);
    w.total = o;
    return w;
}

}  // namespace msda
