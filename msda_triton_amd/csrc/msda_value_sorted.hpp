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
//   K2a total   off[cell] = sum_j part[j][cell]                       (all cells of all planes in parallel)
//   K2b scan    per plane, in LDS: off = exclusive_scan(off); per pixel: the four incident cell lists
//               (start, length), chunks = max(1, ceil(n / kChunk)); work items (pixel, chunk) by a second scan
//   K3 place    (plane, j): cursor[cell] = off[cell] + sum_{j'<j} part[j'][cell] in LDS; every sample ->
//               entries[cursor[cell]++] = {q, dx, dy, a}
//   K4 gather   one G-lane group per work item: walk its window of the pixel's four cell lists (as one
//               virtual list), G entries at a time: each lane fetches one record and turns it into
//               (q, a*fx*fy); the group passes these lane to lane, issues the grad_out row loads (16 bytes per
//               lane, the forward's gather shape) back to back and FMAs into registers; single-chunk pixels
//               store their grad_value row directly, multi-chunk pixels park a partial row in scratch
//   K5 finish   pixels with several chunks: sum their partial rows in chunk order, store
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
constexpr int kItemsPerGroup = 1;    // work items a gather group handles back to back
constexpr int kGatherItemBlock = 256; // threads per workgroup of the gather kernel (no LDS, no barriers)
constexpr int kItemBuckets = 9;      // work items are bucketed by ceil(entries / 8) = 0..8
constexpr int kItemMeta = 32;        // ints of per-plane item metadata: total, bucket starts, bucket cursors
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
    const int ncells = plane_cells(*tab, p.L);
    const int cap = p.cell_cap;

    int *part = p.ws_part + (size_t)pair * p.nsplit * p.nc_cap;  // [slice][cell]
    const int *off = p.ws_off + (size_t)pair * (p.nc_cap + 1);
    Entry<A> *entries = static_cast<Entry<A> *>(p.ws_entries) + (size_t)pair * p.Q * p.LP;
    const T *loc = static_cast<const T *>(p.loc);
    const T *attn = static_cast<const T *>(p.attn);
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
        Pack<T, 2> xy, xy_n;
        T at = TR::from_acc((A)0), at_n = at;
        xy.v[0] = xy.v[1] = at;
        xy_n = xy;
        if (q < qb) {
            const size_t sidx = ((size_t)(b * (size_t)p.Q + q) * p.H + h) * p.LP + sl;
            xy = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
            if constexpr (PLACE) at = attn[sidx];
        }
        while (q < qb) {
            int qn = q + dq, sn = sl + dr;
            if (sn >= p.LP) {
                sn -= p.LP;
                ++qn;
            }
            if (qn < qb) {
                const size_t sidx_n = ((size_t)(b * (size_t)p.Q + qn) * p.H + h) * p.LP + sn;
                xy_n = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx_n);
                if constexpr (PLACE) at_n = attn[sidx_n];
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
                        entries[pos] = e;
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
// K2a: off[cell] = number of samples in the cell (sum over the query slices), and part[j][cell] becomes
// the exclusive prefix over the slices (slice j's first slot relative to the start of the cell's list)
// ------------------------------------------------------------------------------------------
template <typename Tag> __global__ __launch_bounds__(kBlock) void msda_cell_total_kernel(const Params p)
{
    const int per_plane = (p.nc_cap + kBlock - 1) / kBlock;
    const int pair = blockIdx.x / per_plane;
    const int c = (blockIdx.x - pair * per_plane) * kBlock + threadIdx.x;
    if (c >= p.nc_cap) return;  // cells beyond the plane's real count hold garbage that nobody reads
    int *part = p.ws_part + (size_t)pair * p.nsplit * p.nc_cap + c;
    int tot = 0;
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

// ------------------------------------------------------------------------------------------
// K2b: per-plane scans (one 1024-thread workgroup per plane), cell offsets staged in LDS when they fit
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_exclusive_scan(int v, int *s_wave, int &total)
{
    // inclusive scan inside the wave, then across the 16 waves through LDS
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

// The scans proper.  `off` points either at the LDS copy or at global memory; the function is inlined
// at both call sites so every access has a known address space (a run-time pointer select would turn
// them all into slow FLAT operations).
__device__ __forceinline__ void cell_scan_body(int *off, const LevelTab &tab, int *s_wave, const Params &p, int pair,
                                               int nc, int *goff_copy)
{
    const int t = threadIdx.x;
    const int last = p.L - 1;
    // ---- A: exclusive scan of the per-cell totals ----
    {
        const int seg = (nc + kCellBlock - 1) / kCellBlock;
        const int lo = min(nc, t * seg), hi = min(nc, lo + seg);
        int sum = 0;
        for (int i = lo; i < hi; ++i) sum += off[i];
        int total;
        int run = block_exclusive_scan(sum, s_wave, total);
        for (int i = lo; i < hi; ++i) {
            const int c = off[i];
            off[i] = run;
            run += c;
        }
        if (t == 0) off[nc] = total;
    }
    __threadfence_block();
    __syncthreads();
    if (goff_copy != nullptr)
        for (int c = t; c <= nc; c += kCellBlock) goff_copy[c] = off[c];

    // ---- B: per-pixel list records and work items ----
    const int seg = (p.I + kCellBlock - 1) / kCellBlock;
    const int lo = min(p.I, t * seg), hi = min(p.I, lo + seg);
    int2 *pixinfo = p.ws_pixinfo + (size_t)pair * p.I;
    // entries of a pixel = entries of its four incident cells
    auto entries_of = [&](int pix, int &l) {
        while (l < last && pix >= tab.start[l + 1]) ++l;
        const int rel = pix - tab.start[l], w = tab.w[l], cw = w + 1;
        const int y = rel / w, x = rel - y * w;
        const int c11 = tab.cstart[l] + y * cw + x;  // cell (x0 = x-1, y0 = y-1): this pixel is its corner 11
        return (off[c11 + 1] - off[c11]) + (off[c11 + 2] - off[c11 + 1]) + (off[c11 + cw + 1] - off[c11 + cw]) +
               (off[c11 + cw + 2] - off[c11 + cw + 1]);
    };
    // Work items are executed in buckets of equal batch count (ceil(entries / 8) in 0..8) so that the groups of a
    // wave finish together; s_bucket counts the plane's items per bucket.
    int *s_bucket = s_wave + kCellBlock / kWave;  // 9 ints behind the scan scratch (see the kernel's declaration)
    if (t < kItemBuckets) s_bucket[t] = 0;
    // pass 1: count the work items of this thread's pixel segment
    int sum = 0, l = 0;
    int nb_full = 0;
    __syncthreads();
    for (int pix = lo; pix < hi; ++pix) {
        const int n = entries_of(pix, l);
        const int full = n / kChunk, rem = n - full * kChunk;
        sum += max(1, full + (rem > 0));
        nb_full += full;
        if (rem > 0 || n == 0) atomicAdd(&s_bucket[(rem + 7) / 8], 1);
    }
    if (nb_full) atomicAdd(&s_bucket[kItemBuckets - 1], nb_full);
    int total;
    int run = block_exclusive_scan(sum, s_wave, total);
    // pass 2: first item of every pixel (the records themselves are written by msda_item_kernel)
    l = 0;
    for (int pix = lo; pix < hi; ++pix) {
        const int n = entries_of(pix, l);
        const int chunks = max(1, (n + kChunk - 1) / kChunk);
        pixinfo[pix] = make_int2(run, chunks);
        run += chunks;
    }
    __syncthreads();
    if (t == 0) {
        int *meta = p.ws_itemcnt + (size_t)pair * kItemMeta;
        meta[0] = total;
        int start = 0;
        for (int k = 0; k < kItemBuckets; ++k) {
            meta[1 + k] = start;            // bucket start
            meta[1 + kItemBuckets + k] = 0;  // bucket cursor (msda_item_kernel reserves ranges with it)
            start += s_bucket[k];
        }
    }
}

template <typename Tag> __global__ __launch_bounds__(kCellBlock) void msda_cell_scan_kernel(const Params p)
{
    const int pair = blockIdx.x;
    __shared__ LevelTab tab;
    __shared__ int s_wave[kCellBlock / kWave + kItemBuckets];  // scan scratch + item bucket counters
    int *s_off = reinterpret_cast<int *>(msda_smem);
    load_level_table(&tab, p.shapes, p.L);
    __syncthreads();
    const int t = threadIdx.x;
    const int nc = plane_cells(tab, p.L);
    int *goff = p.ws_off + (size_t)pair * (p.nc_cap + 1);
    if (nc <= p.cell_cap) {  // the launch sized the dynamic LDS for cell_cap + 1 ints
        for (int c = t; c < nc; c += kCellBlock) s_off[c] = goff[c];
        __syncthreads();
        cell_scan_body(s_off, tab, s_wave, p, pair, nc, goff);
    } else {
        cell_scan_body(goff, tab, s_wave, p, pair, nc, nullptr);
    }
}

// ------------------------------------------------------------------------------------------
// K2c: work-item records, one thread per pixel (all planes in parallel).  Item k of a pixel covers
// positions [k*kChunk, (k+1)*kChunk) of the pixel's virtual list (its four cell lists back to back);
// the record holds that window already clipped against the four lists: (start, count) per list.
// ------------------------------------------------------------------------------------------
template <typename Tag> __global__ __launch_bounds__(kBlock) void msda_item_kernel(const Params p)
{
    const int slots = (p.I + kBlock - 1) / kBlock;
    int pair, slot;
    if (!decode_block(p.grid3d, p.B * p.H, slots, p.xcd_map, pair, slot)) return;
    __shared__ LevelTab tab;
    load_level_table(&tab, p.shapes, p.L);
    __syncthreads();
    const int pix_raw = slot * kBlock + threadIdx.x;
    const bool pix_ok = pix_raw < p.I;  // idle threads still take part in the barriers below
    const int pix = pix_ok ? pix_raw : p.I - 1;
    int l = 0;
    while (l < p.L - 1 && pix >= tab.start[l + 1]) ++l;
    const int rel = pix - tab.start[l], w = tab.w[l], cw = w + 1;
    const int y = rel / w, x = rel - y * w;
    const int c11 = tab.cstart[l] + y * cw + x;
    // list i = samples for which the pixel is corner i: 00 -> cell (x, y), 01 -> (x-1, y), 10 -> (x, y-1), 11
    const int cells[4] = {c11 + cw + 1, c11 + cw, c11 + 1, c11};
    const int *off = p.ws_off + (size_t)pair * (p.nc_cap + 1);
    int beg[4], len[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        beg[i] = off[cells[i]];
        len[i] = off[cells[i] + 1] - beg[i];
    }
    const int2 info = pix_ok ? p.ws_pixinfo[(size_t)pair * p.I + pix] : make_int2(0, 0);
    const int n = len[0] + len[1] + len[2] + len[3];
    const int full = n / kChunk, rem = n - full * kChunk;
    const int last_bucket = (rem + 7) / 8;             // bucket of the (possibly empty) partial item
    const bool has_last = pix_ok && (rem > 0 || n == 0);
    // reserve slots: first inside the workgroup (LDS), then one global add per bucket and workgroup
    __shared__ int s_cnt[kItemBuckets], s_base[kItemBuckets];
    if (threadIdx.x < kItemBuckets) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    int rank_full = 0, rank_last = 0;
    if (pix_ok && full) rank_full = atomicAdd(&s_cnt[kItemBuckets - 1], full);
    if (has_last) rank_last = atomicAdd(&s_cnt[last_bucket], 1);
    __syncthreads();
    int *meta = p.ws_itemcnt + (size_t)pair * kItemMeta;
    if (threadIdx.x < kItemBuckets) {
        const int c = s_cnt[threadIdx.x];
        s_base[threadIdx.x] = meta[1 + threadIdx.x] + (c ? atomicAdd(&meta[1 + kItemBuckets + threadIdx.x], c) : 0);
    }
    __syncthreads();
    if (!pix_ok) return;
    int4 *items = p.ws_items + (size_t)pair * p.it_cap * 3;
    for (int k = 0; k < info.y; ++k) {
        const int w0 = k * kChunk, w1 = w0 + kChunk;
        int st[4], cn[4], pos = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int lo = max(w0 - pos, 0), hi = min(w1 - pos, len[i]);
            st[i] = beg[i] + lo;
            cn[i] = max(hi - lo, 0);
            pos += len[i];
        }
        const bool is_full = k < full;
        const int slot = is_full ? s_base[kItemBuckets - 1] + rank_full + k : s_base[last_bucket] + rank_last;
        int4 *rec = items + (size_t)slot * 3;
        rec[0] = make_int4(pix, info.y, info.x + k, 0);  // pixel, chunks of the pixel, scratch row of this chunk
        rec[1] = make_int4(st[0], st[1], st[2], st[3]);
        rec[2] = make_int4(cn[0], cn[1], cn[2], cn[3]);
    }
}

// ------------------------------------------------------------------------------------------
// K4: gather.  G lanes per work item, VEC channels per lane (same shape as the forward gather).
// ------------------------------------------------------------------------------------------
template <typename T, int VEC, int G>
__global__ __launch_bounds__(kGatherItemBlock) void msda_value_gather_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    constexpr int NU = kGatherItemBlock / G;
    constexpr int UB = G < 8 ? G : 8;  // row loads in flight per lane
    // A group works through kItemsPerGroup consecutive work items (equal batch counts inside a bucket), fetching
    // the next item's record while it processes the current one: short items would otherwise be all latency.
    const int slots = (p.it_cap + NU * kItemsPerGroup - 1) / (NU * kItemsPerGroup);
    int pair, slot;
    if (!decode_block(p.grid3d, p.B * p.H, slots, p.xcd_map, pair, slot)) return;
    const int nitems = p.ws_itemcnt[(size_t)pair * kItemMeta];
    const int tid = threadIdx.x;
    const int unit = tid / G, j = tid % G;
    const int item0 = (slot * NU + unit) * kItemsPerGroup;
    if (item0 >= nitems) return;
    const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
    const int lane_base = (tid & (kWave - 1)) & ~(G - 1);

    const Entry<A> *entries = static_cast<const Entry<A> *>(p.ws_entries) + (size_t)pair * p.Q * p.LP;
    const T *gout = static_cast<const T *>(p.grad_out) + ((size_t)b * p.Q * p.H + h) * p.D;  // uniform base
    const uint32_t q_stride = (uint32_t)(p.H * p.D) * (uint32_t)sizeof(T);  // bytes; Q*H*D*sizeof < 2^31 is checked on the host
    // grad_out rows of this plane through a buffer descriptor: scalar base + 32-bit byte offset per lane
    const rsrc_t rs_go = make_rsrc(gout, (uint32_t)(((size_t)p.Q * p.H * p.D - (size_t)h * p.D) * sizeof(T)));
    const int nchan_chunks = (p.D + G * VEC - 1) / (G * VEC);

    const int4 *rec = p.ws_items + ((size_t)pair * p.it_cap + item0) * 3;
    int4 hdr = rec[0], beg = rec[1], len = rec[2];
    for (int k = 0; k < kItemsPerGroup && item0 + k < nitems; ++k) {
        int4 hdr_n = hdr, beg_n = beg, len_n = len;
        if (k + 1 < kItemsPerGroup && item0 + k + 1 < nitems) {
            hdr_n = rec[(k + 1) * 3 + 0];
            beg_n = rec[(k + 1) * 3 + 1];
            len_n = rec[(k + 1) * 3 + 2];
        }
        const int pix = hdr.x, nchunks = hdr.y, scratch_row = hdr.z;
        const int c1 = len.x, c2 = c1 + len.y, c3 = c2 + len.z;
        const int w0 = 0, w1 = c3 + len.w;  // the record is already clipped to this item's window

        // position v of the pixel's virtual list (lists 0..3 back to back) -> (element offset of the query's
        // grad_out row inside the plane, a * fx * fy)
        // Positions past the end of the window are padded with the window's last record at weight 0 (so the
        // unrolled batches need no tail loop; an empty window reads row 0 at weight 0).
        auto fetch = [&](int v, uint32_t &q, A &wgt) {
            q = 0;
            wgt = (A)0;
            if (w1 > 0) {
                const int vc = min(v, w1 - 1);
                const int i = (vc >= c1) + (vc >= c2) + (vc >= c3);
                const int base = i == 0 ? beg.x : i == 1 ? beg.y - c1 : i == 2 ? beg.z - c2 : beg.w - c3;
                const Entry<A> e = entries[base + vc];
                const A fx = (i & 1) ? e.dx : (A)1 - e.dx;
                const A fy = (i & 2) ? e.dy : (A)1 - e.dy;
                q = e.q * q_stride;
                wgt = v < w1 ? e.a * (fy * fx) : (A)0;
            }
        };

        for (int cc = 0; cc < nchan_chunks; ++cc) {
            const int c0 = (cc * G + j) * VEC;
            const bool lane_ok = c0 < p.D;
            const uint32_t lane_elem = (lane_ok ? (uint32_t)c0 : 0u) * (uint32_t)sizeof(T);
            A acc[VEC];
    #pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = (A)0;

            uint32_t cur_q, nxt_q;
            A cur_w, nxt_w;
            fetch(w0 + j, cur_q, cur_w);
            for (int v0 = w0; v0 < w1; v0 += G) {
                fetch(v0 + G + j, nxt_q, nxt_w);  // next batch's records are in flight while this one is consumed
                const int cnt = min(G, w1 - v0);
    #pragma unroll
                for (int jj = 0; jj < G; jj += UB) {  // UB row loads issued back to back, then consumed
                    if (jj < cnt) {                   // uniform per group; G == UB: always true
                        A wgt[UB];
                        Pack<T, VEC> g[UB];
    #pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            const uint32_t q = (uint32_t)__shfl((int)cur_q, lane_base + jj + u, kWave);
                            wgt[u] = __shfl(cur_w, lane_base + jj + u, kWave);
                            g[u] = __builtin_bit_cast(Pack<T, VEC>, RawLoad<sizeof(T) * VEC>::load(rs_go, q + lane_elem));
                        }
    #pragma unroll
                        for (int u = 0; u < UB; ++u)
    #pragma unroll
                            for (int v = 0; v < VEC; ++v) acc[v] = fma_t(wgt[u], TR::to_acc(g[u].v[v]), acc[v]);
                    }
                }
                cur_q = nxt_q;
                cur_w = nxt_w;
            }
            if (lane_ok) {
                if (nchunks == 1) {
                    Pack<T, VEC> o;
    #pragma unroll
                    for (int v = 0; v < VEC; ++v) o.v[v] = TR::from_acc(acc[v]);
                    T *dst = static_cast<T *>(p.grad_value) + (((size_t)b * p.I + pix) * p.H + h) * p.D + c0;
                    *reinterpret_cast<Pack<T, VEC> *>(dst) = o;
                } else {
                    Pack<A, VEC> o;
    #pragma unroll
                    for (int v = 0; v < VEC; ++v) o.v[v] = acc[v];
                    A *dst = static_cast<A *>(p.ws_scratch) + ((size_t)pair * p.it_cap + scratch_row) * p.D + c0;
                    *reinterpret_cast<Pack<A, VEC> *>(dst) = o;
                }
            }
        }

        hdr = hdr_n;
        beg = beg_n;
        len = len_n;
    }
}

// ------------------------------------------------------------------------------------------
// K5: pixels that were split into several work items: sum the partial rows in chunk order
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
    const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
    const int unit = threadIdx.x / G, j = threadIdx.x % G;
    const int pix = slot * NU + unit;
    if (pix >= p.I) return;
    const int2 info = p.ws_pixinfo[(size_t)pair * p.I + pix];
    const int item0 = info.x, nchunks = info.y;
    if (nchunks <= 1) return;
    const A *src = static_cast<const A *>(p.ws_scratch) + ((size_t)pair * p.it_cap + item0) * p.D;
    const int nchan_chunks = (p.D + G * VEC - 1) / (G * VEC);
    for (int cc = 0; cc < nchan_chunks; ++cc) {
        const int c0 = (cc * G + j) * VEC;
        if (c0 >= p.D) continue;
        A acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = (A)0;
#pragma unroll 8
        for (int k = 0; k < nchunks; ++k) {
            const Pack<A, VEC> r = *reinterpret_cast<const Pack<A, VEC> *>(src + (size_t)k * p.D + c0);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] += r.v[v];
        }
        Pack<T, VEC> o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o.v[v] = TR::from_acc(acc[v]);
        T *dst = static_cast<T *>(p.grad_value) + (((size_t)b * p.I + pix) * p.H + h) * p.D + c0;
        *reinterpret_cast<Pack<T, VEC> *>(dst) = o;
    }
}

// ------------------------------------------------------------------------------------------
// workspace layout (host + device agree through these helpers)
// ------------------------------------------------------------------------------------------
struct SortedWsLayout {
    int nc_cap, it_cap, nsplit;
    size_t off_part, off_off, off_pixinfo, off_itemcnt, off_items, off_entries, off_scratch, total;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline SortedWsLayout sorted_ws_layout(int64_t B, int64_t I, int64_t H, int64_t D, int64_t Q, int64_t L, int64_t P,
                                       size_t acc_bytes)
{
    SortedWsLayout w;
    const size_t pairs = (size_t)(B * H);
    const size_t samples = (size_t)(Q * L * P);  // per plane
    w.nc_cap = (int)(2 * I + 2 * L);             // (w+1)(h+1) <= 2wh + 2 per level
    w.it_cap = (int)(I + (4 * samples + kChunk - 1) / kChunk + 1);
    // query slices per plane: enough workgroups to fill the chip, at least ~2k samples each
    int64_t ns = pairs ? (int64_t)((256 + pairs - 1) / pairs) : 1;
    const int64_t by_work = (int64_t)((samples + 2047) / 2048);
    if (ns > by_work) ns = by_work;
    if (ns > 64) ns = 64;
    if (ns > Q) ns = Q;
    if (ns < 1) ns = 1;
    w.nsplit = (int)ns;
    const size_t entry_bytes = acc_bytes == 8 ? 32 : 16;
    size_t o = 0;
    w.off_part = o;     o = align_up(o + pairs * w.nsplit * (size_t)w.nc_cap * 4, 256);
    w.off_off = o;      o = align_up(o + pairs * ((size_t)w.nc_cap + 1) * 4, 256);
    w.off_pixinfo = o;  o = align_up(o + pairs * (size_t)I * 8, 256);
    w.off_itemcnt = o;  o = align_up(o + pairs * kItemMeta * 4, 256);
    w.off_items = o;    o = align_up(o + pairs * (size_t)w.it_cap * 48, 256);
    w.off_entries = o;  o = align_up(o + pairs * samples * entry_bytes, 256);
    w.off_scratch = o;  o = align_up(o + pairs * (size_t)w.it_cap * (size_t)D * acc_bytes, 256);
    w.total = o;
    return w;
}

}  // namespace msda
