// msda_value_tiles.hpp — grad_value by TILES: samples binned by 8x8-pixel tile, combined in LDS.
//
// The cell-sorted pipeline (msda_value_sorted.hpp) pays for its fine sort: 5k cells per plane mean a 22 KB
// histogram per workgroup and 16-byte record stores that no two lanes share a cache line for (~1 cycle per
// lane and dword on the CU's address path), then four partial rows per cell through HBM.  Here the global sort is
// COARSE — one bin per 8x8-pixel tile of a pyramid level, 85 bins per plane at c2 — so that
//
//   T1 count   (plane, query slice): per-bin counts, a few hundred LDS counters
//   T2 scan    per plane: bin starts, per-slice offsets; bins are cut into chunks of <= kTileChunk records and
//              appended to ONE global chunk list (one atomic add per plane)
//   T3 place   (plane, slice): rounds of 4096 samples are counting-sorted by bin INSIDE LDS and then copied
//              out in sorted order — neighbouring lanes write neighbouring records: coalesced stores
//   T4 gather  persistent workgroups walk the chunk list: index-sort the chunk by its 81 local cells in LDS,
//              G-lane groups accumulate every cell's four corner rows in registers (one grad_out row load
//              per sample, as the cell-sorted gather), park them in LDS, and the workgroup sums, per pixel of
//              the tile, the four incident cells' rows.  Pixels whose four cells all belong to this chunk are
//              STORED; the tile's rim (cells of a neighbouring tile also touch those pixels) and every pixel
//              of a multi-chunk tile are added with float atomics into the zero-filled grad_value.
//
// STATUS (round 1): correct (same tests as the cell-sorted path, `value_path` option 3) but not faster: at c2-10k
// the place pass drops from 57 to 39 us and the count / scan passes from 36 to 23 us, but the gather takes 130 us
// (local sort and loop overhead 46, cell accumulation 80 at two rounds of 64 groups, rim atomics 13) against
// 62 + 24.5 us for the cell-sorted gather + finish: 198 vs 179 us in total.  Kept as an option, off by default.
//
// A cell (x0, y0), x0 in [-1, W-1], belongs to the tile of pixel (max(x0, 0), max(y0, 0)); inside its tile it has
// local coordinates (x0 - X + 1, y0 - Y + 1) in [0, 8]^2 — 81 local cells, packed into the record next to the
// query index (which is why Q < 2^24 is required).
#pragma once

#include "msda_value_sorted.hpp"

namespace msda {

constexpr int kTile = 8;
constexpr int kTileCells = (kTile + 1) * (kTile + 1);  // 81 local cells, 81 local pixels (own 8x8 + right/bottom rim)
constexpr int kTileChunk = 2048;     // records per gather chunk
constexpr int kTileBlock = 512;      // threads of the gather kernel
constexpr int kTileG = 8;            // lanes per group in the gather kernel
constexpr int kTileRound = 4;        // samples per thread and round of the place pass
constexpr int kTileBinCap = 4096;    // most bins a plane may have on this path (LDS tables of the passes)
constexpr int kTileScanBlock = 256;  // threads of T2
constexpr int kLcellShift = 24;
constexpr int kChunkMulti = 1 << 30;  // chunk record: the bin has several chunks (everything is added atomically)

struct TileTab {
    int tw[kMaxLevels];      // tiles per row of the level
    int tstart[kMaxLevels];  // first bin of the level
    int nbins;
};

// after the level table is visible to the whole workgroup; caller syncs afterwards
__device__ __forceinline__ void load_tile_table(TileTab *tt, const LevelTab *tab, int L)
{
    if (threadIdx.x == 0) {
        int s = 0;
        for (int l = 0; l < L; ++l) {
            const int tw = (tab->w[l] + kTile - 1) / kTile, th = (tab->h[l] + kTile - 1) / kTile;
            tt->tw[l] = tw;
            tt->tstart[l] = s;
            s += tw * th;
        }
        tt->nbins = s;
    }
}

// sample -> (integer cell corner (x0, y0), fractional offsets).  false: the sample touches no pixel.
// Same arithmetic as sample_cell (msda_value_sorted.hpp).
template <typename A>
__device__ __forceinline__ bool sample_corner(A x, A y, int h, int w, bool zeros, bool align, int &x0i, int &y0i, A &dx,
                                              A &dy)
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
    x0i = (int)x0;
    y0i = (int)y0;
    return true;
}

// (bin inside the plane, local cell inside the tile) of a cell
__device__ __forceinline__ void tile_of_cell(const TileTab &tt, int l, int x0, int y0, int &bin, int &lcell)
{
    const int tx = max(x0, 0) / kTile, ty = max(y0, 0) / kTile;
    bin = tt.tstart[l] + ty * tt.tw[l] + tx;
    lcell = (y0 - ty * kTile + 1) * (kTile + 1) + (x0 - tx * kTile + 1);
}

// ------------------------------------------------------------------------------------------
// T1: per-(plane, slice) bin counts
// ------------------------------------------------------------------------------------------
template <typename T> __global__ __launch_bounds__(kCellBlock) void msda_tile_count_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    int pair, slice;
    if (!decode_block(p.grid3d, p.B * p.H, p.nsplit, p.xcd_map, pair, slice)) return;
    const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
    const int qper = (p.Q + p.nsplit - 1) / p.nsplit;
    const int qa = min(p.Q, slice * qper), qb = min(p.Q, qa + qper);

    __shared__ LevelTab tab;
    __shared__ TileTab tt;
    __shared__ int s_cnt[kTileBinCap];
    load_level_table(&tab, p.shapes, p.L);
    __syncthreads();
    load_tile_table(&tt, &tab, p.L);
    __syncthreads();
    const int tid = threadIdx.x;
    const int nb = min(tt.nbins, kTileBinCap);
    for (int i = tid; i < nb; i += kCellBlock) s_cnt[i] = 0;
    __syncthreads();

    const size_t plane_s0 = ((size_t)b * p.Q * p.H + h) * p.LP;
    const T *loc = static_cast<const T *>(p.loc) + 2 * plane_s0;
    const int HLP = p.H * p.LP;
    const float inv_P = 1.0f / (float)p.P, inv_LP = 1.0f / (float)p.LP;
    const int ns = (qb - qa) * p.LP;  // samples of the slice (host: Q*H*L*P*2 < 2^31)
    for (int f = tid; f < ns; f += kCellBlock) {
        const int fq = div_small(f, p.LP, inv_LP);
        const int sl = f - fq * p.LP;
        const int l = div_small(sl, p.P, inv_P);
        const Pack<T, 2> xy = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * ((qa + fq) * HLP + sl));
        int x0, y0;
        A dx, dy;
        if (sample_corner<A>(TR::to_acc(xy.v[0]), TR::to_acc(xy.v[1]), tab.h[l], tab.w[l], p.zeros, p.align, x0, y0, dx,
                             dy)) {
            int bin, lcell;
            tile_of_cell(tt, l, x0, y0, bin, lcell);
            if (bin < nb) atomicAdd(&s_cnt[bin], 1);
        }
    }
    __syncthreads();
    int *part = p.ws_part + ((size_t)pair * p.nsplit + slice) * p.nb_cap;
    for (int i = tid; i < nb; i += kCellBlock) part[i] = s_cnt[i];
}

// ------------------------------------------------------------------------------------------
// T2: per plane: bin totals -> bin starts, per-slice offsets, chunk records appended to the global list
// ------------------------------------------------------------------------------------------
template <typename Tag> __global__ __launch_bounds__(kTileScanBlock) void msda_tile_scan_kernel(const Params p)
{
    const int pair = blockIdx.x;
    __shared__ LevelTab tab;
    __shared__ TileTab tt;
    __shared__ int s_tot[kTileBinCap + 1];  // bin totals, then bin starts (and the plane's total behind them)
    __shared__ int s_coff[kTileBinCap];  // first chunk of every bin (inside the plane)
    __shared__ int s_base;
    load_level_table(&tab, p.shapes, p.L);
    __syncthreads();
    load_tile_table(&tt, &tab, p.L);
    __syncthreads();
    const int t = threadIdx.x;
    const int nb = min(tt.nbins, kTileBinCap);
    int *part = p.ws_part + (size_t)pair * p.nsplit * p.nb_cap;
    for (int bin = t; bin < nb; bin += kTileScanBlock) {
        int tot = 0;
        for (int j = 0; j < p.nsplit; ++j) {
            const int n = part[(size_t)j * p.nb_cap + bin];
            part[(size_t)j * p.nb_cap + bin] = tot;  // slice j's first slot relative to the bin's start
            tot += n;
        }
        s_tot[bin] = tot;
    }
    __syncthreads();
    // exclusive scans over the bins by the first wave: each lane owns a contiguous segment
    if (t < kWave) {
        const int seg = (nb + kWave - 1) / kWave;
        const int lo = min(nb, t * seg), hi = min(nb, lo + seg);
        int sum = 0, csum = 0;
        for (int i = lo; i < hi; ++i) {
            sum += s_tot[i];
            csum += (s_tot[i] + kTileChunk - 1) / kTileChunk;
        }
        int inc = sum, cinc = csum;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int n1 = __shfl_up(inc, d, kWave), n2 = __shfl_up(cinc, d, kWave);
            if (t >= d) {
                inc += n1;
                cinc += n2;
            }
        }
        int run = inc - sum, crun = cinc - csum;
        for (int i = lo; i < hi; ++i) {
            const int n = s_tot[i];
            s_tot[i] = run;
            s_coff[i] = crun;
            run += n;
            crun += (n + kTileChunk - 1) / kTileChunk;
        }
        if (t == kWave - 1) {
            s_tot[nb] = inc;  // (kTileBinCap + 1 would overflow only when nb == kTileBinCap: see below)
            s_base = atomicAdd(p.ws_itemcnt, cinc);  // this plane's range in the global chunk list
        }
    }
    __syncthreads();
    int *binoff = p.ws_off + (size_t)pair * (p.nb_cap + 1);
    const int base = s_base;
    for (int bin = t; bin < nb; bin += kTileScanBlock) {
        const int beg = s_tot[bin];
        const int end = (bin + 1 < nb) ? s_tot[bin + 1] : s_tot[nb];
        binoff[bin] = beg;
        const int n = end - beg;
        const int chunks = (n + kTileChunk - 1) / kTileChunk;
        for (int k = 0; k < chunks; ++k) {
            const int slot = base + s_coff[bin] + k;
            if (slot < p.ch_cap)
                p.ws_chunks[slot] = make_int4(pair, bin, beg + k * kTileChunk,
                                              min(kTileChunk, n - k * kTileChunk) | (chunks > 1 ? kChunkMulti : 0));
        }
    }
}

// ------------------------------------------------------------------------------------------
// T3: place.  Rounds of kTileRound * 1024 samples are counting-sorted by bin in LDS, then written out in sorted
// order (coalesced) to the positions reserved from the per-bin cursors.
// ------------------------------------------------------------------------------------------
template <typename T> __global__ __launch_bounds__(kCellBlock) void msda_tile_place_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    constexpr int kRounds = sizeof(A) == 8 ? kTileRound / 2 : kTileRound;  // 32-byte records: half the round
    constexpr int kRoundN = kRounds * kCellBlock;
    int pair, slice;
    if (!decode_block(p.grid3d, p.B * p.H, p.nsplit, p.xcd_map, pair, slice)) return;
    const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
    const int qper = (p.Q + p.nsplit - 1) / p.nsplit;
    const int qa = min(p.Q, slice * qper), qb = min(p.Q, qa + qper);

    // dynamic LDS: [entries kRoundN][dst kRoundN]; static: tables
    Entry<A> *s_ent = reinterpret_cast<Entry<A> *>(msda_smem);
    int *s_dst = reinterpret_cast<int *>(msda_smem + (size_t)kRoundN * sizeof(Entry<A>));
    __shared__ LevelTab tab;
    __shared__ TileTab tt;
    __shared__ int s_cur[kTileBinCap];   // next free global slot of every bin for this slice
    __shared__ int s_cnt[kTileBinCap];   // records of the round per bin
    __shared__ int s_loff[kTileBinCap];  // first LDS slot of every bin in the round
    __shared__ int s_wave[kCellBlock / kWave];
    load_level_table(&tab, p.shapes, p.L);
    __syncthreads();
    load_tile_table(&tt, &tab, p.L);
    __syncthreads();
    const int tid = threadIdx.x;
    const int nb = min(tt.nbins, kTileBinCap);
    const int *binoff = p.ws_off + (size_t)pair * (p.nb_cap + 1);
    const int *part = p.ws_part + ((size_t)pair * p.nsplit + slice) * p.nb_cap;
    for (int i = tid; i < nb; i += kCellBlock) {
        s_cur[i] = binoff[i] + part[i];
        s_cnt[i] = 0;
    }
    __syncthreads();

    const size_t plane_s0 = ((size_t)b * p.Q * p.H + h) * p.LP;
    const T *loc = static_cast<const T *>(p.loc) + 2 * plane_s0;
    const T *attn = static_cast<const T *>(p.attn) + plane_s0;
    Entry<A> *entries = static_cast<Entry<A> *>(p.ws_entries) + (size_t)pair * p.Q * p.LP;
    const int HLP = p.H * p.LP;
    const float inv_P = 1.0f / (float)p.P, inv_LP = 1.0f / (float)p.LP;
    const int ns = (qb - qa) * p.LP;
    const int bseg = (nb + kCellBlock - 1) / kCellBlock;  // bins per thread in the scan

    for (int r0 = 0; r0 < ns; r0 += kRoundN) {
        // 1. every thread takes kRounds samples, finds their bins, ranks them inside the bin
        Entry<A> e[kRounds];
        int bin[kRounds], rank[kRounds];
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {
            bin[k] = -1;
            rank[k] = 0;
            const int f = r0 + k * kCellBlock + tid;
            if (f < ns) {
                const int fq = div_small(f, p.LP, inv_LP);
                const int sl = f - fq * p.LP;
                const int l = div_small(sl, p.P, inv_P);
                const int sidx = (qa + fq) * HLP + sl;
                const Pack<T, 2> xy = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
                const T at = attn[sidx];
                int x0, y0;
                A dx, dy;
                if (sample_corner<A>(TR::to_acc(xy.v[0]), TR::to_acc(xy.v[1]), tab.h[l], tab.w[l], p.zeros, p.align, x0,
                                     y0, dx, dy)) {
                    int bn, lcell;
                    tile_of_cell(tt, l, x0, y0, bn, lcell);
                    if (bn < nb) {
                        bin[k] = bn;
                        e[k].q = (uint32_t)(qa + fq) | ((uint32_t)lcell << kLcellShift);
                        e[k].dx = dx;
                        e[k].dy = dy;
                        e[k].a = TR::to_acc(at);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < kRounds; ++k)
            if (bin[k] >= 0) rank[k] = atomicAdd(&s_cnt[bin[k]], 1);
        __syncthreads();
        // 2. exclusive scan of the round's bin counts -> LDS slots
        {
            const int lo = min(nb, tid * bseg), hi = min(nb, lo + bseg);
            int sum = 0;
            for (int i = lo; i < hi; ++i) sum += s_cnt[i];
            int total;
            int run = block_exclusive_scan(sum, s_wave, total);
            for (int i = lo; i < hi; ++i) {
                s_loff[i] = run;
                run += s_cnt[i];
            }
            __syncthreads();
            // 3. records into their LDS slots, with their global destination
#pragma unroll
            for (int k = 0; k < kRounds; ++k) {
                if (bin[k] >= 0) {
                    const int slot = s_loff[bin[k]] + rank[k];
                    s_ent[slot] = e[k];
                    s_dst[slot] = s_cur[bin[k]] + rank[k];
                }
            }
            __syncthreads();
            // 4. copy out in sorted order: neighbouring lanes, neighbouring records
            for (int i = tid; i < total; i += kCellBlock) entries[s_dst[i]] = s_ent[i];
            // 5. advance the cursors, clear the counts
            for (int i = tid; i < nb; i += kCellBlock) {
                s_cur[i] += s_cnt[i];
                s_cnt[i] = 0;
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// T4: gather.  Persistent workgroups over the global chunk list.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_add_t(float *p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_t(double *p, double v) { unsafeAtomicAdd(p, v); }

template <typename T, int VEC> __global__ __launch_bounds__(kTileBlock) void msda_tile_gather_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    static_assert(sizeof(T) == sizeof(A), "the tile path stores and atomically adds in the accumulate type");
    constexpr int G = kTileG;
    constexpr int NU = kTileBlock / G;
    constexpr int CP = G * VEC;  // channels per pass
    constexpr int UB = 8;

    __shared__ LevelTab tab;
    __shared__ TileTab tt;
    __shared__ int s_cnt[kTileCells], s_off[kTileCells + 1];
    __shared__ uint16_t s_idx[kTileChunk];
    __shared__ CornerW<A> s_w[kTileBlock];
    __shared__ uint32_t s_q[kTileBlock];
    __shared__ __attribute__((aligned(32))) A s_part[kTileCells * 4 * CP];

    load_level_table(&tab, p.shapes, p.L);
    __syncthreads();
    load_tile_table(&tt, &tab, p.L);
    __syncthreads();
    const int tid = threadIdx.x;
    const int unit = tid / G, j = tid % G;
    const int gbase = tid - j;
    const int nchunks = min(*p.ws_itemcnt, p.ch_cap);
    const int npass = (p.D + CP - 1) / CP;
    const uint32_t q_stride = (uint32_t)(p.H * p.D) * (uint32_t)sizeof(T);

    for (int ci = blockIdx.x; ci < nchunks; ci += gridDim.x) {
        const int4 rec = p.ws_chunks[ci];
        const int pair = rec.x, bin = rec.y, start = rec.z;
        const int count = rec.w & (kChunkMulti - 1);
        const bool multi = (rec.w & kChunkMulti) != 0;
        const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
        int l = 0;
        while (l < p.L - 1 && bin >= tt.tstart[l + 1]) ++l;
        const int trel = bin - tt.tstart[l];
        const int ty = trel / tt.tw[l], tx = trel - ty * tt.tw[l];
        const int X = tx * kTile, Y = ty * kTile, W = tab.w[l], Hh = tab.h[l];

        const Entry<A> *entries = static_cast<const Entry<A> *>(p.ws_entries) + (size_t)pair * p.Q * p.LP + start;
        const T *gout = static_cast<const T *>(p.grad_out) + ((size_t)b * p.Q * p.H + h) * p.D;
        const rsrc_t rs_go = make_rsrc(gout, (uint32_t)(((size_t)p.Q * p.H * p.D - (size_t)h * p.D) * sizeof(T)));

        // ---- A: index-sort the chunk by local cell (one pass over the records: cell and rank stay in registers) ----
        constexpr int kPer = (kTileChunk + kTileBlock - 1) / kTileBlock;
        for (int i = tid; i < kTileCells; i += kTileBlock) s_cnt[i] = 0;
        __syncthreads();
        int lc[kPer], rk[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int i = tid + k * kTileBlock;
            lc[k] = i < count ? (int)(entries[i].q >> kLcellShift) : -1;
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) rk[k] = lc[k] >= 0 ? atomicAdd(&s_cnt[lc[k]], 1) : 0;
        __syncthreads();
        if (tid < kWave) {  // 81 counters: two per lane of the first wave
            const int i0 = 2 * tid, i1 = 2 * tid + 1;
            const int c0 = i0 < kTileCells ? s_cnt[i0] : 0, c1 = i1 < kTileCells ? s_cnt[i1] : 0;
            int inc = c0 + c1;
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const int n = __shfl_up(inc, d, kWave);
                if (tid >= d) inc += n;
            }
            const int ex = inc - (c0 + c1);
            if (i0 < kTileCells) s_off[i0] = ex;
            if (i1 < kTileCells) s_off[i1] = ex + c0;
            if (tid == kWave - 1) s_off[kTileCells] = inc;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPer; ++k)
            if (lc[k] >= 0) s_idx[s_off[lc[k]] + rk[k]] = (uint16_t)(tid + k * kTileBlock);
        __syncthreads();

        for (int cc = 0; cc < npass; ++cc) {
            const int c0 = cc * CP + j * VEC;
            const bool lane_ok = c0 < p.D;
            const uint32_t lane_elem = (lane_ok ? (uint32_t)c0 : 0u) * (uint32_t)sizeof(T);
            // ---- B: every cell's four corner rows, one group per cell ----
            for (int c = unit; c < kTileCells; c += NU) {
                const int beg = s_off[c], n = s_off[c + 1] - beg;
                A acc[4][VEC];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[k][i] = (A)0;
                auto convert = [&](const Entry<A> &e, int v, uint32_t &q, CornerW<A> &cw) {
                    const bool ok = v < n;
                    const A a = ok ? e.a : (A)0;
                    const A ax1 = a * e.dx, ax0 = a - ax1;
                    q = ok ? (e.q & ((1u << kLcellShift) - 1)) * q_stride : 0x80000000u;
                    cw.w[3] = ax1 * e.dy;
                    cw.w[2] = ax0 * e.dy;
                    cw.w[1] = ax1 - cw.w[3];
                    cw.w[0] = ax0 - cw.w[2];
                };
                if (n > 0) {
                    Entry<A> e_cur = entries[s_idx[beg + min(j, n - 1)]];
                    for (int v0 = 0; v0 < n; v0 += G) {
                        uint32_t cur_q;
                        CornerW<A> cur_w;
                        convert(e_cur, v0 + j, cur_q, cur_w);
                        wave_lds_sync();
                        s_q[tid] = cur_q;
                        s_w[tid] = cur_w;
                        e_cur = entries[s_idx[beg + min(v0 + G + j, n - 1)]];
                        wave_lds_sync();
                        Pack<T, VEC> g[UB];
#pragma unroll
                        for (int u = 0; u < UB; ++u)
                            g[u] = __builtin_bit_cast(Pack<T, VEC>,
                                                      RawLoad<sizeof(T) * VEC>::load(rs_go, s_q[gbase + u] + lane_elem));
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            const CornerW<A> w = s_w[gbase + u];
#pragma unroll
                            for (int k = 0; k < 4; ++k)
#pragma unroll
                                for (int v = 0; v < VEC; ++v) acc[k][v] = fma_t(w.w[k], TR::to_acc(g[u].v[v]), acc[k][v]);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    Pack<A, VEC> o;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o.v[v] = acc[k][v];
                    *reinterpret_cast<Pack<A, VEC> *>(&s_part[(c * 4 + k) * CP + j * VEC]) = o;
                }
            }
            __syncthreads();
            // ---- C: per pixel of the tile (own 8x8 plus the right / bottom rim): sum the incident cells' rows ----
            for (int pp = unit; pp < kTileCells; pp += NU) {
                const int py = pp / (kTile + 1), px = pp - py * (kTile + 1);
                const int gx = X + px, gy = Y + py;
                if (gx < W && gy < Hh && lane_ok) {
                    A sum[VEC];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) sum[v] = (A)0;
                    auto add = [&](int cx, int cy, int corner) {
                        if (cx <= kTile && cy <= kTile) {
                            const Pack<A, VEC> r = *reinterpret_cast<const Pack<A, VEC> *>(
                                &s_part[((cy * (kTile + 1) + cx) * 4 + corner) * CP + j * VEC]);
#pragma unroll
                            for (int v = 0; v < VEC; ++v) sum[v] += r.v[v];
                        }
                    };
                    add(px + 1, py + 1, 0);
                    add(px, py + 1, 1);
                    add(px + 1, py, 2);
                    add(px, py, 3);
                    T *dst = static_cast<T *>(p.grad_value) +
                             (((size_t)b * p.I + tab.start[l] + (size_t)gy * W + gx) * p.H + h) * p.D + c0;
                    // complete: all four incident cells belong to this tile and the tile is one chunk
                    const bool complete = !multi && px < kTile && py < kTile && (px > 0 || X == 0) && (py > 0 || Y == 0);
                    if (complete) {
                        Pack<T, VEC> o;
#pragma unroll
                        for (int v = 0; v < VEC; ++v) o.v[v] = TR::from_acc(sum[v]);
                        *reinterpret_cast<Pack<T, VEC> *>(dst) = o;
                    } else {
#pragma unroll
                        for (int v = 0; v < VEC; ++v)
                            if (sum[v] != (A)0) atomic_add_t(reinterpret_cast<A *>(dst) + v, sum[v]);
                    }
                }
            }
            __syncthreads();
        }
    }
}

// workspace of this path (smaller than the cell-sorted layout for the same sizes; the launcher checks)
struct TileWsLayout {
    int nb_cap, ch_cap, nsplit;
    size_t off_part, off_binoff, off_cnt, off_chunks, off_entries, total;
};

inline TileWsLayout tile_ws_layout(int64_t B, int64_t I, int64_t H, int64_t D, int64_t Q, int64_t L, int64_t P,
                                   size_t acc_bytes)
{
    TileWsLayout w;
    const size_t pairs = (size_t)(B * H);
    const size_t samples = (size_t)(Q * L * P);
    // bins of a level: ceil(w/8) ceil(h/8) <= wh/64 + (w + h)/8 + 1 <= wh/64 + (wh + 1)/8 + 1
    w.nb_cap = (int)(I / 64 + (I + L) / 8 + 2 * L + 1);
    w.nsplit = sorted_ws_layout(B, I, H, D, Q, L, P, acc_bytes).nsplit;
    const size_t chunks = pairs * ((size_t)w.nb_cap + samples / kTileChunk + 1);
    w.ch_cap = (int)(chunks < ((size_t)1 << 30) ? chunks : ((size_t)1 << 30));
    const size_t entry_bytes = acc_bytes == 8 ? 32 : 16;
    size_t o = 0;
    w.off_part = o;    o = align_up(o + pairs * w.nsplit * (size_t)w.nb_cap * 4, 256);
    w.off_binoff = o;  o = align_up(o + pairs * ((size_t)w.nb_cap + 1) * 4, 256);
    w.off_cnt = o;     o = align_up(o + 256, 256);
    w.off_chunks = o;  o = align_up(o + (size_t)w.ch_cap * 16, 256);
    w.off_entries = o; o = align_up(o + pairs * samples * entry_bytes, 256);
    w.total = o;
    return w;
}

}  // namespace msda
