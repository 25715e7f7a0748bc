// msda_value_place.hpp — K3 of the sorted grad_value pipeline, level-major.
//
// The plane-major place pass (msda_cell_pass_kernel<T, true>: a workgroup = all levels of a (plane, query slice), 144 KiB
// of LDS, one workgroup per CU) takes 59 us at c2 @ 10k, 42 of them for its 5.1 M scattered 16-byte record stores: every
// workgroup writes into all levels' lists of its plane for the whole kernel, 10 MB of write targets per XCD against
// 4 MB of L2, so partial lines leave L2 before their neighbours arrive.  Here
//
//   a workgroup owns ONE LEVEL of a (plane, query slice) — the slices are the count pass's, so off[cell] + part[slice][cell]
//   (K1 / K2, unchanged) is each cell's first slot exactly as before;
//   the launch is level-major (all workgroups of level 0, then level 1, ...): at any time the chip writes into one
//   level's lists, 2.5 MB per XCD, and the partial lines merge in L2;
//   its LDS cell table covers one level, so several workgroups fit a CU and the chains of latencies (table, samples,
//   LDS atomics) of different workgroups overlap; the level geometry comes from scalar loads of the shapes tensor (no
//   LDS level table, no barrier for it), the table's off[] / part[] loads go out in batches.
//
// Same record array, same gather / finish kernels.
#pragma once

#include "msda_value_sorted.hpp"

namespace msda {

constexpr int kPlaceBlock = 1024;  // threads per workgroup (two per CU at 64 VGPRs; 512 x 3 per CU: 60 us against 44.5 at c2 @ 10k)
constexpr int kPlaceBlockSmall = 256;  // ... for decoder-sized calls (msda_launch.hpp)
constexpr int kPlaceCellsTwoPerCu = 19456;  // LDS table cells (76 KB) that still let two workgroups share a CU's 160 KB

// REPRODUCIBLE ORDER: the order inside a cell's list must not depend on when LDS atomics retire (the gather sums in
// list order).  The workgroup's waves take their cursor atomics in TURNS — wave w of round r goes when the turn counter
// reads r * waves + w and passes it on — so a cell's records of this slice are in (round, wave, sample-in-thread, lane)
// order every time.  Only the few hundred cycles of the atomics themselves are serialised (the cells are computed
// before the turn, the records packed and stored after it): 46.0 us against 44.5 with free-running atomics at c2 @ 10k,
// fwd+bwd unchanged within noise — against 715 us for the one-wave-per-slice kernel of round 3 this replaces.  Lanes
// of ONE wave instruction that hit the same cell are ranked by the LDS hardware in a fixed order.  grad_value of the
// sorted pipeline is therefore bitwise reproducible by default wherever this pass runs.
template <typename T, int TB> __global__ __launch_bounds__(TB) void msda_cell_place_lm_kernel(const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    constexpr int NS = sizeof(A) == 8 ? 3 : 6;  // samples a thread has in flight
    __shared__ int s_turn;
    const int K = p.nsplit;
    int pair, slot;
    if (!decode_block(p.grid3d, p.B * p.H, p.L * K, p.xcd_map, pair, slot)) return;
    const int l = slot / K, k = slot - l * K;  // level-major
    const int b = (int)fast_div((uint32_t)pair, p.div_h), h = pair - b * p.H;
    const int tid = threadIdx.x;

    // the count pass's query slice k
    const int qper = (p.q_end - p.q_begin + K - 1) / K;
    const int qa = min(p.q_end, p.q_begin + k * qper), qb = min(p.q_end, qa + qper);
    if (qb <= qa) return;
    // thread -> (point, query lane): one pass of the workgroup covers dq queries' P samples of this level
    const int pt = tid % p.P, tq = tid / p.P, dq = TB / p.P;
    const bool active = tq < dq;
    const size_t plane_s0 = ((size_t)b * p.Q * p.H + h) * p.LP;
    const T *loc = static_cast<const T *>(p.loc) + 2 * plane_s0;
    const T *attn = static_cast<const T *>(p.attn) + plane_s0;
    const int HLP = p.H * p.LP, sl = l * p.P + pt;

    // ---- level geometry: scalar loads (uniform addresses), as load_level_table computes it ----
    int lh = 0, lw = 0, cs = 0, ps = 0, plane_nc = 0;
    for (int i = 0; i < p.L; ++i) {
        const int hh = (int)p.shapes[2 * i], ww = (int)p.shapes[2 * i + 1];
        if (i < l) {
            cs += (hh + 1) * (ww + 1);
            ps += hh * ww;
        }
        if (i == l) {
            lh = hh;
            lw = ww;
        }
        plane_nc += (hh + 1) * (ww + 1);
    }
    // the level's cells inside the tables (shapes that disagree with I must not push past the workspace)
    const int ncl = min((lh + 1) * (lw + 1), min(plane_nc, p.nc_cap) - cs);
    if (ncl <= 0) return;

    int *s_gb = reinterpret_cast<int *>(msda_smem);
    const int *off = p.ws_off + (size_t)pair * (p.nc_cap + 1) + cs;
    const int *part = p.ws_part + ((size_t)pair * K + k) * p.nc_cap + cs;
    Entry<A> *entries = plane_entries<A>(p, pair);
    const int cc = p.cell_cap;

    for (int c0 = 0; c0 < ncl; c0 += cc) {  // one trip unless the level has more cells than fit in LDS
        const int n = min(cc, ncl - c0);
        if (c0 > 0) __syncthreads();
        // this slice's first slot in every cell list: off[] + part[], four cells per thread requested together
        for (int i0 = 0; i0 < n; i0 += 4 * TB) {
            int va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * TB + tid, n - 1);
                va[u] = off[c0 + i];
                vb[u] = part[c0 + i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * TB + tid;
                if (i < n) s_gb[i] = va[u] + vb[u];
            }
        }
        if (tid == 0) s_turn = 0;
        __syncthreads();
        {
            const int wid = tid / kWave, nw = TB / kWave;
            const int rounds = (qb - qa + NS * dq - 1) / (NS * dq);  // the same for every wave: nobody skips a turn
            for (int r = 0; r < rounds; ++r) {
                const int q0 = qa + tq + r * NS * dq;
                Pack<T, 2> xy[NS];
                T at[NS];
                unsigned rel[NS];
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const int q = q0 + j * dq;
                    xy[j].v[0] = xy[j].v[1] = at[j] = TR::from_acc((A)0);
                    if (active && q < qb) {
                        const int sidx = q * HLP + sl;
                        xy[j] = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
                        at[j] = attn[sidx];
                    }
                }
#pragma unroll
                for (int j = 0; j < NS; ++j) {  // the cells, before the turn
                    const int q = q0 + j * dq;
                    int cell;
                    uint32_t cellw;
                    A dx, dy;
                    rel[j] = 0xFFFFFFFFu;
                    if (active && q < qb && sample_cell<A>(TR::to_acc(xy[j].v[0]), TR::to_acc(xy[j].v[1]), lh, lw, 0, ps, l, p.zeros,
                                                           p.align, cell, cellw, dx, dy)) {
                        const unsigned rr = (unsigned)(cell - c0);
                        if (rr < (unsigned)n) rel[j] = rr;
                    }
                }
                const int my = r * nw + wid;
                while (__hip_atomic_load(&s_turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != my) __builtin_amdgcn_s_sleep(1);
                int pos[NS];
#pragma unroll
                for (int j = 0; j < NS; ++j) pos[j] = rel[j] != 0xFFFFFFFFu ? atomicAdd(&s_gb[rel[j]], 1) : -1;
                // (acquire on the spin, release on the hand-over: the compiler may not move the cursor atomics across either —
                // ADVICE r04; the hardware runs a wave's DS operations in order anyway, so this costs one s_waitcnt lgkmcnt(0))
                if ((tid & (kWave - 1)) == 0) __hip_atomic_store(&s_turn, my + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                for (int j = 0; j < NS; ++j) {  // the records, after the turn
                    if (pos[j] < 0) continue;
                    const int q = q0 + j * dq;
                    int cell;
                    uint32_t cellw;
                    A dx, dy;
                    sample_cell<A>(TR::to_acc(xy[j].v[0]), TR::to_acc(xy[j].v[1]), lh, lw, 0, ps, l, p.zeros, p.align, cell, cellw, dx, dy);
                    entries[pos[j]] = Entry<A>::pack((uint32_t)q, cellw, TR::to_acc(at[j]), dx, dy);
                }
            }
        }
    }
}

// cells of the workgroup's LDS table: every cell a level can have (the host knows the plane's bound only), or what fits
inline int place_cell_cap(int64_t nc_cap, size_t max_lds)
{
    const int64_t fit = (int64_t)(max_lds / 4) / 32 * 32;
    int64_t cells = (nc_cap + 31) / 32 * 32;
    if (cells > fit) cells = fit;
    return (int)cells;
}

}  // namespace msda
