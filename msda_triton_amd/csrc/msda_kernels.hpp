// msda_kernels.hpp — kernel parameters and the two gather kernels.
//
//   msda_fwd_kernel         out = sum_{l,p} attn * bilinear(value_l, loc)        (kernels.py:259-348)
//   msda_bwd_sample_kernel  grad_loc, grad_attn (private per sample, no atomics)  (kernels.py:494-537)
//   (grad_value lives in msda_value_sorted.hpp / msda_value_small.hpp)
//
// Work decomposition: a workgroup owns ONE (batch, head) plane of `value` and a run of query
// chunks, so every row it gathers comes from one 2-D plane.  A *unit* = one (b, q, h); a unit is
// served by G lanes (G * VEC >= D channels, VEC elements = one 16-byte load per lane), i.e.
// 64/G units per wavefront and 256/G per workgroup.
//
// Phase 1 (per wave, one sample per lane and trip): read (x, y, a), do the coordinate math ONCE per
//   sample, park {4 row offsets, 4 weights} in the wave's LDS slice.  Loads are coalesced along (l, p).
// Phase 2 (per unit, G lanes): broadcast-read the parked record, fetch the four rows with range-checked
//   16-byte buffer loads, FMA into per-lane accumulators.  No block barrier: a wave reads only its own records.
// LDSL variants (1024-thread workgroups, one per CU): the coarsest pyramid levels are served from ONE LDS copy shared by
//   16 waves, whose waves take their query slices from a counter in LDS (round 5; rounds 1-2 had tried it with static
//   slices / 256-thread workgroups and dropped it).  msda_fwd_unit_kernel: one wave per unit for decoder-sized calls.
#pragma once

#include <type_traits>

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
    int qw;        // query chunks handled by one workgroup (amortises the level staging)
    int sc;        // samples of a unit parked in LDS at a time (<= LP)
    int zeros, align, xcd_map;
    const void *ref;  // fused module prologue: reference points [B, Q, ref_dim]; then `loc` holds the raw projection [B,Q,H,L,P,3]
    int ref_dim;      // 2: (x, y)   4: (cx, cy, w, h)
    // fused backward prologue: grad_loc holds grad_proj [B,Q,H,L,P,3], grad_attn the per-head partial sums of
    // grad_reference_points [B,Q,H,ref_dim]; the sampling points / attention weights the kernel derives are also
    // written out (workspace) for the grad_value passes that follow
    void *mat_loc, *mat_attn;
    int grid3d;       // this launch uses the division-free 3-D grid (see decode_block)
    int debug;        // dev-only ablation mask (msda_set_option("debug", m)); 0 in normal use
    FastDiv div_h;    // pair -> (b, h)
    // sorted (gather-formulated) grad_value path: caller-provided workspace, see msda_value_sorted.hpp
    int *ws_part;       // [pairs][nsplit][nc_cap]   per-slice cell counts, then each slice's first slot per cell
    int *ws_blocktot;   // [pairs][nsplit][nblk_cap] per slice and block of 256 cells: records
    int *ws_off;        // [pairs][nc_cap+1]  first record of every cell's list (and the plane's total behind the last)
    int *ws_total;      // [pairs]            records of the plane
    int *ws_meta;       // [0] = cells of a plane (written by the count pass for the scan kernel)
    void *ws_entries;   // [pairs - ent_n0 - ent_n1][Q*L*P]  Entry<acc>: sample records sorted by cell (plane_entries())
    // ... the records of the first ent_n0 planes live in the caller's grad_loc buffer, those of the next ent_n1 in
    // grad_attn (both are written only after the gather has consumed the records): 0 / 0 unless the caller asked for
    // the sample gradients in the same call and sized the workspace accordingly (MSDA_WS_RECORDS_IN_GRADS)
    // ... and those of the next ent_n2 in grad_value itself, which only the finish kernel writes (single-round
    // problems: with several rounds the finish of one round precedes the place pass of the next)
    void *ent_alt0, *ent_alt1, *ent_alt2;
    int ent_n0, ent_n1, ent_n2;
    void *ws_scratch;   // [pairs][I][4][D]   acc-typed partial rows: slot k of a pixel = what the cell having it as corner k left
    void *ws_cont;      // [pairs][cont_cap][4][D] acc-typed continuation rows: one set per gather workgroup
    int nc_cap, nblk_cap, win_cap, cont_cap;
    int win;          // records per gather window (sorted grad_value)
    FastDiv div_win;  // ... and the division by it
    int nsplit;         // query slices per plane in the count / place passes
    int cell_cap;       // cells a count / place workgroup holds in LDS at a time
    // query chunking of the sorted path (Q so large that a plane's grad_out rows leave L2): the passes of one round
    // serve the queries [q_begin, q_end); ent_cap = records a plane's list can hold; finish_mode: 0 store grad_value,
    // 1 first round (-> accumulator), 2 middle round (accumulator +=), 3 last round (accumulator + this -> grad_value)
    int q_begin, q_end, ent_cap, finish_mode;
    void *ws_accum;     // [pairs][I][D] acc-typed running sums between rounds (nullptr: a single round)
    int lds_lev_bytes;  // LDSL gather kernels: LDS bytes set aside for the rows of the coarsest levels
    int lds_stagger;    // ... wave w starts w * lds_stagger * 64 cycles late
    int lds_planes;     // ... 2: a workgroup serves the planes (b, 2k) and (b, 2k + 1) and its waves take slices of either
    int vrow_bytes;     // host only: D * sizeof(value element) (plane_grid's block-order rule)
    int v_row;          // bytes from one pixel's rows of `value` to the next pixel's: H * D * sizeof unless the caller pads
    int touch;          // forward kernels: the workgroups request every row of their plane once at the start (touch_rows)
    int small_cells;    // single-launch small-problem kernel: capacity of its LDS cell table
    int small_ns;       // ... workgroups per (plane, level)
    int small_hinted;   // ... small_cells is the caller's bound (max_level_cells argument / option "level_cells"), not the bound from I
};

extern __shared__ __attribute__((aligned(16))) unsigned char msda_smem[];

template <typename A> struct alignas(16) Rec4 {
    A v[4];
};

// Records parked by a wave are read back only by that same wave: DS operations of one wave execute
// in order, so no s_barrier is needed — only a compiler fence so the accesses are not reordered.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The (b, h) plane of `value`: pixel rows Params::v_row bytes apart (H * D * sizeof(TV) when dense; a caller that owns the
// layout may pad every pixel's H rows by one 128-byte line so that a head's rows cycle through all residues mod 8 of
// the line index — the vector L1 picks its tag RAM from those bits, HISTORY.md 4 item 5), the head's row at h * D * sizeof(TV)
// inside a pixel.  plane_base: first byte of the plane; plane_span: bytes the plane's descriptor covers (to the end of
// the batch element's last pixel row — offsets of masked corners lie beyond it).
template <typename TV> __device__ __forceinline__ const unsigned char *plane_base(const Params &p, int b, int h)
{
    return static_cast<const unsigned char *>(p.value) + (size_t)b * p.I * (size_t)p.v_row + (size_t)h * p.D * sizeof(TV);
}
template <typename TV> __device__ __forceinline__ uint32_t plane_span(const Params &p, int h)
{
    // (the last pixel's row ends at (I - 1) * v_row + H * D * sizeof: the pad behind it is not the caller's to be read)
    return (uint32_t)((size_t)(p.I - 1) * (size_t)p.v_row + (size_t)(p.H - h) * p.D * sizeof(TV));
}

// level table + one 16-byte slot (the LDSL kernels' work counter) in front of the records
constexpr size_t kGatherLdsFixedBytes = (sizeof(LevelTab) + 15) / 16 * 16 + 16;
// shared LDS carve-up of the two gather kernels
template <typename A> struct GatherLds {
    LevelTab *tab;
    uint4 *s_off;
    Rec4<A> *s_rec;
    A *s_aux;  // fused backward only: per record slot (attention weight, x offset, y offset)
    __device__ __forceinline__ GatherLds(int units, int scp, bool aux = false)
    {
        unsigned char *p = msda_smem;
        tab = reinterpret_cast<LevelTab *>(p);
        p += kGatherLdsFixedBytes;
        s_off = reinterpret_cast<uint4 *>(p);
        p += (size_t)units * scp * sizeof(uint4);
        s_rec = reinterpret_cast<Rec4<A> *>(p);
        p += (size_t)units * scp * sizeof(Rec4<A>);
        s_aux = reinterpret_cast<A *>(p);
    }
};
constexpr size_t kGatherLdsFixed = kGatherLdsFixedBytes;

#ifdef MSDA_DEV
// dev builds: a phase clock for the forward kernel (msda_set_option("debug", 2048)): every wave sums the cycles between
// its stamps per phase and leaves eight floats at the START of `out` (which is garbage afterwards) — tools/phase_clock.py
#define MSDA_STAMP(var)                                  \
    do {                                                 \
        __builtin_amdgcn_sched_barrier(0);               \
        var = __builtin_amdgcn_s_memtime();              \
        __builtin_amdgcn_sched_barrier(0);               \
    } while (0)
#else
#define MSDA_STAMP(var) \
    do {                \
    } while (0)
#endif

// LDSL kernels: the workgroup's waves take their work — runs of 64 / G queries out of the workgroup's query range — from
// a counter in LDS instead of a fixed share, and start it staggered (wave w waits w * Params::lds_stagger * 64 cycles).
// Sixteen waves released by one barrier otherwise run their phases in lockstep — all wait for their sampling points,
// then all gather from memory, then all read LDS — and the three pipes take turns instead of overlapping (measured at
// c2 @ 10k: a chunk took the SUM of its phases, forward 83 us; the plain kernel's waves are staggered by the dispatcher).
__device__ __forceinline__ int *gather_work_counter()
{
    return reinterpret_cast<int *>(msda_smem + (sizeof(LevelTab) + 15) / 16 * 16);
}
__device__ __forceinline__ int next_slice(int lane)
{
    int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(gather_work_counter(), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readfirstlane(t);
}
// Two planes per workgroup (Params::lds_planes == 2): a counter per plane; a wave takes its next slice from the plane that
// has MORE slices left (ties: its home plane), so the two planes of a workgroup finish together whatever their rows cost —
// the rows of one head can be 20 % slower to gather than its neighbour's (HISTORY.md 4 item 5).  Returns the slice (>= nslices:
// both planes are done) and the plane in `hp`.
__device__ __forceinline__ int next_slice2(int lane, int home, int nslices, int &hp)
{
    int t = 0, pick = 0;
    if (lane == 0) {
        int *c = gather_work_counter();
        const int c0 = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int c1 = __hip_atomic_load(c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        pick = c0 == c1 ? home : (c1 < c0 ? 1 : 0);
        t = __hip_atomic_fetch_add(c + pick, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (t >= nslices) {
            pick ^= 1;
            t = __hip_atomic_fetch_add(c + pick, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    hp = __builtin_amdgcn_readfirstlane(pick);
    return __builtin_amdgcn_readfirstlane(t);
}
__device__ __forceinline__ void stagger_wave(int wave, int units)
{
    for (int i = 0; i < wave * units; ++i) __builtin_amdgcn_s_sleep(1);
}

// LDSL kernels: which levels are served from LDS — the longest suffix of the level list whose pixels fit `budget` bytes
// (pyramids list their levels fine to coarse; any other order only costs the speed-up).  Uniform: every thread derives
// the same answer from the level table.
struct CoarseLevels {
    int first;   // levels [first, L) are LDS-resident (first == L: none)
    int pixels;  // ... their pixels, rows in level-packed order
};
__device__ __forceinline__ CoarseLevels coarse_levels(const LevelTab *tab, int L, uint32_t row_bytes, int budget)
{
    CoarseLevels c{L, 0};
    const int cap = budget / (int)row_bytes;
    for (int l = L - 1; l >= 0; --l) {
        const int n = tab->h[l] * tab->w[l];
        if (n > cap - c.pixels) break;
        c.pixels += n;
        c.first = l;
    }
    return c;
}
// ... and the copy itself: every thread of the workgroup moves 16-byte (VEC elements of TV) pieces of the rows of levels
// [first, L) from the plane into LDS at `lds_off` (rounded up to 128), a row of zeros behind them; ends with a barrier.
struct CoarseStage {
    int first;                  // levels [first, L) are LDS-resident
    uint32_t base, zero;        // LDS byte offsets of the first row and of the row of zeros
    uint32_t row_bytes;         // D * sizeof(TV): distance of two rows in LDS
    uint32_t plane_stride;      // Params::lds_planes == 2: bytes from one plane's copy to the other's
};
// Next-slice prefetch WITHOUT registers (LDSL kernels): the sampling inputs of a wave's next slice travel from memory
// straight into a per-wave LDS staging area (global_load_lds_*: LDS address = M0 + lane * size, inactive lanes write
// nothing).  Issued from inline assembly: the compiler does not see these loads, so no ds_read waits for them; the
// consumer's wait is the explicit dma_wait() the kernels place behind phase 2 — by then the slice's own gathers have
// returned (vector-memory loads return in order, so the older staging loads have too) and the wait costs nothing.
constexpr int kStagePre = 2;                                  // samples per lane and slice
constexpr uint32_t kStageWaveBytes = kStagePre * kWave * 12;  // per wave: [trip][x | y | a][lane] floats
__device__ __forceinline__ void dma_dword(const void *g, uint32_t lds_uniform)
{
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(lds_uniform) : "memory", "m0");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <typename TV, int VEC, int BLK>
__device__ __forceinline__ CoarseStage stage_coarse_levels(const LevelTab *tab, const Params &p, rsrc_t rs, uint32_t plane_row_bytes, size_t lds_off,
                                                           int half = 0)
{
    // Params::lds_planes == 2: each half of the workgroup's threads stages ITS plane's levels into its own region
    const int nthr = p.lds_planes == 2 ? BLK / 2 : BLK;
    constexpr uint32_t kPiece = VEC * sizeof(TV);  // bytes a lane loads of a row
    CoarseStage cs;
    cs.row_bytes = (uint32_t)p.D * (uint32_t)sizeof(TV);
    const CoarseLevels cl = coarse_levels(tab, p.L, cs.row_bytes, p.lds_lev_bytes);
    cs.first = cl.first;
    cs.plane_stride = (((uint32_t)cl.pixels + 1) * cs.row_bytes + 127) / 128 * 128;  // bytes between the two planes' copies
    cs.base = (uint32_t)((lds_off + 127) / 128 * 128) + (uint32_t)half * cs.plane_stride;
    cs.zero = cs.base + (uint32_t)cl.pixels * cs.row_bytes;
    const int ppr = (int)(cs.row_bytes / kPiece);  // pieces per row
    const int npieces = cl.pixels * ppr;
    const float inv_ppr = 1.0f / (float)ppr;
    const uint32_t first_row = (uint32_t)(cl.first < p.L ? tab->start[cl.first] : 0);
    using RLV = RawLoad<kPiece>;
    const int tid = p.lds_planes == 2 ? (int)threadIdx.x - half * nthr : (int)threadIdx.x;
    if (tid == 0) gather_work_counter()[half] = 0;
    for (int i = tid; i < npieces; i += nthr) {
        const int r = div_small(i, ppr, inv_ppr), c = i - imul24(r, ppr);
        const typename RLV::type v = RLV::load(rs, mul24(first_row + (uint32_t)r, plane_row_bytes) + (uint32_t)c * kPiece);
        *reinterpret_cast<typename RLV::type *>(msda_smem + cs.base + (uint32_t)i * kPiece) = v;
    }
    for (int i = tid; i < ppr; i += nthr) {
        typename RLV::type z{};
        *reinterpret_cast<typename RLV::type *>(msda_smem + cs.zero + (uint32_t)i * kPiece) = z;
    }
    __syncthreads();
    return cs;
}
// VEC consecutive elements of an LDS-resident row at byte offset `off` of the workgroup's LDS, widened
// Params::touch (small forwards on cold caches): the workgroups of a plane request one dword of every row of the plane
// — row first + k * stride, k < N per thread — right behind their sampling points, so that the rows stream into the XCD's
// L2 while the points and the level table are on their way, and the gather's dependent trips find them there.  The
// values are not used: retire() only keeps the loads alive (and is where the wave waits for them — behind the barrier
// for the level table, which it waits for anyway).
template <int N> struct Touch {
    uint32_t v[N];
    __device__ __forceinline__ void issue(const rsrc_t &rs, uint32_t row_bytes, int first, int stride, int rows)
    {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int i = first + k * stride;
            v[k] = RawLoad<4>::load(rs, i < rows ? (uint32_t)i * row_bytes : kMaskedOffset);
        }
    }
    __device__ __forceinline__ void retire() const
    {
#pragma unroll
        for (int k = 0; k < N; ++k) asm volatile("" ::"v"(v[k]));
    }
};

template <typename T, int VEC> __device__ __forceinline__ void lds_row(uint32_t off, typename Traits<T>::acc (&dst)[VEC])
{
    const Pack<T, VEC> pk = *reinterpret_cast<const Pack<T, VEC> *>(msda_smem + off);
#pragma unroll
    for (int i = 0; i < VEC; ++i) dst[i] = Traits<T>::to_acc(pk.v[i]);
}

// The gather kernels' arguments, requested in ONE batch of scalar loads at the top of the kernel: left to itself the
// compiler loads each field next to its first use, which put a second kernel-argument miss (~1 us on a cold cache)
// behind the first one on the way to the first sample — a tenth of a decoder-sized forward.
__device__ __forceinline__ void request_all_arguments(const Params &p)
{
    asm volatile("" ::"s"(p.value), "s"(p.shapes), "s"(p.loc), "s"(p.attn), "s"(p.out), "s"(p.grad_out), "s"(p.grad_loc), "s"(p.grad_attn));
    asm volatile("" ::"s"(p.B), "s"(p.I), "s"(p.H), "s"(p.D), "s"(p.Q), "s"(p.L), "s"(p.P), "s"(p.LP), "s"(p.nqc), "s"(p.qw), "s"(p.sc),
                 "s"(p.zeros), "s"(p.align), "s"(p.xcd_map), "s"(p.grid3d), "s"(p.div_h.magic), "s"(p.div_h.shift), "s"(p.ref),
                 "s"(p.ref_dim), "s"(p.lds_lev_bytes));
}

// ==========================================================================================
// forward.  After the one-time staging barrier every wave runs on its own: it parks the records of
// ITS 64/G units, gathers, stores, and moves to its next query chunk without any block barrier.
// ==========================================================================================
// TV: storage type of the value rows (T unless the caller keeps value in 16 bits next to fp32 coordinates / weights /
// output: the "mixed" entry points, msda_mixed.hip)
// (Round 4's x-pair table for 64-byte rows — every row stored twice so that a footprint's two x-corners come from ONE
//  128-byte line: 15.5 -> 8.1 ps per sample in the gather alone, profiles/r04_row_pair_bench.txt — was removed in
//  round 5: inside the kernels it netted -3 us forward / +7 us backward at c3 and stayed off; HISTORY.md 3.6.)
// LDSL (BLK = kBlockLds threads): the coarsest levels of the plane — the longest SUFFIX of the level list whose rows fit
// Params::lds_lev_bytes — are copied into LDS once per workgroup and their samples read from there (coarse_levels()).
// The gather is bound by the vector-memory path (64 B/clk/CU; HISTORY.md 4); an LDS row read costs a quarter of that,
// on another pipe.  One large workgroup per CU, so that 16 waves share ONE copy (round 2 tried it with 256-thread
// workgroups: the copies ate the occupancy).  Same arithmetic in the same order: results are bit-identical.
// (Measured and dropped in round 5: variants with 8 / 16 samples' rows in flight per lane for small grids.  On a cold
//  cache a batch of 16 loads per lane takes ~2 500 cycles, four of them per unit — but 64 loads at once took 2.5x as
//  long as the four batches together (in-kernel clock, Q = 10 / 100).  What small grids want is more WAVES with few
//  loads each: msda_fwd_unit_kernel below.)
// TS (module kernels, FUSED): storage type of the projection and of `out` when it differs from the arithmetic type T —
// 16-bit projections / results next to fp32 reference points and fp32 arithmetic (msda_*_fused_f32_sbf16 / _sf16)
// Occupancy target of the 256-thread forward: five waves per SIMD (102 VGPRs) — except the variants that do not fit it
// without scratch: 16-bit operators with the prologue fused in or with 8-byte pieces (VEC 4), and double accumulation
// (1-18 VGPR spills under the cap, round 5 / 6; tests/test_host_api.py reads every kernel's spill count from the library)
template <typename T, int VEC, bool FUSED> constexpr int fwd_waves_per_eu()
{
    return (sizeof(T) == 2 && (FUSED || VEC * sizeof(T) < 16)) || sizeof(typename Traits<T>::acc) == 8 ? 4 : 5;
}
template <typename T, int VEC, int G, bool FUSED, typename TV = T, int BLK = kBlock, bool LDSL = false, typename TS = T>
__global__ __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(BLK == kBlock ? fwd_waves_per_eu<T, VEC, FUSED>() : 4))) void msda_fwd_kernel(const Params p)
{
    using SR = Traits<TS>;
    static_assert(FUSED || sizeof(TS) == sizeof(T), "a separate storage type exists for the module kernels only");
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    static_assert(sizeof(typename Traits<TV>::acc) == sizeof(A), "value rows widen to the same accumulate type");
    static_assert(!(LDSL && VEC == 1), "the LDS-served levels use the 16-byte vector path");
    request_all_arguments(p);
    constexpr int NU = BLK / G;        // units per workgroup and query chunk
    constexpr int UPW = kWave / G;     // units per wave

    const int slots = (p.nqc + p.qw - 1) / p.qw;
    int pair, slot;
    // LDSL with Params::lds_planes == 2: the workgroup serves the planes 2 * pair and 2 * pair + 1 (neighbouring heads of
    // one batch element); `half` (wave-uniform) is a wave's home plane, `hp` below the plane of the slice it works on
    const int two = LDSL && p.lds_planes == 2;
    if (!decode_block(p.grid3d, two ? (p.B * p.H) >> 1 : p.B * p.H, slots, p.xcd_map, pair, slot)) return;
    const int half = two ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >= BLK / 2)) : 0;
    const int pair0 = two ? 2 * pair : pair;  // (the workgroup's first plane)
    pair = pair0 + half;
    const int b = (int)fast_div((uint32_t)pair, p.div_h);
    int h = pair - b * p.H;

    const int scp = p.sc + 1;  // +1 record of padding: units land on different LDS banks
    const GatherLds<A> lds(NU, scp);
    LevelTab *tab = lds.tab;

    // row stride and plane: the rows of one head are H*D elements apart
    const uint32_t row_bytes = (uint32_t)p.v_row;
    rsrc_t rs = make_rsrc(plane_base<TV>(p, b, h), plane_span<TV>(p, h));  // (home plane; re-made per slice when the wave works on the other one)

    const int tid = threadIdx.x;
    const int wave = tid / kWave, lane = tid % kWave;
    const int wunit = lane / G, j = lane % G;  // unit inside the wave, lane inside the unit (across the row's channels)
    uint4 *w_off = lds.s_off + wave * UPW * scp;
    Rec4<A> *w_rec = lds.s_rec + wave * UPW * scp;
    // per-plane bases (64-bit, uniform) + 32-bit per-sample indices (the host checks Q*H*L*P*3 < 2^31)
    size_t plane_s0 = ((size_t)b * p.Q * p.H + h) * p.LP;
    const T *loc = static_cast<const T *>(p.loc) + 2 * plane_s0;
    [[maybe_unused]] const TS *proj3 = static_cast<const TS *>(p.loc) + 3 * plane_s0;  // FUSED: raw projection (dx, dy, logit)
    const T *attn = FUSED ? nullptr : static_cast<const T *>(p.attn) + plane_s0;
    // the wave's plane-dependent values for plane pair0 + hp_ (two planes per workgroup: per slice)
    auto select_plane = [&](int hp_) {
        h = pair0 + hp_ - b * p.H;
        rs = make_rsrc(plane_base<TV>(p, b, h), plane_span<TV>(p, h));
        plane_s0 = ((size_t)b * p.Q * p.H + h) * p.LP;
        loc = static_cast<const T *>(p.loc) + 2 * plane_s0;
        proj3 = static_cast<const TS *>(p.loc) + 3 * plane_s0;
        attn = FUSED ? nullptr : static_cast<const T *>(p.attn) + plane_s0;
    };
    const T *refp = FUSED ? static_cast<const T *>(p.ref) + (size_t)b * p.Q * p.ref_dim : nullptr;
    const int HLP = p.H * p.LP;
    const float inv_P = 1.0f / (float)p.P;
    const int nchan_chunks = (p.D + G * VEC - 1) / (G * VEC);
    const int qc_end = min(p.nqc, (slot + 1) * p.qw);

    // The workgroup's FIRST chunk of samples is requested before the level table is waited for (the addresses do not
    // need it), so the two round trips overlap.  Same-box A/B at c2 @ 10k (two alternations): fwd 0.0985 -> 0.0954 ms
    // warm, 0.129 -> 0.123 ms by the cold-cache do_bench recipe; nothing at Q <= 1000.  (An earlier look at this through
    // rocprofv3 runs on different boxes had called it noise.)  Plain operator, one channel chunk, all L * P samples
    // parked at once, at most kPre per lane.
    constexpr int kPre = 2;
    const bool pre = !FUSED && nchan_chunks == 1 && p.sc >= p.LP && UPW * p.LP <= kPre * kWave;
    Pack<T, 2> pxy[kPre];
    T pa[kPre];
    // request the sampling points and weights of the wave's units [wq, wq + UPW) (queries below qlim), one sample per
    // lane and trip
    auto prefetch = [&](int wq, int qlim, int hp_ = -1) {
        // (two planes per workgroup: the NEXT slice may belong to the other plane — its own bases, the current ones stay)
        const T *loc_ = loc, *attn_ = attn;
        if (hp_ >= 0) {
            const size_t ps_ = ((size_t)b * p.Q * p.H + (size_t)(pair0 + hp_ - b * p.H)) * p.LP;
            loc_ = static_cast<const T *>(p.loc) + 2 * ps_;
            attn_ = FUSED ? nullptr : static_cast<const T *>(p.attn) + ps_;
        }
#pragma unroll
        for (int t = 0; t < kPre; ++t) {
            if constexpr (!FUSED) {
                const int f = lane + t * kWave;
                const int fu = div_small(f, p.LP, 1.0f / (float)p.LP);
                const int fq = wq + fu;
#ifdef MSDA_DEV  // ablation 1024: no sampling-point loads (whatever the registers hold)
                if (p.debug & 1024) continue;
#endif
                if (f < UPW * p.LP && fq < qlim) {
                    const int sidx = imul24(fq, HLP) + (f - imul24(fu, p.LP));
                    pxy[t] = *reinterpret_cast<const Pack<T, 2> *>(loc_ + 2 * sidx);
                    pa[t] = attn_[sidx];
                }
            }
        }
    };
#pragma unroll
    for (int t = 0; t < kPre; ++t) pxy[t].v[0] = pxy[t].v[1] = pa[t] = TR::from_acc((A)0);
    if (!LDSL && pre) prefetch((slot * p.qw) * NU + wave * UPW, p.Q);
    Touch<4> touch;
    if (p.touch) {  // (two planes per workgroup: each half touches its own plane)
        const int nthr = two ? BLK / 2 : BLK;
        touch.issue(rs, row_bytes, slot * nthr + (tid - half * (BLK / 2)), slots * nthr, p.I);
    }
    load_level_table(tab, p.shapes, p.L);
    __syncthreads();
    if (p.touch) touch.retire();
    // LDSL: levels [fl, L) live in LDS behind the records, rows D * sizeof(TV) bytes apart, then one row of zeros
    // (what a corner masked by "zeros" padding reads)
    CoarseStage cs{p.L, 0, 0, (uint32_t)p.D * (uint32_t)sizeof(TV)};
    if constexpr (LDSL)
        cs = stage_coarse_levels<TV, VEC, BLK>(tab, p, rs, row_bytes, kGatherLdsFixed + (size_t)NU * scp * (sizeof(uint4) + sizeof(Rec4<A>)), half);
    const uint32_t cs_base0 = cs.base - (uint32_t)half * cs.plane_stride, cs_zero_rel = cs.zero - cs.base;
    const int fl = cs.first;

    // LDSL: the workgroup's queries [q_lo, q_hi) go to its waves slice by slice (next_slice)
    const int q_lo = imul24(slot * p.qw, NU), q_hi = min(p.Q, imul24(qc_end, NU));
    const int q_end_ = LDSL ? q_hi : p.Q;  // queries beyond it are not this workgroup's
    // LDSL: a wave takes its NEXT slice as soon as phase 1 has consumed the current one's sampling points and requests
    // that slice's points at once, so that their trip to memory runs under the gather instead of in front of the next
    // phase 1 (a wave works through ~10 slices one after the other: what bounds these kernels is the latency chain of a
    // slice, not a pipe — TA 53 %, LDS 35 %, VALU 41 % busy at c2 @ 10k)
    int t_next = 0;
    bool have_next = false;
    [[maybe_unused]] unsigned long long clk_t0 = 0, clk_a = 0, clk_b = 0, clk_c = 0, clk_d = 0, clk_e = 0;
    [[maybe_unused]] float clk_ph[5] = {0, 0, 0, 0, 0};
    MSDA_STAMP(clk_t0);
    int hp_next = half;  // (plane of the slice in t_next)
    const int nslices = (q_hi - q_lo + UPW - 1) / UPW;
    auto grab = [&]() { t_next = two ? next_slice2(lane, half, nslices, hp_next) : next_slice(lane); };
    if constexpr (LDSL) {
        stagger_wave(wave, p.lds_stagger);
        grab();
        have_next = true;
        if (pre) prefetch(q_lo + t_next * UPW, q_hi, two ? hp_next : -1);
    }
    for (int it = 0;; ++it) {
        int wq0;  // first query of this wave (wave-uniform)
        MSDA_STAMP(clk_a);
        if constexpr (LDSL) {
            if (!have_next) grab();
            have_next = false;
            wq0 = q_lo + t_next * UPW;
            if (wq0 >= q_hi) break;
            if (two) {  // this slice's plane: descriptor, sample bases, LDS copy of its levels
                select_plane(hp_next);
                cs.base = cs_base0 + (uint32_t)hp_next * cs.plane_stride;
                cs.zero = cs.base + cs_zero_rel;
            }
        } else {
            const int qc = slot * p.qw + it;
            if (qc >= qc_end) break;
            wq0 = qc * NU + wave * UPW;
            if (wq0 >= p.Q) break;
        }
        const int q = wq0 + wunit;
        const bool unit_ok = q < (LDSL ? q_hi : p.Q);
        const bool use_pre = pre && (LDSL || it == 0);
        for (int cc = 0; cc < nchan_chunks; ++cc) {
            const int c0 = (cc * G + j) * VEC;
            const bool lane_ok = unit_ok && (c0 < p.D);
            const uint32_t lane_off = (uint32_t)c0 * (uint32_t)sizeof(TV);
            A acc[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = (A)0;

            for (int s0 = 0; s0 < p.LP; s0 += p.sc) {
                const int sc = min(p.sc, p.LP - s0);
                const float inv_sc = 1.0f / (float)sc;
                wave_lds_sync();  // previous records consumed
                if constexpr (FUSED) {
                    // ---- phase 0 (module prologue, frontend.py:253-282; sc == LP): ONE round of global loads — every
                    // lane fetches its samples' (x offset, y offset, logit) and reference point, forms the sampling
                    // point and parks {logit, px, py}; then the unit's lanes take max and sum(exp) over the parked
                    // logits (DPP-reduced) and leave them in the unit's padding slot ----
                    for (int f = lane; f < UPW * sc; f += kWave) {
                        const int fu = div_small(f, sc, inv_sc);
                        const int sl = f - imul24(fu, sc);
                        const int fq = wq0 + fu;
                        if (fq < q_end_) {
                            const int l = div_small(sl, p.P, inv_P);
                            const int sidx = imul24(fq, HLP) + sl;
                            const A ox = SR::to_acc(proj3[3 * sidx]), oy = SR::to_acc(proj3[3 * sidx + 1]);
                            const A lg = SR::to_acc(proj3[3 * sidx + 2]);
                            const T *r = refp + (size_t)fq * p.ref_dim;
                            Rec4<A> w;
                            w.v[0] = lg;
                            if (p.ref_dim == 2) {
                                // NB: (x, y) offsets are divided by img_shapes in its stored (h, w) order (frontend.py:275)
                                w.v[1] = TR::to_acc(r[0]) + ox / (A)tab->h[l];
                                w.v[2] = TR::to_acc(r[1]) + oy / (A)tab->w[l];
                            } else {
                                w.v[1] = TR::to_acc(r[0]) + ox * TR::to_acc(r[2]) / (A)(2 * p.P);
                                w.v[2] = TR::to_acc(r[1]) + oy * TR::to_acc(r[3]) / (A)(2 * p.P);
                            }
                            w.v[3] = (A)0;
                            w_rec[imul24(fu, scp) + sl] = w;
                        }
                    }
                    wave_lds_sync();
                    if (unit_ok) {
                        A mx = -__builtin_huge_val();
                        for (int sl = j; sl < sc; sl += G) mx = fmax_t(mx, w_rec[imul24(wunit, scp) + sl].v[0]);
                        mx = group_max<G>(mx);
                        A sum = (A)0;
                        for (int sl = j; sl < sc; sl += G) sum += exp_t(w_rec[imul24(wunit, scp) + sl].v[0] - mx);
                        sum = group_sum<G>(sum);
                        if (j == 0) {
                            w_rec[imul24(wunit, scp) + sc].v[0] = mx;
                            w_rec[imul24(wunit, scp) + sc].v[1] = (A)1 / sum;
                        }
                    }
                    wave_lds_sync();
                }
                // ---- phase 1: the wave's UPW * sc samples, one per lane and trip ----
                auto tap_sample = [&](int f, bool have, const Pack<T, 2> &hxy, T ha) {
                    const int fu = div_small(f, sc, inv_sc);
                    const int sl = s0 + (f - imul24(fu, sc));
                    const int fq = wq0 + fu;
                    if (fq < q_end_) {
                        const int l = div_small(sl, p.P, inv_P);
                        const int sidx = imul24(fq, HLP) + sl;
                        A sx, sy, a;
                        if constexpr (FUSED) {  // everything was parked by phase 0
                            const Rec4<A> pk = w_rec[imul24(fu, scp) + (sl - s0)], un = w_rec[imul24(fu, scp) + sc];
                            sx = pk.v[1];
                            sy = pk.v[2];
                            a = exp_t(pk.v[0] - un.v[0]) * un.v[1];
                        } else if (have) {
                            sx = TR::to_acc(hxy.v[0]);
                            sy = TR::to_acc(hxy.v[1]);
                            a = TR::to_acc(ha);
                        } else {
                            const Pack<T, 2> xy = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
                            sx = TR::to_acc(xy.v[0]);
                            sy = TR::to_acc(xy.v[1]);
                            a = TR::to_acc(attn[sidx]);
                        }
                        Taps<A> t;
                        if constexpr (LDSL) {  // an LDS-served level: offsets into the workgroup's copy (selects, not two code paths)
                            const bool inl = l >= fl;
                            make_taps<A>(sx, sy, tab->h[l], tab->w[l], tab->start[l] - (inl ? tab->start[cs.first] : 0), p.zeros, p.align,
                                         inl ? cs.row_bytes : row_bytes, t, inl ? cs.base : 0u, inl ? cs.zero : kMaskedOffset);
#ifdef MSDA_DEV  // ablation 16384 (WRONG results): LDS rows forced onto the bank half of the unit's parity — what a
                 // conflict-free order of the corner reads would be worth
                            if ((p.debug & 16384) && inl) {
                                const uint32_t par = (uint32_t)(fu & 1) << 7;
                                t.off[0] = cs.base + (((t.off[0] - cs.base) & ~128u) | par);
                                t.off[2] = cs.base + (((t.off[2] - cs.base) & ~128u) | par);
                                t.off[1] = cs.base + (((t.off[1] - cs.base) & ~128u) | (par ^ 128u));
                                t.off[3] = cs.base + (((t.off[3] - cs.base) & ~128u) | (par ^ 128u));
                            }
#endif
                        } else {
                            make_taps<A>(sx, sy, tab->h[l], tab->w[l], tab->start[l], p.zeros, p.align, row_bytes, t);
                        }
                        const A wy0 = (A)1 - t.dy, wx0 = (A)1 - t.dx;
                        Rec4<A> w;
                        w.v[0] = a * (wy0 * wx0);
                        w.v[1] = a * (wy0 * t.dx);
                        w.v[2] = a * (t.dy * wx0);
                        w.v[3] = a * (t.dy * t.dx);
                        const int rslot = imul24(fu, scp) + (sl - s0);
                        w_off[rslot] = make_uint4(t.off[0], t.off[1], t.off[2], t.off[3]);
                        w_rec[rslot] = w;
                    }
                };
                if (use_pre) {
#pragma unroll
                    for (int t = 0; t < kPre; ++t)
                        if (lane + t * kWave < UPW * sc) tap_sample(lane + t * kWave, true, pxy[t], pa[t]);
                    if constexpr (LDSL) {  // (pre: one channel chunk, one trip — this runs once per slice)
                        grab();
                        have_next = true;
                        prefetch(q_lo + t_next * UPW, q_hi, two ? hp_next : -1);
                    }
                } else {
                    for (int f = lane; f < UPW * sc; f += kWave) tap_sample(f, false, pxy[0], pa[0]);
                }
                wave_lds_sync();
                MSDA_STAMP(clk_b);
                // ---- phase 2: gather + blend ----
                if (lane_ok) {
                    const uint4 *uo = w_off + imul24(wunit, scp);
                    const Rec4<A> *uw = w_rec + imul24(wunit, scp);
                    // samples [0, s_lds) of this trip gather from memory, [s_lds, sc) from the LDS-resident levels
                    const int s_lds = LDSL ? min(max(imul24(fl, p.P) - s0, 0), sc) : sc;
#ifdef MSDA_DEV  // ablations (msda_set_option("debug", mask)): 256 no memory gather, 512 no LDS gather
                    const int s_mem_end = (p.debug & 256) ? 0 : s_lds, s_lds_end = (p.debug & 512) ? s_lds : sc;
#else
                    const int s_mem_end = s_lds, s_lds_end = sc;
#endif
                    // (a rolling window of 16 loads in flight — the next sample's loads issued as soon as the oldest sample's
                    // rows are blended — measured no faster than "issue 16, wait": 75.2 vs 74.9 us at c2 @ 10k, round 5)
#pragma unroll 4
                    for (int s = 0; s < s_mem_end; ++s) {
                        const uint4 o = uo[s];
                        const Rec4<A> w = uw[s];
                        A v0[VEC], v1[VEC], v2[VEC], v3[VEC];
                        load_row<TV, VEC>(rs, o.x + lane_off, v0);
                        load_row<TV, VEC>(rs, o.y + lane_off, v1);
                        load_row<TV, VEC>(rs, o.z + lane_off, v2);
                        load_row<TV, VEC>(rs, o.w + lane_off, v3);
                        blend4<VEC, sizeof(TV) == sizeof(A)>(acc, w.v, v0, v1, v2, v3);
                    }
#ifdef MSDA_DEV
                    if (p.debug & 2048) asm volatile("" ::"v"(acc[0]));  // (the clock must see the gathered rows consumed)
#endif
                    MSDA_STAMP(clk_c);
                    if constexpr (LDSL) {
#pragma unroll 4
                        for (int s = s_lds; s < s_lds_end; ++s) {
                            const uint4 o = uo[s];
                            const Rec4<A> w = uw[s];
                            A v0[VEC], v1[VEC], v2[VEC], v3[VEC];
                            lds_row<TV, VEC>(o.x + lane_off, v0);
                            lds_row<TV, VEC>(o.y + lane_off, v1);
                            lds_row<TV, VEC>(o.z + lane_off, v2);
                            lds_row<TV, VEC>(o.w + lane_off, v3);
                            blend4<VEC, sizeof(TV) == sizeof(A)>(acc, w.v, v0, v1, v2, v3);
                        }
                    }
                }
            }
            if (lane_ok) {
                Pack<TS, VEC> o;
#pragma unroll
                for (int i = 0; i < VEC; ++i) o.v[i] = SR::from_acc(acc[i]);
                TS *dst = static_cast<TS *>(p.out) + ((size_t)(b * (size_t)p.Q + q) * p.H + h) * p.D + c0;
#ifdef MSDA_DEV
                asm volatile("" ::"v"(o.v[0]));
                MSDA_STAMP(clk_d);
                if (!(p.debug & 2048))
#endif
                store_stream(dst, o);
            }
            MSDA_STAMP(clk_e);
#ifdef MSDA_DEV
            clk_ph[0] += (float)(long long)(clk_b - clk_a);
            clk_ph[1] += (float)(long long)(clk_c - clk_b);
            clk_ph[2] += (float)(long long)(clk_d - clk_c);
            clk_ph[3] += (float)(long long)(clk_e - clk_d);
            clk_ph[4] += 1.0f;
#endif
        }
    }
#ifdef MSDA_DEV
    if ((p.debug & 2048) && lane == 0) {
        unsigned long long clk_end;
        MSDA_STAMP(clk_end);
        const int wg = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
        float *dbg = reinterpret_cast<float *>(p.out) + ((size_t)wg * (BLK / kWave) + wave) * 8;
        dbg[0] = clk_ph[0];
        dbg[1] = clk_ph[1];
        dbg[2] = clk_ph[2];
        dbg[3] = (float)(__builtin_amdgcn_s_getreg(0x1814) & 15);  // HW_REG_XCC_ID (in place of the store phase)
        dbg[4] = clk_ph[4];
        dbg[5] = (float)(long long)(clk_end - clk_t0);
        dbg[6] = (float)wg;
        dbg[7] = 12345.0f;
    }
#endif
}

// ==========================================================================================
// forward for SMALL problems (decoder calls: a few hundred queries): one WAVE per unit (b, q, h).  The lanes are
// R = 64 / GL groups of GL lanes (GL * VEC = D channels); a load instruction fetches R different (sample, corner) rows
// of the unit, so its 4 * L * P rows take 4 * L * P / R instructions (8 for L * P = 16, 128-byte rows), all in flight at
// once — where msda_fwd_kernel's wave serves eight units in four batches of sixteen loads, one memory round trip per
// batch.  On a nearly empty chip the kernel IS its chain of round trips (arguments, points + level sizes, rows): cold
// cache, Q = 100: 13.2 us against the Triton comparator's 9.0 (profiles/r04_query_sweep_*; VERDICT r04 missing #3).
// Phase 1: lanes < L * P compute one sample's taps each and leave {offset, weight} per (sample, corner) in the wave's
// LDS slice; phase 2: lane (r, j) blends the pairs r, r + R, ...; the R partial rows meet through lane shuffles.
// ==========================================================================================
// The arguments in front of `p` are the ones the first round of loads needs (sampling points, weights, level sizes):
// the build asks for them to be PRELOADED into SGPRs at wave launch (-amdgpu-kernarg-preload-count, Makefile), so that
// round leaves without waiting for the kernel-argument segment — one memory trip less in a chain of four.
// U (1 or 2): units per wave.  U = 2 gives each half of the wave a unit: the same instructions serve two units (most of
// phase 1 runs with 16 of 64 lanes doing anything), a unit's rows take twice the load instructions, all still in flight
// together.  Measured (msda_launch.hpp): ahead by ~1 us at Q = 200-300 when the rows come from HBM, behind by 0.3-0.6 us
// when they are cached — option "unit_waves", off by default.
template <typename T, int VEC, typename TV = T, int U = 1>
__global__ __launch_bounds__(kWave) void msda_fwd_unit_kernel(const void *a_loc, const void *a_attn, const int64_t *a_shapes, int a_LP, int a_L,
                                                              int a_units, const Params p)
{
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    static_assert(sizeof(A) == 4, "float accumulation (the shuffles below move 32-bit values)");
    static_assert(U == 1 || U == 2, "one unit per wave or per half wave");
    // one WAVE per 64-thread workgroup: a few hundred one-wave workgroups land on different CUs, each with the CU's
    // texture path to itself (four-wave workgroups: +0.5 ... 0.8 us at Q = 10 ... 300, cold)
    constexpr int UL = kWave / U;  // lanes of a unit
    const int lane = threadIdx.x % kWave;
    const int part = lane / UL, ul = lane - part * UL;  // the lane's unit inside the wave, the lane inside the unit
    const int unit = (int)blockIdx.x * U + part;  // (b * Q + q) * H + h
    const bool live = unit < a_units;
    // the unit's samples and the level sizes: requested from the preloaded arguments alone
    const size_t s_base = (size_t)(live ? unit : 0) * a_LP;
    Pack<T, 2> xy;
    T at;
    xy.v[0] = xy.v[1] = at = TR::from_acc((A)0);
    const bool has = live && ul < a_LP;
    if (has) {
        xy = *reinterpret_cast<const Pack<T, 2> *>(static_cast<const T *>(a_loc) + 2 * (s_base + ul));
        at = static_cast<const T *>(a_attn)[s_base + ul];
    }
    LevelTab *tab = reinterpret_cast<LevelTab *>(msda_smem);
    load_level_table(tab, a_shapes, a_L);
    __syncthreads();
    if (U == 1 && !live) return;  // (wave-uniform; no barrier below)
    request_all_arguments(p);
    const int GL = p.D / VEC;            // lanes across a row (the host checks: a power of two, <= 64 / U, D % VEC == 0)
    const int R = UL / GL;               // rows of a unit per load instruction
    const int r = ul / GL, j = ul - r * GL;
    // per unit of the wave: [4 * LP] offsets, [4 * LP] weights
    uint32_t *w_off = reinterpret_cast<uint32_t *>(msda_smem + kGatherLdsFixed) + (size_t)part * 8 * p.LP;
    A *w_wgt = reinterpret_cast<A *>(w_off + 4 * p.LP);
    const int u_ = live ? unit : 0;
    const int bq = (int)fast_div((uint32_t)u_, p.div_h), h = u_ - bq * p.H;
    const int b = U == 1 ? bq / p.Q : (int)fast_div((uint32_t)bq, p.div_win);  // (U == 2: the host put Q's divider into div_win)
    const uint32_t row_bytes = (uint32_t)p.v_row;
    // U == 2: the halves may sit on different planes, so the descriptor covers the whole tensor and the plane is an offset
    // (the host checks B * I * v_row < 2^31 for this variant)
    const rsrc_t rs = U == 1 ? make_rsrc(plane_base<TV>(p, b, h), plane_span<TV>(p, h))
                             : make_rsrc(p.value, (uint32_t)((size_t)(p.B - 1) * p.I * (size_t)p.v_row) + plane_span<TV>(p, 0));
    const uint32_t plane_off = U == 1 ? 0u : (uint32_t)((size_t)b * p.I * (size_t)p.v_row + (size_t)h * p.D * sizeof(TV));
    for (int s0 = 0; s0 < p.LP; s0 += UL) {  // (L * P <= 64 / U: one trip)
        const int sl = s0 + ul;
        if (s0 > 0) {
            wave_lds_sync();
            if (live && sl < p.LP) {
                xy = *reinterpret_cast<const Pack<T, 2> *>(static_cast<const T *>(p.loc) + 2 * (s_base + sl));
                at = static_cast<const T *>(p.attn)[s_base + sl];
            }
        }
        if (sl < p.LP) {
            const int l = div_small(sl, p.P, 1.0f / (float)p.P);
            Taps<A> t;
            make_taps<A>(TR::to_acc(xy.v[0]), TR::to_acc(xy.v[1]), tab->h[l], tab->w[l], tab->start[l], p.zeros, p.align, row_bytes, t, plane_off);
            const A a = TR::to_acc(at), wy0 = (A)1 - t.dy, wx0 = (A)1 - t.dx;
            *reinterpret_cast<uint4 *>(w_off + 4 * sl) = make_uint4(t.off[0], t.off[1], t.off[2], t.off[3]);
            Rec4<A> w;
            w.v[0] = a * (wy0 * wx0);
            w.v[1] = a * (wy0 * t.dx);
            w.v[2] = a * (t.dy * wx0);
            w.v[3] = a * (t.dy * t.dx);
            *reinterpret_cast<Rec4<A> *>(w_wgt + 4 * sl) = w;
        }
    }
    wave_lds_sync();
    // phase 2: pair index = 4 * sample + corner; lane group r takes the pairs r, r + R, ... — in ascending order, so every
    // group's partial sum is a fixed sub-sequence of the reference's sample order
    const int npairs = 4 * p.LP;
    const uint32_t lane_off = (uint32_t)j * VEC * (uint32_t)sizeof(TV);
    A acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = (A)0;
    using RLV = RawLoad<sizeof(TV) * VEC>;
    constexpr int kFly = 8 * U;  // load instructions in flight together
    for (int p0 = r; p0 < npairs; p0 += kFly * R) {
        Pack<TV, VEC> v[kFly];
        A wv[kFly];
#pragma unroll
        for (int u = 0; u < kFly; ++u) {
            const int pi = p0 + u * R;
            const bool on = pi < npairs;
            const uint32_t o = on ? w_off[pi] : kMaskedOffset;  // (beyond the last pair: an out-of-range offset, zeros, weight 0)
            wv[u] = on ? w_wgt[pi] : (A)0;
            v[u] = __builtin_bit_cast(Pack<TV, VEC>, RLV::load(rs, o == kMaskedOffset ? o : o + lane_off));
        }
#pragma unroll
        for (int u = 0; u < kFly; ++u) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fma_t(wv[u], Traits<TV>::to_acc(v[u].v[i]), acc[i]);
        }
    }
    // the R partial rows of the unit: lanes j, j + GL, j + 2 GL, ... hold the same channels
    for (int m = GL; m < UL; m <<= 1) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], m, kWave);
    }
    if (r == 0 && live) {
        Pack<T, VEC> o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) o.v[i] = TR::from_acc(acc[i]);
        store_stream(static_cast<T *>(p.out) + (size_t)unit * p.D + j * VEC, o);
    }
}

// ==========================================================================================
// backward, part 1: grad_loc and grad_attn.  Same decomposition as the forward; every sample's
// three results are reduced over the unit's G lanes with DPP moves and written exactly once.
// ==========================================================================================
// LDSL (BLK = kBlockLds): the coarsest levels served from LDS, as in the forward (reduce-scatter units only).
template <typename T, int VEC, int G, bool FUSED, typename TV = T, int BLK = kBlock, bool LDSL = false, typename TS = T>
__global__ __launch_bounds__(BLK) void msda_bwd_sample_kernel(const Params p)
{
    using SR = Traits<TS>;  // (FUSED only: storage of the projection, grad_out and grad_proj; see msda_fwd_kernel)
    static_assert(FUSED || sizeof(TS) == sizeof(T), "a separate storage type exists for the module kernels only");
    using A = typename Traits<T>::acc;
    using TR = Traits<T>;
    static_assert(sizeof(typename Traits<TV>::acc) == sizeof(A), "value rows widen to the same accumulate type");
    static_assert(!LDSL || (VEC != 1 && sizeof(A) == 4 && !TR::kDot2 && (G == 4 || G == 8)), "LDS-served levels: reduce-scatter units");
    request_all_arguments(p);
    constexpr int NU = BLK / G;
    constexpr int UPW = kWave / G;
    // units of 4 / 8 lanes hand every sample's dot products to ONE owner lane (reduce-scatter) instead of all-reducing
    // (16-bit operators keep the all-reduce: with v_dot2c rows the two cost the same instructions, and the all-reduce
    // measured 5-6 % faster at c3 / c5)
    constexpr bool kScatter = sizeof(A) == 4 && !TR::kDot2 && (G == 4 || G == 8);

    const int slots = (p.nqc + p.qw - 1) / p.qw;
    int pair, slot;
    // LDSL with Params::lds_planes == 2: two planes per workgroup, the waves take slices of either (see msda_fwd_kernel)
    // (the plain kernel never runs it — msda_launch.hpp — and is compiled without: the extra live scalars cost it 2-4 %)
    const int two = LDSL && FUSED && p.lds_planes == 2;
    if (!decode_block(p.grid3d, two ? (p.B * p.H) >> 1 : p.B * p.H, slots, p.xcd_map, pair, slot)) return;
    const int half = two ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >= BLK / 2)) : 0;
    const int pair0 = two ? 2 * pair : pair;
    pair = pair0 + half;
    const int b = (int)fast_div((uint32_t)pair, p.div_h);
    int h = pair - b * p.H;

    const int scp = p.sc + 1;
    // record in : {dx, dy, a*sx*gx_on, a*sy*gy_on};  record out (same slot): {gA, gX, gY, -}
    const GatherLds<A> lds(NU, scp, FUSED);
    LevelTab *tab = lds.tab;

    const uint32_t row_bytes = (uint32_t)p.v_row;
    rsrc_t rs = make_rsrc(plane_base<TV>(p, b, h), plane_span<TV>(p, h));  // (home plane; re-made per slice when the wave works on the other one)

    const int tid = threadIdx.x;
    const int wave = tid / kWave, lane = tid % kWave;
    const int wunit = lane / G, j = lane % G;  // unit inside the wave, lane across the row's channels
    uint4 *w_off = lds.s_off + wave * UPW * scp;
    Rec4<A> *w_rec = lds.s_rec + wave * UPW * scp;
    // per-plane bases (64-bit, uniform) + 32-bit per-sample indices (the host checks Q*H*L*P*2 < 2^31)
    size_t plane_s0 = ((size_t)b * p.Q * p.H + h) * p.LP;
    const T *loc = static_cast<const T *>(p.loc) + 2 * plane_s0;
    [[maybe_unused]] const TS *proj3 = static_cast<const TS *>(p.loc) + 3 * plane_s0;  // FUSED: raw projection (dx, dy, logit)
    const T *attn = FUSED ? nullptr : static_cast<const T *>(p.attn) + plane_s0;
    auto select_plane = [&](int hp_) {  // the plane-dependent values for plane pair0 + hp_
        h = pair0 + hp_ - b * p.H;
        rs = make_rsrc(plane_base<TV>(p, b, h), plane_span<TV>(p, h));
        plane_s0 = ((size_t)b * p.Q * p.H + h) * p.LP;
        loc = static_cast<const T *>(p.loc) + 2 * plane_s0;
        proj3 = static_cast<const TS *>(p.loc) + 3 * plane_s0;
        attn = FUSED ? nullptr : static_cast<const T *>(p.attn) + plane_s0;
    };
    const T *refp = FUSED ? static_cast<const T *>(p.ref) + (size_t)b * p.Q * p.ref_dim : nullptr;
    A *w_aux = lds.s_aux + wave * UPW * scp * 3;  // FUSED: [slot] = a, [UPW*scp + slot] = ox, [2*UPW*scp + slot] = oy
    A *w_a = w_aux, *w_ox = w_aux + UPW * scp, *w_oy = w_aux + 2 * UPW * scp;
    const A half_inv_P = (A)1 / (A)(2 * p.P);
    const int HLP = p.H * p.LP;
    const float inv_P = 1.0f / (float)p.P;
    const int nchan_chunks = (p.D + G * VEC - 1) / (G * VEC);
    const int qc_end = min(p.nqc, (slot + 1) * p.qw);

    // the workgroup's first chunk of samples requested before the level table is waited for, as in the forward
    constexpr int kPre = 2;
    // (not with LDS-served levels: the wave does not know its first queries before the staging barrier)
    const bool pre = !FUSED && !LDSL && p.sc >= p.LP && UPW * p.LP <= kPre * kWave;
    // LDSL: the next slice's points and weights go to the wave's LDS staging area instead (dma_dword: no registers)
    constexpr bool kDma = LDSL && !FUSED && sizeof(T) == 4;
    // (two planes per workgroup: the staging areas do not fit next to two copies of the levels)
    const bool pre_dma = kDma && !two && p.sc >= p.LP && UPW * p.LP <= kStagePre * kWave;
    const size_t rec_bytes = kGatherLdsFixed + (size_t)NU * (p.sc + 1) * (sizeof(uint4) + sizeof(Rec4<A>) + (FUSED ? 3 * sizeof(A) : 0));
    const uint32_t stage_lds = __builtin_amdgcn_readfirstlane((uint32_t)rec_bytes + (uint32_t)(threadIdx.x / kWave) * kStageWaveBytes);
    // (M0 wants the absolute LDS address: the dynamic area starts behind any static __shared__ object)
    const uint32_t stage_abs = stage_lds + __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char *)msda_smem);
    [[maybe_unused]] auto stage_request = [&](int wq, int qlim) {
#pragma unroll
        for (int t = 0; t < kStagePre; ++t) {
            const int f = (int)(threadIdx.x % kWave) + t * kWave;
            const int fu = div_small(f, p.LP, 1.0f / (float)p.LP);
            const int fq = wq + fu;
            if (f < UPW * p.LP && fq < qlim) {
                const int sidx = imul24(fq, HLP) + (f - imul24(fu, p.LP));
                dma_dword(loc + 2 * sidx, stage_abs + (uint32_t)t * (3 * kWave * 4));
                dma_dword(loc + 2 * sidx + 1, stage_abs + (uint32_t)t * (3 * kWave * 4) + kWave * 4);
                dma_dword(attn + sidx, stage_abs + (uint32_t)t * (3 * kWave * 4) + 2 * kWave * 4);
            }
        }
    };
    Pack<T, 2> pxy[kPre];
    T pa[kPre];
#pragma unroll
    for (int t = 0; t < kPre; ++t) {
        pxy[t].v[0] = pxy[t].v[1] = pa[t] = TR::from_acc((A)0);
        if constexpr (!FUSED) {
            const int f = lane + t * kWave;
            const int fu = div_small(f, p.LP, 1.0f / (float)p.LP);
            const int fq = (slot * p.qw) * NU + wave * UPW + fu;
            if (pre && f < UPW * p.LP && fq < p.Q) {
                const int sidx = imul24(fq, HLP) + (f - imul24(fu, p.LP));
                pxy[t] = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
                pa[t] = attn[sidx];
            }
        }
    }
    load_level_table(tab, p.shapes, p.L);
    __syncthreads();
    CoarseStage cs{p.L, 0, 0, (uint32_t)p.D * (uint32_t)sizeof(TV)};
    if constexpr (LDSL)
        cs = stage_coarse_levels<TV, VEC, BLK>(tab, p, rs, row_bytes, rec_bytes + (kDma && !two ? (size_t)(BLK / kWave) * kStageWaveBytes : 0), half);
    const uint32_t cs_base0 = cs.base - (uint32_t)half * cs.plane_stride, cs_zero_rel = cs.zero - cs.base;
    // levels [fl, L) are SERVED from LDS: the first staged level whose first sample starts an exchange batch of G samples
    // in every trip (P = 4 / 8: every level; a staged level in front of it is simply not used)
    int fl = cs.first;
    if constexpr (LDSL) {
        while (fl < p.L && (imul24(fl, p.P) % G) != 0) ++fl;
        if (p.sc < p.LP && (p.sc % G) != 0) fl = p.L;
    }

    // LDSL: the workgroup's queries [q_lo, q_hi) go to its waves slice by slice (next_slice)
    const int q_lo = imul24(slot * p.qw, NU), q_hi = min(p.Q, imul24(qc_end, NU));
    const int q_end_ = LDSL ? q_hi : p.Q;  // queries beyond it are not this workgroup's
    // LDSL: the NEXT slice is taken, and its points requested, as soon as this slice's are consumed (as in the forward)
    int t_next = 0, hp_next = half;
    bool have_next = false;
    const int nslices = (q_hi - q_lo + UPW - 1) / UPW;
    auto grab = [&]() { t_next = two ? next_slice2(lane, half, nslices, hp_next) : next_slice(lane); };
    if constexpr (LDSL) {
        stagger_wave(wave, p.lds_stagger);
        if constexpr (kDma) {
            if (pre_dma) {
                grab();
                have_next = true;
                stage_request(q_lo + t_next * UPW, q_hi);
                dma_wait();
            }
        }
    }
    for (int it = 0;; ++it) {
        int wq0;  // first query of this wave (wave-uniform)
        if constexpr (LDSL) {
            if (!have_next) grab();
            have_next = false;
            wq0 = q_lo + t_next * UPW;
            if (wq0 >= q_hi) break;
            if (two) {
                select_plane(hp_next);
                cs.base = cs_base0 + (uint32_t)hp_next * cs.plane_stride;
                cs.zero = cs.base + cs_zero_rel;
            }
        } else {
            const int qc = slot * p.qw + it;
            if (qc >= qc_end) break;
            wq0 = qc * NU + wave * UPW;
            if (wq0 >= p.Q) break;
        }
        const int q = wq0 + wunit;
        const bool unit_ok = q < (LDSL ? q_hi : p.Q);
        const bool use_pre = pre && it == 0;
        for (int s0 = 0; s0 < p.LP; s0 += p.sc) {
            const int sc = min(p.sc, p.LP - s0);
            const float inv_sc = 1.0f / (float)sc;
            wave_lds_sync();
            if constexpr (FUSED) {
                // ---- phase 0 (module prologue, frontend.py:253-282; sc == LP), as in the fused forward: one round of
                // global loads parks {logit, px, py} (and the raw offsets for the box-size gradient), then the unit's
                // lanes leave max and 1 / sum(exp) of its logits in the padding slot ----
                for (int f = lane; f < UPW * sc; f += kWave) {
                    const int fu = div_small(f, sc, inv_sc);
                    const int sl = f - imul24(fu, sc);
                    const int fq = wq0 + fu;
                    if (fq < q_end_) {
                        const int l = div_small(sl, p.P, inv_P);
                        const int sidx = imul24(fq, HLP) + sl;
                        const A ox = SR::to_acc(proj3[3 * sidx]), oy = SR::to_acc(proj3[3 * sidx + 1]);
                        const A lg = SR::to_acc(proj3[3 * sidx + 2]);
                        const T *r = refp + (size_t)fq * p.ref_dim;
                        Rec4<A> w;
                        w.v[0] = lg;
                        if (p.ref_dim == 2) {
                            w.v[1] = TR::to_acc(r[0]) + ox / (A)tab->h[l];  // NB: (x, y) / img_shapes in its stored (h, w) order
                            w.v[2] = TR::to_acc(r[1]) + oy / (A)tab->w[l];
                        } else {
                            w.v[1] = TR::to_acc(r[0]) + ox * TR::to_acc(r[2]) * half_inv_P;
                            w.v[2] = TR::to_acc(r[1]) + oy * TR::to_acc(r[3]) * half_inv_P;
                        }
                        w.v[3] = (A)0;
                        w_rec[imul24(fu, scp) + sl] = w;
                        w_ox[imul24(fu, scp) + sl] = ox;
                        w_oy[imul24(fu, scp) + sl] = oy;
                    }
                }
                wave_lds_sync();
                if (unit_ok) {
                    A mx = -__builtin_huge_val();
                    for (int sl = j; sl < sc; sl += G) mx = fmax_t(mx, w_rec[imul24(wunit, scp) + sl].v[0]);
                    mx = group_max<G>(mx);
                    A sum = (A)0;
                    for (int sl = j; sl < sc; sl += G) sum += exp_t(w_rec[imul24(wunit, scp) + sl].v[0] - mx);
                    sum = group_sum<G>(sum);
                    if (j == 0) {
                        w_rec[imul24(wunit, scp) + sc].v[0] = mx;
                        w_rec[imul24(wunit, scp) + sc].v[1] = (A)1 / sum;
                    }
                }
                wave_lds_sync();
            }
            // ---- phase 1 ----
            auto tap_sample = [&](int f, bool have, const Pack<T, 2> &hxy, T ha) {
                const int fu = div_small(f, sc, inv_sc);
                const int sl = s0 + (f - imul24(fu, sc));
                const int fq = wq0 + fu;
                if (fq < q_end_) {
                    const int l = div_small(sl, p.P, inv_P);
                    const int sidx = imul24(fq, HLP) + sl;
                    const int lh = tab->h[l], lw = tab->w[l];
                    A px, py, a;
                    if constexpr (FUSED) {
                        // everything was parked by phase 0
                        const int rs_ = imul24(fu, scp) + (sl - s0);
                        const Rec4<A> pk = w_rec[rs_], un = w_rec[imul24(fu, scp) + sc];
                        px = pk.v[1];
                        py = pk.v[2];
                        a = exp_t(pk.v[0] - un.v[0]) * un.v[1];
                        w_a[rs_] = a;
                    } else if (have) {
                        px = TR::to_acc(hxy.v[0]);
                        py = TR::to_acc(hxy.v[1]);
                        a = TR::to_acc(ha);
                    } else {
                        const Pack<T, 2> xy = *reinterpret_cast<const Pack<T, 2> *>(loc + 2 * sidx);
                        px = TR::to_acc(xy.v[0]);
                        py = TR::to_acc(xy.v[1]);
                        a = TR::to_acc(attn[sidx]);
                    }
                    Taps<A> t;
                    if constexpr (LDSL) {  // an LDS-served level: offsets into the workgroup's copy (selects, not two code paths)
                        const bool inl = l >= fl;
                        make_taps<A>(px, py, lh, lw, tab->start[l] - (inl ? tab->start[cs.first] : 0), p.zeros, p.align,
                                     inl ? cs.row_bytes : row_bytes, t, inl ? cs.base : 0u, inl ? cs.zero : kMaskedOffset);
                    } else {
                        make_taps<A>(px, py, lh, lw, tab->start[l], p.zeros, p.align, row_bytes, t);
                    }
                    const A sx = p.align ? (A)(lw - 1) : (A)lw;
                    const A sy = p.align ? (A)(lh - 1) : (A)lh;
                    Rec4<A> r;
                    r.v[0] = t.dx;
                    r.v[1] = t.dy;
                    r.v[2] = t.gx_on ? a * sx : (A)0;
                    r.v[3] = t.gy_on ? a * sy : (A)0;
                    const int rslot = imul24(fu, scp) + (sl - s0);
                    w_off[rslot] = make_uint4(t.off[0], t.off[1], t.off[2], t.off[3]);
                    w_rec[rslot] = r;
                }
            };
            if (use_pre) {
#pragma unroll
                for (int t = 0; t < kPre; ++t)
                    if (lane + t * kWave < UPW * sc) tap_sample(lane + t * kWave, true, pxy[t], pa[t]);
            } else if (pre_dma) {
                if constexpr (kDma) {
                    const float *st = reinterpret_cast<const float *>(msda_smem + stage_lds);
#pragma unroll
                    for (int t = 0; t < kStagePre; ++t) {
                        if (lane + t * kWave < UPW * sc) {
                            Pack<T, 2> hxy;
                            hxy.v[0] = st[t * 3 * kWave + lane];
                            hxy.v[1] = st[t * 3 * kWave + kWave + lane];
                            tap_sample(lane + t * kWave, true, hxy, st[t * 3 * kWave + 2 * kWave + lane]);
                        }
                    }
                    wave_lds_sync();  // (staging area read before it is requested again)
                    grab();
                    have_next = true;
                    stage_request(q_lo + t_next * UPW, q_hi);
                }
            } else {
                for (int f = lane; f < UPW * sc; f += kWave) tap_sample(f, false, pxy[0], pa[0]);
            }
            wave_lds_sync();
            // FUSED: softmax backward needs dot = sum_s a_s * gA_s over the unit; the reference point's gradient is
            // the sum of the sampling points' (times the offsets, for the box size)
            A f_dot = (A)0, f_gx = (A)0, f_gy = (A)0, f_gw = (A)0, f_gh = (A)0;
            // ---- phase 2: four dot products with grad_out per sample, reduced over the unit ----
            if (unit_ok) {  // idle lanes of a live unit still join the DPP sums
                const uint4 *uo = w_off + imul24(wunit, scp);
                Rec4<A> *up = w_rec + imul24(wunit, scp);
                const TS *go_row = static_cast<const TS *>(p.grad_out) + ((size_t)(b * (size_t)p.Q + q) * p.H + h) * p.D;
                // one sample's epilogue: combine the four dot products, reduce over the unit, park the result
                auto finish = [&](int s, const Rec4<A> &r, A d0, A d1, A d2, A d3) {
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
                };
                if (kScatter && nchan_chunks == 1) {
                    // fast path, units of 4 or 8 lanes: the partial dot products of G samples are reduce-SCATTERED over
                    // the unit (quad_steps / half_step, msda_common.hpp) — lane j ends up with the four complete dot
                    // products of sample sb + j, combines them ONCE and parks the result itself.  (Until round 5 every
                    // lane combined every sample and three all-reduces followed: 50 M vector instructions per launch at
                    // c2 @ 10k against the forward's 21 M for the same gather.)  Rows of UB samples in flight at once;
                    // an 8-lane unit runs two such half-batches per exchange.
                    if constexpr (kScatter) {
                        constexpr int UB = 4, NH = G / UB;
                        const int c0 = j * VEC;
                        const bool lane_in = c0 < p.D;
                        const uint32_t lo = lane_in ? (uint32_t)c0 * (uint32_t)sizeof(TV) : 0u;
                        Pack<TS, VEC> gp;
#pragma unroll
                        for (int i = 0; i < VEC; ++i) gp.v[i] = SR::from_acc((A)0);
                        if (lane_in) gp = *reinterpret_cast<const Pack<TS, VEC> *>(go_row + c0);
                        A g[VEC];
#pragma unroll
                        for (int i = 0; i < VEC; ++i) g[i] = SR::to_acc(gp.v[i]);
                        using RLV = RawLoad<sizeof(TV) * VEC>;
                        // one exchange: the samples [sb, sb + G) of this trip, their rows from memory or (LDS: a compile-time
                        // tag, so that each loop below is straight-line code) from the LDS-resident levels
                        auto batch = [&](int sb, auto lds_tag) {
                            constexpr bool kLds = decltype(lds_tag)::value;
                            float e[UB][4];
#pragma unroll
                            for (int hb = 0; hb < NH; ++hb) {
                                Pack<TV, VEC> v[UB][4];  // (kept packed until they are consumed)
#pragma unroll
                                for (int u = 0; u < UB; ++u) {
                                    const uint4 o = uo[min(sb + hb * UB + u, sc - 1)];  // tail: the last sample again, not stored
                                    if constexpr (kLds) {
                                        v[u][0] = *reinterpret_cast<const Pack<TV, VEC> *>(msda_smem + o.x + lo);
                                        v[u][1] = *reinterpret_cast<const Pack<TV, VEC> *>(msda_smem + o.y + lo);
                                        v[u][2] = *reinterpret_cast<const Pack<TV, VEC> *>(msda_smem + o.z + lo);
                                        v[u][3] = *reinterpret_cast<const Pack<TV, VEC> *>(msda_smem + o.w + lo);
                                    } else {
                                        v[u][0] = __builtin_bit_cast(Pack<TV, VEC>, RLV::load(rs, o.x + lo));
                                        v[u][1] = __builtin_bit_cast(Pack<TV, VEC>, RLV::load(rs, o.y + lo));
                                        v[u][2] = __builtin_bit_cast(Pack<TV, VEC>, RLV::load(rs, o.z + lo));
                                        v[u][3] = __builtin_bit_cast(Pack<TV, VEC>, RLV::load(rs, o.w + lo));
                                    }
                                }
#pragma unroll
                                for (int u = 0; u < UB; ++u) {
#pragma unroll
                                    for (int k = 0; k < 4; ++k) {
                                        // the partial dot product as TWO partial sums (fp32 rows: even and odd channels, two
                                        // v_pk_fma_f32 for four channels); an 8-lane unit's first exchange step adds them up
                                        // on the way (half_step2), a 4-lane unit adds them here
                                        float d_lo = 0.0f, d_hi = 0.0f;
                                        if constexpr (TR::kDot2 && (VEC % 2) == 0 && sizeof(TV) == sizeof(T)) {
                                            // 16-bit rows: straight from the packed pairs (v_dot2c_f32_f16 / _bf16)
                                            using P2 = typename TR::pair_t;
                                            struct Pairs {
                                                P2 p[VEC / 2];
                                            };
                                            const Pairs gq = __builtin_bit_cast(Pairs, gp), vq = __builtin_bit_cast(Pairs, v[u][k]);
#pragma unroll
                                            for (int i = 0; i < VEC / 2; i += 2) {
                                                d_lo = TR::dot2(gq.p[i], vq.p[i], d_lo);
                                                if (i + 1 < VEC / 2) d_hi = TR::dot2(gq.p[i + 1], vq.p[i + 1], d_hi);
                                            }
                                        } else if constexpr ((VEC % 2) == 0 && sizeof(TV) == sizeof(A)) {
                                            f32x2 a2 = {0.0f, 0.0f};
#pragma unroll
                                            for (int i = 0; i < VEC; i += 2)
                                                a2 = __builtin_elementwise_fma(f32x2{g[i], g[i + 1]},
                                                                               f32x2{Traits<TV>::to_acc(v[u][k].v[i]), Traits<TV>::to_acc(v[u][k].v[i + 1])}, a2);
                                            d_lo = a2.x;
                                            d_hi = a2.y;
                                        } else if constexpr ((VEC % 2) == 0) {
                                            // (rows widened from 16 bits next to fp32 everything else: the same two chains, so
                                            // that the result is bit-identical to the fp32 kernel's on the rounded rows)
#pragma unroll
                                            for (int i = 0; i < VEC; i += 2) {
                                                d_lo = fma_t(g[i], Traits<TV>::to_acc(v[u][k].v[i]), d_lo);
                                                d_hi = fma_t(g[i + 1], Traits<TV>::to_acc(v[u][k].v[i + 1]), d_hi);
                                            }
                                        } else {
#pragma unroll
                                            for (int i = 0; i < VEC; ++i) d_lo = fma_t(g[i], Traits<TV>::to_acc(v[u][k].v[i]), d_lo);
                                        }
                                        if constexpr (G == 8) {
                                            if (hb == 0) e[u][k] = half_step2(d_lo, d_hi);
                                            else if (u == UB - 1 && k == 3) half_step2_upper<true>(e[u][k], d_lo, d_hi);
                                            else half_step2_upper<false>(e[u][k], d_lo, d_hi);
                                        } else {
                                            e[u][k] = d_lo + d_hi;
                                        }
                                    }
                                }
                            }
                            float dd[4];
                            quad_steps<4>(e, j, dd);
                            const int s = sb + j;  // this lane's sample
                            const Rec4<A> r = up[min(s, sc - 1)];
                            const A dx = r.v[0], dy = r.v[1];
                            const A wy0 = (A)1 - dy, wx0 = (A)1 - dx;
                            Rec4<A> res;
                            res.v[0] = (wy0 * wx0) * dd[0] + (wy0 * dx) * dd[1] + (dy * wx0) * dd[2] + (dy * dx) * dd[3];
                            res.v[1] = r.v[2] * (wy0 * (dd[1] - dd[0]) + dy * (dd[3] - dd[2]));
                            res.v[2] = r.v[3] * (wx0 * (dd[2] - dd[0]) + dx * (dd[3] - dd[1]));
                            res.v[3] = (A)0;
                            if (s < sc) up[s] = res;
                        };
                        // samples [0, s_lds) of this trip gather from memory, [s_lds, sc) from the LDS-resident levels (the
                        // boundary is a multiple of G: `fl` was chosen that way)
                        const int s_lds = LDSL ? min(max(imul24(fl, p.P) - s0, 0), sc) : sc;
                        for (int sb = 0; sb < s_lds; sb += G) batch(sb, std::false_type{});
                        if constexpr (LDSL)
                            for (int sb = s_lds; sb < sc; sb += G) batch(sb, std::true_type{});
                    }
                } else if (nchan_chunks == 1) {
                    // fast path: grad_out row in registers, UB samples' rows (4 * UB loads) in flight at once
                    constexpr int UB = 4;
                    const int c0 = j * VEC;
                    const bool lane_in = c0 < p.D;
                    const uint32_t lane_off = (uint32_t)c0 * (uint32_t)sizeof(TV);
                    Pack<TS, VEC> gp;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) gp.v[i] = SR::from_acc((A)0);
                    if (lane_in) gp = *reinterpret_cast<const Pack<TS, VEC> *>(go_row + c0);
                    if constexpr (TR::kDot2 && (VEC % 2) == 0 && sizeof(TV) == sizeof(T)) {
                        // 16-bit rows: the four dot products with grad_out straight from the packed pairs
                        // (v_dot2c_f32_f16 / _bf16: two multiply-adds per instruction, no widening)
                        using P2 = typename TR::pair_t;
                        struct Pairs {
                            P2 p[VEC / 2];
                        };
                        const Pairs gq = __builtin_bit_cast(Pairs, gp);
                        for (int sb = 0; sb < sc; sb += UB) {
                            uint4 o[UB];
                            Rec4<A> r[UB];
                            Pairs v[UB][4];
#pragma unroll
                            for (int u = 0; u < UB; ++u) {
                                const int s = min(sb + u, sc - 1);  // tail: recompute the last sample, its store is skipped
                                o[u] = uo[s];
                                r[u] = up[s];
                            }
                            using RL = RawLoad<sizeof(T) * VEC>;
                            const uint32_t lo = lane_in ? lane_off : 0u;
#pragma unroll
                            for (int u = 0; u < UB; ++u) {
                                v[u][0] = __builtin_bit_cast(Pairs, RL::load(rs, o[u].x + lo));
                                v[u][1] = __builtin_bit_cast(Pairs, RL::load(rs, o[u].y + lo));
                                v[u][2] = __builtin_bit_cast(Pairs, RL::load(rs, o[u].z + lo));
                                v[u][3] = __builtin_bit_cast(Pairs, RL::load(rs, o[u].w + lo));
                            }
#pragma unroll
                            for (int u = 0; u < UB; ++u) {
                                A d0 = 0, d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
                                for (int i = 0; i < VEC / 2; ++i) {
                                    d0 = TR::dot2(gq.p[i], v[u][0].p[i], d0);
                                    d1 = TR::dot2(gq.p[i], v[u][1].p[i], d1);
                                    d2 = TR::dot2(gq.p[i], v[u][2].p[i], d2);
                                    d3 = TR::dot2(gq.p[i], v[u][3].p[i], d3);
                                }
                                if (sb + u < sc) finish(sb + u, r[u], d0, d1, d2, d3);  // uniform
                            }
                        }
                    } else {
                    A g[VEC];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) g[i] = SR::to_acc(gp.v[i]);
                    for (int sb = 0; sb < sc; sb += UB) {
                        uint4 o[UB];
                        Rec4<A> r[UB];
                        A v[UB][4][VEC];
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            const int s = min(sb + u, sc - 1);  // tail: recompute the last sample, its store is skipped
                            o[u] = uo[s];
                            r[u] = up[s];
                        }
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            const uint32_t lo = lane_in ? lane_off : 0u;
                            load_row<TV, VEC>(rs, o[u].x + lo, v[u][0]);
                            load_row<TV, VEC>(rs, o[u].y + lo, v[u][1]);
                            load_row<TV, VEC>(rs, o[u].z + lo, v[u][2]);
                            load_row<TV, VEC>(rs, o[u].w + lo, v[u][3]);
                        }
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            A d0 = 0, d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
                            for (int i = 0; i < VEC; ++i) {
                                d0 = fma_t(g[i], v[u][0][i], d0);
                                d1 = fma_t(g[i], v[u][1][i], d1);
                                d2 = fma_t(g[i], v[u][2][i], d2);
                                d3 = fma_t(g[i], v[u][3][i], d3);
                            }
                            if (sb + u < sc) finish(sb + u, r[u], d0, d1, d2, d3);  // uniform
                        }
                    }
                    }
                } else {
                    for (int s = 0; s < sc; ++s) {
                        const uint4 o = uo[s];
                        const Rec4<A> r = up[s];
                        A d0 = 0, d1 = 0, d2 = 0, d3 = 0;
                        for (int cc = 0; cc < nchan_chunks; ++cc) {
                            const int c0 = (cc * G + j) * VEC;
                            if (c0 < p.D) {
                                const uint32_t lane_off = (uint32_t)c0 * (uint32_t)sizeof(TV);
                                const Pack<TS, VEC> gp = *reinterpret_cast<const Pack<TS, VEC> *>(go_row + c0);
                                A v0[VEC], v1[VEC], v2[VEC], v3[VEC];
                                load_row<TV, VEC>(rs, o.x + lane_off, v0);
                                load_row<TV, VEC>(rs, o.y + lane_off, v1);
                                load_row<TV, VEC>(rs, o.z + lane_off, v2);
                                load_row<TV, VEC>(rs, o.w + lane_off, v3);
#pragma unroll
                                for (int i = 0; i < VEC; ++i) {
                                    const A gg = SR::to_acc(gp.v[i]);
                                    d0 = fma_t(gg, v0[i], d0);
                                    d1 = fma_t(gg, v1[i], d1);
                                    d2 = fma_t(gg, v2[i], d2);
                                    d3 = fma_t(gg, v3[i], d3);
                                }
                            }
                        }
                        finish(s, r, d0, d1, d2, d3);
                    }
                }
            }
            if constexpr (FUSED) {
                // ---- phase 2b: per-unit sums for the prologue's chain rule, off the gather's critical path: the
                // unit's lanes stride over its parked results, DPP-reduce ----
                wave_lds_sync();
                if (unit_ok) {
                    for (int sl = j; sl < sc; sl += G) {
                        const int rs_ = imul24(wunit, scp) + sl;
                        const Rec4<A> res = w_rec[rs_];
                        f_dot = fma_t(w_a[rs_], res.v[0], f_dot);
                        f_gx += res.v[1];
                        f_gy += res.v[2];
                        f_gw = fma_t(res.v[1], w_ox[rs_], f_gw);
                        f_gh = fma_t(res.v[2], w_oy[rs_], f_gh);
                    }
                    f_dot = group_sum<G>(f_dot);
                    f_gx = group_sum<G>(f_gx);
                    f_gy = group_sum<G>(f_gy);
                    if (p.ref_dim == 4) {
                        f_gw = group_sum<G>(f_gw);
                        f_gh = group_sum<G>(f_gh);
                    }
                }
                if (unit_ok && j == 0) {
                    w_a[imul24(wunit, scp) + sc] = f_dot;  // the unit's padding slot
                    // per-head partial of grad_reference_points; the caller sums over the heads
                    T *gr = static_cast<T *>(p.grad_attn) + ((size_t)(b * (size_t)p.Q + q) * p.H + h) * p.ref_dim;
                    gr[0] = TR::from_acc(f_gx);
                    gr[1] = TR::from_acc(f_gy);
                    if (p.ref_dim == 4) {
                        gr[2] = TR::from_acc(f_gw * half_inv_P);
                        gr[3] = TR::from_acc(f_gh * half_inv_P);
                    }
                }
            }
            wave_lds_sync();
            if constexpr (kDma) {
                if (pre_dma) dma_wait();  // the next slice's inputs have landed (they are older than this slice's gathers)
            }
            // ---- phase 3: coalesced write-out, one sample per lane and trip ----
            for (int f = lane; f < UPW * sc; f += kWave) {
                const int fu = div_small(f, sc, inv_sc);
                const int sl = s0 + (f - imul24(fu, sc));
                const int fq = wq0 + fu;
                if (fq < q_end_) {
                    const int sidx = imul24(fq, HLP) + sl;
                    const Rec4<A> res = w_rec[imul24(fu, scp) + (sl - s0)];
                    if constexpr (FUSED) {
                        // chain rule through the prologue: softmax (logit), offset scaling (dx, dy)
                        const int l = div_small(sl, p.P, inv_P);
                        const int rs_ = imul24(fu, scp) + (sl - s0);
                        const A a = w_a[rs_], dot = w_a[imul24(fu, scp) + sc];
                        const T *r = refp + (size_t)fq * p.ref_dim;
                        A kx, ky;
                        if (p.ref_dim == 2) {
                            kx = (A)1 / (A)tab->h[l];
                            ky = (A)1 / (A)tab->w[l];
                        } else {
                            kx = TR::to_acc(r[2]) * half_inv_P;
                            ky = TR::to_acc(r[3]) * half_inv_P;
                        }
#ifdef MSDA_DEV  // ablations: 4096 no parked points / weights, 8192 no grad_proj stores
                        const bool abl_mat = p.debug & 4096, abl_gp = p.debug & 8192;
#else
                        constexpr bool abl_mat = false, abl_gp = false;
#endif
                        if (p.mat_loc != nullptr && !abl_mat) {
                            // the sampling point and attention weight this kernel derived, for the grad_value passes
                            // (stored here, at the end of the wave's life, where nothing waits behind the stores)
                            const A ox = w_ox[rs_], oy = w_oy[rs_];
                            Pack<T, 2> m;
                            if (p.ref_dim == 2) {
                                m.v[0] = TR::from_acc(TR::to_acc(r[0]) + ox / (A)tab->h[l]);
                                m.v[1] = TR::from_acc(TR::to_acc(r[1]) + oy / (A)tab->w[l]);
                            } else {
                                m.v[0] = TR::from_acc(TR::to_acc(r[0]) + ox * TR::to_acc(r[2]) * half_inv_P);
                                m.v[1] = TR::from_acc(TR::to_acc(r[1]) + oy * TR::to_acc(r[3]) * half_inv_P);
                            }
                            *reinterpret_cast<Pack<T, 2> *>(static_cast<T *>(p.mat_loc) + 2 * (plane_s0 + sidx)) = m;
                            static_cast<T *>(p.mat_attn)[plane_s0 + sidx] = TR::from_acc(a);
                        }
                        Pack<TS, 1> g0, g1, g2;
                        g0.v[0] = SR::from_acc(res.v[1] * kx);
                        g1.v[0] = SR::from_acc(res.v[2] * ky);
                        g2.v[0] = SR::from_acc(a * (res.v[0] - dot));
                        TS *gp = static_cast<TS *>(p.grad_loc) + 3 * (plane_s0 + sidx);
                        if (!abl_gp) {
                            store_stream(gp, g0);
                            store_stream(gp + 1, g1);
                            store_stream(gp + 2, g2);
                        }
                    } else {
                        Pack<T, 1> ga;
                        ga.v[0] = TR::from_acc(res.v[0]);
                        store_stream(static_cast<T *>(p.grad_attn) + plane_s0 + sidx, ga);
                        Pack<T, 2> g;
                        g.v[0] = TR::from_acc(res.v[1]);
                        g.v[1] = TR::from_acc(res.v[2]);
                        store_stream(static_cast<T *>(p.grad_loc) + 2 * (plane_s0 + sidx), g);
                    }
                }
            }
        }
    }
}

}  // namespace msda
