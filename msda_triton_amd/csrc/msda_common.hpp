// msda_common.hpp — device-side building blocks shared by the forward and backward kernels.
// gfx950 (MI355X / CDNA4) only: wave64, buffer loads with hardware range checking, DPP lane moves.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msda_hip.h"

namespace msda {

constexpr int kWave = 64;
constexpr int kBlock = 256;          // threads per workgroup of the gather kernels
constexpr int kBlockLds = 1024;      // ... of their variants that keep the coarsest levels in LDS (one workgroup per CU)
constexpr int kMaxLevels = MSDA_MAX_LEVELS;
constexpr uint32_t kMaskedOffset = 0x80000000u;  // >= any buffer size we accept -> buffer_load returns 0

// ------------------------------------------------------------------------------------------
// storage type traits: how many bytes, what we accumulate in, how to convert
// ------------------------------------------------------------------------------------------
template <typename T> struct Traits;
template <> struct Traits<float> {
    using acc = float;
    static constexpr bool kDot2 = false;
    static __device__ __forceinline__ float to_acc(float v) { return v; }
    static __device__ __forceinline__ float from_acc(float v) { return v; }
};
template <> struct Traits<double> {
    using acc = double;
    static constexpr bool kDot2 = false;
    static __device__ __forceinline__ double to_acc(double v) { return v; }
    static __device__ __forceinline__ double from_acc(double v) { return v; }
};
template <> struct Traits<_Float16> {
    using acc = float;
    // c + a.x * b.x + a.y * b.y in one instruction (v_dot2c_f32_f16): dot products of two 16-bit rows need no widening
    static constexpr bool kDot2 = true;
    typedef _Float16 pair_t __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ float dot2(pair_t a, pair_t b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
    static __device__ __forceinline__ float to_acc(_Float16 v) { return (float)v; }
    static __device__ __forceinline__ _Float16 from_acc(float v) { return (_Float16)v; }
};
template <> struct Traits<__bf16> {
    using acc = float;
    static constexpr bool kDot2 = true;  // v_dot2c_f32_bf16
    typedef __bf16 pair_t __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ float dot2(pair_t a, pair_t b, float c)
    {
        return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false);
    }
    static __device__ __forceinline__ float to_acc(__bf16 v) { return (float)v; }
    static __device__ __forceinline__ __bf16 from_acc(float v) { return (__bf16)v; }  // v_cvt_pk_bf16_f32, RNE
};

// a*b + c in the accumulate type (one v_fma_f32 / v_fma_f64, never a silent promotion to double)
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
// low 32 bits of the product of two values < 2^24 (v_mul_u32_u24)
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
// the same for non-negative ints below 2^24 (indices: the host checks Q, I < 2^24): v_mul_u32_u24 runs at full rate,
// v_mul_lo_u32 at a quarter
__device__ __forceinline__ int imul24(int a, int b) { return (int)__umul24((uint32_t)a, (uint32_t)b); }
__device__ __forceinline__ float floor_t(float a) { return __builtin_floorf(a); }
__device__ __forceinline__ double floor_t(double a) { return __builtin_floor(a); }
__device__ __forceinline__ float fmin_t(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double fmin_t(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float fmax_t(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double fmax_t(double a, double b) { return __builtin_fmax(a, b); }

// acc[i] += w[0] v0[i] + w[1] v1[i] + w[2] v2[i] + w[3] v3[i], four chained FMAs per channel.  fp32: two channels per
// instruction (v_pk_fma_f32 with the weight broadcast by op_sel) — every vector instruction of a wave64 holds its SIMD
// for four cycles on gfx950, packed or not (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1 quad-cycle, round 5), so the pair
// halves the blend's share of the forward's issue time.  Written as 2-vectors by hand: the SLP vectoriser finds the
// same pairs but also builds others out of register shuffles (the fp32 translation units run without it).
typedef float f32x2 __attribute__((ext_vector_type(2)));
// kPacked: only where the rows ARE fp32 — rows widened from 16 bits take the scalar chain (fp16: v_fma_mix_f32 converts
// inside the FMA, and the pairs would cost a conversion each: c5's forward 2.99 -> 3.77 ms when they were packed too)
template <int VEC, bool kPacked>
__device__ __forceinline__ void blend4(float (&acc)[VEC], const float (&w)[4], const float (&v0)[VEC], const float (&v1)[VEC],
                                       const float (&v2)[VEC], const float (&v3)[VEC])
{
    if constexpr (kPacked && (VEC % 2) == 0) {
#pragma unroll
        for (int i = 0; i < VEC; i += 2) {
            f32x2 a = {acc[i], acc[i + 1]};
            a = __builtin_elementwise_fma(f32x2{w[0], w[0]}, f32x2{v0[i], v0[i + 1]}, a);
            a = __builtin_elementwise_fma(f32x2{w[1], w[1]}, f32x2{v1[i], v1[i + 1]}, a);
            a = __builtin_elementwise_fma(f32x2{w[2], w[2]}, f32x2{v2[i], v2[i + 1]}, a);
            a = __builtin_elementwise_fma(f32x2{w[3], w[3]}, f32x2{v3[i], v3[i + 1]}, a);
            acc[i] = a.x;
            acc[i + 1] = a.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            acc[i] = fma_t(w[0], v0[i], acc[i]);
            acc[i] = fma_t(w[1], v1[i], acc[i]);
            acc[i] = fma_t(w[2], v2[i], acc[i]);
            acc[i] = fma_t(w[3], v3[i], acc[i]);
        }
    }
}
template <int VEC, bool kPacked>
__device__ __forceinline__ void blend4(double (&acc)[VEC], const double (&w)[4], const double (&v0)[VEC], const double (&v1)[VEC],
                                       const double (&v2)[VEC], const double (&v3)[VEC])
{
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        acc[i] = fma_t(w[0], v0[i], acc[i]);
        acc[i] = fma_t(w[1], v1[i], acc[i]);
        acc[i] = fma_t(w[2], v2[i], acc[i]);
        acc[i] = fma_t(w[3], v3[i], acc[i]);
    }
}

// N elements of T with the natural alignment of the whole pack (so one load/store instruction).
template <typename T, int N> struct alignas(sizeof(T) * N) Pack {
    T v[N];
};

// ------------------------------------------------------------------------------------------
// buffer resource + range-checked raw loads.  A masked bilinear corner carries kMaskedOffset,
// which is out of range for every descriptor we build, so the hardware returns zeros for it:
// this is the reference's tl.where(mask, img, 0) (kernels.py:220-231) at no instruction cost.
// ------------------------------------------------------------------------------------------
using rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, uint32_t bytes)
{
    // base/bytes must be wave-uniform (T20): callers derive them from kernargs and blockIdx only.
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), /*stride*/ 0, (int)bytes, 0x00020000);
}

template <int BYTES> struct RawLoad;
template <> struct RawLoad<16> {
    using type = __attribute__((ext_vector_type(4))) uint32_t;
    static __device__ __forceinline__ type load(rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
};
template <> struct RawLoad<8> {
    using type = __attribute__((ext_vector_type(2))) uint32_t;
    static __device__ __forceinline__ type load(rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); }
};
template <> struct RawLoad<4> {
    using type = uint32_t;
    static __device__ __forceinline__ type load(rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0); }
};
template <> struct RawLoad<2> {
    using type = uint16_t;
    static __device__ __forceinline__ type load(rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0); }
};

// Streaming (nontemporal) store of a small POD: for outputs that are written once and not read again by the
// kernels of this call, so they do not displace the gathered rows in L2.
template <int BYTES> struct StreamBits {
    using type = __attribute__((ext_vector_type(BYTES / 4))) uint32_t;
};
template <> struct StreamBits<4> {
    using type = uint32_t;
};
template <> struct StreamBits<2> {
    using type = uint16_t;
};
template <typename V> __device__ __forceinline__ void store_stream(void *dst, const V &v)
{
    if constexpr (sizeof(V) <= 16) {
        using B = typename StreamBits<sizeof(V)>::type;
        __builtin_nontemporal_store(__builtin_bit_cast(B, v), static_cast<B *>(dst));
    } else {
        struct Halves {
            unsigned char a[sizeof(V) / 2], b[sizeof(V) / 2];
        };
        using B = typename StreamBits<sizeof(V) / 2>::type;
        struct Two {
            B lo, hi;
        };
        const Two t = __builtin_bit_cast(Two, v);
        __builtin_nontemporal_store(t.lo, static_cast<B *>(dst));
        __builtin_nontemporal_store(t.hi, static_cast<B *>(dst) + 1);
    }
}

// Load VEC consecutive elements of T at byte offset `off` and widen them to the accumulate type.
template <typename T, int VEC>
__device__ __forceinline__ void load_row(rsrc_t r, uint32_t off, typename Traits<T>::acc (&dst)[VEC])
{
    using RL = RawLoad<sizeof(T) * VEC>;
    typename RL::type raw = RL::load(r, off);
    Pack<T, VEC> p = __builtin_bit_cast(Pack<T, VEC>, raw);
#pragma unroll
    for (int i = 0; i < VEC; ++i) dst[i] = Traits<T>::to_acc(p.v[i]);
}

// ------------------------------------------------------------------------------------------
// level table in LDS: (h, w) from the device-resident shapes tensor, start = exclusive cumsum
// of h*w (reference: load_shapes_and_level_offsets, kernels.py:44-64).  No host sync.
// ------------------------------------------------------------------------------------------
struct LevelTab {
    int h[kMaxLevels];
    int w[kMaxLevels];
    int start[kMaxLevels];
    int cstart[kMaxLevels];  // exclusive cumsum of (h+1)*(w+1): the level's first bilinear *cell* (backward)
};

__device__ __forceinline__ void load_level_table(LevelTab *tab, const int64_t *shapes, int L)
{
    // ONE round trip to memory: lane l of the first wave fetches level l's (h, w) — vector loads, all in flight
    // together — and the exclusive sums come from a shuffle scan over those lanes.  (Until round 5 thread t looped
    // over the levels below it: the compiler made that a waterfall of scalar loads, one dependent miss per level and
    // a vector load of the thread's own level behind them — on a cold cache four round trips in front of the first
    // sample, which is most of a decoder-sized launch: HISTORY.md 9, small-Q forward.)
    const int t = threadIdx.x;
    if (t < kWave) {
        int lh = 0, lw = 0;
        if (t < L) {
            lh = (int)shapes[2 * t];
            lw = (int)shapes[2 * t + 1];
        }
        const int n = lh * lw, c = (lh + 1) * (lw + 1);
        int sn = n, sc = c;
        if (L <= 16) {  // inclusive scan inside a DPP row: row_shr fills with zeros, no LDS crossbar trip
            sn += __builtin_amdgcn_update_dpp(0, sn, 0x111, 0xF, 0xF, true);  // row_shr:1
            sc += __builtin_amdgcn_update_dpp(0, sc, 0x111, 0xF, 0xF, true);
            if (L > 2) {
                sn += __builtin_amdgcn_update_dpp(0, sn, 0x112, 0xF, 0xF, true);  // row_shr:2
                sc += __builtin_amdgcn_update_dpp(0, sc, 0x112, 0xF, 0xF, true);
            }
            if (L > 4) {
                sn += __builtin_amdgcn_update_dpp(0, sn, 0x114, 0xF, 0xF, true);  // row_shr:4
                sc += __builtin_amdgcn_update_dpp(0, sc, 0x114, 0xF, 0xF, true);
            }
            if (L > 8) {
                sn += __builtin_amdgcn_update_dpp(0, sn, 0x118, 0xF, 0xF, true);  // row_shr:8
                sc += __builtin_amdgcn_update_dpp(0, sc, 0x118, 0xF, 0xF, true);
            }
        } else {
            for (int d = 1; d < L; d <<= 1) {  // inclusive scan over the lanes 0 .. L-1 (L <= kMaxLevels <= 64)
                const int a = __shfl_up(sn, d, kWave), b = __shfl_up(sc, d, kWave);
                if (t >= d) {
                    sn += a;
                    sc += b;
                }
            }
        }
        if (t < L) {
            tab->h[t] = lh;
            tab->w[t] = lw;
            tab->start[t] = sn - n;
            tab->cstart[t] = sc - c;
        }
    }
}

// n / d for 0 <= n < 2^22, using a precomputed float reciprocal (exact after the fix-up steps).
__device__ __forceinline__ int div_small(int n, int d, float inv_d)
{
    // (24-bit multiplies: v_mul_lo_u32 runs at a quarter of the rate, and these sit in every sample's phase 1)
    int q = (int)((float)n * inv_d);
    if (imul24(q, d) > n) --q;
    if (imul24(q + 1, d) <= n) ++q;
    return q;
}

// ------------------------------------------------------------------------------------------
// one sample's bilinear taps (reference: sample_bilinear, kernels.py:120-252; semantics §9.1
// of SURVEY.md).  Offsets are byte offsets of the four corner rows relative to the (b, h) plane
// base; a corner that "zeros" padding masks out gets kMaskedOffset.
// ------------------------------------------------------------------------------------------
template <typename A> struct Taps {
    uint32_t off[4];  // 00 (y0,x0), 01 (y0,x1), 10 (y1,x0), 11 (y1,x1)
    A dx, dy;
    bool gx_on, gy_on;  // location gradient alive (grid_sample border clipping zeroes it)
};

// a * b + c for a, b < 2^24 in one full-rate instruction (the compiler turns `c + mul24(a, b)` into the quarter-rate
// v_mad_u64_u32)
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// base / masked: added to every corner's offset / what a corner masked by "zeros" padding gets instead (defaults: the
// plane itself; the LDSL kernels pass the LDS position of their copy and of its row of zeros)
template <typename A>
__device__ __forceinline__ void make_taps(A x, A y, int h, int w, int start, bool zeros, bool align,
                                          uint32_t row_bytes, Taps<A> &t, uint32_t base = 0, uint32_t masked = kMaskedOffset)
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
    const A x0 = floor_t(px), y0 = floor_t(py);
    const A x1 = x0 + (A)1, y1 = y0 + (A)1;
    // clamp in floating point before the int conversion: far-OOB / NaN coordinates cannot overflow
    const A xm = W - (A)1, ym = Hh - (A)1;
    const int x0c = (int)fmin_t(fmax_t(x0, (A)0), xm);
    const int x1c = (int)fmin_t(fmax_t(x1, (A)0), xm);
    const int y0c = (int)fmin_t(fmax_t(y0, (A)0), ym);
    const int y1c = (int)fmin_t(fmax_t(y1, (A)0), ym);
    // 24-bit multiplies are full rate (v_mul_lo_u32 is quarter rate); pixel indices and row sizes are < 2^24 (host check)
    const uint32_t r0 = mad24((uint32_t)y0c, (uint32_t)w, (uint32_t)start), r1 = mad24((uint32_t)y1c, (uint32_t)w, (uint32_t)start);
    t.off[0] = mad24(r0 + (uint32_t)x0c, row_bytes, base);
    t.off[1] = mad24(r0 + (uint32_t)x1c, row_bytes, base);
    t.off[2] = mad24(r1 + (uint32_t)x0c, row_bytes, base);
    t.off[3] = mad24(r1 + (uint32_t)x1c, row_bytes, base);
    if (zeros) {
        const bool mx0 = (x0 >= (A)0) && (x0 <= xm), mx1 = (x1 >= (A)0) && (x1 <= xm);
        const bool my0 = (y0 >= (A)0) && (y0 <= ym), my1 = (y1 >= (A)0) && (y1 <= ym);
        if (!(my0 && mx0)) t.off[0] = masked;
        if (!(my0 && mx1)) t.off[1] = masked;
        if (!(my1 && mx0)) t.off[2] = masked;
        if (!(my1 && mx1)) t.off[3] = masked;
        t.gx_on = true;
        t.gy_on = true;
    } else {
        t.gx_on = (px > (A)0) && (px < xm);
        t.gy_on = (py > (A)0) && (py < ym);
    }
    t.dx = px - x0;
    t.dy = py - y0;
}

// ------------------------------------------------------------------------------------------
// sum over the G lanes of a unit (G a power of two, groups are G-aligned inside the wave).
// Stages 1,2 are quad permutes, 4 is row_half_mirror, 8 is row_mirror (valid because after
// the lower stages every lane of an aligned sub-group already holds that sub-group's sum);
// 16 and 32 go through ds_bpermute.  All lanes of the group end with the same bits.
// ------------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int G> __device__ __forceinline__ float group_sum(float v)
{
    if constexpr (G >= 2) v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
    if constexpr (G >= 4) v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v += dpp_f32<0x141>(v);  // row_half_mirror
    if constexpr (G >= 16) v += dpp_f32<0x140>(v); // row_mirror
    if constexpr (G >= 32) v += __shfl_xor(v, 16, kWave);
    if constexpr (G >= 64) v += __shfl_xor(v, 32, kWave);
    return v;
}

// ------------------------------------------------------------------------------------------
// Reduce-SCATTER over the lanes of a unit (G = 4 or 8): every lane brings K partial sums for each of G items
// (e[item][k]); afterwards lane j of the unit holds the K COMPLETE sums of item j.  A butterfly whose every step
// halves the items a lane keeps: G = 8 costs 16 + 8 + 4 exchanged adds per k-quadruple instead of the 8 * 3 * K / 4
// of all-reducing item by item, and the work behind the sums is then done once per item, on its owner lane, instead
// of G times.  quarter_step / pair_step are the two quad-internal steps; the 8-lane step (row_half_mirror: lane j
// meets lane 7 - j, lanes 0..3 keep items 0..3, lanes 4..7 items 4..7) is split so that the kernel can run it on one
// half of the items as soon as that half's partial sums exist (half_step).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float half_step(float v) { return v + dpp_f32<0x141>(v); }  // row_half_mirror partner
// ... of a value that arrives as two partial sums: x = lo + partner's hi, then x + partner's x = lo + hi + lo' + hi' — two
// DPP adds, the same count as adding lo + hi first
__device__ __forceinline__ float half_step2(float lo, float hi)
{
    const float x = lo + dpp_f32<0x141>(hi);
    return x + dpp_f32<0x141>(x);
}
// ... written only on lanes 4..7 of every 8 (DPP bank mask 0xA: the banks are groups of four lanes), the others keep
// `e`: the second half-batch's step lands in the registers of the first without a select per value
// (Inline assembly, because no builtin writes a DPP result under a bank mask into a register that keeps its other lanes.
// The hazard pass does not look into it: a DPP operand written by the previous vector instruction needs two wait states
// — `s_nop 1` in front; and `e` may be read through DPP right behind the LAST of a run of these — kLast adds the wait
// states there.  volatile keeps the run in program order, so "last" means what it says.)
template <bool kLast> __device__ __forceinline__ void half_step2_upper(float &e, float lo, float hi)
{
    const float x = lo + dpp_f32<0x141>(hi);
    if constexpr (kLast)
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n\ts_nop 1" : "+v"(e) : "v"(x));
    else
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa" : "+v"(e) : "v"(x));
}
template <int K> __device__ __forceinline__ void quad_steps(const float (&e)[4][K], int j, float (&out)[K])
{
    const bool hi2 = (j & 2) != 0, hi1 = (j & 1) != 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        // lanes {0,1} of a quad keep items 0, 1; lanes {2,3} items 2, 3 (partner: lane ^ 2)
        const float a0 = e[0][k] + dpp_f32<0x4E>(e[0][k]), a1 = e[1][k] + dpp_f32<0x4E>(e[1][k]);
        const float b0 = e[2][k] + dpp_f32<0x4E>(e[2][k]), b1 = e[3][k] + dpp_f32<0x4E>(e[3][k]);
        const float f0 = hi2 ? b0 : a0, f1 = hi2 ? b1 : a1;
        // even lanes keep the first of their two items, odd lanes the second (partner: lane ^ 1)
        const float c0 = f0 + dpp_f32<0xB1>(f0), c1 = f1 + dpp_f32<0xB1>(f1);
        out[k] = hi1 ? c1 : c0;
    }
}

// max over the G lanes of a unit (same lane choreography as group_sum)
template <int G> __device__ __forceinline__ float group_max(float v)
{
    if constexpr (G >= 2) v = fmax_t(v, dpp_f32<0xB1>(v));
    if constexpr (G >= 4) v = fmax_t(v, dpp_f32<0x4E>(v));
    if constexpr (G >= 8) v = fmax_t(v, dpp_f32<0x141>(v));
    if constexpr (G >= 16) v = fmax_t(v, dpp_f32<0x140>(v));
    if constexpr (G >= 32) v = fmax_t(v, __shfl_xor(v, 16, kWave));
    if constexpr (G >= 64) v = fmax_t(v, __shfl_xor(v, 32, kWave));
    return v;
}
template <int G> __device__ __forceinline__ double group_max(double v)
{
#pragma unroll
    for (int m = 1; m < G; m <<= 1) v = fmax_t(v, __shfl_xor(v, m, kWave));
    return v;
}
__device__ __forceinline__ float exp_t(float a) { return ::expf(a); }
__device__ __forceinline__ double exp_t(double a) { return ::exp(a); }

template <int G> __device__ __forceinline__ double group_sum(double v)
{
#pragma unroll
    for (int m = 1; m < G; m <<= 1) v += __shfl_xor(v, m, kWave);
    return v;
}

// n / d for any 32-bit n with a host-made (magic, shift) pair (round-up method, branch-free core).
struct FastDiv {
    uint32_t magic, shift;  // shift == 0: d is 1
};
__device__ __forceinline__ uint32_t fast_div(uint32_t n, FastDiv fd)
{
    if (fd.shift == 0) return n;
    const uint32_t t = __umulhi(n, fd.magic);
    return (((n - t) >> 1) + t) >> (fd.shift - 1);
}

// blockIdx -> (pair = b*H + h, slot).  With xcd_map, workgroups whose linear ids are congruent mod 8
// (they share an XCD and therefore an L2 under round-robin dispatch) work on the same (b, h)
// planes, and each XCD walks its planes one after another, so a plane is pulled into exactly
// one L2.  Pure speed: any placement gives the same results.  Returns false for padding blocks.
// grid3d: the launch used dim3(8, slots, ceil(pairs/8)) (xcd_map) or dim3(slots, pairs), whose
// linear dispatch order equals the 1-D formula below, so no integer division is needed.
__device__ __forceinline__ bool decode_block(int grid3d, int npairs, int slots, int xcd_map, int &pair, int &slot)
{
    if (grid3d) {
        if (xcd_map) {
            // (xcd_map 2: the planes of one XCD rotate through the heads instead of all having the same one; by 3 per batch
            //  element, so that with H = 8 neighbouring heads — which can be slow together, c3's 6 and 7 — do not meet again)
            pair = (int)(blockIdx.z * 8 + (xcd_map == 2 ? ((blockIdx.x + 3 * blockIdx.z) & 7) : blockIdx.x));
            slot = (int)blockIdx.y;
        } else {
            pair = (int)blockIdx.y;
            slot = (int)blockIdx.x;
        }
    } else {
        const int bid = (int)blockIdx.x;
        if (xcd_map) {
            const int x = bid & 7, t = bid >> 3;
            slot = t % slots;
            pair = (t / slots) * 8 + x;
        } else {
            pair = bid / slots;
            slot = bid - pair * slots;
        }
    }
    return pair < npairs;
}

}  // namespace msda
