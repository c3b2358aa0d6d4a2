// Shared device code of the bf16 GEMM translation units (gemm.hip: the product kernels and the dispatch; gemm_experiments.hip: the
// measured-and-rejected kernel variants, kept selectable for the kernels-agree test and A/B timing): operand / epilogue types,
// activations, the XCD-aware tile order, the LDS-staged epilogue with its per-family row loops, ring constants and counted waits.
#pragma once
#include <type_traits>
#include "common.h"


typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define BK 64

struct GemmEpi {
    const bf16_t* bias;
    const float* row_gate;
    const void* residual;
    int residual_dtype;
    int64_t ld_res;
    int act;
    int swiglu;
    int use_scale;
    float scale;
    int out_dtype;
    const float* a_scale;       // fp8 GEMM only: per-row scale of the quantised activations ...
    const float* w_scale;       // ... and per-output-channel scale of the quantised weights (NULL for bf16 operands)
};

// Activations evaluated on bf16-rounded inputs and rounded to bf16 again by the caller, so ~1e-6 relative
// accuracy is ample; the libm erff/tanhf/expf bodies are 3-5x more VALU work (the GELU epilogue of the ViT fc1
// GEMM measured 2x the MFMA time with erff).
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) for a bf16-valued x, to be rounded to bf16 by the caller (round 3; before: Abramowitz-Stegun
// 7.1.26 everywhere, one v_rcp + one v_exp per value, 99.74 % of all finite bf16 inputs rounding as the exact function does):
//   * |z| <= 2.2 (z = x / sqrt 2, |x| <= 3.11) and z > 2.2: erf(z) = z P(z^2), P of degree 8, z clamped to [-2.2, 2.2] - eleven
//     multiply-adds, no transcendental instruction; for z > 2.2 the clamp leaves 0.5 x (1 + erf(2.2)) = 0.99907 x, which rounds to
//     bf16 as x Phi(x) does.  Over ALL bf16 inputs x >= -3.11: one input (-2.6875) comes out one ulp off, every other identical.
//   * z < -2.2, where 1 + erf cancels: 0.5 x erfcx(|z|) exp(-z^2), erfcx by a degree-6 polynomial in |z| - 2.2 (clamped at
//     |z| = 4.6: beyond, |GELU| < 2.2e-10): every bf16 input down to x = -6.5 identical to the exact function.
// Same operations in the scalar and the packed form (explicit fma): bit-identical to each other.
#define GELU_ZM 2.2f
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = x * 0.70710678118654752440f;
    const float zc = __builtin_amdgcn_fmed3f(z, -GELU_ZM, GELU_ZM);
    const float u = zc * zc;
    float p = 2.258253176e-07f;
    p = __builtin_fmaf(p, u, -6.468885658e-06f);
    p = __builtin_fmaf(p, u, 8.784283273e-05f);
    p = __builtin_fmaf(p, u, -7.748150965e-04f);
    p = __builtin_fmaf(p, u, 5.104614887e-03f);
    p = __builtin_fmaf(p, u, -2.676485851e-02f);
    p = __builtin_fmaf(p, u, 1.127952486e-01f);
    p = __builtin_fmaf(p, u, -3.761198521e-01f);
    p = __builtin_fmaf(p, u, 1.128379107e+00f);
    const float h = 0.5f * x;
    float y = h * (1.0f + zc * p);
    if (z < -GELU_ZM) {
        const float az = -z;
        const float s = __builtin_amdgcn_fmed3f(az - GELU_ZM, 0.0f, 2.4f);
        float q = 4.732637171e-05f;                           // erfcx(2.2 + s), degree 6 (relative-error fit on [0, 2.4])
        q = __builtin_fmaf(q, s, -5.468023592e-04f);
        q = __builtin_fmaf(q, s, 2.993027214e-03f);
        q = __builtin_fmaf(q, s, -1.105889119e-02f);
        q = __builtin_fmaf(q, s, 3.343209252e-02f);
        q = __builtin_fmaf(q, s, -9.172994643e-02f);
        q = __builtin_fmaf(q, s, 2.355919182e-01f);
        y = h * (q * __expf(-(az * az)));
    }
    return y;
}
// The same GELU on two values at once: identical operations per element, written on 2-vectors so they issue as v_pk_mul_f32 /
// v_pk_fma_f32 / v_pk_add_f32 - the epilogue of the fc1 GEMM carries 128 of these per lane and tile.  The tail branch is taken
// by the whole wave when any lane needs it.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 x) {
    const f32x2 z = x * 0.70710678118654752440f;
    const f32x2 zc = f32x2{__builtin_amdgcn_fmed3f(z[0], -GELU_ZM, GELU_ZM), __builtin_amdgcn_fmed3f(z[1], -GELU_ZM, GELU_ZM)};
    const f32x2 u = zc * zc;
    f32x2 p = f32x2{2.258253176e-07f, 2.258253176e-07f};
    p = __builtin_elementwise_fma(p, u, f32x2{-6.468885658e-06f, -6.468885658e-06f});
    p = __builtin_elementwise_fma(p, u, f32x2{8.784283273e-05f, 8.784283273e-05f});
    p = __builtin_elementwise_fma(p, u, f32x2{-7.748150965e-04f, -7.748150965e-04f});
    p = __builtin_elementwise_fma(p, u, f32x2{5.104614887e-03f, 5.104614887e-03f});
    p = __builtin_elementwise_fma(p, u, f32x2{-2.676485851e-02f, -2.676485851e-02f});
    p = __builtin_elementwise_fma(p, u, f32x2{1.127952486e-01f, 1.127952486e-01f});
    p = __builtin_elementwise_fma(p, u, f32x2{-3.761198521e-01f, -3.761198521e-01f});
    p = __builtin_elementwise_fma(p, u, f32x2{1.128379107e+00f, 1.128379107e+00f});
    const f32x2 h = 0.5f * x;
    f32x2 y = h * (1.0f + zc * p);
    if (__any((z[0] < -GELU_ZM) | (z[1] < -GELU_ZM))) {
        const f32x2 az = -z;
        const f32x2 sm = az - GELU_ZM;
        const f32x2 s = f32x2{__builtin_amdgcn_fmed3f(sm[0], 0.0f, 2.4f), __builtin_amdgcn_fmed3f(sm[1], 0.0f, 2.4f)};
        f32x2 q = f32x2{4.732637171e-05f, 4.732637171e-05f};
        q = __builtin_elementwise_fma(q, s, f32x2{-5.468023592e-04f, -5.468023592e-04f});
        q = __builtin_elementwise_fma(q, s, f32x2{2.993027214e-03f, 2.993027214e-03f});
        q = __builtin_elementwise_fma(q, s, f32x2{-1.105889119e-02f, -1.105889119e-02f});
        q = __builtin_elementwise_fma(q, s, f32x2{3.343209252e-02f, 3.343209252e-02f});
        q = __builtin_elementwise_fma(q, s, f32x2{-9.172994643e-02f, -9.172994643e-02f});
        q = __builtin_elementwise_fma(q, s, f32x2{2.355919182e-01f, 2.355919182e-01f});
        const f32x2 a2 = az * az;
        const f32x2 t = h * (q * f32x2{__expf(-a2[0]), __expf(-a2[1])});
        y = f32x2{z[0] < -GELU_ZM ? t[0] : y[0], z[1] < -GELU_ZM ? t[1] : y[1]};
    }
    return y;
}
__device__ __forceinline__ float act_apply(float y, int act) {
    switch (act) {
        case 1: return gelu_erf_fast(y);
        case 2: { const float u = 0.7978845608028654f * (y + 0.044715f * y * y * y);
                  const float th = 1.0f - 2.0f * fast_rcp(1.0f + __expf(2.0f * u));      // tanh(u)
                  return 0.5f * y * (1.0f + th); }
        case 3: return fmaxf(y, 0.0f);
        default: return y;
    }
}
__device__ __forceinline__ float silu_fast(float v) { return v * fast_rcp(1.0f + __expf(-v)); }

__device__ __forceinline__ int lds_off(int row, int chunk) {       // bytes within a [rows][64] bf16 tile
    return row * 128 + ((chunk ^ (row & 7)) << 4);
}

// XCD-aware tile id: blocks b and b+8 share an XCD; give each XCD a contiguous run of tiles, and order tiles
// in groups of GROUP tile-rows (M fastest inside a group) so a run is a compact 2-D patch of the tile grid.
__device__ __forceinline__ void tile_coords(int bid, int tiles_m, int tiles_n, int& tm, int& tn, int GROUP = 8) {
    const int nwg = tiles_m * tiles_n;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per_group = GROUP * tiles_n;
    const int first_m = (bid / per_group) * GROUP;
    const int gsize = min(tiles_m - first_m, GROUP);
    tm = first_m + (bid % per_group) % gsize;
    tn = (bid % per_group) / gsize;
}

// compile-time loop: accumulator arrays must only ever be indexed by constants (a runtime index puts them in scratch)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// Phase B, generic form: every epilogue option under run-time flags (used only for combinations without a specialisation).
template <int TM, int TN, int NWAVES, int EPL>            // EPL: output elements per lane (8 = bf16 out, 4 = fp32 out)
__device__ __noinline__ void epilogue_rows_generic(const GemmEpi& ep, void* __restrict__ C, int64_t ldc, int M, int N, int m0, int n0,
                                           int wave, int lane, const char* smem) {
    constexpr int YS = TN * 2 + 16;
    constexpr bool F32 = (EPL == 4);
    const bool sw = ep.swiglu != 0;
    const int tcols = sw ? TN / 2 : TN;                   // output columns this tile produces
    const int on = sw ? (N >> 1) : N;
    const int oc0 = sw ? (n0 >> 1) : n0;
    const int lpr = tcols / EPL;                          // lanes per row
    const int rpi = 64 / lpr;                             // rows per wave-instruction
    const int lr = lane / lpr, lcol = (lane % lpr) * EPL;
    for (int rb = wave * rpi; rb < TM; rb += NWAVES * rpi) {
        const int row = rb + lr;
        const int m = m0 + row;
        const int c = oc0 + lcol;
        if (m >= M || c >= on) continue;
        const char* yrow = smem + row * YS;
        float y[EPL];
        if (!sw) {
            if (F32) {
                const uint2 v = *reinterpret_cast<const uint2*>(yrow + lcol * 2);
                y[0] = __uint_as_float(v.x << 16); y[1] = __uint_as_float(v.x & 0xffff0000u);
                y[2] = __uint_as_float(v.y << 16); y[3] = __uint_as_float(v.y & 0xffff0000u);
            } else {
                const u32x4 v = *reinterpret_cast<const u32x4*>(yrow + lcol * 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) { y[2 * e] = __uint_as_float(v[e] << 16); y[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); }
            }
            if (ep.act) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) y[e] = rbf(act_apply(y[e], ep.act));
            }
        } else {
            // output col lcol..lcol+EPL-1 lives in packed cols (lcol/16)*32 + lcol%16 (gate) and +16 (up)
            const int pc = (lcol >> 4) * 32 + (lcol & 15);
            float gv[EPL], uv[EPL];
            if (F32) {
                const uint2 g2 = *reinterpret_cast<const uint2*>(yrow + pc * 2);
                const uint2 u2 = *reinterpret_cast<const uint2*>(yrow + (pc + 16) * 2);
                gv[0] = __uint_as_float(g2.x << 16); gv[1] = __uint_as_float(g2.x & 0xffff0000u);
                gv[2] = __uint_as_float(g2.y << 16); gv[3] = __uint_as_float(g2.y & 0xffff0000u);
                uv[0] = __uint_as_float(u2.x << 16); uv[1] = __uint_as_float(u2.x & 0xffff0000u);
                uv[2] = __uint_as_float(u2.y << 16); uv[3] = __uint_as_float(u2.y & 0xffff0000u);
            } else {
                const u32x4 g4 = *reinterpret_cast<const u32x4*>(yrow + pc * 2);
                const u32x4 u4 = *reinterpret_cast<const u32x4*>(yrow + (pc + 16) * 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gv[2 * e] = __uint_as_float(g4[e] << 16); gv[2 * e + 1] = __uint_as_float(g4[e] & 0xffff0000u);
                    uv[2 * e] = __uint_as_float(u4[e] << 16); uv[2 * e + 1] = __uint_as_float(u4[e] & 0xffff0000u);
                }
            }
#pragma unroll
            for (int e = 0; e < EPL; ++e) y[e] = rbf(rbf(silu_fast(gv[e])) * uv[e]);
        }
        if (ep.row_gate && ep.row_gate[m] == 0.0f) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) y[e] = 0.0f;
        }
        if (ep.use_scale) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) y[e] = rbf(ep.scale * y[e]);
        }
        const int nv = min(EPL, on - c);
        if (ep.residual) {
            if (ep.residual_dtype == LICV_F32) {
                const float* rp = reinterpret_cast<const float*>(ep.residual) + (int64_t)m * ep.ld_res + c;
                if (nv == EPL) {
#pragma unroll
                    for (int q = 0; q < EPL / 4; ++q) {
                        const floatx4 rv = *reinterpret_cast<const floatx4*>(rp + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[4 * q + e] = rv[e] + y[4 * q + e];
                    }
                } else for (int e = 0; e < nv; ++e) y[e] = rp[e] + y[e];
            } else {
                const bf16_t* rp = reinterpret_cast<const bf16_t*>(ep.residual) + (int64_t)m * ep.ld_res + c;
                if (nv == EPL && !F32) {
                    const u32x4 rv = *reinterpret_cast<const u32x4*>(rp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        y[2 * e] = rbf(__uint_as_float(rv[e] << 16) + y[2 * e]);
                        y[2 * e + 1] = rbf(__uint_as_float(rv[e] & 0xffff0000u) + y[2 * e + 1]);
                    }
                } else for (int e = 0; e < nv; ++e) y[e] = rbf(bf2f(rp[e]) + y[e]);
            }
        }
        if (F32) {
            float* cp = reinterpret_cast<float*>(C) + (int64_t)m * ldc + c;
            if (nv == 4) *reinterpret_cast<floatx4*>(cp) = floatx4{y[0], y[1], y[2], y[3]};
            else for (int e = 0; e < nv; ++e) cp[e] = y[e];
        } else {
            bf16_t* cp = reinterpret_cast<bf16_t*>(C) + (int64_t)m * ldc + c;
            if (nv == 8) {
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (uint32_t)f2bf(y[2 * e]) | ((uint32_t)f2bf(y[2 * e + 1]) << 16);
                *reinterpret_cast<u32x4*>(cp) = o;
            } else for (int e = 0; e < nv; ++e) cp[e] = f2bf(y[e]);
        }
    }
}


// Phase B of the staged epilogue (see below): compact run-time loops over the rows of the LDS image, one specialisation
// per epilogue family.  Measured on the all-flags-at-run-time form: 600 basic blocks / 22 KiB of code, 7.6 us per
// 256 x 256 tile just to copy a finished bf16 image out (13 us with a residual, 19 us with GELU) on an otherwise idle
// chip — per-element branches on ep.act / ep.swiglu / residual dtype the compiler cannot hoist out of a noinline
// body.  Each family below is branch-free inside its row loop.
struct RowMap {                 // lane -> (row group, first output column) for EPL consecutive output columns per lane
    int lr, lcol, rstep, c, nv;
};
template <int TN, int NWAVES, int EPL>
__device__ __forceinline__ RowMap row_map(int tcols, int oc0, int on, int wave, int lane, int& rb_first) {
    const int lpr = tcols / EPL, rpi = 64 / lpr;
    RowMap r;
    r.lr = lane / lpr; r.lcol = (lane % lpr) * EPL; r.rstep = NWAVES * rpi; r.c = oc0 + r.lcol; r.nv = min(EPL, on - r.c);
    rb_first = wave * rpi;
    return r;
}
__device__ __forceinline__ void unpack8(const u32x4& v, float (&y)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { y[2 * e] = __uint_as_float(v[e] << 16); y[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); }
}
__device__ __forceinline__ u32x4 pack8(const float (&y)[8]) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (uint32_t)f2bf(y[2 * e]) | ((uint32_t)f2bf(y[2 * e + 1]) << 16);
    return o;
}
__device__ __forceinline__ void store_bf16_row(bf16_t* cp, const float (&y)[8], int nv) {
    if (nv == 8) *reinterpret_cast<u32x4*>(cp) = pack8(y);
    else for (int e = 0; e < nv; ++e) cp[e] = f2bf(y[e]);
}

// no activation / residual / gate, bf16 out: the LDS image already holds the result -> 16-byte copies
template <int TM, int TN, int NWAVES>
__device__ __noinline__ void epilogue_rows_plain(void* __restrict__ C, int64_t ldc, int M, int N, int m0, int n0, int wave, int lane,
                                                 const char* smem) {
    constexpr int YS = TN * 2 + 16;
    int rb0;
    const RowMap r = row_map<TN, NWAVES, 8>(TN, n0, N, wave, lane, rb0);
    if (r.c >= N) return;
    for (int rb = rb0; rb < TM; rb += r.rstep) {
        const int row = rb + r.lr, m = m0 + row;
        if (m >= M) break;
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * YS + r.lcol * 2);
        bf16_t* cp = reinterpret_cast<bf16_t*>(C) + (int64_t)m * ldc + r.c;
        if (r.nv == 8) *reinterpret_cast<u32x4*>(cp) = v;
        else for (int e = 0; e < r.nv; ++e) cp[e] = (bf16_t)(e & 1 ? v[e >> 1] >> 16 : v[e >> 1] & 0xffffu);
    }
}

// activation only (ACT: 1 erf-GELU, 2 tanh-GELU, 3 ReLU), bf16 out
template <int TM, int TN, int NWAVES, int ACT>
__device__ __noinline__ void epilogue_rows_act(void* __restrict__ C, int64_t ldc, int M, int N, int m0, int n0, int wave, int lane,
                                               const char* smem) {
    constexpr int YS = TN * 2 + 16;
    int rb0;
    const RowMap r = row_map<TN, NWAVES, 8>(TN, n0, N, wave, lane, rb0);
    if (r.c >= N) return;
    for (int rb = rb0; rb < TM; rb += r.rstep) {
        const int row = rb + r.lr, m = m0 + row;
        if (m >= M) break;
        float y[8];
        unpack8(*reinterpret_cast<const u32x4*>(smem + row * YS + r.lcol * 2), y);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = act_apply(y[e], ACT);
        store_bf16_row(reinterpret_cast<bf16_t*>(C) + (int64_t)m * ldc + r.c, y, r.nv);
    }
}

// SwiGLU pairing of the interleaved gate|up image, bf16 out (N/2 output columns)
template <int TM, int TN, int NWAVES>
__device__ __noinline__ void epilogue_rows_swiglu(void* __restrict__ C, int64_t ldc, int M, int N, int m0, int n0, int wave, int lane,
                                                  const char* smem) {
    constexpr int YS = TN * 2 + 16;
    int rb0;
    const RowMap r = row_map<TN, NWAVES, 8>(TN / 2, n0 >> 1, N >> 1, wave, lane, rb0);
    if (r.c >= (N >> 1)) return;
    const int pc = (r.lcol >> 4) * 32 + (r.lcol & 15);    // output cols lcol.. live in packed cols pc.. (gate) and pc+16.. (up)
    for (int rb = rb0; rb < TM; rb += r.rstep) {
        const int row = rb + r.lr, m = m0 + row;
        if (m >= M) break;
        const char* yrow = smem + row * YS;
        float g[8], u[8], y[8];
        unpack8(*reinterpret_cast<const u32x4*>(yrow + pc * 2), g);
        unpack8(*reinterpret_cast<const u32x4*>(yrow + (pc + 16) * 2), u);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = rbf(silu_fast(g[e])) * u[e];
        store_bf16_row(reinterpret_cast<bf16_t*>(C) + (int64_t)m * ldc + r.c, y, r.nv);
    }
}

// residual add (same dtype in and out: bf16 stream or fp32 stream), optional row gate and tanh-gate scale.  The residual
// rows are fetched PF row-groups ahead: nothing else runs on the CU to hide their latency.
template <int TM, int TN, int NWAVES, bool F32>
__device__ __noinline__ void epilogue_rows_res(const GemmEpi& ep, void* __restrict__ C, int64_t ldc, int M, int N, int m0, int n0,
                                               int wave, int lane, const char* smem) {
    constexpr int YS = TN * 2 + 16;
    constexpr int EPL = F32 ? 4 : 8, ESZ = F32 ? 4 : 2, PF = F32 ? 8 : 4;
    int rb0;
    const RowMap r = row_map<TN, NWAVES, EPL>(TN, n0, N, wave, lane, rb0);
    if (r.c >= N) return;
    const bool full = r.nv == EPL;
    const char* res = reinterpret_cast<const char*>(ep.residual);
    const float* gate = ep.row_gate;
    const bool scaled = ep.use_scale != 0;
    const float scale = ep.scale;
    // rolling prefetch: PF row-groups of residual in flight; slot u is refilled for group g + PF right after group g
    // has consumed it
    auto fetch = [&](int rb) -> u32x4 {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        const int m = m0 + rb + r.lr;
        if (rb < TM && m < M) {
            const char* rp = res + ((int64_t)m * ep.ld_res + r.c) * ESZ;
            if (full) v = *reinterpret_cast<const u32x4*>(rp);
            else for (int e = 0; e < r.nv; ++e) {
                if (F32) v[e] = reinterpret_cast<const uint32_t*>(rp)[e];
                else v[e >> 1] |= (uint32_t)reinterpret_cast<const bf16_t*>(rp)[e] << ((e & 1) * 16);
            }
        }
        return v;
    };
    u32x4 rv[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) rv[u] = fetch(rb0 + u * r.rstep);
    for (; rb0 < TM; rb0 += PF * r.rstep) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int rb = rb0 + u * r.rstep;
            const int row = rb + r.lr, m = m0 + row;
            if (rb >= TM || m >= M) break;
            const u32x4 rcur = rv[u];
            rv[u] = fetch(rb + PF * r.rstep);
            float y[EPL];
            if (F32) {
                const uint2 v = *reinterpret_cast<const uint2*>(smem + row * YS + r.lcol * 2);
                y[0] = __uint_as_float(v.x << 16); y[1] = __uint_as_float(v.x & 0xffff0000u);
                y[2] = __uint_as_float(v.y << 16); y[3] = __uint_as_float(v.y & 0xffff0000u);
            } else {
                float t[8];
                unpack8(*reinterpret_cast<const u32x4*>(smem + row * YS + r.lcol * 2), t);
#pragma unroll
                for (int e = 0; e < EPL; ++e) y[e] = t[e];
            }
            if (gate && gate[m] == 0.0f) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) y[e] = 0.0f;
            }
            if (scaled) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) y[e] = rbf(scale * y[e]);
            }
            if (F32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = __uint_as_float(rcur[e]) + y[e];
                float* cp = reinterpret_cast<float*>(C) + (int64_t)m * ldc + r.c;
                if (full) *reinterpret_cast<floatx4*>(cp) = floatx4{y[0], y[1], y[2], y[3]};
                else for (int e = 0; e < r.nv; ++e) cp[e] = y[e];
            } else {
                float q[8];
                unpack8(rcur, q);
                float z[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = q[e] + y[e < EPL ? e : 0];
                store_bf16_row(reinterpret_cast<bf16_t*>(C) + (int64_t)m * ldc + r.c, z, r.nv);
            }
        }
    }
}

template <int TM, int TN, int NWAVES>
__device__ __forceinline__ void epilogue_rows(const GemmEpi& ep, void* __restrict__ C, int64_t ldc, int M, int N, int m0, int n0,
                                              int wave, int lane, const char* smem) {
    const bool bf_out = ep.out_dtype == LICV_BF16;
    const bool simple = !ep.residual && !ep.row_gate && !ep.use_scale && bf_out;
    if (simple && ep.swiglu)                 epilogue_rows_swiglu<TM, TN, NWAVES>(C, ldc, M, N, m0, n0, wave, lane, smem);
    else if (simple && ep.act == 0)          epilogue_rows_plain<TM, TN, NWAVES>(C, ldc, M, N, m0, n0, wave, lane, smem);
    else if (simple && ep.act == 1)          epilogue_rows_act<TM, TN, NWAVES, 1>(C, ldc, M, N, m0, n0, wave, lane, smem);
    else if (simple && ep.act == 2)          epilogue_rows_act<TM, TN, NWAVES, 2>(C, ldc, M, N, m0, n0, wave, lane, smem);
    else if (simple && ep.act == 3)          epilogue_rows_act<TM, TN, NWAVES, 3>(C, ldc, M, N, m0, n0, wave, lane, smem);
    else if (ep.residual && !ep.swiglu && !ep.act && ep.residual_dtype == LICV_BF16 && bf_out)
        epilogue_rows_res<TM, TN, NWAVES, false>(ep, C, ldc, M, N, m0, n0, wave, lane, smem);
    else if (ep.residual && !ep.swiglu && !ep.act && ep.residual_dtype == LICV_F32 && !bf_out)
        epilogue_rows_res<TM, TN, NWAVES, true>(ep, C, ldc, M, N, m0, n0, wave, lane, smem);
    else if (bf_out) epilogue_rows_generic<TM, TN, NWAVES, 8>(ep, C, ldc, M, N, m0, n0, wave, lane, smem);
    else             epilogue_rows_generic<TM, TN, NWAVES, 4>(ep, C, ldc, M, N, m0, n0, wave, lane, smem);
}

// The same dispatch with every family inlined at the call (the statement attribute overrides the callee's noinline): for a kernel whose
// register budget is half a SIMD's file (the tall kernel: eight waves, 128 accumulators) - as separate functions the families keep
// their own ~150 arch VGPRs, which are added to the kernel's accumulator file in its resource record.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wignored-attributes"
template <int TM, int TN, int NWAVES>
__device__ __forceinline__ void epilogue_rows_inlined(const GemmEpi& ep, void* __restrict__ C, int64_t ldc, int M, int N, int m0, int n0,
                                                      int wave, int lane, const char* smem) {
    const bool bf_out = ep.out_dtype == LICV_BF16;
    const bool simple = !ep.residual && !ep.row_gate && !ep.use_scale && bf_out;
    if (simple && ep.swiglu)                 { [[clang::always_inline]] epilogue_rows_swiglu<TM, TN, NWAVES>(C, ldc, M, N, m0, n0, wave, lane, smem); }
    else if (simple && ep.act == 0)          { [[clang::always_inline]] epilogue_rows_plain<TM, TN, NWAVES>(C, ldc, M, N, m0, n0, wave, lane, smem); }
    else if (simple && ep.act == 1)          { [[clang::always_inline]] epilogue_rows_act<TM, TN, NWAVES, 1>(C, ldc, M, N, m0, n0, wave, lane, smem); }
    else if (simple && ep.act == 2)          { [[clang::always_inline]] epilogue_rows_act<TM, TN, NWAVES, 2>(C, ldc, M, N, m0, n0, wave, lane, smem); }
    else if (simple && ep.act == 3)          { [[clang::always_inline]] epilogue_rows_act<TM, TN, NWAVES, 3>(C, ldc, M, N, m0, n0, wave, lane, smem); }
    else if (ep.residual && !ep.swiglu && !ep.act && ep.residual_dtype == LICV_BF16 && bf_out)
        { [[clang::always_inline]] epilogue_rows_res<TM, TN, NWAVES, false>(ep, C, ldc, M, N, m0, n0, wave, lane, smem); }
    else if (ep.residual && !ep.swiglu && !ep.act && ep.residual_dtype == LICV_F32 && !bf_out)
        { [[clang::always_inline]] epilogue_rows_res<TM, TN, NWAVES, true>(ep, C, ldc, M, N, m0, n0, wave, lane, smem); }
    else if (bf_out) { [[clang::always_inline]] epilogue_rows_generic<TM, TN, NWAVES, 8>(ep, C, ldc, M, N, m0, n0, wave, lane, smem); }
    else             { [[clang::always_inline]] epilogue_rows_generic<TM, TN, NWAVES, 4>(ep, C, ldc, M, N, m0, n0, wave, lane, smem); }
}
#pragma clang diagnostic pop

// ------------------------------------------------------------------------------------------------
// LDS-staged epilogue (all kernels).  Two measured problems of storing straight from the accumulator layout:
// 32-byte row fragments per store, and — far worse — code size: the element-wise epilogue (erf / tanh / exp
// bodies under run-time flags) unrolled over every accumulator register is hundreds of KiB of straight-line
// code that misses the instruction cache on every tile (~30 us per 256x256 tile, more than the MFMAs of a
// K=1280 tile).  So: phase A (unrolled, tiny) only does y0 = bf16(acc + bias) and parks the wave's block in
// an LDS image of the tile (row stride +16 B against bank conflicts); after one barrier, phase B is a compact
// run-time LOOP over full rows — 16 B per lane, whole 256/512-byte row segments — that applies activation /
// SwiGLU pairing / row gate / tanh-gate scale / residual on the bf16 values (exactly where the unfused torch
// ops would round) and stores bf16 or fp32.
// ------------------------------------------------------------------------------------------------
// timing-only instrumentation (tools/gemm_phases.py): when set, wave 0 of every pingpong workgroup records
// wall_clock64() at [0] start, [1] stage 0 published, [2] main loop done, [3] output image in LDS, [4] end
// (one copy of the pointer per translation unit: device variables are not shared between code objects; licv_gemm_debug_timestamps
// in gemm.hip sets this one and, through licv_gemm_exp_debug_timestamps, the experiments' copy)
static __device__ long long* g_dbg_ts = nullptr;
static int set_dbg_ts(void* dev_buffer) {
    long long* p = (long long*)dev_buffer;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_ts), &p, sizeof(p)) == hipSuccess ? LICV_OK : LICV_E_HIP;
}

// phase A alone: registers -> LDS image of y0 = bf16(acc + bias) (the tall kernel runs it on its four multiplying waves only and
// phase B on all eight)
template <int TN, int MT, int NT, bool SCALED = false>
__device__ __forceinline__ void epilogue_image(floatx4 (&acc)[MT][NT], const GemmEpi& ep, int M, int N, int m0, int n0, int wrow0, int wcol0,
                                               int lane, char* smem, const float (*bias_pre)[4] = nullptr) {
    constexpr int YS = TN * 2 + 16;                       // LDS row stride in bytes
    {
        const int rl = wrow0 + (lane & 15);
        const int cq = (lane >> 4) * 4;
        float bv[NT][4];
        static_for<0, NT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int ncol = n0 + wcol0 + j * 16 + cq;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                bv[j][r] = bias_pre ? bias_pre[j][r] : ((ep.bias && ncol + r < N) ? bf2f(ep.bias[ncol + r]) : 0.f);
        });
        float cs[NT][4];
        if (SCALED) {
            static_for<0, NT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int ncol = n0 + wcol0 + j * 16 + cq;
#pragma unroll
                for (int r = 0; r < 4; ++r) cs[j][r] = ncol + r < N ? ep.w_scale[ncol + r] : 0.f;
            });
        }
        static_for<0, MT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const float rs = SCALED ? ep.a_scale[min(m0 + rl + i * 16, M - 1)] : 1.0f;
            static_for<0, NT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                uint2 u;
                if (SCALED) {       // fp8 operands: C = (Aq . Wq^T) * a_scale[m] * w_scale[n]
                    u.x = (uint32_t)f2bf(acc[i][j][0] * rs * cs[j][0] + bv[j][0]) | ((uint32_t)f2bf(acc[i][j][1] * rs * cs[j][1] + bv[j][1]) << 16);
                    u.y = (uint32_t)f2bf(acc[i][j][2] * rs * cs[j][2] + bv[j][2]) | ((uint32_t)f2bf(acc[i][j][3] * rs * cs[j][3] + bv[j][3]) << 16);
                } else {
                    u.x = (uint32_t)f2bf(acc[i][j][0] + bv[j][0]) | ((uint32_t)f2bf(acc[i][j][1] + bv[j][1]) << 16);
                    u.y = (uint32_t)f2bf(acc[i][j][2] + bv[j][2]) | ((uint32_t)f2bf(acc[i][j][3] + bv[j][3]) << 16);
                }
                *reinterpret_cast<uint2*>(smem + (rl + i * 16) * YS + (wcol0 + j * 16 + cq) * 2) = u;
            });
        });
    }
}

template <int TM, int TN, int NWAVES, int MT, int NT, bool SCALED = false>
__device__ __forceinline__ void epilogue_staged(floatx4 (&acc)[MT][NT], const GemmEpi& ep, void* __restrict__ C, int64_t ldc,
                                                int M, int N, int m0, int n0, int wrow0, int wcol0, int wave, int lane, char* smem,
                                                long long* ts = nullptr, const float (*bias_pre)[4] = nullptr) {
    // ---- phase A: registers -> LDS image of y0 = bf16(acc + bias)
    epilogue_image<TN, MT, NT, SCALED>(acc, ep, M, N, m0, n0, wrow0, wcol0, lane, smem, bias_pre);
    __syncthreads();
    if (ts) ts[3] = wall_clock64();
    // ---- phase B: compact loop over rows; lane -> 8 (bf16 out) or 4 (fp32 out) consecutive OUTPUT columns
    epilogue_rows<TM, TN, NWAVES>(ep, C, ldc, M, N, m0, n0, wave, lane, smem);
}

// ---- 5-slot LDS-DMA ring of 32-deep K stages (lean / flow / fp8 kernels; experiments: ring, ping-pong, quad, pair, persist)
#define RING_STAGES 5
#define RING_STAGE_BYTES 32768

__device__ __forceinline__ int ring_off(int row, int chunk) {      // bytes within a [256][32] bf16 half-stage
    return row * 64 + ((chunk ^ (((row >> 2) & 1) << 1)) << 4);
}

__device__ __forceinline__ void wait_vmcnt(int n) {                 // n is wave-uniform: 12, 8, 4 or 0
    if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- four-wave kernels (quad, quad64, flow64)
__device__ __forceinline__ void wait_vmcnt8(int n) {                // waits vmcnt(8 * n); n is wave-uniform, 0..3
    if (n >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The accumulators are pinned to the AGPR half of the register file through the instruction's operand constraint: left to
// itself the allocator spreads 256 accumulators over both halves and shuffles them with v_accvgpr moves inside the loop.
#define QUAD_MFMA(c, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))

// arguments of one dense GEMM launch as the dispatch hands them to a kernel family
struct GemmArgs {
    const void* A; int64_t lda; const void* W; int64_t ldw; void* C; int64_t ldc;
    int M, N, K;
    GemmEpi ep;
    hipStream_t stream;
    int pp_group;       // tile-rows per XCD patch group (0 = the default heuristic)
    int num_cus;
};
