// Row-wise, HBM-bound kernels of the L-ICV hot path for gfx950: the ICV hook (inject + norm-preserve),
// RMSNorm / LayerNorm, rotary, and the small gather / layout kernels.
//
// Shape of every row kernel: ONE 64-lane wave owns one row, keeps it in registers (NCH chunks of
// 4 elements per lane, coalesced 16-B / 8-B accesses), reduces with cross-lane shuffles (no LDS, no
// barrier), and writes the row once: algorithmic traffic = 1 read + 1 write.  4 waves per workgroup.
#include "common.h"

#define WAVES_PER_BLOCK 4

template <int DT> struct RowIO;
template <> struct RowIO<LICV_F32> {
    static __device__ __forceinline__ floatx4 load4(const void* p, int64_t i) {
        return *reinterpret_cast<const floatx4*>(reinterpret_cast<const float*>(p) + i);
    }
};
template <> struct RowIO<LICV_BF16> {
    static __device__ __forceinline__ floatx4 load4(const void* p, int64_t i) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p) + i);
        floatx4 r;
        r[0] = __uint_as_float(u.x << 16); r[1] = __uint_as_float(u.x & 0xffff0000u);
        r[2] = __uint_as_float(u.y << 16); r[3] = __uint_as_float(u.y & 0xffff0000u);
        return r;
    }
};
__device__ __forceinline__ void store4_bf16(void* p, int64_t i, floatx4 v) {
    uint2 u;
    u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p) + i) = u;
}

// Per-lane sums over a row's NCH chunks are formed in FOUR GROUPS of NCH / 4 consecutive chunks (sequentially inside a group, then
// ((g0 + g1) + g2) + g3) whenever NCH % 4 == 0, and only then reduced across the 64 lanes.  The "wide" form of a row kernel (one
// row per 256-thread workgroup, wave w owning group w: *_wide_k below) then produces the same fp32 sums bit for bit, so a 24-row
// decode call can spread a row over four waves without leaving the results of the one-wave form.
template <int NCH> struct GroupSum {
    static constexpr int G = (NCH % 4 == 0) ? 4 : 1;
    static constexpr int PER = NCH / G;
    float p[G];
    __device__ __forceinline__ GroupSum() {
#pragma unroll
        for (int g = 0; g < G; ++g) p[g] = 0.f;
    }
    __device__ __forceinline__ void add(int c, float v) { p[c / PER] += v; }        // c is a compile-time constant after unrolling
    __device__ __forceinline__ float total() const { float t = p[0];
#pragma unroll
        for (int g = 1; g < G; ++g) t += p[g];
        return t; }
};

// ------------------------------------------------------------------------------------------------
// ICV hook forward.  ref:icv_src/icv_model/icv_intervention.py:62-84
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// fp8 image of a normalised row, written by the kernel that produced the row (BASELINE configs[4]: the GEMM that follows takes OCP
// e4m3 operands with one scale per row).  Exactly licv_quantize_rows_fp8 applied to the row's bf16 values - amax, scale =
// max(amax, 1e-12) / 448, e4m3(y / scale) - without the separate pass over it (that pass cost as much as the fp8 GEMMs it fed:
// 7 % of the Idefics2 32-shot step).  q == nullptr: off.  The bf16 row itself need not be written at all then (out == nullptr).
// ------------------------------------------------------------------------------------------------
struct Q8Out { uint8_t* q; float* scale; };

template <int NCH>       // y: the row as NCH x 4 values per lane, chunk c at element (c * 64 + lane) * 4, already rounded to bf16
__device__ __forceinline__ void emit_row_fp8(const floatx4 (&y)[NCH], int dim, int lane, uint8_t* __restrict__ qrow, float* __restrict__ scale_slot) {
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const bool ok = (c * 64 + lane) * 4 < dim;
#pragma unroll
        for (int j = 0; j < 4; ++j) amax = fmaxf(amax, ok ? fabsf(y[c][j]) : 0.f);
    }
    amax = wave_max(amax);
    const float sc = fmaxf(amax, 1e-12f) / 448.0f;
    if (lane == 0) *scale_slot = sc;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < dim) {
            int w = __builtin_amdgcn_cvt_pk_fp8_f32(y[c][0] / sc, y[c][1] / sc, 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(y[c][2] / sc, y[c][3] / sc, w, true);
            *reinterpret_cast<int*>(qrow + i) = w;
        }
    }
}

// A branch that still lies in the workspace of a split-K GEMM (licv_gemm_bf16_splitk_produce): value (m, col) = bf16(slice 0 + slice 1 +
// ... in slice order) — exactly what skinny_finalize_k would have written for a plain epilogue.  The WS forms of the row kernels below
// run ONE ROW PER WORKGROUP (blockDim 256): all four waves sum the slices of the row into an LDS image of the bf16 branch (every load
// of a thread in flight at once), then wave 0 alone runs the unchanged one-wave row algorithm with the image as its branch operand —
// the same per-lane summation chains and the same butterfly, so the results are bit-identical to finalize + row kernel, one launch
// instead of two (and no bf16 round trip of the branch through memory).
struct WsSrc { const float* ws; int splits; int64_t slice; int64_t stride; };
template <int NCH>
__device__ __forceinline__ void ws_branch_to_lds(const WsSrc& s, int64_t row, int dim, bf16_t* img) {
    constexpr int U = (NCH + 3) / 4;                                // 1024 columns per pass of the 256 threads
    const float* p = s.ws + row * s.stride;
    floatx4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = (u * 256 + (int)threadIdx.x) * 4, ii = i < dim ? i : 0;
        v[u] = *reinterpret_cast<const floatx4*>(p + ii);
    }
    for (int sp = 1; sp < s.splits; ++sp) {
        floatx4 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = (u * 256 + (int)threadIdx.x) * 4, ii = i < dim ? i : 0;
            t[u] = *reinterpret_cast<const floatx4*>(p + (int64_t)sp * s.slice + ii);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] += t[u];                   // slice order, as the finalize kernel adds them
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = (u * 256 + (int)threadIdx.x) * 4;
        if (i < dim) store4_bf16(img, i, v[u]);
    }
    __syncthreads();
}

template <int DT, int NCH, bool FUSE_NORM, bool WS = false>
__global__ __launch_bounds__(WS ? 256 : 64 * WAVES_PER_BLOCK)
void inject_renorm_fwd_k(const void* __restrict__ h, const float* __restrict__ icv, const float* __restrict__ alpha,
                         float* __restrict__ out, int64_t rows, int hidden,
                         const bf16_t* __restrict__ norm_w, bf16_t* __restrict__ xn, float eps,
                         const void* __restrict__ res, int res_dt, int norm_flavour, const bf16_t* __restrict__ pre = nullptr,
                         Q8Out q8 = Q8Out{nullptr, nullptr}, WsSrc wsrc = WsSrc{nullptr, 0, 0, 0}) {
    const int lane = threadIdx.x & 63;
    const int64_t row = WS ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t base = row * hidden;
    __shared__ __attribute__((aligned(16))) bf16_t ws_img[WS ? NCH * 256 : 8];
    const float a = alpha ? *alpha : 1.0f;
    // Every load of a phase is issued before the first value is used: chunk indices past the row are clamped to the row's first
    // chunk and their values discarded by a select, not skipped by a branch.  (A guard `if (i < hidden) { load; use; }` per chunk
    // made each chunk its own basic block with a full vmcnt(0) wait: 16 dependent round trips per row — 11 us of the 12 a 24-row
    // call took in a decode step, and fewer bytes in flight per wave at every size.)
    // The optional operands (`pre`, `res`) are wave-uniform: their loads are issued unconditionally from a stand-in address inside
    // `h` when absent and dropped by a select — no branch per chunk either.
    floatx4 x[NCH], hvv[NCH], bvv[NCH], vvv[NCH];
    GroupSum<NCH> gss, ghh;
    const bool has_pre = WS || pre != nullptr;
    const void* pre_p = has_pre ? (const void*)pre : h;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4, ii = i < hidden ? i : 0;
        hvv[c] = RowIO<DT>::load4(h, base + ii);
        vvv[c] = *reinterpret_cast<const floatx4*>(icv + ii);
    }
    if constexpr (WS) {                                    // `pre` = the branch summed from the split-K slices (see WsSrc); the loads above are in flight meanwhile
        ws_branch_to_lds<NCH>(wsrc, row, hidden, ws_img);
        if (threadIdx.x >= 64) return;
        pre_p = ws_img;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4, ii = i < hidden ? i : 0;
        bvv[c] = RowIO<LICV_BF16>::load4(pre_p, (WS ? 0 : base) + ii);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        const bool ok = i < hidden;
        floatx4 hv = hvv[c];
        // the layer's last residual add, h + branch in the stream's dtype, folded in (it was the down projection's epilogue)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sum = (DT == LICV_BF16) ? rbf(hv[j] + bvv[c][j]) : hv[j] + bvv[c][j];
            hv[j] = has_pre ? sum : hv[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = hv[j] + a * vvv[c][j];
            ghh.add(c, ok ? hv[j] * hv[j] : 0.f);
            gss.add(c, ok ? s * s : 0.f);
            x[c][j] = s;
        }
    }
    const float ss = wave_sum(gss.total());
    const float hh = wave_sum(ghh.total());
    // shifted / ||shifted|| * ||h||  in that order, as the reference writes it.  torch's .norm() returns
    // the input dtype: for a bf16 stream ||h|| is ROUNDED to bf16 (||h+v|| is fp32: the sum was promoted).
    const float ns = sqrtf(ss);
    const float nh = (DT == LICV_BF16) ? rbf(sqrtf(hh)) : sqrtf(hh);
    GroupSum<NCH> gq2;
    floatx4 rvv[NCH], wvv[NCH];
    const bool has_res = res != nullptr;                 // hooked BRANCH output (Idefics2 `.mlp`): stream = residual + edited branch
    const bool res32 = has_res && res_dt == LICV_F32;
    const void* res_p = has_res ? res : h;
    if (res32) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = (c * 64 + lane) * 4, ii = i < hidden ? i : 0;
            rvv[c] = RowIO<LICV_F32>::load4(res_p, base + ii);
        }
    } else {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = (c * 64 + lane) * 4, ii = i < hidden ? i : 0;
            rvv[c] = RowIO<LICV_BF16>::load4(res_p, base + ii);
        }
    }
    if (FUSE_NORM) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = (c * 64 + lane) * 4, ii = i < hidden ? i : 0;
            wvv[c] = RowIO<LICV_BF16>::load4(norm_w, ii);
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        const bool ok = i < hidden;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float e = x[c][j] / ns * nh;
            x[c][j] = has_res ? rvv[c][j] + e : e;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) gq2.add(c, ok ? x[c][j] * x[c][j] : 0.f);
        if (ok) *reinterpret_cast<floatx4*>(out + base + i) = x[c];
    }
    if (FUSE_NORM) {
        const float q2 = wave_sum(gq2.total());
        const float rs = rsqrtf(q2 / (float)hidden + eps);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = (c * 64 + lane) * 4;
            if (i < hidden) {
                floatx4 y;
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = wvv[c][j] * (norm_flavour == 1 ? x[c][j] * rs : rbf(x[c][j] * rs));
                if (xn) store4_bf16(xn, base + i, y);
#pragma unroll
                for (int j = 0; j < 4; ++j) x[c][j] = rbf(y[j]);
            }
        }
        if (q8.q) emit_row_fp8<NCH>(x, hidden, lane, q8.q + base, q8.scale + row);
    }
}

// ICV hook backward: grad wrt h (optional) and per-wave partial sums of grad wrt v.
template <int DT, int NCH>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK)
void inject_renorm_bwd_k(const void* __restrict__ h, const float* __restrict__ icv, const float* __restrict__ alpha,
                         const float* __restrict__ go, float* __restrict__ gh, float* __restrict__ gv_part,
                         int64_t rows, int hidden) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * WAVES_PER_BLOCK;
    const float a = alpha ? *alpha : 1.0f;
    floatx4 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int64_t row = wave; row < rows; row += nwaves) {
        const int64_t base = row * hidden;
        floatx4 s[NCH], g[NCH], hv[NCH];
        float ss = 0.f, hh = 0.f, gs = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = (c * 64 + lane) * 4;
            if (i < hidden) {
                hv[c] = RowIO<DT>::load4(h, base + i);
                g[c] = *reinterpret_cast<const floatx4*>(go + base + i);
                const floatx4 vv = *reinterpret_cast<const floatx4*>(icv + i);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[c][j] = hv[c][j] + a * vv[j];
                    ss += s[c][j] * s[c][j];
                    hh += hv[c][j] * hv[c][j];
                    gs += g[c][j] * s[c][j];
                }
            }
        }
        ss = wave_sum(ss); hh = wave_sum(hh); gs = wave_sum(gs);
        const float ns = sqrtf(ss);
        const float nh = (DT == LICV_BF16) ? rbf(sqrtf(hh)) : sqrtf(hh);
        const float r = nh / ns;                 // d out / d s = r * (I - u u^T),  u = s/ns
        const float gu = gs / ns;                // <g, u>
        const float kh = gu / nh;                // d out / d h (through ||h||) = u h^T / nh
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = (c * 64 + lane) * 4;
            if (i < hidden) {
                floatx4 dh;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dsj = r * (g[c][j] - s[c][j] / ns * gu);
                    acc[c][j] += dsj;
                    dh[j] = dsj + kh * hv[c][j];
                }
                if (gh) *reinterpret_cast<floatx4*>(gh + base + i) = dh;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < hidden) *reinterpret_cast<floatx4*>(gv_part + (int64_t)wave * hidden + i) = acc[c];
    }
}

// ------------------------------------------------------------------------------------------------
// RMSNorm.  hf:idefics/modeling_idefics.py:342-350 (flavour 0), hf:mistral/modeling_mistral.py:182-199 (1)
// ------------------------------------------------------------------------------------------------
template <int DT, int NCH>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK)
void rmsnorm_fwd_k(const void* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ out,
                   int64_t rows, int dim, int64_t inner, int64_t ld_x, int64_t ld_out, float eps, int flavour, Q8Out q8 = Q8Out{nullptr, nullptr}) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t ro = row / inner, ri = row % inner;
    const int64_t xb = ro * ld_x + ri * dim, ob = ro * ld_out + ri * dim;
    floatx4 v[NCH], wv[NCH];                                      // all loads first, no per-chunk branch (see inject_renorm_fwd_k)
    GroupSum<NCH> gss;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4, ii = i < dim ? i : 0;
        v[c] = RowIO<DT>::load4(x, xb + ii);
        wv[c] = RowIO<LICV_BF16>::load4(w, ii);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const bool ok = (c * 64 + lane) * 4 < dim;
#pragma unroll
        for (int j = 0; j < 4; ++j) gss.add(c, ok ? v[c][j] * v[c][j] : 0.f);
    }
    const float ss = wave_sum(gss.total());
    const float rs = rsqrtf(ss / (float)dim + eps);
    const bool single_round = (flavour == 1 && DT == LICV_F32);   // Mistral norm on an fp32 stream
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < dim) {
            floatx4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float n = v[c][j] * rs;
                y[j] = wv[c][j] * (single_round ? n : rbf(n));
            }
            if (out) store4_bf16(out, ob + i, y);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[c][j] = rbf(y[j]);
        }
    }
    if (q8.q) emit_row_fp8<NCH>(v, dim, lane, q8.q + row * dim, q8.scale + row);       // (contiguous rows: the q8 entry point takes inner = 1)
}

// h += branch (in place, in the stream's dtype: a bf16 stream rounds the sum — the o-projection's residual epilogue, folded in here so
// that GEMM writes its bf16 branch through the register-direct epilogue), then the RMSNorm of the new h.
template <int DT, int NCH, bool WS = false>
__global__ __launch_bounds__(WS ? 256 : 64 * WAVES_PER_BLOCK)
void add_rmsnorm_fwd_k(void* __restrict__ h, const bf16_t* __restrict__ branch, const bf16_t* __restrict__ w, bf16_t* __restrict__ out,
                       int64_t rows, int dim, float eps, int flavour, const float* __restrict__ row_gate, int use_scale, float scale,
                       Q8Out q8 = Q8Out{nullptr, nullptr}, WsSrc wsrc = WsSrc{nullptr, 0, 0, 0}) {
    const int lane = threadIdx.x & 63;
    const int64_t row = WS ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t base = row * dim;
    __shared__ __attribute__((aligned(16))) bf16_t ws_img[WS ? NCH * 256 : 8];
    const bool closed = row_gate && row_gate[row] == 0.0f;           // gated cross-attention: a token that attends no image adds nothing
    floatx4 v[NCH], bvv[NCH], wv[NCH];                             // all loads first, no per-chunk branch (see inject_renorm_fwd_k)
    GroupSum<NCH> gss;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4, ii = i < dim ? i : 0;
        v[c] = RowIO<DT>::load4(h, base + ii);
        wv[c] = RowIO<LICV_BF16>::load4(w, ii);
    }
    if constexpr (WS) {                                    // the branch summed from the split-K slices (see WsSrc); the loads above are in flight meanwhile
        ws_branch_to_lds<NCH>(wsrc, row, dim, ws_img);
        if (threadIdx.x >= 64) return;
        branch = ws_img;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4, ii = i < dim ? i : 0;
        bvv[c] = RowIO<LICV_BF16>::load4(branch, (WS ? 0 : base) + ii);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        const bool ok = i < dim;
        floatx4 bv = bvv[c];
        if (closed) bv = floatx4{0.f, 0.f, 0.f, 0.f};
        if (use_scale) {                                          // tanh(alpha) gate, rounded to bf16 as the GEMM epilogue does
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = rbf(scale * bv[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[c][j] = (DT == LICV_BF16) ? rbf(v[c][j] + bv[j]) : v[c][j] + bv[j];
            gss.add(c, ok ? v[c][j] * v[c][j] : 0.f);
        }
        if (ok) {
            if (DT == LICV_BF16) store4_bf16(h, base + i, v[c]);
            else *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(h) + base + i) = v[c];
        }
    }
    const float ss = wave_sum(gss.total());
    const float rs = rsqrtf(ss / (float)dim + eps);
    const bool single_round = (flavour == 1 && DT == LICV_F32);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < dim) {
            floatx4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float n = v[c][j] * rs;
                y[j] = wv[c][j] * (single_round ? n : rbf(n));
            }
            if (out) store4_bf16(out, base + i, y);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[c][j] = rbf(y[j]);
        }
    }
    if (q8.q) emit_row_fp8<NCH>(v, dim, lane, q8.q + base, q8.scale + row);
}

// ------------------------------------------------------------------------------------------------
// WIDE forms for the split-K slice consumers of a decode step (24 rows): one row per 256-thread workgroup, wave w owns chunk group w
// (see GroupSum), every lane sums the split-K slices of its OWN elements in slice order (what skinny_finalize_k adds, rounded to bf16
// like the branch it would have written) and the three row statistics are combined across the waves through LDS in group order before
// the 64-lane butterfly: bit-identical to the one-wave kernels above, with a quarter of the serial work per wave (the one-wave form
// spent ~10 us of a 13 us call in wave 0's 16 dependent chunks).
// ------------------------------------------------------------------------------------------------
template <int PER>
__device__ __forceinline__ void wide_branch_from_slices(const WsSrc& s, int64_t row, int dim, int wave, int lane, floatx4 (&bv)[PER]) {
    const float* p = s.ws + row * s.stride;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = ((wave * PER + k) * 64 + lane) * 4, ii = i < dim ? i : 0;
        bv[k] = *reinterpret_cast<const floatx4*>(p + ii);
    }
    for (int sp = 1; sp < s.splits; ++sp) {
        floatx4 t[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = ((wave * PER + k) * 64 + lane) * 4, ii = i < dim ? i : 0;
            t[k] = *reinterpret_cast<const floatx4*>(p + (int64_t)sp * s.slice + ii);
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) bv[k] += t[k];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[k][j] = rbf(bv[k][j]);
}
// the four group partials of a lane, combined in group order, then the butterfly over the lanes (== wave_sum(GroupSum::total()))
__device__ __forceinline__ float wide_combine(float partial, float (*red)[64], int wave, int lane) {
    red[wave][lane] = partial;
    __syncthreads();
    const float t = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    return wave_sum(t);
}

template <int DT, int NCH>
__global__ __launch_bounds__(256)
void inject_renorm_wide_k(const void* __restrict__ h, const float* __restrict__ icv, const float* __restrict__ alpha, float* __restrict__ out,
                          int hidden, const bf16_t* __restrict__ norm_w, bf16_t* __restrict__ xn, float eps, WsSrc src) {
    constexpr int PER = NCH / 4;
    __shared__ float red[3][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = blockIdx.x, base = row * hidden;
    const float a = alpha ? *alpha : 1.0f;
    floatx4 hv[PER], vv[PER], wv[PER], bv[PER], x[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = ((wave * PER + k) * 64 + lane) * 4, ii = i < hidden ? i : 0;
        hv[k] = RowIO<DT>::load4(h, base + ii);
        vv[k] = *reinterpret_cast<const floatx4*>(icv + ii);
        wv[k] = RowIO<LICV_BF16>::load4(norm_w, ii);
    }
    wide_branch_from_slices<PER>(src, row, hidden, wave, lane, bv);
    float pss = 0.f, phh = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const bool ok = ((wave * PER + k) * 64 + lane) * 4 < hidden;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float hj = (DT == LICV_BF16) ? rbf(hv[k][j] + bv[k][j]) : hv[k][j] + bv[k][j];
            const float sj = hj + a * vv[k][j];
            phh += ok ? hj * hj : 0.f;
            pss += ok ? sj * sj : 0.f;
            x[k][j] = sj;
        }
    }
    red[1][wave][lane] = phh;
    const float ss = wide_combine(pss, red[0], wave, lane);          // (its barrier also publishes red[1])
    const float hh = wave_sum(((red[1][0][lane] + red[1][1][lane]) + red[1][2][lane]) + red[1][3][lane]);
    const float ns = sqrtf(ss);
    const float nh = (DT == LICV_BF16) ? rbf(sqrtf(hh)) : sqrtf(hh);
    float pq2 = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = ((wave * PER + k) * 64 + lane) * 4;
        const bool ok = i < hidden;
#pragma unroll
        for (int j = 0; j < 4; ++j) x[k][j] = x[k][j] / ns * nh;
#pragma unroll
        for (int j = 0; j < 4; ++j) pq2 += ok ? x[k][j] * x[k][j] : 0.f;
        if (ok) *reinterpret_cast<floatx4*>(out + base + i) = x[k];
    }
    const float q2 = wide_combine(pq2, red[2], wave, lane);
    const float rs = rsqrtf(q2 / (float)hidden + eps);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = ((wave * PER + k) * 64 + lane) * 4;
        if (i < hidden) {
            floatx4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = wv[k][j] * rbf(x[k][j] * rs);
            store4_bf16(xn, base + i, y);
        }
    }
}

template <int DT, int NCH>
__global__ __launch_bounds__(256)
void add_rmsnorm_wide_k(void* __restrict__ h, const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int dim, float eps, int flavour,
                        const float* __restrict__ row_gate, int use_scale, float scale, WsSrc src) {
    constexpr int PER = NCH / 4;
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = blockIdx.x, base = row * dim;
    const bool closed = row_gate && row_gate[row] == 0.0f;
    floatx4 v[PER], wv[PER], bv[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = ((wave * PER + k) * 64 + lane) * 4, ii = i < dim ? i : 0;
        v[k] = RowIO<DT>::load4(h, base + ii);
        wv[k] = RowIO<LICV_BF16>::load4(w, ii);
    }
    wide_branch_from_slices<PER>(src, row, dim, wave, lane, bv);
    float pss = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = ((wave * PER + k) * 64 + lane) * 4;
        const bool ok = i < dim;
        floatx4 b = bv[k];
        if (closed) b = floatx4{0.f, 0.f, 0.f, 0.f};
        if (use_scale) {
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = rbf(scale * b[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[k][j] = (DT == LICV_BF16) ? rbf(v[k][j] + b[j]) : v[k][j] + b[j];
            pss += ok ? v[k][j] * v[k][j] : 0.f;
        }
        if (ok) {
            if (DT == LICV_BF16) store4_bf16(h, base + i, v[k]);
            else *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(h) + base + i) = v[k];
        }
    }
    const float ss = wide_combine(pss, red, wave, lane);
    const float rs = rsqrtf(ss / (float)dim + eps);
    const bool single_round = (flavour == 1 && DT == LICV_F32);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = ((wave * PER + k) * 64 + lane) * 4;
        if (i < dim) {
            floatx4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float n = v[k][j] * rs;
                y[j] = wv[k][j] * (single_round ? n : rbf(n));
            }
            store4_bf16(out, base + i, y);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm on bf16 (nn.LayerNorm): fp32 statistics, one rounding at the end.  16-byte accesses: each lane
// owns 8 consecutive elements per 512-element chunk (dim % 8 == 0), or 4 per 256-chunk otherwise.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_r;
__device__ __forceinline__ void load8_bf16(const bf16_t* p, float (&o)[8]) {
    const u32x4_r u = *reinterpret_cast<const u32x4_r*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = __uint_as_float(u[e] << 16); o[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
}
__device__ __forceinline__ void store8_bf16(bf16_t* p, const float (&v)[8]) {
    u32x4_r u;
#pragma unroll
    for (int e = 0; e < 4; ++e) u[e] = (uint32_t)f2bf(v[2 * e]) | ((uint32_t)f2bf(v[2 * e + 1]) << 16);
    *reinterpret_cast<u32x4_r*>(p) = u;
}

template <int NCH>      // chunks of 512 elements
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK)
void layernorm8_fwd_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const bf16_t* __restrict__ b,
                      bf16_t* __restrict__ out, int64_t rows, int dim, int64_t inner, int64_t ld_x, int64_t ld_out,
                      int64_t out_group, int64_t out_group_extra, float eps, Q8Out q8 = Q8Out{nullptr, nullptr}) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t ro = row / inner, ri = row % inner;
    const bf16_t* xp = x + ro * ld_x + ri * dim;
    bf16_t* op = out + ro * ld_out + ri * dim + (out_group > 0 ? (row / out_group) * out_group_extra : 0);
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 8;
        if (i < dim) {
            load8_bf16(xp + i, v[c]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[c][j];
        }
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 8;
        if (i < dim) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)dim + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 8;
        if (i < dim) {
            float wv[8], bv[8], y[8];
            load8_bf16(w + i, wv);
            load8_bf16(b + i, bv);
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = (v[c][j] - mean) * rstd * wv[j] + bv[j];
            if (out) store8_bf16(op + i, y);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = rbf(y[j]);
        }
    }
    if (q8.q) {                                              // the fp8 image of the row (contiguous rows), see emit_row_fp8
        float amax = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const bool ok = (c * 64 + lane) * 8 < dim;
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, ok ? fabsf(v[c][j]) : 0.f);
        }
        amax = wave_max(amax);
        const float sc = fmaxf(amax, 1e-12f) / 448.0f;
        if (lane == 0) q8.scale[row] = sc;
        uint8_t* qrow = q8.q + row * dim;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = (c * 64 + lane) * 8;
            if (i < dim) {
                int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][0] / sc, v[c][1] / sc, 0, false);
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][2] / sc, v[c][3] / sc, w0, true);
                int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][4] / sc, v[c][5] / sc, 0, false);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][6] / sc, v[c][7] / sc, w1, true);
                *reinterpret_cast<int2*>(qrow + i) = int2{w0, w1};
            }
        }
    }
}

template <int NCH>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK)
void layernorm_fwd_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const bf16_t* __restrict__ b,
                     bf16_t* __restrict__ out, int64_t rows, int dim, int64_t inner, int64_t ld_x, int64_t ld_out,
                     int64_t out_group, int64_t out_group_extra, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t ro = row / inner, ri = row % inner;
    const int64_t xb = ro * ld_x + ri * dim;
    const int64_t ob = ro * ld_out + ri * dim + (out_group > 0 ? (row / out_group) * out_group_extra : 0);
    floatx4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < dim) {
            v[c] = RowIO<LICV_BF16>::load4(x, xb + i);
            s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
        }
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < dim) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[c][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)dim + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < dim) {
            const floatx4 wv = RowIO<LICV_BF16>::load4(w, i);
            const floatx4 bv = RowIO<LICV_BF16>::load4(b, i);
            floatx4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = (v[c][j] - mean) * rstd * wv[j] + bv[j];
            store4_bf16(out, ob + i, y);
        }
    }
}

// Per-head q/k norms (dim = head_dim <= 128, millions of short rows): 4 rows per wave, 16 lanes x 8 elements (16 B) per
// row, reductions inside the 16-lane group.  One wave per 96-element row left 40 of 64 lanes idle and took 385 us for
// the perceiver's 1.36 M rows (2.7 ms per headline step); this form runs at the HBM rate.
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <bool LN>      // LN: LayerNorm with bias; else RMSNorm (y = w * bf16(x * rstd))
__global__ __launch_bounds__(256)
void headnorm_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const bf16_t* __restrict__ b, bf16_t* __restrict__ out,
                int64_t rows, int dim, int64_t inner, int64_t ld_x, int64_t ld_out, float eps) {
    const int sub = threadIdx.x & 15;
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool live = row < rows && sub * 8 < dim;
    const int64_t ro = row / inner, ri = row % inner;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) load8_bf16(x + ro * ld_x + ri * dim + sub * 8, v);
    float mean = 0.f;
    if (LN) {
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sm += v[j];
        mean = group16_sum(sm) / (float)dim;
    }
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = live ? v[j] - mean : 0.f; q += d * d; }
    const float rstd = rsqrtf(group16_sum(q) / (float)dim + eps);
    if (!live) return;
    float wv[8], y[8];
    load8_bf16(w + sub * 8, wv);
    if (LN) {
        float bv[8];
        load8_bf16(b + sub * 8, bv);
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = (v[j] - mean) * rstd * wv[j] + bv[j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = wv[j] * rbf(v[j] * rstd);
    }
    store8_bf16(out + ro * ld_out + ri * dim + sub * 8, y);
}

// Dynamic per-row quantisation to OCP fp8 e4m3 for the fp8 GEMM (BASELINE configs[4]): scale[r] = amax(x[r, :]) / 448,
// q[r, k] = e4m3(x[r, k] / scale[r]) (round to nearest even; |x / scale| <= 448 by construction, so nothing saturates).
// The same kernel quantises weights offline (rows = output channels).  One wave per row, the row held in registers.
template <int DT>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK)
void quantize_rows_fp8_k(const void* __restrict__ x, uint8_t* __restrict__ q, float* __restrict__ scale, int64_t rows, int dim,
                         int64_t ld_x, int64_t ld_q) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    float amax = 0.f;
    for (int i = lane * 4; i < dim; i += 256) {
        const floatx4 v = RowIO<DT>::load4(x, row * ld_x + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fabsf(v[j]));
    }
    amax = wave_max(amax);
    const float sc = fmaxf(amax, 1e-12f) / 448.0f;
    if (lane == 0) scale[row] = sc;
    for (int i = lane * 4; i < dim; i += 256) {                 // second sweep: the row (<= 64 KB) is still in L2
        const floatx4 v = RowIO<DT>::load4(x, row * ld_x + i);
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] / sc, v[1] / sc, 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] / sc, v[3] / sc, w, true);
        *reinterpret_cast<int*>(q + row * ld_q + i) = w;
    }
}

// The same quantiser for bf16 rows of up to 16384 elements (dim % 8 == 0), the row held in registers: every 16-byte load of the row
// is issued before the first value is used (buffer loads, out-of-range chunks read as zero), ONE pass over memory.  The two-sweep
// kernel above keeps one 8-byte load per lane in flight (28 dependent round trips for a 14336-wide SwiGLU output row): as the
// last separate quantiser calls of the fp8 step (attention outputs, SwiGLU / GELU outputs) it was 13 % of the Idefics2 32-shot step.
template <int NCH>      // chunks of 512 elements
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK)
void quantize_rows_fp8_wide_k(const bf16_t* __restrict__ x, uint8_t* __restrict__ q, float* __restrict__ scale, int64_t rows, int dim,
                              int64_t ld_x, int64_t ld_q) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x + row * ld_x), 0, (int)(dim * 2), 0x00020000);   // past the row: zeros
    u32x4_r raw[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) raw[c] = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)((c * 64 + lane) * 16), 0, 0);
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            amax = fmaxf(amax, fabsf(__uint_as_float(raw[c][e] << 16)));
            amax = fmaxf(amax, fabsf(__uint_as_float(raw[c][e] & 0xffff0000u)));
        }
    amax = wave_max(amax);
    const float sc = fmaxf(amax, 1e-12f) / 448.0f;
    if (lane == 0) scale[row] = sc;
    uint8_t* qrow = q + row * ld_q;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 8;
        if (i < dim) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(raw[c][e] << 16); v[2 * e + 1] = __uint_as_float(raw[c][e] & 0xffff0000u); }
            int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] / sc, v[1] / sc, 0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] / sc, v[3] / sc, w0, true);
            int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] / sc, v[5] / sc, 0, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] / sc, v[7] / sc, w1, true);
            *reinterpret_cast<int2*>(qrow + i) = int2{w0, w1};
        }
    }
}

// ViT embeddings + pre-LN: hf:idefics/vision.py:152-166 then :369
template <int NCH>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK)
void vit_embed_ln_k(const bf16_t* __restrict__ patches, const bf16_t* __restrict__ cls, const bf16_t* __restrict__ pos,
                    const bf16_t* __restrict__ w, const bf16_t* __restrict__ b, bf16_t* __restrict__ out,
                    int64_t n_img, int n_patch, int dim, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int T = n_patch + 1;
    if (row >= n_img * T) return;
    const int64_t img = row / T;
    const int t = (int)(row % T);
    const bf16_t* src = (t == 0) ? cls : patches + (img * n_patch + (t - 1)) * (int64_t)dim;
    floatx4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < dim) {
            const floatx4 a = RowIO<LICV_BF16>::load4(src, i);
            const floatx4 p = RowIO<LICV_BF16>::load4(pos, (int64_t)t * dim + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[c][j] = rbf(a[j] + p[j]); s += v[c][j]; }
        }
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < dim) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[c][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)dim + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < dim) {
            const floatx4 wv = RowIO<LICV_BF16>::load4(w, i);
            const floatx4 bv = RowIO<LICV_BF16>::load4(b, i);
            floatx4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = (v[c][j] - mean) * rstd * wv[j] + bv[j];
            store4_bf16(out, row * dim + i, y);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Rotary, rotate_half form.  hf:idefics/modeling_idefics.py:396-428.  Each bf16 torch op rounds:
// q*cos, rotate_half(q)*sin, and the sum.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void rotary_fwd_k(bf16_t* __restrict__ x, const bf16_t* __restrict__ cosT, const bf16_t* __restrict__ sinT,
                  const int64_t* __restrict__ pos, int64_t rows, int n_heads, int head_dim, int64_t ld,
                  int64_t tensor_stride, int n_tensors, int64_t n_pos) {
    const int half = head_dim >> 1;
    const int qper = half >> 2;                       // 4-wide groups per head
    const int64_t total = rows * n_tensors * (int64_t)n_heads * qper;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(idx % qper);
        int64_t r = idx / qper;
        const int hd = (int)(r % n_heads); r /= n_heads;
        const int t = (int)(r % n_tensors);
        const int64_t row = r / n_tensors;
        int64_t p = pos[row];
        p = p < 0 ? 0 : (p >= n_pos ? n_pos - 1 : p);
        const int i = g * 4;
        bf16_t* base = x + row * ld + t * tensor_stride + (int64_t)hd * head_dim;
        const floatx4 lo = RowIO<LICV_BF16>::load4(base, i);
        const floatx4 hi = RowIO<LICV_BF16>::load4(base, i + half);
        const floatx4 c = RowIO<LICV_BF16>::load4(cosT, p * head_dim + i);
        const floatx4 s = RowIO<LICV_BF16>::load4(sinT, p * head_dim + i);
        floatx4 olo, ohi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            olo[j] = rbf(lo[j] * c[j]) + rbf(-hi[j] * s[j]);
            ohi[j] = rbf(hi[j] * c[j]) + rbf(lo[j] * s[j]);
        }
        store4_bf16(base, i, olo);
        store4_bf16(base, i + half, ohi);
    }
}

// Decode / prefill with a KV cache: rotary on the Q and K heads of the fused projection and the append of (rotated K | V) to the
// cache in ONE launch (one workgroup per row).  Q is rotated in place, K goes straight to cache[b, past + s, 0:H], V is copied to
// cache[b, past + s, H:2H]; the arithmetic is rotary_fwd_k's, value for value (the K columns of qkv itself are left unrotated:
// with a cache nothing reads them).
__global__ __launch_bounds__(256)
void rotary_kv_append_k(bf16_t* __restrict__ qkv, const bf16_t* __restrict__ cosT, const bf16_t* __restrict__ sinT,
                        const int64_t* __restrict__ pos, int64_t S, int n_heads, int head_dim, int64_t H, int64_t n_pos,
                        bf16_t* __restrict__ cache, int64_t max_len, int64_t past) {
    const int64_t row = blockIdx.x;                       // b * S + s
    const int64_t b = row / S, sq = row - b * S;
    const int half = head_dim >> 1, qper = half >> 2;
    int64_t p = pos[row];
    p = p < 0 ? 0 : (p >= n_pos ? n_pos - 1 : p);
    bf16_t* crow = cache + (b * max_len + past + sq) * 2 * H;
    for (int idx = threadIdx.x; idx < 2 * n_heads * qper; idx += blockDim.x) {
        const int g = idx % qper, hd = (idx / qper) % n_heads, t = idx / (qper * n_heads);
        const int i = g * 4;
        bf16_t* base = qkv + row * 3 * H + t * H + (int64_t)hd * head_dim;
        const floatx4 lo = RowIO<LICV_BF16>::load4(base, i);
        const floatx4 hi = RowIO<LICV_BF16>::load4(base, i + half);
        const floatx4 c = RowIO<LICV_BF16>::load4(cosT, p * head_dim + i);
        const floatx4 s = RowIO<LICV_BF16>::load4(sinT, p * head_dim + i);
        floatx4 olo, ohi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            olo[j] = rbf(lo[j] * c[j]) + rbf(-hi[j] * s[j]);
            ohi[j] = rbf(hi[j] * c[j]) + rbf(lo[j] * s[j]);
        }
        bf16_t* dst = t == 0 ? base : crow + (int64_t)hd * head_dim;
        store4_bf16(dst, i, olo);
        store4_bf16(dst, i + half, ohi);
    }
    const uint4* vsrc = reinterpret_cast<const uint4*>(qkv + row * 3 * H + 2 * H);
    uint4* vdst = reinterpret_cast<uint4*>(crow + H);
    for (int64_t i = threadIdx.x; i < H / 8; i += blockDim.x) vdst[i] = vsrc[i];
}

// The same from the split-K slices of the QKV projection (see WsSrc): q / k / v = bf16(sum of the slices), rotated Q written to `qkv`
// (the attention kernel's Q operand), K and V straight into the cache — the projection's finalize launch and the bf16 round trip of
// its output are gone.  One thread per item (a 4 + 4 element rotary pair, or 4 V elements): grid (rows, ceil(3 * n_heads * head_dim / 8 / 256)).
__global__ __launch_bounds__(256)
void rotary_kv_append_ws_k(WsSrc src, bf16_t* __restrict__ qkv, const bf16_t* __restrict__ cosT, const bf16_t* __restrict__ sinT,
                           const int64_t* __restrict__ pos, int64_t S, int n_heads, int head_dim, int64_t H, int64_t n_pos,
                           bf16_t* __restrict__ cache, int64_t max_len, int64_t past) {
    const int64_t row = blockIdx.x;                       // b * S + s
    const int64_t b = row / S, sq = row - b * S;
    const int half = head_dim >> 1, qper = half >> 2;
    const int n_rot = 2 * n_heads * qper, n_v = (int)(H >> 2);
    const int idx = blockIdx.y * 256 + threadIdx.x;
    if (idx >= n_rot + n_v) return;
    bf16_t* crow = cache + (b * max_len + past + sq) * 2 * H;
    const float* wrow = src.ws + row * src.stride;
    auto sum4 = [&](int64_t col) -> floatx4 {
        floatx4 v = *reinterpret_cast<const floatx4*>(wrow + col);
        for (int sp = 1; sp < src.splits; ++sp) v += *reinterpret_cast<const floatx4*>(wrow + (int64_t)sp * src.slice + col);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rbf(v[j]);
        return v;
    };
    if (idx >= n_rot) {                                   // V: copy
        const int64_t i = (int64_t)(idx - n_rot) * 4;
        store4_bf16(crow + H, i, sum4(2 * H + i));
        return;
    }
    int64_t p = pos[row];
    p = p < 0 ? 0 : (p >= n_pos ? n_pos - 1 : p);
    const int g = idx % qper, hd = (idx / qper) % n_heads, t = idx / (qper * n_heads);
    const int i = g * 4;
    const int64_t col = t * H + (int64_t)hd * head_dim;
    floatx4 lo = *reinterpret_cast<const floatx4*>(wrow + col + i), hi = *reinterpret_cast<const floatx4*>(wrow + col + i + half);
    for (int sp = 1; sp < src.splits; ++sp) {
        lo += *reinterpret_cast<const floatx4*>(wrow + (int64_t)sp * src.slice + col + i);
        hi += *reinterpret_cast<const floatx4*>(wrow + (int64_t)sp * src.slice + col + i + half);
    }
    const floatx4 c = RowIO<LICV_BF16>::load4(cosT, p * head_dim + i);
    const floatx4 s = RowIO<LICV_BF16>::load4(sinT, p * head_dim + i);
    floatx4 olo, ohi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float l = rbf(lo[j]), h = rbf(hi[j]);
        olo[j] = rbf(l * c[j]) + rbf(-h * s[j]);
        ohi[j] = rbf(h * c[j]) + rbf(l * s[j]);
    }
    bf16_t* dst = t == 0 ? qkv + row * 3 * H + (int64_t)hd * head_dim : crow + (int64_t)hd * head_dim;
    store4_bf16(dst, i, olo);
    store4_bf16(dst, i + half, ohi);
}

// ------------------------------------------------------------------------------------------------
// gathers / layout
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void embed_gather_k(const int64_t* __restrict__ ids, const bf16_t* __restrict__ table, const bf16_t* __restrict__ extra,
                    bf16_t* __restrict__ out, int64_t n_tokens, int dim, int64_t vocab, int64_t n_extra) {
    const int vec = dim >> 3;                          // 16-byte chunks per row
    const int64_t total = n_tokens * vec;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t tok = idx / vec;
        const int c = (int)(idx % vec);
        int64_t id = ids[tok];
        const bf16_t* src;
        if (id >= vocab) { int64_t e = id - vocab; e = e >= n_extra ? n_extra - 1 : e; src = extra + e * dim; }
        else { id = id < 0 ? 0 : id; src = table + id * dim; }
        reinterpret_cast<uint4*>(out + tok * dim)[c] = reinterpret_cast<const uint4*>(src)[c];
    }
}

__global__ __launch_bounds__(256)
void im2col_k(const bf16_t* __restrict__ pix, bf16_t* __restrict__ out, int64_t n_img, int H, int W, int P, int64_t ld_out) {
    const int gh = H / P, gw = W / P;
    const int kdim = 3 * P * P;
    const int64_t total = n_img * gh * gw * ld_out;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(idx % ld_out);
        const int64_t r = idx / ld_out;
        bf16_t val = 0;
        if (col < kdim) {
            const int c = col / (P * P), rem = col % (P * P);
            const int py = rem / P, px = rem % P;
            const int gx = (int)(r % gw);
            const int gy = (int)((r / gw) % gh);
            const int64_t img = r / ((int64_t)gw * gh);
            val = pix[((img * 3 + c) * H + (gy * P + py)) * (int64_t)W + gx * P + px];
        }
        out[idx] = val;
    }
}

__global__ __launch_bounds__(256)
void tile_rows_k(const bf16_t* __restrict__ src, bf16_t* __restrict__ out, int64_t rows, int dim, int64_t period) {
    const int vec = dim >> 3;
    const int64_t total = rows * vec;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / vec;
        const int c = (int)(idx % vec);
        reinterpret_cast<uint4*>(out + r * dim)[c] = reinterpret_cast<const uint4*>(src + (r % period) * dim)[c];
    }
}

// out[idx[i], :] = src[i, :]   (hf:idefics2/modeling_idefics2.py:789-815 inputs_merger: image states into <image> slots)
__global__ __launch_bounds__(256)
void scatter_rows_k(const bf16_t* __restrict__ src, const int64_t* __restrict__ idx, bf16_t* __restrict__ out, int64_t n, int dim) {
    const int vec = dim >> 3;
    const int64_t total = n * vec;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / vec;
        const int c = (int)(i % vec);
        reinterpret_cast<uint4*>(out + idx[r] * dim)[c] = reinterpret_cast<const uint4*>(src + r * dim)[c];
    }
}

__global__ __launch_bounds__(256)
void swiglu_k(const bf16_t* __restrict__ gu, bf16_t* __restrict__ out, int64_t rows, int64_t inter) {
    const int64_t total = rows * inter;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / inter, c = idx % inter;
        const float g = bf2f(gu[r * 2 * inter + c]);
        const float u = bf2f(gu[r * 2 * inter + inter + c]);
        const float s = rbf(g * __builtin_amdgcn_rcpf(1.0f + __expf(-g)));
        out[idx] = f2bf(s * u);
    }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static inline int pick_nch(int64_t dim) {
    for (int n = 1; n <= 32; n <<= 1) if ((int64_t)n * 256 >= dim) return n;
    return 0;
}
static inline int row_blocks(int64_t rows) { return (int)((rows + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK); }
static inline int flat_blocks(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

#define DISPATCH_NCH(nch, CALL) switch (nch) { \
    case 1: { constexpr int N = 1; CALL; } break; case 2: { constexpr int N = 2; CALL; } break; \
    case 4: { constexpr int N = 4; CALL; } break; case 8: { constexpr int N = 8; CALL; } break; \
    case 16: { constexpr int N = 16; CALL; } break; case 32: { constexpr int N = 32; CALL; } break; \
    default: return licv_set_error(LICV_E_UNSUPPORTED, "row length %lld too large", (long long)dim_); }

static int inject_fwd_impl(const void* h, int h_dtype, const float* icv_row, const float* alpha, float* out, int64_t rows, int64_t hidden,
                           const void* norm_w, void* xn_out, float norm_eps, const void* res, int res_dt, int norm_flavour, void* stream,
                           const void* pre, Q8Out q8);

extern "C" int licv_inject_renorm_fwd(const void* h, int h_dtype, const float* icv_row, const float* alpha,
                                      float* out, int64_t rows, int64_t hidden,
                                      const void* norm_w, void* xn_out, float norm_eps, void* stream) {
    return inject_fwd_impl(h, h_dtype, icv_row, alpha, out, rows, hidden, norm_w, xn_out, norm_eps, nullptr, 0, 0, stream, nullptr, Q8Out{nullptr, nullptr});
}

// The same hook with the layer's last residual add folded in: the edited tensor is h + branch (the stream's dtype: a bf16 stream
// rounds the sum, an fp32 stream does not — exactly what the down projection's residual epilogue produced), so that GEMM can
// write its bf16 branch through the register-direct epilogue instead of a read-modify-write of the fp32 stream.
extern "C" int licv_inject_renorm_pre_fwd(const void* h, int h_dtype, const void* branch_bf16, const float* icv_row, const float* alpha,
                                          float* out, int64_t rows, int64_t hidden,
                                          const void* norm_w, void* xn_out, float norm_eps, void* stream) {
    LICV_CHECK_ARG(branch_bf16, "inject_renorm_pre_fwd: null branch");
    return inject_fwd_impl(h, h_dtype, icv_row, alpha, out, rows, hidden, norm_w, xn_out, norm_eps, nullptr, 0, 0, stream, branch_bf16, Q8Out{nullptr, nullptr});
}

extern "C" int licv_inject_renorm_add_fwd(const void* branch, int branch_dtype, const float* icv_row, const float* alpha,
                                          const void* residual, int residual_dtype, float* out, int64_t rows, int64_t hidden,
                                          const void* norm_w, void* xn_out, float norm_eps, int norm_flavour, void* stream) {
    LICV_CHECK_ARG(residual, "inject_renorm_add_fwd: null residual");
    LICV_CHECK_ARG(residual_dtype == LICV_BF16 || residual_dtype == LICV_F32, "inject_renorm_add_fwd: bad residual dtype");
    LICV_CHECK_ARG(norm_flavour == 0 || norm_flavour == 1, "inject_renorm_add_fwd: bad norm flavour");
    return inject_fwd_impl(branch, branch_dtype, icv_row, alpha, out, rows, hidden, norm_w, xn_out, norm_eps, residual, residual_dtype,
                           norm_flavour, stream, nullptr, Q8Out{nullptr, nullptr});
}

// ... with the fp8 image of the normalised rows (and, optionally, no bf16 copy of them: xn_out may be NULL)
extern "C" int licv_inject_renorm_add_fwd_q8(const void* branch, int branch_dtype, const float* icv_row, const float* alpha,
                                             const void* residual, int residual_dtype, float* out, int64_t rows, int64_t hidden,
                                             const void* norm_w, void* xn_out, void* q_fp8, float* q_scale, float norm_eps, int norm_flavour, void* stream) {
    LICV_CHECK_ARG(residual && norm_w && q_fp8 && q_scale, "inject_renorm_add_fwd_q8: null pointer");
    LICV_CHECK_ARG(residual_dtype == LICV_BF16 || residual_dtype == LICV_F32, "inject_renorm_add_fwd_q8: bad residual dtype");
    LICV_CHECK_ARG(norm_flavour == 0 || norm_flavour == 1, "inject_renorm_add_fwd_q8: bad norm flavour");
    return inject_fwd_impl(branch, branch_dtype, icv_row, alpha, out, rows, hidden, norm_w, xn_out, norm_eps, residual, residual_dtype,
                           norm_flavour, stream, nullptr, Q8Out{(uint8_t*)q_fp8, q_scale});
}

static int inject_fwd_impl(const void* h, int h_dtype, const float* icv_row, const float* alpha, float* out, int64_t rows, int64_t hidden,
                           const void* norm_w, void* xn_out, float norm_eps, const void* res, int res_dt, int norm_flavour, void* stream,
                           const void* pre, Q8Out q8) {
    LICV_CHECK_ARG(h && icv_row && out, "inject_renorm_fwd: null pointer");
    LICV_CHECK_ARG(hidden > 0 && hidden % 4 == 0, "inject_renorm_fwd: hidden (%lld) must be a positive multiple of 4", (long long)hidden);
    LICV_CHECK_ARG(h_dtype == LICV_BF16 || h_dtype == LICV_F32, "inject_renorm_fwd: bad dtype %d", h_dtype);
    LICV_CHECK_ARG((norm_w == nullptr) == (xn_out == nullptr && q8.q == nullptr), "inject_renorm_fwd: norm_w and xn_out (or the fp8 image) go together");
    if (rows <= 0) return LICV_OK;
    const int64_t dim_ = hidden;
    const int nch = pick_nch(hidden);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(row_blocks(rows)), block(64 * WAVES_PER_BLOCK);
    const bool fuse = norm_w != nullptr;
#define LAUNCH_INJ(DTV, FUSE) inject_renorm_fwd_k<DTV, N, FUSE><<<grid, block, 0, st>>>( \
        h, icv_row, alpha, out, rows, (int)hidden, (const bf16_t*)norm_w, (bf16_t*)xn_out, norm_eps, res, res_dt, norm_flavour, (const bf16_t*)pre, q8)
    if (h_dtype == LICV_F32) { if (fuse) { DISPATCH_NCH(nch, LAUNCH_INJ(LICV_F32, true)); } else { DISPATCH_NCH(nch, LAUNCH_INJ(LICV_F32, false)); } }
    else                     { if (fuse) { DISPATCH_NCH(nch, LAUNCH_INJ(LICV_BF16, true)); } else { DISPATCH_NCH(nch, LAUNCH_INJ(LICV_BF16, false)); } }
#undef LAUNCH_INJ
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

// ... with the branch still in the workspace of its split-K GEMM (licv_gemm_bf16_splitk_produce): see WsSrc.  One workgroup per row.
static int check_ws(const char* who, const float* ws, int splits, int64_t slice_elems, int64_t row_stride, int64_t dim) {
    LICV_CHECK_ARG(ws && splits >= 1 && slice_elems > 0 && row_stride >= dim, "%s: bad split-K workspace description", who);
    LICV_CHECK_ARG(((uintptr_t)ws & 15) == 0 && slice_elems % 4 == 0 && row_stride % 4 == 0, "%s: workspace slices must be 16-byte aligned", who);
    return LICV_OK;
}
extern "C" int licv_inject_renorm_pre_fwd_ws(const void* h, int h_dtype, const float* ws, int splits, int64_t slice_elems, int64_t row_stride,
                                             const float* icv_row, const float* alpha, float* out, int64_t rows, int64_t hidden,
                                             const void* norm_w, void* xn_out, float norm_eps, void* stream) {
    LICV_CHECK_ARG(h && icv_row && out && norm_w && xn_out, "inject_renorm_pre_fwd_ws: null pointer");
    LICV_CHECK_ARG(hidden > 0 && hidden % 4 == 0, "inject_renorm_pre_fwd_ws: hidden (%lld) must be a positive multiple of 4", (long long)hidden);
    LICV_CHECK_ARG(h_dtype == LICV_BF16 || h_dtype == LICV_F32, "inject_renorm_pre_fwd_ws: bad dtype %d", h_dtype);
    { const int rc = check_ws("inject_renorm_pre_fwd_ws", ws, splits, slice_elems, row_stride, hidden); if (rc != LICV_OK) return rc; }
    if (rows <= 0) return LICV_OK;
    const int64_t dim_ = hidden;
    const int nch = pick_nch(hidden);
    LICV_CHECK_ARG(nch > 0, "inject_renorm_pre_fwd_ws: hidden %lld unsupported", (long long)hidden);
    hipStream_t st = (hipStream_t)stream;
    const WsSrc src{ws, splits, slice_elems, row_stride};
    if (nch % 4 == 0) {                                    // the wide form: the row over the four waves of its workgroup
#define LAUNCH_INJWIDE(DTV, NC) inject_renorm_wide_k<DTV, NC><<<dim3((unsigned)rows), dim3(256), 0, st>>>( \
            h, icv_row, alpha, out, (int)hidden, (const bf16_t*)norm_w, (bf16_t*)xn_out, norm_eps, src)
        if (h_dtype == LICV_F32) { switch (nch) { case 4: LAUNCH_INJWIDE(LICV_F32, 4); break; case 8: LAUNCH_INJWIDE(LICV_F32, 8); break; case 16: LAUNCH_INJWIDE(LICV_F32, 16); break; default: LAUNCH_INJWIDE(LICV_F32, 32); break; } }
        else                     { switch (nch) { case 4: LAUNCH_INJWIDE(LICV_BF16, 4); break; case 8: LAUNCH_INJWIDE(LICV_BF16, 8); break; case 16: LAUNCH_INJWIDE(LICV_BF16, 16); break; default: LAUNCH_INJWIDE(LICV_BF16, 32); break; } }
#undef LAUNCH_INJWIDE
        LICV_LAUNCH_CHECK();
        return LICV_OK;
    }
#define LAUNCH_INJW(DTV) inject_renorm_fwd_k<DTV, N, true, true><<<dim3((unsigned)rows), dim3(256), 0, st>>>( \
        h, icv_row, alpha, out, rows, (int)hidden, (const bf16_t*)norm_w, (bf16_t*)xn_out, norm_eps, nullptr, 0, 0, nullptr, Q8Out{nullptr, nullptr}, src)
    if (h_dtype == LICV_F32) { DISPATCH_NCH(nch, LAUNCH_INJW(LICV_F32)); } else { DISPATCH_NCH(nch, LAUNCH_INJW(LICV_BF16)); }
#undef LAUNCH_INJW
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

static inline int bwd_blocks(int64_t rows) { int b = row_blocks(rows); return b > 128 ? 128 : (b < 1 ? 1 : b); }
extern "C" int64_t licv_inject_bwd_partials(int64_t rows) { return (int64_t)bwd_blocks(rows) * WAVES_PER_BLOCK; }

extern "C" int licv_inject_renorm_bwd(const void* h, int h_dtype, const float* icv_row, const float* alpha,
                                      const float* grad_out, float* grad_h, float* grad_v_partial,
                                      int64_t rows, int64_t hidden, void* stream) {
    LICV_CHECK_ARG(h && icv_row && grad_out && grad_v_partial, "inject_renorm_bwd: null pointer");
    LICV_CHECK_ARG(hidden > 0 && hidden % 4 == 0, "inject_renorm_bwd: hidden must be a multiple of 4");
    LICV_CHECK_ARG(h_dtype == LICV_BF16 || h_dtype == LICV_F32, "inject_renorm_bwd: bad dtype %d", h_dtype);
    const int64_t dim_ = hidden;
    const int nch = pick_nch(hidden);
    LICV_CHECK_ARG(nch > 0 && nch <= 16, "inject_renorm_bwd: hidden %lld unsupported", (long long)hidden);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(bwd_blocks(rows)), block(64 * WAVES_PER_BLOCK);
#define LAUNCH_INJB(DTV) inject_renorm_bwd_k<DTV, N><<<grid, block, 0, st>>>(h, icv_row, alpha, grad_out, grad_h, grad_v_partial, rows, (int)hidden)
    if (h_dtype == LICV_F32) { DISPATCH_NCH(nch, LAUNCH_INJB(LICV_F32)); } else { DISPATCH_NCH(nch, LAUNCH_INJB(LICV_BF16)); }
#undef LAUNCH_INJB
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_rmsnorm_fwd(const void* x, int x_dtype, const void* w, void* out, int64_t rows, int64_t dim,
                                int64_t inner, int64_t ld_x, int64_t ld_out, float eps, int flavour, void* stream) {
    LICV_CHECK_ARG(x && w && out, "rmsnorm_fwd: null pointer");
    LICV_CHECK_ARG(dim > 0 && dim % 4 == 0 && inner >= 1, "rmsnorm_fwd: dim (%lld) must be a multiple of 4", (long long)dim);
    LICV_CHECK_ARG(ld_x % 4 == 0 && ld_out % 4 == 0, "rmsnorm_fwd: leading dims must be multiples of 4");
    LICV_CHECK_ARG(x_dtype == LICV_BF16 || x_dtype == LICV_F32, "rmsnorm_fwd: bad dtype %d", x_dtype);
    LICV_CHECK_ARG(flavour == 0 || flavour == 1, "rmsnorm_fwd: bad flavour %d", flavour);
    if (rows <= 0) return LICV_OK;
    if (x_dtype == LICV_BF16 && dim <= 128 && dim % 8 == 0 && ld_x % 8 == 0 && ld_out % 8 == 0 &&
        (((uintptr_t)x | (uintptr_t)out | (uintptr_t)w) & 15) == 0) {
        headnorm_k<false><<<(unsigned)((rows * 16 + 255) / 256), 256, 0, (hipStream_t)stream>>>(
            (const bf16_t*)x, (const bf16_t*)w, nullptr, (bf16_t*)out, rows, (int)dim, inner, ld_x, ld_out, eps);
        LICV_LAUNCH_CHECK();
        return LICV_OK;
    }
    const int64_t dim_ = dim;
    const int nch = pick_nch(dim);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(row_blocks(rows)), block(64 * WAVES_PER_BLOCK);
#define LAUNCH_RMS(DTV) rmsnorm_fwd_k<DTV, N><<<grid, block, 0, st>>>(x, (const bf16_t*)w, (bf16_t*)out, rows, (int)dim, inner, ld_x, ld_out, eps, flavour)
    if (x_dtype == LICV_F32) { DISPATCH_NCH(nch, LAUNCH_RMS(LICV_F32)); } else { DISPATCH_NCH(nch, LAUNCH_RMS(LICV_BF16)); }
#undef LAUNCH_RMS
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_add_rmsnorm_fwd(void* h, int h_dtype, const void* branch_bf16, const float* row_gate, int use_scale, float scale,
                                    const void* w, void* out, int64_t rows, int64_t dim, float eps, int flavour, void* stream) {
    LICV_CHECK_ARG(h && branch_bf16 && w && out, "add_rmsnorm_fwd: null pointer");
    LICV_CHECK_ARG(dim > 0 && dim % 4 == 0, "add_rmsnorm_fwd: dim (%lld) must be a multiple of 4", (long long)dim);
    LICV_CHECK_ARG(h_dtype == LICV_BF16 || h_dtype == LICV_F32, "add_rmsnorm_fwd: bad dtype %d", h_dtype);
    LICV_CHECK_ARG(flavour == 0 || flavour == 1, "add_rmsnorm_fwd: bad flavour %d", flavour);
    if (rows <= 0) return LICV_OK;
    const int64_t dim_ = dim;
    const int nch = pick_nch(dim);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(row_blocks(rows)), block(64 * WAVES_PER_BLOCK);
#define LAUNCH_ARMS(DTV) add_rmsnorm_fwd_k<DTV, N><<<grid, block, 0, st>>>(h, (const bf16_t*)branch_bf16, (const bf16_t*)w, (bf16_t*)out, rows, (int)dim, eps, flavour, \
                                                                           row_gate, use_scale, scale)
    if (h_dtype == LICV_F32) { DISPATCH_NCH(nch, LAUNCH_ARMS(LICV_F32)); } else { DISPATCH_NCH(nch, LAUNCH_ARMS(LICV_BF16)); }
#undef LAUNCH_ARMS
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

// ---- norms that also (or only: out may be NULL) write the fp8 image of their rows; contiguous rows of `dim` elements
extern "C" int licv_add_rmsnorm_fwd_ws(void* h, int h_dtype, const float* ws, int splits, int64_t slice_elems, int64_t row_stride,
                                       const float* row_gate, int use_scale, float scale, const void* w, void* out,
                                       int64_t rows, int64_t dim, float eps, int flavour, void* stream) {
    LICV_CHECK_ARG(h && w && out, "add_rmsnorm_fwd_ws: null pointer");
    LICV_CHECK_ARG(dim > 0 && dim % 4 == 0, "add_rmsnorm_fwd_ws: dim (%lld) must be a multiple of 4", (long long)dim);
    LICV_CHECK_ARG(h_dtype == LICV_BF16 || h_dtype == LICV_F32, "add_rmsnorm_fwd_ws: bad dtype %d", h_dtype);
    LICV_CHECK_ARG(flavour == 0 || flavour == 1, "add_rmsnorm_fwd_ws: bad flavour %d", flavour);
    { const int rc = check_ws("add_rmsnorm_fwd_ws", ws, splits, slice_elems, row_stride, dim); if (rc != LICV_OK) return rc; }
    if (rows <= 0) return LICV_OK;
    const int64_t dim_ = dim;
    const int nch = pick_nch(dim);
    LICV_CHECK_ARG(nch > 0, "add_rmsnorm_fwd_ws: dim %lld unsupported", (long long)dim);
    hipStream_t st = (hipStream_t)stream;
    const WsSrc src{ws, splits, slice_elems, row_stride};
    if (nch % 4 == 0) {                                    // the wide form: the row over the four waves of its workgroup
#define LAUNCH_ARMSWIDE(DTV, NC) add_rmsnorm_wide_k<DTV, NC><<<dim3((unsigned)rows), dim3(256), 0, st>>>(h, (const bf16_t*)w, (bf16_t*)out, (int)dim, eps, flavour, \
            row_gate, use_scale, scale, src)
        if (h_dtype == LICV_F32) { switch (nch) { case 4: LAUNCH_ARMSWIDE(LICV_F32, 4); break; case 8: LAUNCH_ARMSWIDE(LICV_F32, 8); break; case 16: LAUNCH_ARMSWIDE(LICV_F32, 16); break; default: LAUNCH_ARMSWIDE(LICV_F32, 32); break; } }
        else                     { switch (nch) { case 4: LAUNCH_ARMSWIDE(LICV_BF16, 4); break; case 8: LAUNCH_ARMSWIDE(LICV_BF16, 8); break; case 16: LAUNCH_ARMSWIDE(LICV_BF16, 16); break; default: LAUNCH_ARMSWIDE(LICV_BF16, 32); break; } }
#undef LAUNCH_ARMSWIDE
        LICV_LAUNCH_CHECK();
        return LICV_OK;
    }
#define LAUNCH_ARMSW(DTV) add_rmsnorm_fwd_k<DTV, N, true><<<dim3((unsigned)rows), dim3(256), 0, st>>>(h, nullptr, (const bf16_t*)w, (bf16_t*)out, rows, (int)dim, eps, flavour, \
        row_gate, use_scale, scale, Q8Out{nullptr, nullptr}, src)
    if (h_dtype == LICV_F32) { DISPATCH_NCH(nch, LAUNCH_ARMSW(LICV_F32)); } else { DISPATCH_NCH(nch, LAUNCH_ARMSW(LICV_BF16)); }
#undef LAUNCH_ARMSW
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_rmsnorm_fwd_q8(const void* x, int x_dtype, const void* w, void* out, void* q_fp8, float* q_scale, int64_t rows, int64_t dim,
                                   float eps, int flavour, void* stream) {
    LICV_CHECK_ARG(x && w && q_fp8 && q_scale, "rmsnorm_fwd_q8: null pointer");
    LICV_CHECK_ARG(dim >= 256 && dim % 4 == 0, "rmsnorm_fwd_q8: dim (%lld) must be a multiple of 4, >= 256", (long long)dim);
    LICV_CHECK_ARG(x_dtype == LICV_BF16 || x_dtype == LICV_F32, "rmsnorm_fwd_q8: bad dtype %d", x_dtype);
    LICV_CHECK_ARG(flavour == 0 || flavour == 1, "rmsnorm_fwd_q8: bad flavour %d", flavour);
    if (rows <= 0) return LICV_OK;
    const int64_t dim_ = dim;
    const int nch = pick_nch(dim);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(row_blocks(rows)), block(64 * WAVES_PER_BLOCK);
    const Q8Out q8{(uint8_t*)q_fp8, q_scale};
#define LAUNCH_RMSQ(DTV) rmsnorm_fwd_k<DTV, N><<<grid, block, 0, st>>>(x, (const bf16_t*)w, (bf16_t*)out, rows, (int)dim, 1, dim, dim, eps, flavour, q8)
    if (x_dtype == LICV_F32) { DISPATCH_NCH(nch, LAUNCH_RMSQ(LICV_F32)); } else { DISPATCH_NCH(nch, LAUNCH_RMSQ(LICV_BF16)); }
#undef LAUNCH_RMSQ
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_add_rmsnorm_fwd_q8(void* h, int h_dtype, const void* branch_bf16, const void* w, void* out, void* q_fp8, float* q_scale,
                                       int64_t rows, int64_t dim, float eps, int flavour, void* stream) {
    LICV_CHECK_ARG(h && branch_bf16 && w && q_fp8 && q_scale, "add_rmsnorm_fwd_q8: null pointer");
    LICV_CHECK_ARG(dim > 0 && dim % 4 == 0, "add_rmsnorm_fwd_q8: dim (%lld) must be a multiple of 4", (long long)dim);
    LICV_CHECK_ARG(h_dtype == LICV_BF16 || h_dtype == LICV_F32, "add_rmsnorm_fwd_q8: bad dtype %d", h_dtype);
    LICV_CHECK_ARG(flavour == 0 || flavour == 1, "add_rmsnorm_fwd_q8: bad flavour %d", flavour);
    if (rows <= 0) return LICV_OK;
    const int64_t dim_ = dim;
    const int nch = pick_nch(dim);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(row_blocks(rows)), block(64 * WAVES_PER_BLOCK);
    const Q8Out q8{(uint8_t*)q_fp8, q_scale};
#define LAUNCH_ARMSQ(DTV) add_rmsnorm_fwd_k<DTV, N><<<grid, block, 0, st>>>(h, (const bf16_t*)branch_bf16, (const bf16_t*)w, (bf16_t*)out, rows, (int)dim, eps, flavour, \
                                                                            nullptr, 0, 0.f, q8)
    if (h_dtype == LICV_F32) { DISPATCH_NCH(nch, LAUNCH_ARMSQ(LICV_F32)); } else { DISPATCH_NCH(nch, LAUNCH_ARMSQ(LICV_BF16)); }
#undef LAUNCH_ARMSQ
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_layernorm_fwd_q8(const void* x, const void* w, const void* b, void* out, void* q_fp8, float* q_scale, int64_t rows, int64_t dim,
                                     float eps, void* stream) {
    LICV_CHECK_ARG(x && w && b && q_fp8 && q_scale, "layernorm_fwd_q8: null pointer");
    LICV_CHECK_ARG(dim >= 256 && dim <= 4096 && dim % 8 == 0, "layernorm_fwd_q8: dim (%lld) must be a multiple of 8 in [256, 4096]", (long long)dim);
    LICV_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)b & 15) == 0 && (!out || ((uintptr_t)out & 15) == 0) && ((uintptr_t)q_fp8 & 7) == 0,
                   "layernorm_fwd_q8: misaligned pointer");
    if (rows <= 0) return LICV_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(row_blocks(rows)), block(64 * WAVES_PER_BLOCK);
    const Q8Out q8{(uint8_t*)q_fp8, q_scale};
    const int n8 = (int)((dim + 511) / 512);
#define LAUNCH_LN8Q(NC) layernorm8_fwd_k<NC><<<grid, block, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)out, rows, (int)dim, 1, dim, dim, 0, 0, eps, q8)
    switch (n8) { case 1: LAUNCH_LN8Q(1); break; case 2: LAUNCH_LN8Q(2); break; case 3: LAUNCH_LN8Q(3); break; case 4: LAUNCH_LN8Q(4); break;
                  case 5: LAUNCH_LN8Q(5); break; case 6: LAUNCH_LN8Q(6); break; case 7: LAUNCH_LN8Q(7); break; default: LAUNCH_LN8Q(8); break; }
#undef LAUNCH_LN8Q
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_quantize_rows_fp8(const void* x, int x_dtype, void* q_fp8, float* scale, int64_t rows, int64_t dim,
                                      int64_t ld_x, int64_t ld_q, void* stream) {
    LICV_CHECK_ARG(x && q_fp8 && scale, "quantize_rows_fp8: null pointer");
    LICV_CHECK_ARG(dim > 0 && dim % 4 == 0 && ld_x % 4 == 0 && ld_q % 4 == 0 && ld_x >= dim && ld_q >= dim,
                   "quantize_rows_fp8: dim and leading dims must be multiples of 4 (dim %lld)", (long long)dim);
    LICV_CHECK_ARG(x_dtype == LICV_BF16 || x_dtype == LICV_F32, "quantize_rows_fp8: bad dtype %d", x_dtype);
    if (rows <= 0) return LICV_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(row_blocks(rows)), block(64 * WAVES_PER_BLOCK);
    if (x_dtype == LICV_BF16 && dim % 8 == 0 && dim <= 16384 && ld_x % 8 == 0 && ld_q % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)q_fp8 & 7) == 0 &&
        ld_x * 2 < (1ll << 31)) {
        const int n8 = (int)((dim + 511) / 512);
#define LAUNCH_QW(NC) quantize_rows_fp8_wide_k<NC><<<grid, block, 0, st>>>((const bf16_t*)x, (uint8_t*)q_fp8, scale, rows, (int)dim, ld_x, ld_q)
        if (n8 <= 2) LAUNCH_QW(2); else if (n8 <= 3) LAUNCH_QW(3); else if (n8 <= 4) LAUNCH_QW(4); else if (n8 <= 8) LAUNCH_QW(8);
        else if (n8 <= 9) LAUNCH_QW(9); else if (n8 <= 16) LAUNCH_QW(16); else if (n8 <= 24) LAUNCH_QW(24); else if (n8 <= 28) LAUNCH_QW(28); else LAUNCH_QW(32);
#undef LAUNCH_QW
        LICV_LAUNCH_CHECK();
        return LICV_OK;
    }
    if (x_dtype == LICV_F32) quantize_rows_fp8_k<LICV_F32><<<grid, block, 0, st>>>(x, (uint8_t*)q_fp8, scale, rows, (int)dim, ld_x, ld_q);
    else                     quantize_rows_fp8_k<LICV_BF16><<<grid, block, 0, st>>>(x, (uint8_t*)q_fp8, scale, rows, (int)dim, ld_x, ld_q);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_layernorm_fwd(const void* x, const void* w, const void* b, void* out, int64_t rows, int64_t dim,
                                  int64_t inner, int64_t ld_x, int64_t ld_out, int64_t out_group,
                                  int64_t out_group_extra, float eps, void* stream) {
    LICV_CHECK_ARG(x && w && b && out, "layernorm_fwd: null pointer");
    LICV_CHECK_ARG(dim > 0 && dim % 4 == 0 && inner >= 1, "layernorm_fwd: dim (%lld) must be a multiple of 4", (long long)dim);
    LICV_CHECK_ARG(ld_x % 4 == 0 && ld_out % 4 == 0 && out_group_extra % 4 == 0, "layernorm_fwd: strides must be multiples of 4");
    if (rows <= 0) return LICV_OK;
    if (dim <= 128 && dim % 8 == 0 && ld_x % 8 == 0 && ld_out % 8 == 0 && out_group == 0 &&
        (((uintptr_t)x | (uintptr_t)out | (uintptr_t)w | (uintptr_t)b) & 15) == 0) {
        headnorm_k<true><<<(unsigned)((rows * 16 + 255) / 256), 256, 0, (hipStream_t)stream>>>(
            (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)out, rows, (int)dim, inner, ld_x, ld_out, eps);
        LICV_LAUNCH_CHECK();
        return LICV_OK;
    }
    const int64_t dim_ = dim;
    const int nch = pick_nch(dim);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(row_blocks(rows)), block(64 * WAVES_PER_BLOCK);
    const bool wide = dim % 8 == 0 && ld_x % 8 == 0 && ld_out % 8 == 0 && out_group_extra % 8 == 0 && dim >= 256 && dim <= 4096 &&
                      ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)b & 15) == 0;
    if (wide) {
        const int n8 = (int)((dim + 511) / 512);
#define LAUNCH_LN8(NC) layernorm8_fwd_k<NC><<<grid, block, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, \
                 (bf16_t*)out, rows, (int)dim, inner, ld_x, ld_out, out_group, out_group_extra, eps)
        switch (n8) { case 1: LAUNCH_LN8(1); break; case 2: LAUNCH_LN8(2); break; case 3: LAUNCH_LN8(3); break; case 4: LAUNCH_LN8(4); break;
                      case 5: LAUNCH_LN8(5); break; case 6: LAUNCH_LN8(6); break; case 7: LAUNCH_LN8(7); break; default: LAUNCH_LN8(8); break; }
#undef LAUNCH_LN8
    } else {
        DISPATCH_NCH(nch, (layernorm_fwd_k<N><<<grid, block, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b,
                     (bf16_t*)out, rows, (int)dim, inner, ld_x, ld_out, out_group, out_group_extra, eps)));
    }
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_vit_embed_ln(const void* patches, const void* cls, const void* pos, const void* ln_w, const void* ln_b,
                                 void* out, int64_t n_img, int64_t n_patch, int64_t dim, float eps, void* stream) {
    LICV_CHECK_ARG(patches && cls && pos && ln_w && ln_b && out, "vit_embed_ln: null pointer");
    LICV_CHECK_ARG(dim > 0 && dim % 4 == 0, "vit_embed_ln: dim must be a multiple of 4");
    const int64_t rows = n_img * (n_patch + 1);
    if (rows <= 0) return LICV_OK;
    const int64_t dim_ = dim;
    const int nch = pick_nch(dim);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(row_blocks(rows)), block(64 * WAVES_PER_BLOCK);
    DISPATCH_NCH(nch, (vit_embed_ln_k<N><<<grid, block, 0, st>>>((const bf16_t*)patches, (const bf16_t*)cls, (const bf16_t*)pos,
                 (const bf16_t*)ln_w, (const bf16_t*)ln_b, (bf16_t*)out, n_img, (int)n_patch, (int)dim, eps)));
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_rotary_fwd(void* x, const void* cosT, const void* sinT, const int64_t* position_ids, int64_t rows,
                               int64_t n_heads, int64_t head_dim, int64_t ld, int64_t tensor_stride, int n_tensors,
                               int64_t n_pos, void* stream) {
    LICV_CHECK_ARG(x && cosT && sinT && position_ids, "rotary_fwd: null pointer");
    LICV_CHECK_ARG(head_dim > 0 && head_dim % 8 == 0, "rotary_fwd: head_dim (%lld) must be a multiple of 8", (long long)head_dim);
    LICV_CHECK_ARG(ld % 4 == 0 && tensor_stride % 4 == 0 && n_tensors >= 1 && n_pos > 0, "rotary_fwd: bad strides");
    if (rows <= 0) return LICV_OK;
    const int64_t total = rows * n_tensors * n_heads * (head_dim / 8);
    rotary_fwd_k<<<flat_blocks(total), 256, 0, (hipStream_t)stream>>>((bf16_t*)x, (const bf16_t*)cosT, (const bf16_t*)sinT,
        position_ids, rows, (int)n_heads, (int)head_dim, ld, tensor_stride, n_tensors, n_pos);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_rotary_kv_append(void* qkv, const void* cosT, const void* sinT, const int64_t* position_ids, int64_t batch, int64_t S,
                                     int64_t n_heads, int64_t head_dim, int64_t n_pos, void* cache, int64_t cache_max_len, int64_t past,
                                     void* stream) {
    LICV_CHECK_ARG(qkv && cosT && sinT && position_ids && cache, "rotary_kv_append: null pointer");
    LICV_CHECK_ARG(head_dim > 0 && head_dim % 8 == 0 && n_heads > 0 && n_pos > 0, "rotary_kv_append: head_dim (%lld) must be a multiple of 8", (long long)head_dim);
    LICV_CHECK_ARG(past >= 0 && S > 0 && past + S <= cache_max_len, "rotary_kv_append: %lld + %lld tokens do not fit a cache of %lld", (long long)past, (long long)S, (long long)cache_max_len);
    LICV_CHECK_ARG(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)cache & 15) == 0, "rotary_kv_append: misaligned pointer");
    if (batch <= 0) return LICV_OK;
    rotary_kv_append_k<<<(unsigned)(batch * S), 256, 0, (hipStream_t)stream>>>((bf16_t*)qkv, (const bf16_t*)cosT, (const bf16_t*)sinT,
        position_ids, S, (int)n_heads, (int)head_dim, n_heads * head_dim, n_pos, (bf16_t*)cache, cache_max_len, past);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_rotary_kv_append_ws(const float* ws, int splits, int64_t slice_elems, int64_t row_stride, void* qkv,
                                        const void* cosT, const void* sinT, const int64_t* position_ids, int64_t batch, int64_t S,
                                        int64_t n_heads, int64_t head_dim, int64_t n_pos, void* cache, int64_t cache_max_len, int64_t past,
                                        void* stream) {
    LICV_CHECK_ARG(qkv && cosT && sinT && position_ids && cache, "rotary_kv_append_ws: null pointer");
    LICV_CHECK_ARG(head_dim > 0 && head_dim % 8 == 0 && n_heads > 0 && n_pos > 0, "rotary_kv_append_ws: head_dim (%lld) must be a multiple of 8", (long long)head_dim);
    LICV_CHECK_ARG(past >= 0 && S > 0 && past + S <= cache_max_len, "rotary_kv_append_ws: %lld + %lld tokens do not fit a cache of %lld", (long long)past, (long long)S, (long long)cache_max_len);
    LICV_CHECK_ARG(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)cache & 15) == 0, "rotary_kv_append_ws: misaligned pointer");
    const int64_t H = n_heads * head_dim;
    { const int rc = check_ws("rotary_kv_append_ws", ws, splits, slice_elems, row_stride, 3 * H); if (rc != LICV_OK) return rc; }
    if (batch <= 0) return LICV_OK;
    const int64_t items = 2 * n_heads * (head_dim / 8) + H / 4;
    rotary_kv_append_ws_k<<<dim3((unsigned)(batch * S), (unsigned)((items + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
        WsSrc{ws, splits, slice_elems, row_stride}, (bf16_t*)qkv, (const bf16_t*)cosT, (const bf16_t*)sinT,
        position_ids, S, (int)n_heads, (int)head_dim, H, n_pos, (bf16_t*)cache, cache_max_len, past);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_embed_gather(const int64_t* ids, const void* table, const void* extra, void* out, int64_t n_tokens,
                                 int64_t dim, int64_t vocab, int64_t n_extra, void* stream) {
    LICV_CHECK_ARG(ids && table && out, "embed_gather: null pointer");
    LICV_CHECK_ARG(dim > 0 && dim % 8 == 0, "embed_gather: dim must be a multiple of 8");
    LICV_CHECK_ARG(n_extra == 0 || extra, "embed_gather: additional table missing");
    if (n_tokens <= 0) return LICV_OK;
    embed_gather_k<<<flat_blocks(n_tokens * (dim / 8)), 256, 0, (hipStream_t)stream>>>(ids, (const bf16_t*)table,
        (const bf16_t*)extra, (bf16_t*)out, n_tokens, (int)dim, n_extra > 0 ? vocab : INT64_MAX, n_extra);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_im2col_patches(const void* pix, void* out, int64_t n_img, int64_t height, int64_t width, int64_t patch,
                                   int64_t ld_out, void* stream) {
    LICV_CHECK_ARG(pix && out, "im2col_patches: null pointer");
    LICV_CHECK_ARG(patch > 0 && height >= patch && width >= patch, "im2col_patches: image smaller than one patch");   // a ragged edge is dropped like a stride-P conv does
    LICV_CHECK_ARG(ld_out >= 3 * patch * patch, "im2col_patches: ld_out too small");
    const int64_t total = n_img * (height / patch) * (width / patch) * ld_out;
    if (total <= 0) return LICV_OK;
    im2col_k<<<flat_blocks(total), 256, 0, (hipStream_t)stream>>>((const bf16_t*)pix, (bf16_t*)out, n_img, (int)height,
        (int)width, (int)patch, ld_out);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_tile_rows(const void* src, void* out, int64_t rows, int64_t dim, int64_t period, void* stream) {
    LICV_CHECK_ARG(src && out && period > 0, "tile_rows: bad argument");
    LICV_CHECK_ARG(dim > 0 && dim % 8 == 0, "tile_rows: dim must be a multiple of 8");
    if (rows <= 0) return LICV_OK;
    tile_rows_k<<<flat_blocks(rows * (dim / 8)), 256, 0, (hipStream_t)stream>>>((const bf16_t*)src, (bf16_t*)out, rows, (int)dim, period);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_scatter_rows(const void* src, const int64_t* idx, void* out, int64_t n, int64_t dim, void* stream) {
    LICV_CHECK_ARG(src && idx && out, "scatter_rows: null pointer");
    LICV_CHECK_ARG(dim > 0 && dim % 8 == 0, "scatter_rows: dim must be a multiple of 8");
    if (n <= 0) return LICV_OK;
    scatter_rows_k<<<flat_blocks(n * (dim / 8)), 256, 0, (hipStream_t)stream>>>((const bf16_t*)src, idx, (bf16_t*)out, n, (int)dim);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_swiglu(const void* gu, void* out, int64_t rows, int64_t inter, void* stream) {
    LICV_CHECK_ARG(gu && out && inter > 0, "swiglu: bad argument");
    if (rows <= 0) return LICV_OK;
    swiglu_k<<<flat_blocks(rows * inter), 256, 0, (hipStream_t)stream>>>((const bf16_t*)gu, (bf16_t*)out, rows, inter);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}
