// Backward kernels for the L-ICV student pass (ref:icv_src/icv_module.py:97-98 runs the hooked forward with grad;
// only `icv` and `alpha` are trainable, so what is needed is d loss / d hidden-state through the frozen LMM).
// Dense-layer input gradients reuse the forward MFMA GEMM on transposed weight copies; this file holds the
// rest: RMSNorm, SwiGLU, rotary (inverse rotation = forward kernel with -sin), small-sequence attention, the
// masked-KL rows, and the residual-branch cast.  The student sequence is the query only (tens of tokens), so
// these kernels favour simplicity over peak throughput; fp32 maths, bf16 where the forward rounded.
#include "common.h"

#define BW_WAVES 4

static int g_attn_bwd_staged = 1;      // licv_backward_option 0
static int g_rmsnorm_bwd_wide = 1;     // licv_backward_option 1

__device__ __forceinline__ floatx4 ld4(const void* p, int dt, int64_t i) {
    if (dt == LICV_F32) return *reinterpret_cast<const floatx4*>(reinterpret_cast<const float*>(p) + i);
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p) + i);
    floatx4 r;
    r[0] = __uint_as_float(u.x << 16); r[1] = __uint_as_float(u.x & 0xffff0000u);
    r[2] = __uint_as_float(u.y << 16); r[3] = __uint_as_float(u.y & 0xffff0000u);
    return r;
}
__device__ __forceinline__ void st4(void* p, int dt, int64_t i, floatx4 v) {
    if (dt == LICV_F32) { *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(p) + i) = v; return; }
    uint2 u;
    u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p) + i) = u;
}

// ------------------------------------------------------------------------------------------------
// RMSNorm backward (hf:idefics/modeling_idefics.py:342-350 forward: y = w * bf16(x * rsqrt(mean(x^2)+eps))).
// g = bf16(dy * w); dx = rs * (g - xhat * mean(g * xhat));  dx is ADDED to `dx_acc` when accumulate != 0.
// round_g = 0: g stays fp32 - the Mistral flavour on an fp32 stream (hf:mistral/modeling_mistral.py:182-199 returns
// weight * x.to(input_dtype): with the fp32 stream behind a hook the product, and so its gradient, is fp32; the Idefics flavour casts
// to the weight's bf16 BEFORE the product whatever the stream's dtype).
// Rows addressed like the forward kernel (inner / ld) so the per-head q norm works in place.
// ------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(64 * BW_WAVES)
void rmsnorm_bwd_k(const void* __restrict__ x, int x_dt, const bf16_t* __restrict__ w, const void* __restrict__ dy, int dy_dt,
                   void* __restrict__ dx, int dx_dt, int64_t rows, int dim, int64_t inner, int64_t ld_x, int64_t ld_dy,
                   int64_t ld_dx, float eps, int accumulate, int round_g) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * BW_WAVES + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t ro = row / inner, ri = row % inner;
    const int64_t xb = ro * ld_x + ri * dim, yb = ro * ld_dy + ri * dim, db = ro * ld_dx + ri * dim;
    floatx4 xv[NCH], gv[NCH];
    // The two row statistics are formed in NG = 4 groups of NCH / 4 consecutive chunks (one group where NCH < 4), each summed over the
    // wave, then added in group order: the order rmsnorm_bwd_wide_k (one wave per group) produces, so the two kernels agree bit for bit.
    constexpr int NG = NCH >= 4 ? 4 : 1, PG = NCH / NG;
    float sg[NG];
#pragma unroll
    for (int q = 0; q < NG; ++q) sg[q] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < dim) {
            xv[c] = ld4(x, x_dt, xb + i);
            const floatx4 d = ld4(dy, dy_dt, yb + i);
            const floatx4 wv = ld4(w, LICV_BF16, i);
#pragma unroll
            for (int j = 0; j < 4; ++j) { sg[c / PG] += xv[c][j] * xv[c][j]; const float gw = d[j] * wv[j]; gv[c][j] = round_g ? rbf(gw) : gw; }
        }
    }
    float ss = wave_sum(sg[0]);
#pragma unroll
    for (int q = 1; q < NG; ++q) ss += wave_sum(sg[q]);
    const float rs = rsqrtf(ss / (float)dim + eps);
#pragma unroll
    for (int q = 0; q < NG; ++q) sg[q] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < dim) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sg[c / PG] += gv[c][j] * xv[c][j] * rs;
        }
    }
    float dot = wave_sum(sg[0]);
#pragma unroll
    for (int q = 1; q < NG; ++q) dot += wave_sum(sg[q]);
    dot = dot / (float)dim;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * 64 + lane) * 4;
        if (i < dim) {
            floatx4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = rs * (gv[c][j] - xv[c][j] * rs * dot);
            if (accumulate) { const floatx4 a = ld4(dx, dx_dt, db + i);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] += a[j]; }
            st4(dx, dx_dt, db + i, o);
        }
    }
}

// Four waves per row (NCH % 4 == 0: rows of 1024 elements and more), wave q on chunk group q: the student's 256 rows x 4096 were 64
// workgroups of one-wave rows, sixteen dependent chunks each - 22 us per call, 81 calls per backward pass.  Same sums in the same
// order as the one-wave kernel.
// WS: dy is still the fp32 split-K slices of the dgrad GEMM that produced it (licv_gemm_bf16_splitk_produce): element = bf16(sum of the
// slices in slice order) - what the finalize launch would have written and this kernel read back; one launch less per norm.
struct BwdWs { const float* ws; int splits; int64_t slice; int64_t stride; };
template <int NCH, bool WS = false>
__global__ __launch_bounds__(256)
void rmsnorm_bwd_wide_k(const void* __restrict__ x, int x_dt, const bf16_t* __restrict__ w, const void* __restrict__ dy, int dy_dt,
                        void* __restrict__ dx, int dx_dt, int64_t rows, int dim, int64_t inner, int64_t ld_x, int64_t ld_dy,
                        int64_t ld_dx, float eps, int accumulate, int round_g, BwdWs src = BwdWs{nullptr, 0, 0, 0}) {
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t row = blockIdx.x;
    const int64_t ro = row / inner, ri = row % inner;
    const int64_t xb = ro * ld_x + ri * dim, yb = ro * ld_dy + ri * dim, db = ro * ld_dx + ri * dim;
    constexpr int PG = NCH / 4;
    floatx4 xv[PG], gv[PG];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < PG; ++c) {
        const int i = ((q * PG + c) * 64 + lane) * 4;
        if (i < dim) {
            xv[c] = ld4(x, x_dt, xb + i);
            floatx4 d;
            if (WS) {
                const float* p = src.ws + row * src.stride + i;
                d = *reinterpret_cast<const floatx4*>(p);
                for (int sp = 1; sp < src.splits; ++sp) d += *reinterpret_cast<const floatx4*>(p + (int64_t)sp * src.slice);
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] = rbf(d[j]);
            } else d = ld4(dy, dy_dt, yb + i);
            const floatx4 wv = ld4(w, LICV_BF16, i);
#pragma unroll
            for (int j = 0; j < 4; ++j) { ss += xv[c][j] * xv[c][j]; const float gw = d[j] * wv[j]; gv[c][j] = round_g ? rbf(gw) : gw; }
        }
    }
    ss = wave_sum(ss);
    if (lane == 0) red[0][q] = ss;
    __syncthreads();
    ss = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    const float rs = rsqrtf(ss / (float)dim + eps);
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < PG; ++c) {
        const int i = ((q * PG + c) * 64 + lane) * 4;
        if (i < dim) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dot += gv[c][j] * xv[c][j] * rs;
        }
    }
    dot = wave_sum(dot);
    if (lane == 0) red[1][q] = dot;
    __syncthreads();
    dot = (((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]) / (float)dim;
#pragma unroll
    for (int c = 0; c < PG; ++c) {
        const int i = ((q * PG + c) * 64 + lane) * 4;
        if (i < dim) {
            floatx4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = rs * (gv[c][j] - xv[c][j] * rs * dot);
            if (accumulate) { const floatx4 a = ld4(dx, dx_dt, db + i);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] += a[j]; }
            st4(dx, dx_dt, db + i, o);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// SwiGLU backward on the UNFUSED (rows, 2I) [gate | up] buffer: act = bf16(silu(g)) * u.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void swiglu_bwd_k(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ dact, bf16_t* __restrict__ dgu, int64_t rows, int64_t inter) {
    const int64_t total = rows * inter;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / inter, c = idx % inter;
        const float g = bf2f(gu[r * 2 * inter + c]);
        const float u = bf2f(gu[r * 2 * inter + inter + c]);
        const float d = bf2f(dact[idx]);
        const float sig = 1.0f / (1.0f + __expf(-g));
        const float s = rbf(g * sig);
        const float ds = rbf(d * u);                                // grad wrt silu(g) (bf16 like the autograd product)
        const float dsilu = sig * (1.0f + g * (1.0f - sig));
        dgu[r * 2 * inter + c] = f2bf(ds * dsilu);
        dgu[r * 2 * inter + inter + c] = f2bf(d * s);
    }
}

// residual branch grad: out = bf16(bf16(dh) * scale), rows with gate == 0 zeroed (gated cross-attention / plain branch)
__global__ __launch_bounds__(256)
void branch_grad_k(const float* __restrict__ dh, bf16_t* __restrict__ out, int64_t rows, int64_t dim, float scale, int use_scale,
                   const float* __restrict__ row_gate) {
    const int64_t total = rows * dim;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / dim;
        float v = rbf(dh[idx]);
        if (use_scale) v = rbf(v * scale);
        if (row_gate && row_gate[r] == 0.0f) v = 0.0f;
        out[idx] = f2bf(v);
    }
}

// ------------------------------------------------------------------------------------------------
// Attention backward for SHORT sequences (Sq * Sk <= 16384): one workgroup per (batch, head); P and dS live in
// LDS as fp32 (Sq x Sk each).  Thread i owns query row i for P/dS/dQ, then thread j owns key j for dK/dV.
// mask_mode as in the forward kernel (0 none, 1 causal + key_valid, 3 image mask).  want_dkv = 0 skips dK/dV
// (cross-attention: K/V come from the frozen vision side).
// ------------------------------------------------------------------------------------------------
struct AttnBwdP {
    const bf16_t* q; int64_t q_bs, q_rs;
    const bf16_t* k; const bf16_t* v; int64_t kv_bs, kv_rs;
    const bf16_t* dout;                 // (B, Sq, nh*hd) dense
    bf16_t* dq; int64_t dq_bs, dq_rs;
    bf16_t* dk; bf16_t* dv; int64_t dkv_bs, dkv_rs;
    int B, Sq, Sk, nh, nkv, hd;
    float scale;
    int mask_mode;
    const int32_t* key_valid;
    const int32_t* img_mask; int n_img, img_len;
    int want_dkv;
};

// One workgroup per (batch, head); every phase spreads its (row, column) or (row, 8-channel chunk) items over all its
// lanes (1024 in the staged form: the phases are chains of dependent FMAs and LDS reads, and one wave per SIMD left each
// latency exposed - 38 us with 256 lanes).  P and dS (Sq x Sk fp32 each) live in LDS.  STAGED (round 4): so do the head's Q, K, V and dO rows (rows of hd + 8 bf16, so
// the 16-byte reads of 16 consecutive rows fall into 64 different banks) - the 32-token student's 1024 (query, key) pairs each walked
// two 128-long dot products straight from global memory, dependent 16-byte loads one L2 round trip apiece: 67 us per layer for a
// few MFLOP.  The arithmetic and its order are the same in both forms: bit-identical results (tests/test_backward_ops_gpu.py).
// Not STAGED (the operands do not fit beside P and dS): Q / K / V / dO stay in global memory (L1 / L2 resident).
template <bool STAGED>
__global__ __launch_bounds__(1024)
void attn_bwd_small_k(AttnBwdP a) {
    extern __shared__ __attribute__((aligned(16))) float bsm[];
    const int head = blockIdx.x % a.nh, b = blockIdx.x / a.nh;
    const int kvh = head / (a.nh / a.nkv);
    const int coff = a.Sk - a.Sq;
    const int hd = a.hd, Sq = a.Sq, Sk = a.Sk, tid = threadIdx.x, nthr = blockDim.x;
    const int nch = hd >> 3;
    const int npair = (Sq * Sk + 3) & ~3;     // (keeps the operand rows behind P and dS 16-byte aligned)
    float* P = bsm;                          // Sq x Sk
    float* dS = bsm + npair;                 // Sq x Sk
    const int RS = hd + 8;
    bf16_t* sQ = reinterpret_cast<bf16_t*>(bsm + 2 * npair);
    bf16_t* sG = sQ + Sq * RS;
    bf16_t* sK = sG + Sq * RS;
    bf16_t* sV = sK + Sk * RS;
    const bf16_t* qb = a.q + (int64_t)b * a.q_bs + (int64_t)head * hd;
    const bf16_t* kb = a.k + (int64_t)b * a.kv_bs + (int64_t)kvh * hd;
    const bf16_t* vb = a.v + (int64_t)b * a.kv_bs + (int64_t)kvh * hd;
    const int64_t do_rs = (int64_t)a.nh * hd;
    const bf16_t* dob = a.dout + (int64_t)b * Sq * do_rs + (int64_t)head * hd;
    if (STAGED) {
        for (int idx = tid; idx < 2 * (Sq + Sk) * nch; idx += nthr) {
            const int r = idx / nch, c = (idx - r * nch) * 8;
            uint4 v;
            bf16_t* dst;
            if (r < Sq) { v = *reinterpret_cast<const uint4*>(qb + (int64_t)r * a.q_rs + c); dst = sQ + r * RS + c; }
            else if (r < 2 * Sq) { v = *reinterpret_cast<const uint4*>(dob + (int64_t)(r - Sq) * do_rs + c); dst = sG + (r - Sq) * RS + c; }
            else if (r < 2 * Sq + Sk) { v = *reinterpret_cast<const uint4*>(kb + (int64_t)(r - 2 * Sq) * a.kv_rs + c); dst = sK + (r - 2 * Sq) * RS + c; }
            else { v = *reinterpret_cast<const uint4*>(vb + (int64_t)(r - 2 * Sq - Sk) * a.kv_rs + c); dst = sV + (r - 2 * Sq - Sk) * RS + c; }
            *reinterpret_cast<uint4*>(dst) = v;
        }
        __syncthreads();
    }
    // 8 consecutive channels (16 bytes) of a row of Q / dO / K / V
    auto q8 = [&](int i, int c) -> uint4 { return STAGED ? *reinterpret_cast<const uint4*>(sQ + i * RS + c) : *reinterpret_cast<const uint4*>(qb + (int64_t)i * a.q_rs + c); };
    auto g8 = [&](int i, int c) -> uint4 { return STAGED ? *reinterpret_cast<const uint4*>(sG + i * RS + c) : *reinterpret_cast<const uint4*>(dob + (int64_t)i * do_rs + c); };
    auto k8 = [&](int j, int c) -> uint4 { return STAGED ? *reinterpret_cast<const uint4*>(sK + j * RS + c) : *reinterpret_cast<const uint4*>(kb + (int64_t)j * a.kv_rs + c); };
    auto v8 = [&](int j, int c) -> uint4 { return STAGED ? *reinterpret_cast<const uint4*>(sV + j * RS + c) : *reinterpret_cast<const uint4*>(vb + (int64_t)j * a.kv_rs + c); };
    auto dot = [&](auto&& fx, int i, auto&& fy, int j) -> float {      // (the order of dot8)
        float acc = 0.f;
        for (int d = 0; d < hd; d += 8) {
            const uint4 xv = fx(i, d), yv = fy(j, d);
            const uint32_t xu[4] = {xv.x, xv.y, xv.z, xv.w}, yu[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc += __uint_as_float(xu[e] << 16) * __uint_as_float(yu[e] << 16);
                acc += __uint_as_float(xu[e] & 0xffff0000u) * __uint_as_float(yu[e] & 0xffff0000u);
            }
        }
        return acc;
    };
    // phase 1: masked scores and dP = dO V^T
    for (int idx = tid; idx < Sq * Sk; idx += nthr) {
        const int i = idx / Sk, j = idx - i * Sk;
        bool ok = true;
        if (a.mask_mode == 1) ok = (j <= i + coff) && (!a.key_valid || a.key_valid[(int64_t)b * Sk + j] != 0);
        else if (a.mask_mode == 2) ok = !a.key_valid || a.key_valid[(int64_t)b * Sk + j] != 0;
        else if (a.mask_mode == 3) ok = a.img_mask[((int64_t)b * Sq + i) * a.n_img + j / a.img_len] != 0;
        float s = -INFINITY, dp = 0.f;
        if (ok) {
            s = dot(q8, i, k8, j) * a.scale;
            dp = dot(g8, i, v8, j);
        }
        P[idx] = s;
        dS[idx] = dp;
    }
    __syncthreads();
    // phase 2: one wave per query row: softmax (probabilities rounded to bf16 like the forward), D = sum_j p dP, dS
    const int lane = tid & 63, wave = tid >> 6;
    for (int i = wave; i < Sq; i += (nthr >> 6)) {
        float mx = -INFINITY;
        for (int j = lane; j < Sk; j += 64) mx = fmaxf(mx, P[i * Sk + j]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < Sk; j += 64) {
            const float sc = P[i * Sk + j];
            const float e = (sc == -INFINITY) ? 0.f : __expf(sc - mx);
            P[i * Sk + j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
        float D = 0.f;
        for (int j = lane; j < Sk; j += 64) {
            const float p = rbf(P[i * Sk + j] * inv);
            P[i * Sk + j] = p;
            D += p * dS[i * Sk + j];
        }
        D = wave_sum(D);
        for (int j = lane; j < Sk; j += 64) dS[i * Sk + j] = P[i * Sk + j] * (dS[i * Sk + j] - D) * a.scale;
    }
    __syncthreads();
    // phase 3: dQ_i = sum_j dS_ij K_j, 8 channels per item
    for (int idx = tid; idx < Sq * nch; idx += nthr) {
        const int i = idx / nch, c = (idx - i * nch) * 8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < Sk; ++j) {
            const float ds = dS[i * Sk + j];
            if (ds == 0.f) continue;
            const uint4 kv = k8(j, c);
            const uint32_t ku[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[2 * e] += ds * __uint_as_float(ku[e] << 16); acc[2 * e + 1] += ds * __uint_as_float(ku[e] & 0xffff0000u); }
        }
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (uint32_t)f2bf(acc[2 * e]) | ((uint32_t)f2bf(acc[2 * e + 1]) << 16);
        *reinterpret_cast<uint4*>(a.dq + (int64_t)b * a.dq_bs + (int64_t)i * a.dq_rs + (int64_t)head * hd + c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    if (!a.want_dkv) return;
    // phase 4: dK_j = sum_i dS_ij Q_i ; dV_j = sum_i P_ij dO_i  (written per QUERY head; GQA groups are reduced by head_group_sum_k)
    for (int idx = tid; idx < Sk * nch; idx += nthr) {
        const int j = idx / nch, c = (idx - j * nch) * 8;
        float ak[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, av[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < Sq; ++i) {
            const float ds = dS[i * Sk + j], p = P[i * Sk + j];
            if (ds == 0.f && p == 0.f) continue;
            const uint4 qv = q8(i, c);
            const uint4 gv = g8(i, c);
            const uint32_t qu[4] = {qv.x, qv.y, qv.z, qv.w}, gu[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ak[2 * e] += ds * __uint_as_float(qu[e] << 16); ak[2 * e + 1] += ds * __uint_as_float(qu[e] & 0xffff0000u);
                av[2 * e] += p * __uint_as_float(gu[e] << 16);  av[2 * e + 1] += p * __uint_as_float(gu[e] & 0xffff0000u);
            }
        }
        uint32_t ok_[4], ov[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ok_[e] = (uint32_t)f2bf(ak[2 * e]) | ((uint32_t)f2bf(ak[2 * e + 1]) << 16);
            ov[e] = (uint32_t)f2bf(av[2 * e]) | ((uint32_t)f2bf(av[2 * e + 1]) << 16);
        }
        const int64_t off = (int64_t)b * a.dkv_bs + (int64_t)j * a.dkv_rs + (int64_t)head * hd + c;
        *reinterpret_cast<uint4*>(a.dk + off) = make_uint4(ok_[0], ok_[1], ok_[2], ok_[3]);
        *reinterpret_cast<uint4*>(a.dv + off) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
    }
}

// ------------------------------------------------------------------------------------------------
// d/d student-logits of  mean_rows( sum_v p (log(p+eps) - log(q+eps)) ) * T^2   (ref:icv_src/icv_module.py:121-134)
// grad rows are written densely: (n_rows, vocab) bf16.  upstream = d loss / d kl (normally 1).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float blk_reduce(float v, bool is_max, float* red) {
    v = is_max ? wave_max(v) : wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}

template <bool BF>
__global__ __launch_bounds__(256)
void kl_rows_bwd_k(const void* __restrict__ stu, const void* __restrict__ tea, const int64_t* __restrict__ srows,
                   const int64_t* __restrict__ trows, int64_t vocab, int64_t ld_s, int64_t ld_t, float T, float eps,
                   float coef, const float* __restrict__ coef_dev, bf16_t* __restrict__ grad, int64_t ld_g) {
    __shared__ float red[8];
    const int64_t row = blockIdx.x;
    const int64_t sb = srows[row] * ld_s, tb = trows[row] * ld_t;
    if (coef_dev) coef *= *coef_dev;                      // upstream gradient that lives on the device (autograd path)
    auto ld = [&](const void* p, int64_t i) -> float {
        return BF ? bf2f(reinterpret_cast<const bf16_t*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
    };
    float ms = -INFINITY, mt = -INFINITY;
    for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) { ms = fmaxf(ms, ld(stu, sb + i) / T); mt = fmaxf(mt, ld(tea, tb + i) / T); }
    ms = blk_reduce(ms, true, red); mt = blk_reduce(mt, true, red);
    float zs = 0.f, zt = 0.f;
    for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) { zs += __expf(ld(stu, sb + i) / T - ms); zt += __expf(ld(tea, tb + i) / T - mt); }
    zs = blk_reduce(zs, false, red); zt = blk_reduce(zt, false, red);
    // g_v = d kl_row / d q_v = -p_v / (q_v + eps);  d kl_row / d z_u = (1/T) q_u (g_u - sum_v q_v g_v)
    float qg = 0.f;
    for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) {
        const float q = __expf(ld(stu, sb + i) / T - ms) / zs;
        const float p = __expf(ld(tea, tb + i) / T - mt) / zt;
        qg += q * (-p / (q + eps));
    }
    qg = blk_reduce(qg, false, red);
    for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) {
        const float q = __expf(ld(stu, sb + i) / T - ms) / zs;
        const float p = __expf(ld(tea, tb + i) / T - mt) / zt;
        const float gz = q * ((-p / (q + eps)) - qg) / T;
        grad[row * ld_g + i] = f2bf(gz * coef);
    }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static inline int nch_for(int64_t dim) { for (int n = 1; n <= 32; n <<= 1) if ((int64_t)n * 256 >= dim) return n; return 0; }
static inline int flat_grid(int64_t total) { int64_t b = (total + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

extern "C" int licv_rmsnorm_bwd(const void* x, int x_dtype, const void* w_bf16, const void* dy, int dy_dtype, void* dx, int dx_dtype,
                                int64_t rows, int64_t dim, int64_t inner, int64_t ld_x, int64_t ld_dy, int64_t ld_dx, float eps,
                                int accumulate, int flavour, void* stream) {
    LICV_CHECK_ARG(x && w_bf16 && dy && dx, "rmsnorm_bwd: null pointer");
    LICV_CHECK_ARG(flavour == 0 || flavour == 1, "rmsnorm_bwd: flavour must be 0 (Idefics) or 1 (Mistral)");
    const int round_g = (flavour == 1 && x_dtype == LICV_F32) ? 0 : 1;
    LICV_CHECK_ARG(dim > 0 && dim % 4 == 0 && inner >= 1 && ld_x % 4 == 0 && ld_dy % 4 == 0 && ld_dx % 4 == 0, "rmsnorm_bwd: dims must be multiples of 4");
    if (rows <= 0) return LICV_OK;
    const int nch = nch_for(dim);
    LICV_CHECK_ARG(nch > 0 && nch <= 16, "rmsnorm_bwd: row length %lld unsupported", (long long)dim);
    const dim3 grid((unsigned)((rows + BW_WAVES - 1) / BW_WAVES)), block(64 * BW_WAVES);
    hipStream_t st = (hipStream_t)stream;
#define L(NC) rmsnorm_bwd_k<NC><<<grid, block, 0, st>>>(x, x_dtype, (const bf16_t*)w_bf16, dy, dy_dtype, dx, dx_dtype, rows, (int)dim, inner, ld_x, ld_dy, ld_dx, eps, accumulate, round_g)
#define LW(NC) rmsnorm_bwd_wide_k<NC><<<dim3((unsigned)rows), dim3(256), 0, st>>>(x, x_dtype, (const bf16_t*)w_bf16, dy, dy_dtype, dx, dx_dtype, rows, (int)dim, inner, ld_x, ld_dy, ld_dx, eps, accumulate, round_g)
    if (g_rmsnorm_bwd_wide && nch >= 4) {
        switch (nch) { case 4: LW(4); break; case 8: LW(8); break; default: LW(16); break; }
    } else
    switch (nch) { case 1: L(1); break; case 2: L(2); break; case 4: L(4); break; case 8: L(8); break; default: L(16); break; }
#undef L
#undef LW
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_rmsnorm_bwd_ws(const void* x, int x_dtype, const void* w_bf16, const float* ws, int splits, int64_t slice_elems,
                                   int64_t row_stride, void* dx, int dx_dtype, int64_t rows, int64_t dim, float eps, int accumulate,
                                   int flavour, void* stream) {
    LICV_CHECK_ARG(x && w_bf16 && ws && dx, "rmsnorm_bwd_ws: null pointer");
    LICV_CHECK_ARG((x_dtype == LICV_BF16 || x_dtype == LICV_F32) && (dx_dtype == LICV_BF16 || dx_dtype == LICV_F32), "rmsnorm_bwd_ws: bad dtype");
    LICV_CHECK_ARG(flavour == 0 || flavour == 1, "rmsnorm_bwd_ws: bad flavour %d", flavour);
    LICV_CHECK_ARG(splits >= 1 && slice_elems > 0 && row_stride >= dim && row_stride % 4 == 0 && slice_elems % 4 == 0 && ((uintptr_t)ws & 15) == 0,
                   "rmsnorm_bwd_ws: bad split-K workspace description");
    LICV_CHECK_ARG(dim >= 1024 && dim % 4 == 0, "rmsnorm_bwd_ws: rows of %lld elements (needs >= 1024, a multiple of 4)", (long long)dim);
    if (rows <= 0) return LICV_OK;
    const int nch = nch_for(dim);
    LICV_CHECK_ARG(nch >= 4 && nch <= 16, "rmsnorm_bwd_ws: row length %lld unsupported", (long long)dim);
    const int round_g = (flavour == 1 && x_dtype == LICV_F32) ? 0 : 1;
    hipStream_t st = (hipStream_t)stream;
    const BwdWs src{ws, splits, slice_elems, row_stride};
#define LW(NC) rmsnorm_bwd_wide_k<NC, true><<<dim3((unsigned)rows), dim3(256), 0, st>>>(x, x_dtype, (const bf16_t*)w_bf16, nullptr, LICV_BF16, dx, dx_dtype, rows, (int)dim, 1, dim, dim, dim, eps, accumulate, round_g, src)
    switch (nch) { case 4: LW(4); break; case 8: LW(8); break; default: LW(16); break; }
#undef LW
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_swiglu_bwd(const void* gu_bf16, const void* dact_bf16, void* dgu_bf16, int64_t rows, int64_t inter, void* stream) {
    LICV_CHECK_ARG(gu_bf16 && dact_bf16 && dgu_bf16 && inter > 0, "swiglu_bwd: bad argument");
    if (rows <= 0) return LICV_OK;
    swiglu_bwd_k<<<flat_grid(rows * inter), 256, 0, (hipStream_t)stream>>>((const bf16_t*)gu_bf16, (const bf16_t*)dact_bf16, (bf16_t*)dgu_bf16, rows, inter);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_branch_grad(const float* dh, void* out_bf16, int64_t rows, int64_t dim, float scale, int use_scale,
                                const float* row_gate, void* stream) {
    LICV_CHECK_ARG(dh && out_bf16 && dim > 0, "branch_grad: bad argument");
    if (rows <= 0) return LICV_OK;
    branch_grad_k<<<flat_grid(rows * dim), 256, 0, (hipStream_t)stream>>>(dh, (bf16_t*)out_bf16, rows, dim, scale, use_scale, row_gate);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

// A/B timing and tests (results are bit-identical either way): option 0 = 0: attn_bwd_small keeps Q / K / V / dO in global memory;
// option 1 = 0: rmsnorm_bwd on one wave per row at every row length
extern "C" int licv_backward_option(int option, int value) {
    if (option == 0) { g_attn_bwd_staged = value; return LICV_OK; }
    if (option == 1) { g_rmsnorm_bwd_wide = value; return LICV_OK; }
    return licv_set_error(LICV_E_BADARG, "backward_option: unknown option %d", option);
}

extern "C" int licv_attn_bwd_small(const licv_attn_args* x, const void* dout, void* dq, int64_t dq_bs, int64_t dq_rs,
                                   void* dk, void* dv, int64_t dkv_bs, int64_t dkv_rs, void* stream) {
    LICV_CHECK_ARG(x && x->q && x->k && x->v && dout && dq, "attn_bwd_small: null pointer");
    LICV_CHECK_ARG(x->Sq > 0 && x->Sk > 0 && x->Sq * x->Sk <= 16384, "attn_bwd_small: Sq*Sk = %lld exceeds the short-sequence limit 16384",
                   (long long)(x->Sq * x->Sk));
    LICV_CHECK_ARG(x->mask_mode >= 0 && x->mask_mode <= 3, "attn_bwd_small: bad mask mode");
    LICV_CHECK_ARG(x->head_dim % 8 == 0 && x->q_rs % 8 == 0 && x->kv_rs % 8 == 0 && x->q_bs % 8 == 0 && x->kv_bs % 8 == 0 &&
                   dq_bs % 8 == 0 && dq_rs % 8 == 0 && dkv_bs % 8 == 0 && dkv_rs % 8 == 0,
                   "attn_bwd_small: head_dim and strides must be multiples of 8 elements");
    LICV_CHECK_ARG((((uintptr_t)x->q | (uintptr_t)x->k | (uintptr_t)x->v | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
                   "attn_bwd_small: pointers must be 16-byte aligned");
    LICV_CHECK_ARG(x->mask_mode != 3 || (x->img_mask && x->n_img > 0 && x->img_len > 0), "attn_bwd_small: image mask arguments missing");
    const int want = (dk && dv) ? 1 : 0;
    // dK / dV are written PER QUERY HEAD (n_heads x head_dim columns); with GQA the caller reduces each group of
    // n_heads / n_kv_heads query heads with licv_head_group_sum (the backward of repeat_kv)
    AttnBwdP p;
    p.q = (const bf16_t*)x->q; p.q_bs = x->q_bs; p.q_rs = x->q_rs;
    p.k = (const bf16_t*)x->k; p.v = (const bf16_t*)x->v; p.kv_bs = x->kv_bs; p.kv_rs = x->kv_rs;
    p.dout = (const bf16_t*)dout; p.dq = (bf16_t*)dq; p.dq_bs = dq_bs; p.dq_rs = dq_rs;
    p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.dkv_bs = dkv_bs; p.dkv_rs = dkv_rs;
    p.B = (int)x->B; p.Sq = (int)x->Sq; p.Sk = (int)x->Sk; p.nh = (int)x->n_heads; p.nkv = (int)x->n_kv_heads; p.hd = (int)x->head_dim;
    p.scale = x->scale; p.mask_mode = x->mask_mode; p.key_valid = x->key_valid; p.img_mask = x->img_mask;
    p.n_img = (int)x->n_img; p.img_len = (int)x->img_len; p.want_dkv = want;
    const size_t npair = ((size_t)x->Sq * x->Sk + 3) & ~(size_t)3;
    const size_t lds = 2 * npair * sizeof(float);
    // ... plus the head's Q, dO, K, V rows where they fit beside P and dS (knob: licv_backward_option 0 = 0 keeps them in global memory)
    const size_t lds_staged = lds + 2 * (size_t)(x->Sq + x->Sk) * (x->head_dim + 8) * 2;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_small_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)attn_bwd_small_k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    if (g_attn_bwd_staged && lds_staged <= 160 * 1024)
        attn_bwd_small_k<true><<<(unsigned)(x->B * x->n_heads), g_attn_bwd_staged == 2 ? 256 : 1024, lds_staged, (hipStream_t)stream>>>(p);
    else
        attn_bwd_small_k<false><<<(unsigned)(x->B * x->n_heads), 256, lds, (hipStream_t)stream>>>(p);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

// ------------------------------------------------------------------------------------------------
// Cross-entropy rows ("hard" loss, ref:icv_src/icv_module.py:94-95,111-117 -> HF ForCausalLMLoss: logits upcast to fp32,
// mean over the kept positions).  One workgroup per row: loss = logsumexp(z) - z[label]; gradient coef * (softmax(z) - onehot).
// ------------------------------------------------------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(256)
void ce_rows_k(const void* __restrict__ logits, const int64_t* __restrict__ rows, const int64_t* __restrict__ labels, int64_t vocab,
               int64_t ld, float* __restrict__ loss_out, float coef, const float* __restrict__ coef_dev, bf16_t* __restrict__ grad, int64_t ld_g,
               const int64_t* __restrict__ grad_rows, int accumulate) {
    __shared__ float red[8];
    if (coef_dev) coef *= *coef_dev;
    const int64_t r = blockIdx.x;
    const int64_t base = rows[r] * ld;
    const int64_t label = labels[r];
    auto ldv = [&](int64_t i) -> float {
        return BF ? bf2f(reinterpret_cast<const bf16_t*>(logits)[base + i]) : reinterpret_cast<const float*>(logits)[base + i];
    };
    float mx = -INFINITY;
    for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) mx = fmaxf(mx, ldv(i));
    mx = blk_reduce(mx, true, red);
    float z = 0.f;
    for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) z += __expf(ldv(i) - mx);
    z = blk_reduce(z, false, red);
    if (loss_out && threadIdx.x == 0) loss_out[r] = logf(z) + mx - ldv(label);
    if (grad) {
        bf16_t* g = grad + (grad_rows ? grad_rows[r] : r) * ld_g;
        const float inv = coef / z;
        for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) {
            float v = __expf(ldv(i) - mx) * inv - (i == label ? coef : 0.f);
            if (accumulate) v += bf2f(g[i]);
            g[i] = f2bf(v);
        }
    }
}

extern "C" int licv_ce_rows(const void* logits, int dtype, const int64_t* rows, const int64_t* labels, int64_t n_rows, int64_t vocab,
                            int64_t ld, float* loss_rows, float grad_coef, const float* grad_coef_dev, void* grad_bf16, int64_t ld_grad,
                            const int64_t* grad_rows, int accumulate, void* stream) {
    LICV_CHECK_ARG(logits && rows && labels && (loss_rows || grad_bf16), "ce_rows: null pointer");
    LICV_CHECK_ARG(dtype == LICV_BF16 || dtype == LICV_F32, "ce_rows: bad dtype");
    LICV_CHECK_ARG(vocab > 0 && ld >= vocab && (!grad_bf16 || ld_grad >= vocab), "ce_rows: bad vocab / leading dims");
    if (n_rows <= 0) return LICV_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LICV_BF16) ce_rows_k<true><<<(unsigned)n_rows, 256, 0, st>>>(logits, rows, labels, vocab, ld, loss_rows, grad_coef, grad_coef_dev, (bf16_t*)grad_bf16, ld_grad, grad_rows, accumulate);
    else                    ce_rows_k<false><<<(unsigned)n_rows, 256, 0, st>>>(logits, rows, labels, vocab, ld, loss_rows, grad_coef, grad_coef_dev, (bf16_t*)grad_bf16, ld_grad, grad_rows, accumulate);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

// out[r, g*hd + d] = bf16( sum_{j < rep} src[r, (g*rep + j)*hd + d] )   (backward of repeat_kv, hf:mistral/modeling_mistral.py)
__global__ __launch_bounds__(256)
void head_group_sum_k(const bf16_t* __restrict__ src, bf16_t* __restrict__ out, int64_t rows, int n_groups, int rep, int hd,
                      int64_t ld_src, int64_t ld_out) {
    const int64_t total = rows * n_groups * hd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % hd);
        const int g = (int)((i / hd) % n_groups);
        const int64_t r = i / ((int64_t)hd * n_groups);
        float acc = 0.f;
        for (int j = 0; j < rep; ++j) acc += bf2f(src[r * ld_src + (int64_t)(g * rep + j) * hd + d]);
        out[r * ld_out + (int64_t)g * hd + d] = f2bf(acc);
    }
}

extern "C" int licv_head_group_sum(const void* src, void* out, int64_t rows, int64_t n_groups, int64_t rep, int64_t head_dim,
                                   int64_t ld_src, int64_t ld_out, void* stream) {
    LICV_CHECK_ARG(src && out, "head_group_sum: null pointer");
    LICV_CHECK_ARG(n_groups > 0 && rep > 0 && head_dim > 0 && ld_src >= n_groups * rep * head_dim && ld_out >= n_groups * head_dim,
                   "head_group_sum: bad shape");
    const int64_t total = rows * n_groups * head_dim;
    if (total <= 0) return LICV_OK;
    int64_t b = (total + 255) / 256; b = b > 8192 ? 8192 : b;
    head_group_sum_k<<<(unsigned)b, 256, 0, (hipStream_t)stream>>>((const bf16_t*)src, (bf16_t*)out, rows, (int)n_groups, (int)rep,
                                                                  (int)head_dim, ld_src, ld_out);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_kl_rows_bwd(const void* stu_logits, const void* tea_logits, int dtype, const int64_t* stu_rows, const int64_t* tea_rows,
                                int64_t n_rows, int64_t vocab, int64_t ld_stu, int64_t ld_tea, float temperature, float eps, float upstream,
                                const float* upstream_dev, void* grad_rows_bf16, int64_t ld_grad, void* stream) {
    LICV_CHECK_ARG(stu_logits && tea_logits && stu_rows && tea_rows && grad_rows_bf16, "kl_rows_bwd: null pointer");
    LICV_CHECK_ARG(dtype == LICV_BF16 || dtype == LICV_F32, "kl_rows_bwd: bad dtype");
    LICV_CHECK_ARG(ld_grad >= vocab, "kl_rows_bwd: ld_grad smaller than vocab");
    if (n_rows <= 0) return LICV_OK;
    const float coef = upstream * temperature * temperature / (float)n_rows;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LICV_BF16) kl_rows_bwd_k<true><<<(unsigned)n_rows, 256, 0, st>>>(stu_logits, tea_logits, stu_rows, tea_rows, vocab, ld_stu, ld_tea, temperature, eps, coef, upstream_dev, (bf16_t*)grad_rows_bf16, ld_grad);
    else                    kl_rows_bwd_k<false><<<(unsigned)n_rows, 256, 0, st>>>(stu_logits, tea_logits, stu_rows, tea_rows, vocab, ld_stu, ld_tea, temperature, eps, coef, upstream_dev, (bf16_t*)grad_rows_bf16, ld_grad);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}
