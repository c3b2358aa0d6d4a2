// Distillation loss rows and the fused AdamW of the L-ICV trainer.
//   KL: ref:icv_src/icv_module.py:121-134  (softmax in the logits' dtype, eps INSIDE the log, * T^2 by the caller)
//   AdamW: ref:icv_src/icv_module.py:171-209 (torch.optim.AdamW; two lr groups: alpha | icv)
#include "common.h"

template <bool BF>
__device__ __forceinline__ float rnd(float x) { return BF ? rbf(x) : x; }

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
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
void kl_rows_fwd_k(const void* __restrict__ stu, const void* __restrict__ tea, const int64_t* __restrict__ srows,
                   const int64_t* __restrict__ trows, int64_t vocab, int64_t ld_s, int64_t ld_t, float T, float eps,
                   float* __restrict__ out) {
    __shared__ float red[8];
    const int64_t row = blockIdx.x;
    const int64_t sb = srows[row] * ld_s, tb = trows[row] * ld_t;
    auto ld = [&](const void* p, int64_t i) -> float {
        return BF ? bf2f(reinterpret_cast<const bf16_t*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
    };
    float ms = -INFINITY, mt = -INFINITY;
    for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) {
        ms = fmaxf(ms, rnd<BF>(ld(stu, sb + i) / T));
        mt = fmaxf(mt, rnd<BF>(ld(tea, tb + i) / T));
    }
    ms = block_reduce(ms, true, red);
    mt = block_reduce(mt, true, red);
    float zs = 0.f, zt = 0.f;
    for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) {
        zs += expf(rnd<BF>(ld(stu, sb + i) / T) - ms);
        zt += expf(rnd<BF>(ld(tea, tb + i) / T) - mt);
    }
    zs = block_reduce(zs, false, red);
    zt = block_reduce(zt, false, red);
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) {
        const float q = rnd<BF>(expf(rnd<BF>(ld(stu, sb + i) / T) - ms) / zs);
        const float p = rnd<BF>(expf(rnd<BF>(ld(tea, tb + i) / T) - mt) / zt);
        const float lp = rnd<BF>(logf(rnd<BF>(p + eps)));
        const float lq = rnd<BF>(logf(rnd<BF>(q + eps)));
        acc += rnd<BF>(p * rnd<BF>(lp - lq));
    }
    acc = block_reduce(acc, false, red);
    if (threadIdx.x == 0) out[row] = rnd<BF>(acc);
}

// d/dT of one row's  f = sum_v p (log(p + eps) - log(q + eps)),  p = softmax(tea / T), q = softmax(stu / T)
// (ref:icv_src/icv_module.py:49-52: `temperature` is an nn.Parameter with requires_grad = learnable_t; :121-134).  With
// t = tea / T, s = stu / T:  dp_v/dT = -(p_v / T)(t_v - E_p[t]),  dq_v/dT = -(q_v / T)(s_v - E_q[s]),  so
//   df/dT = sum_v { -(p_v / T)(t_v - E_p[t]) [log(p_v + eps) - log(q_v + eps) + p_v / (p_v + eps)] + p_v (q_v / T)(s_v - E_q[s]) / (q_v + eps) }.
// fp32 throughout on the (bf16-rounded, as the forward reads them) logits: a derivative, no torch rounding points to mirror.
template <bool BF>
__global__ __launch_bounds__(256)
void kl_rows_dtemp_k(const void* __restrict__ stu, const void* __restrict__ tea, const int64_t* __restrict__ srows,
                     const int64_t* __restrict__ trows, int64_t vocab, int64_t ld_s, int64_t ld_t, float T, float eps,
                     float* __restrict__ out) {
    __shared__ float red[8];
    const int64_t row = blockIdx.x;
    const int64_t sb = srows[row] * ld_s, tb = trows[row] * ld_t;
    auto ld = [&](const void* p, int64_t i) -> float {
        return BF ? bf2f(reinterpret_cast<const bf16_t*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
    };
    float ms = -INFINITY, mt = -INFINITY;
    for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) {
        ms = fmaxf(ms, rnd<BF>(ld(stu, sb + i) / T));
        mt = fmaxf(mt, rnd<BF>(ld(tea, tb + i) / T));
    }
    ms = block_reduce(ms, true, red);
    mt = block_reduce(mt, true, red);
    float zs = 0.f, zt = 0.f, es = 0.f, et = 0.f;
    for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) {
        const float s = rnd<BF>(ld(stu, sb + i) / T), t = rnd<BF>(ld(tea, tb + i) / T);
        const float a = expf(s - ms), b = expf(t - mt);
        zs += a; zt += b; es += a * s; et += b * t;
    }
    zs = block_reduce(zs, false, red);
    zt = block_reduce(zt, false, red);
    es = block_reduce(es, false, red) / zs;              // E_q[s]
    et = block_reduce(et, false, red) / zt;              // E_p[t]
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) {
        const float s = rnd<BF>(ld(stu, sb + i) / T), t = rnd<BF>(ld(tea, tb + i) / T);
        const float q = expf(s - ms) / zs, p = expf(t - mt) / zt;
        const float dp = -(p / T) * (t - et), dq = -(q / T) * (s - es);
        acc += dp * (logf(p + eps) - logf(q + eps) + p / (p + eps)) - p * dq / (q + eps);
    }
    acc = block_reduce(acc, false, red);
    if (threadIdx.x == 0) out[row] = acc;
}

__global__ __launch_bounds__(256)
void adamw_step_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                  int64_t n, int64_t n0, float lr0, float lr1, float b1, float b2, float eps, float wd,
                  float bc1, float bc2_sqrt, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float lr = i < n0 ? lr0 : lr1;
        const float gi = g[i] * gscale;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = m[i] * b1 + (1.0f - b1) * gi;
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

extern "C" int licv_kl_rows_fwd(const void* stu, const void* tea, int dtype, const int64_t* stu_rows, const int64_t* tea_rows,
                                int64_t n_rows, int64_t vocab, int64_t ld_stu, int64_t ld_tea, float temperature, float eps,
                                float* out_rows, void* stream) {
    LICV_CHECK_ARG(stu && tea && stu_rows && tea_rows && out_rows, "kl_rows_fwd: null pointer");
    LICV_CHECK_ARG(dtype == LICV_BF16 || dtype == LICV_F32, "kl_rows_fwd: bad dtype %d", dtype);
    LICV_CHECK_ARG(vocab > 0 && temperature > 0.f, "kl_rows_fwd: bad vocab/temperature");
    if (n_rows <= 0) return LICV_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LICV_BF16) kl_rows_fwd_k<true><<<(unsigned)n_rows, 256, 0, st>>>(stu, tea, stu_rows, tea_rows, vocab, ld_stu, ld_tea, temperature, eps, out_rows);
    else                    kl_rows_fwd_k<false><<<(unsigned)n_rows, 256, 0, st>>>(stu, tea, stu_rows, tea_rows, vocab, ld_stu, ld_tea, temperature, eps, out_rows);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_kl_rows_dtemp(const void* stu, const void* tea, int dtype, const int64_t* stu_rows, const int64_t* tea_rows,
                                  int64_t n_rows, int64_t vocab, int64_t ld_stu, int64_t ld_tea, float temperature, float eps,
                                  float* out_rows, void* stream) {
    LICV_CHECK_ARG(stu && tea && stu_rows && tea_rows && out_rows, "kl_rows_dtemp: null pointer");
    LICV_CHECK_ARG(dtype == LICV_BF16 || dtype == LICV_F32, "kl_rows_dtemp: bad dtype %d", dtype);
    LICV_CHECK_ARG(vocab > 0 && temperature > 0.f, "kl_rows_dtemp: bad vocab/temperature");
    if (n_rows <= 0) return LICV_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LICV_BF16) kl_rows_dtemp_k<true><<<(unsigned)n_rows, 256, 0, st>>>(stu, tea, stu_rows, tea_rows, vocab, ld_stu, ld_tea, temperature, eps, out_rows);
    else                    kl_rows_dtemp_k<false><<<(unsigned)n_rows, 256, 0, st>>>(stu, tea, stu_rows, tea_rows, vocab, ld_stu, ld_tea, temperature, eps, out_rows);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_group0, float lr0, float lr1,
                               float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
    LICV_CHECK_ARG(p && g && m && v, "adamw_step: null pointer");
    LICV_CHECK_ARG(step >= 1 && n_group0 >= 0 && n_group0 <= n, "adamw_step: bad step/group split");
    if (n <= 0) return LICV_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    int64_t b = (n + 255) / 256; b = b > 1024 ? 1024 : b;
    adamw_step_k<<<(int)b, 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, n_group0, lr0, lr1, beta1, beta2, eps, weight_decay,
                                                          (float)bc1, (float)sqrt(bc2), grad_scale);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}
