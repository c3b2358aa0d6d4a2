// Achievable-peak probes for the roofline denominators (SURVEY.md §8d asks for a build-owned MFMA-loop and stream-copy figure next
// to the vendor peaks).  mfma_loop_k: every wave issues independent v_mfma_f32_16x16x32_bf16 back to back from registers only.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_p;

__global__ __launch_bounds__(256)
void mfma_loop_k(float* __restrict__ sink, int iters) {
    bf16x8_p a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    floatx4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        // written as asm so that each accumulator stays in its own registers (the builtin form was compiled into a rotating
        // accumulator chain with copies, which measures the copies)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) sink[0] = s;                   // keeps the loop alive, never true in practice
}

// launches `blocks` workgroups of 4 waves running iters x 8 MFMAs each; FLOP = blocks * 4 * iters * 8 * 16384
extern "C" int licv_probe_mfma_loop(void* sink, int blocks, int iters, void* stream) {
    LICV_CHECK_ARG(sink && blocks > 0 && iters > 0, "probe_mfma_loop: bad arguments");
    mfma_loop_k<<<blocks, 256, 0, (hipStream_t)stream>>>((float*)sink, iters);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

// semantics probe for v_permlane16_swap as the flow GEMM's epilogue uses it: out[lane] = {a', b'} for a = lane, b = 100 + lane
__global__ void permlane16_swap_probe_k(uint32_t* __restrict__ out) {
    const uint32_t a = threadIdx.x, b = 100 + threadIdx.x;
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[2 * threadIdx.x] = r[0];
    out[2 * threadIdx.x + 1] = r[1];
}
extern "C" int licv_probe_permlane16_swap(void* out_u32_128, void* stream) {
    LICV_CHECK_ARG(out_u32_128, "probe_permlane16_swap: null pointer");
    permlane16_swap_probe_k<<<1, 64, 0, (hipStream_t)stream>>>((uint32_t*)out_u32_128);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}
