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

// ------------------------------------------------------------------------------------------------
// Weight-stream probe: what HBM rate does a given per-instruction ACCESS SHAPE reach when 4-wave workgroups stream a cold [N, K]
// bf16 matrix the way the weight-streaming GEMMs do (each wave owns 16 rows x a K range, `UN` 16-byte loads per lane in flight in
// each of two register sets, nothing but an XOR done with the data)?
//   shape 0: 16 rows x 64 B per instruction (MFMA fragment order, what gemm_bf16_skinny_k issues)
//   shape 1:  8 rows x 128 B   shape 2: 2 rows x 512 B   shape 3: 1 row x 1 KB
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_p;

template <int SHAPE, int UN>
__global__ __launch_bounds__(256)
void weight_stream_probe_k(const bf16_t* __restrict__ W, int64_t ldw, int N, int K, int kr, unsigned* __restrict__ sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 64 + wave * 16;
    const int k0 = blockIdx.y * kr;
    // instruction i of this wave covers IB bytes of each of IR rows; the wave's 16 rows x kr*2 bytes take 16*kr*2/1024 instructions
    constexpr int IR = SHAPE == 0 ? 16 : SHAPE == 1 ? 8 : SHAPE == 2 ? 2 : 1;
    constexpr int LPR = 64 / IR;                                 // lanes per row
    const int r_in = lane / LPR, c_in = lane % LPR;
    const int row_groups = 16 / IR;                              // instructions needed to cover the 16 rows at one K position
    const int n_instr = 16 * kr * 2 / 1024;
    u32x4_p acc = u32x4_p{0u, 0u, 0u, 0u};
    u32x4_p buf[2][UN];
    auto issue = [&](int i0, u32x4_p (&dst)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = min(i0 + u, n_instr - 1);
            const int kpos = i / row_groups, rg = i % row_groups;        // walk the rows first, then along K
            const int row = min(n0 + rg * IR + r_in, N - 1);
            const int k = k0 + kpos * (LPR * 8) + c_in * 8;
            dst[u] = *reinterpret_cast<const u32x4_p*>(W + (int64_t)row * ldw + min(k, K - 8));
        }
    };
    issue(0, buf[0]);
    for (int i = 0; i < n_instr; i += 2 * UN) {
        issue(i + UN, buf[1]);
#pragma unroll
        for (int u = 0; u < UN; ++u) acc ^= buf[0][u];
        issue(i + 2 * UN, buf[0]);
#pragma unroll
        for (int u = 0; u < UN; ++u) acc ^= buf[1][u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u) sink[0] = acc.x;   // keeps the loads alive
}

extern "C" int licv_probe_weight_stream(const void* W, int64_t ldw, int64_t N, int64_t K, int splits, int shape, int depth, void* sink, void* stream) {
    LICV_CHECK_ARG(W && sink && N % 64 == 0 && splits > 0 && K % (splits * 512) == 0, "probe_weight_stream: N %% 64, K %% (splits * 512) must be 0");
    const dim3 grid((unsigned)(N / 64), (unsigned)splits);
    const int kr = (int)(K / splits);
    hipStream_t st = (hipStream_t)stream;
#define WS_LAUNCH(S, U) weight_stream_probe_k<S, U><<<grid, 256, 0, st>>>((const bf16_t*)W, ldw, (int)N, (int)K, kr, (unsigned*)sink)
    if (depth <= 4)       { if (shape == 0) WS_LAUNCH(0, 4); else if (shape == 1) WS_LAUNCH(1, 4); else if (shape == 2) WS_LAUNCH(2, 4); else WS_LAUNCH(3, 4); }
    else if (depth <= 8)  { if (shape == 0) WS_LAUNCH(0, 8); else if (shape == 1) WS_LAUNCH(1, 8); else if (shape == 2) WS_LAUNCH(2, 8); else WS_LAUNCH(3, 8); }
    else                  { if (shape == 0) WS_LAUNCH(0, 16); else if (shape == 1) WS_LAUNCH(1, 16); else if (shape == 2) WS_LAUNCH(2, 16); else WS_LAUNCH(3, 16); }
#undef WS_LAUNCH
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

// ------------------------------------------------------------------------------------------------
// The same stream through LDS-DMA (`buffer_load_dwordx4 ... lds`, 8 rows x 128 B per instruction = the GEMM kernels' pieces): how
// many pieces per wave does the hardware keep in flight?  Each wave owns 16 rows x a K range and keeps DEPTH pieces outstanding
// (issue one, `s_waitcnt vmcnt(DEPTH - 1)`); the LDS destination cycles through the wave's 32 KiB (nothing reads it).
// ------------------------------------------------------------------------------------------------
template <int DEPTH>
__global__ __launch_bounds__(256)
void lds_dma_probe_k(const bf16_t* __restrict__ W, int64_t ldw, int N, int K, int kr) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 64 + wave * 16;
    const int k0 = blockIdx.y * kr;
    const int n_instr = 16 * kr * 2 / 1024;                              // 8 rows x 128 B each: two per 64-element K step of the 16 rows
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)n0 * ldw + k0), 0, 0xFFFFFFFF, 0x00020000);
    const int lds_base = __builtin_amdgcn_readfirstlane((int)(uintptr_t)(__attribute__((address_space(3))) char*)dsm) + wave * 32768;
    const int row_off = (lane >> 3) * (int)ldw * 2 + (lane & 7) * 16;
    for (int i = 0; i < n_instr; ++i) {
        const int kpos = i >> 1, half = i & 1;
        const int voff = row_off + half * 8 * (int)ldw * 2 + kpos * 128;
        const int m0 = lds_base + (i & 31) * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(%3)"
                     :: "s"(m0), "v"(voff), "s"(rs), "i"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

extern "C" int licv_probe_lds_dma_stream(const void* W, int64_t ldw, int64_t N, int64_t K, int splits, int depth, void* stream) {
    LICV_CHECK_ARG(W && N % 64 == 0 && splits > 0 && K % (splits * 512) == 0, "probe_lds_dma_stream: N %% 64, K %% (splits * 512) must be 0");
    LICV_CHECK_ARG((int64_t)16 * ldw * 2 + K * 2 < (1ll << 31), "probe_lds_dma_stream: offsets exceed 32 bits");
    const dim3 grid((unsigned)(N / 64), (unsigned)splits);
    const int kr = (int)(K / splits);
    hipStream_t st = (hipStream_t)stream;
#define DMA_LAUNCH(D) do { (void)hipFuncSetAttribute((const void*)lds_dma_probe_k<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); \
        lds_dma_probe_k<D><<<grid, 256, 131072, st>>>((const bf16_t*)W, ldw, (int)N, (int)K, kr); } while (0)
    if (depth <= 4) DMA_LAUNCH(4); else if (depth <= 8) DMA_LAUNCH(8); else if (depth <= 16) DMA_LAUNCH(16); else if (depth <= 32) DMA_LAUNCH(32); else DMA_LAUNCH(48);
#undef DMA_LAUNCH
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

// ------------------------------------------------------------------------------------------------
// How many bytes per second can ONE CU take in from an L2 / Infinity-Cache resident buffer, by path?  Every workgroup (256 threads, one
// per CU when the grid is 256) reads the SAME `bytes` of `buf` `reps` times: mode 0 = 16-byte buffer loads to VGPRs (all four waves, 8 in
// flight per lane), mode 1 = LDS-DMA pieces of 8 rows x 128 B (all four waves, 16 in flight per wave), mode 2 = waves 0-1 by LDS-DMA
// and waves 2-3 to VGPRs at once (each pair reads the whole buffer: twice the bytes of modes 0 / 1 per workgroup).  The question behind
// it: does the M = 256 GEMM gain per-CU operand bandwidth if W bypasses LDS while A stays on LDS-DMA?
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_ip;
__global__ __launch_bounds__(256)
void l2_ingest_probe_k(const char* __restrict__ buf, int64_t bytes, int reps, int mode, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char isx[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (int)bytes, 0x00020000);
    const bool dma = mode == 1 || (mode == 2 && wave < 2);
    const int nw = mode == 2 ? 2 : 4, w = mode == 2 ? (wave & 1) : wave;     // waves sharing one pass over the buffer, my index among them
    u32x4_ip acc = {0u, 0u, 0u, 0u};
    if (dma) {
        const int lds_base = __builtin_amdgcn_readfirstlane((int)(uintptr_t)(__attribute__((address_space(3))) char*)isx) + wave * 16384;
        const int n_piece = (int)(bytes / 1024);                              // 1 KiB per instruction
        for (int r = 0; r < reps; ++r)
            for (int i = w; i < n_piece; i += nw) {
                const int voff = i * 1024 + lane * 16;
                const int m0 = lds_base + ((i / nw) & 15) * 1024;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(15)" :: "s"(m0), "v"(voff), "s"(rs) : "memory");
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        const int n16 = (int)(bytes / 16);
        for (int r = 0; r < reps; ++r)
            for (int i0 = w * 64 + lane; i0 < n16; i0 += nw * 64 * 8) {
                u32x4_ip v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + u * nw * 64; v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)(i < n16 ? i : i0) * 16u, 0, 0); }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc ^= v[u];
            }
    }
    if (sink && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u) *sink = 1u;
}

extern "C" int licv_probe_l2_ingest(const void* buf, int64_t bytes, int reps, int mode, int blocks, void* sink_u32, void* stream) {
    LICV_CHECK_ARG(buf && bytes >= 65536 && bytes % 65536 == 0 && bytes < (1ll << 31) && reps > 0 && mode >= 0 && mode <= 2 && blocks > 0, "probe_l2_ingest: bad arguments");
    (void)hipFuncSetAttribute((const void*)l2_ingest_probe_k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    l2_ingest_probe_k<<<blocks, 256, 65536, (hipStream_t)stream>>>((const char*)buf, bytes, reps, mode, (unsigned*)sink_u32);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}
