// bf16 MFMA GEMM for every nn.Linear of the path:  C[M,N] = epilogue(A[M,K] · W[N,K]^T), fp32 accumulate.
//
// gfx950 design (v1, "tile128"): 128x128x64 tile per 256-thread workgroup (4 waves in 2x2, 64x64 per
// wave = 4x4 accumulators of v_mfma_f32_16x16x32_bf16).  Both operands are K-contiguous, so a lane's
// MFMA fragment (8 consecutive k of one row) is one 16-byte LDS read.  Tiles are staged
// global -> VGPR -> LDS (register staging so ragged M/N/K edges cost nothing: buffer loads return 0
// out of range), double-buffered with ONE barrier per K-tile; the next tile's global loads are issued
// before the MFMAs of the current one and written to LDS after them.  LDS rows are 128 B with the
// 16-byte chunk index XOR-swizzled by (row & 7): conflict-free for the ds_read_b128 lane groups.
// The MFMA is issued with W as the A operand and A as the B operand, so each lane ends up with 4
// CONSECUTIVE output columns of one row: 8-byte bf16 / 16-byte fp32 stores, vector bias/residual loads.
// Workgroup ids are remapped so that the 8 XCDs (private L2s) each own a compact block of the tile grid.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define BM 128
#define BN 128
#define BK 64
#define GEMM_THREADS 256

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
};

__device__ __forceinline__ float act_apply(float y, int act) {
    switch (act) {
        case 1: return 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
        case 2: { const float k0 = 0.7978845608028654f, k1 = 0.044715f;
                  return 0.5f * y * (1.0f + tanhf(k0 * (y + k1 * y * y * y))); }
        case 3: return fmaxf(y, 0.0f);
        default: return y;
    }
}

__device__ __forceinline__ int lds_off(int row, int chunk) {       // bytes within one 128x64 bf16 tile
    return row * 128 + ((chunk ^ (row & 7)) << 4);
}

__global__ __launch_bounds__(GEMM_THREADS, 2)
void gemm_bf16_tile128_k(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw,
                         void* __restrict__ C, int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n, GemmEpi ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [2 stages][A 16 KiB | B 16 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware tile id: blocks b and b+8 share an XCD; give each XCD a contiguous run of tiles, and
    // order tiles in groups of 8 tile-rows (N fastest inside a group) so a run is a compact 2-D patch.
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int gidx = bid / per_group;
    const int first_m = gidx * GROUP;
    const int gsize = min(tiles_m - first_m, GROUP);
    const int tm = first_m + (bid % per_group) % gsize;
    const int tn = (bid % per_group) / gsize;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- buffer descriptors: out-of-range rows read as 0; K tail handled through the offset
    const int64_t a_bytes = (int64_t)(M - m0) * lda * 2, w_bytes = (int64_t)(N - n0) * ldw * 2;
    const int lim = 0x7ffffff0;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (int64_t)m0 * lda), 0, (int)(a_bytes < lim ? a_bytes : lim), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)n0 * ldw), 0, (int)(w_bytes < lim ? w_bytes : lim), 0x00020000);

    // staging map: thread -> (row = tid/8 + 32*i, chunk = tid%8), i = 0..3, for A and for W
    const int srow = tid >> 3, schunk = tid & 7;
    u32x4 ra[4], rw[4];
    auto load_tile = [&](int kt) {
        const int k = kt * BK + schunk * 8;
        const bool kin = k < K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            const unsigned offA = kin ? (unsigned)(((int64_t)row * lda + k) * 2) : 0x80000000u;
            const unsigned offW = kin ? (unsigned)(((int64_t)row * ldw + k) * 2) : 0x80000000u;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, offA, 0, 0);
            rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, offW, 0, 0);
        }
    };
    auto store_tile = [&](int stage) {
        char* sa = smem + stage * 32768;
        char* sw = sa + 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            *reinterpret_cast<u32x4*>(sa + lds_off(row, schunk)) = ra[i];
            *reinterpret_cast<u32x4*>(sw + lds_off(row, schunk)) = rw[i];
        }
    };

    floatx4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nkt = (K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frow = lane & 15, fchunk = lane >> 4;
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
        const char* sa = smem + cur * 32768 + (wm * 64) * 128;
        const char* sw = smem + cur * 32768 + 16384 + (wn * 64) * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[4], fw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *reinterpret_cast<const bf16x8*>(sa + lds_off(i * 16 + frow, kk * 4 + fchunk));
                fw[i] = *reinterpret_cast<const bf16x8*>(sw + lds_off(i * 16 + frow, kk * 4 + fchunk));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nkt) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m = .. + (lane&15)][n = .. + (lane>>4)*4 + r], r = 0..3
    const int rbase = m0 + wm * 64 + (lane & 15);
    const int cq = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = rbase + i * 16;
        if (m >= M) continue;
        const float gate = ep.row_gate ? ep.row_gate[m] : 1.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ep.swiglu && (j & 1)) continue;
            const int ncol = n0 + wn * 64 + j * 16 + cq;       // column in the (packed) N space
            if (ncol >= N) continue;
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r];
                if (ep.bias) v += (ncol + r < N) ? bf2f(ep.bias[ncol + r]) : 0.f;
                v = rbf(v);
                if (ep.act) v = rbf(act_apply(v, ep.act));
                if (ep.swiglu) {
                    const float u = rbf(acc[i][j + 1][r]);
                    const float s = rbf(v / (1.0f + expf(-v)));
                    v = rbf(s * u);
                }
                if (ep.row_gate && gate == 0.0f) v = 0.0f;
                if (ep.use_scale) v = rbf(ep.scale * v);
                y[r] = v;
            }
            // output column: swiglu halves the column space (16-wide gate/up blocks alternate)
            const int oc = ep.swiglu ? ((n0 + wn * 64) >> 1) + (j >> 1) * 16 + cq : ncol;
            const int on = ep.swiglu ? (N >> 1) : N;
            const int nvalid = min(4, on - oc);
            if (ep.residual) {
                if (ep.residual_dtype == LICV_F32) {
                    const float* rp = reinterpret_cast<const float*>(ep.residual) + (int64_t)m * ep.ld_res + oc;
                    if (nvalid == 4) { const floatx4 rv = *reinterpret_cast<const floatx4*>(rp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[r] = rv[r] + y[r]; }
                    else for (int r = 0; r < nvalid; ++r) y[r] = rp[r] + y[r];
                } else {
                    const bf16_t* rp = reinterpret_cast<const bf16_t*>(ep.residual) + (int64_t)m * ep.ld_res + oc;
                    for (int r = 0; r < nvalid; ++r) y[r] = rbf(bf2f(rp[r]) + y[r]);
                }
            }
            if (ep.out_dtype == LICV_F32) {
                float* cp = reinterpret_cast<float*>(C) + (int64_t)m * ldc + oc;
                if (nvalid == 4) *reinterpret_cast<floatx4*>(cp) = floatx4{y[0], y[1], y[2], y[3]};
                else for (int r = 0; r < nvalid; ++r) cp[r] = y[r];
            } else {
                bf16_t* cp = reinterpret_cast<bf16_t*>(C) + (int64_t)m * ldc + oc;
                if (nvalid == 4) {
                    uint2 u;
                    u.x = (uint32_t)f2bf(y[0]) | ((uint32_t)f2bf(y[1]) << 16);
                    u.y = (uint32_t)f2bf(y[2]) | ((uint32_t)f2bf(y[3]) << 16);
                    *reinterpret_cast<uint2*>(cp) = u;
                } else for (int r = 0; r < nvalid; ++r) cp[r] = f2bf(y[r]);
            }
        }
    }
}

// gate/up rows interleaved in blocks of 16: packed[32b + i] = gate[16b + i], packed[32b + 16 + i] = up[16b + i]
__global__ __launch_bounds__(256)
void pack_gate_up_k(const bf16_t* __restrict__ g, const bf16_t* __restrict__ u, bf16_t* __restrict__ out, int64_t inter, int64_t K) {
    const int64_t vec = K >> 3, total = 2 * inter * vec;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t prow = idx / vec, c = idx % vec;
        const int64_t blk = prow >> 5, within = prow & 31;
        const bf16_t* src = (within < 16 ? g : u) + (blk * 16 + (within & 15)) * K;
        reinterpret_cast<uint4*>(out + prow * K)[c] = reinterpret_cast<const uint4*>(src)[c];
    }
}

extern "C" int licv_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                              int64_t M, int64_t N, int64_t K, const licv_gemm_epilogue* e, void* stream) {
    LICV_CHECK_ARG(A && W && C && e, "gemm_bf16: null pointer");
    LICV_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm_bf16: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    LICV_CHECK_ARG(K % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0, "gemm_bf16: K, lda, ldw must be multiples of 8 (K=%lld lda=%lld ldw=%lld)",
                   (long long)K, (long long)lda, (long long)ldw);
    LICV_CHECK_ARG(lda >= K && ldw >= K, "gemm_bf16: leading dimension smaller than K");
    LICV_CHECK_ARG(ldc % 4 == 0, "gemm_bf16: ldc (%lld) must be a multiple of 4", (long long)ldc);
    LICV_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0, "gemm_bf16: pointers must be 16-byte aligned");
    LICV_CHECK_ARG(e->out_dtype == LICV_BF16 || e->out_dtype == LICV_F32, "gemm_bf16: bad out dtype");
    LICV_CHECK_ARG(e->act >= 0 && e->act <= 3, "gemm_bf16: bad activation %d", e->act);
    LICV_CHECK_ARG(!e->swiglu || (N % 32 == 0 && !e->bias_bf16 && !e->act), "gemm_bf16: swiglu needs N %% 32 == 0, no bias/act");
    LICV_CHECK_ARG(!e->residual || (e->ld_res % 4 == 0 && ((uintptr_t)e->residual & 15) == 0), "gemm_bf16: residual misaligned");
    LICV_CHECK_ARG(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "gemm_bf16: dimension too large");
    if (M == 0) return LICV_OK;
    GemmEpi ep;
    ep.bias = (const bf16_t*)e->bias_bf16; ep.row_gate = e->row_gate; ep.residual = e->residual;
    ep.residual_dtype = e->residual_dtype; ep.ld_res = e->ld_res; ep.act = e->act; ep.swiglu = e->swiglu;
    ep.use_scale = e->use_scale; ep.scale = e->scale; ep.out_dtype = e->out_dtype;
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
    const size_t lds = 2 * 32768;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_bf16_tile128_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    gemm_bf16_tile128_k<<<dim3(tiles_m * tiles_n), dim3(GEMM_THREADS), lds, (hipStream_t)stream>>>(
        (const bf16_t*)A, lda, (const bf16_t*)W, ldw, C, ldc, (int)M, (int)N, (int)K, tiles_m, tiles_n, ep);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_pack_gate_up(const void* gate, const void* up, void* packed, int64_t inter, int64_t K, void* stream) {
    LICV_CHECK_ARG(gate && up && packed, "pack_gate_up: null pointer");
    LICV_CHECK_ARG(inter % 16 == 0 && K % 8 == 0, "pack_gate_up: inter must be a multiple of 16 and K of 8");
    const int64_t total = 2 * inter * (K / 8);
    int64_t b = (total + 255) / 256; b = b > 4096 ? 4096 : b;
    pack_gate_up_k<<<(int)b, 256, 0, (hipStream_t)stream>>>((const bf16_t*)gate, (const bf16_t*)up, (bf16_t*)packed, inter, K);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}
