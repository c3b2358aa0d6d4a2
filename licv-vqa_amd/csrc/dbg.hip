// Timing-only micro-kernels (not part of the product path): how fast can workgroups write 256x256 bf16 tiles?
#include "common.h"
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// each 512-thread block writes one 256x256 tile of a row-major (M x N) matrix, 2 rows (2 x 512 B) per
// wave-instruction — the store pattern of the staged GEMM epilogue.  `lds_bytes` of dynamic LDS only limit
// how many workgroups fit on a CU (160 KiB -> one).  mode 2 uses nontemporal stores.
__global__ __launch_bounds__(512)
void dbg_tile_write_k(bf16_t* C, int64_t ldc, int M, int N, int tiles_n, int mode) {
    extern __shared__ char dbg_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    if (mode == 99) dbg_smem[tid] = 1;                     // keep the LDS allocation referenced
    const u32x4 v = u32x4{(unsigned)tid, 1u, 2u, 3u};
    for (int it = 0; it < 16; ++it) {
        const int row = (it * 8 + wave) * 2 + (lane >> 5);
        const int m = tm * 256 + row, c = tn * 256 + (lane & 31) * 8;
        if (m < M && c + 8 <= N) {
            u32x4* p = reinterpret_cast<u32x4*>(C + (int64_t)m * ldc + c);
            if (mode == 2) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}

extern "C" int licv_dbg_tile_write(void* C, int64_t ldc, int64_t M, int64_t N, int mode, int lds_bytes, void* stream) {
    if (!C || ldc < N || M <= 0 || N <= 0 || lds_bytes < 0 || lds_bytes > 163840) return LICV_E_BADARG;
    const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)((N + 255) / 256);
    (void)hipFuncSetAttribute((const void*)dbg_tile_write_k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    dbg_tile_write_k<<<tiles_m * tiles_n, 512, lds_bytes, (hipStream_t)stream>>>((bf16_t*)C, ldc, (int)M, (int)N, tiles_n, mode);
    return LICV_OK;
}
