// bf16 MFMA GEMM for every nn.Linear of the path:  C[M,N] = epilogue(A[M,K] · W[N,K]^T), fp32 accumulate.
//
// Two gfx950 kernels share one epilogue:
//
//  * gemm_bf16_tile256_k — the throughput kernel (M >= 512, K % 64 == 0).  256x256x64 tile per 512-thread
//    workgroup (8 waves as 2(M) x 4(N), 128x64 per wave = 8x4 accumulators of v_mfma_f32_16x16x32_bf16),
//    one workgroup per CU.  Tiles are staged global -> LDS directly by LDS-DMA (global_load_lds_dwordx4:
//    no VGPR round trip, no ds_write), two 64 KiB stages; the DMA for K-tile t+1 is issued right after the
//    single barrier of K-tile t and lands under that tile's 64 MFMAs per wave.  The LDS image is the
//    DMA's lane-linear order; the 16-byte chunk XOR swizzle (chunk ^ (row & 7)) is applied on the per-lane
//    SOURCE address and again on the ds_read_b128 address (same involution), so fragment reads are
//    bank-conflict free.  Ragged M/N edges clamp the source row (those outputs are never stored).
//  * gemm_bf16_tile128_k — the general kernel (any M, N; K % 8 == 0): 128x128x64 tile, 4 waves, register
//    staged with buffer loads that return 0 out of range, same swizzle, one barrier per K-tile.
//
// Both issue the MFMA with W as the A operand and A as the B operand, so each lane ends up with 4
// CONSECUTIVE output columns of one row: 8-byte bf16 / 16-byte fp32 stores, vector bias/residual loads.
// Workgroup ids are remapped so that the 8 XCDs (private L2s) each own a compact patch of the tile grid.
#include <mutex>
#include "gemm_common.h"

// ------------------------------------------------------------------------------------------------
// "lean" ping-pong kernel: the ping-pong schedule, ring and K order unchanged (bit-identical results), with the per-stage
// overhead of the LOAD phase taken out.  The stamped timeline of the kernel above (tools/gemm_segments.py) shows where a stage
// goes: the interval in which the trailing group loads is 870 cycles against 700 for the other one — a wave that begins its
// load phase just as its SIMD partner begins an MFMA burst loses ~140 cycles before its first LDS read issues, and that head
// was vector-ALU address arithmetic (two 32-bit adds per fragment base, a 64-bit add per DMA piece) plus a chain of scalar
// branches for the counted wait.  Here
//   * a DMA piece is `global_load_lds  v_offset, s[base:base+1]`: the lane part (row * ld + chunk, loop-invariant, 32-bit) stays
//     in a VGPR and the K advance is scalar — no vector ALU per piece;
//   * the fragment-read base of stage s+1 is formed at the END of stage s's compute phase (inside the wave's own priority
//     window), so a load phase starts with its ds_reads;
//   * the steady-state loop (s + 4 < ns) has no data-dependent branch: issue 4 pieces, `s_waitcnt vmcnt(12)`; only the last
//     four stages run the variable wait.
// SPLITK = 1 is the split-K producer (blockIdx.y selects `group` stages; C is the fp32 workspace, [split][M_pad][N_pad]).
// ------------------------------------------------------------------------------------------------
template <int SPLITK, int VAR = 0>     // VAR: load-phase order experiments (1: counted wait before the DMA issue, 2: DMA before the reads)
__global__ __launch_bounds__(512, 2)
void gemm_bf16_lean_k(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw,
                      void* __restrict__ C, int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n, GemmEpi ep, int group) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [5 stages][A 16 KiB | W 16 KiB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                         // waves 0-3 lead, 4-7 trail by half a stage
    // VAR 8 (diagnostic build): lane 0 of wave 0 records the 100 MHz wall clock at the tile's phase boundaries (slots 0-4, as the
    // ping-pong kernel does) and the shader clock around the main loop (slots 5, 6) into g_dbg_ts[block * 8 ...]
    long long* ts = (VAR == 8 && g_dbg_ts && tid == 0) ? g_dbg_ts + (int64_t)blockIdx.x * 8 : nullptr;
    if (ts) ts[0] = wall_clock64();
    int tm, tn;
    tile_coords(blockIdx.x, tiles_m, tiles_n, tm, tn, SPLITK ? 8 : group);
    const int m0 = tm * 256, n0 = tn * 256;
    const int kbase = SPLITK ? (int)blockIdx.y * group : 0;
    const int ns = SPLITK ? min(group, K / 32 - kbase) : K / 32;     // >= 4, host-guaranteed

    // DMA sources: wave-uniform corner of the tile (scalar registers) + a per-lane byte offset that never changes
    uint32_t offA[2], offW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 32 + i * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ (((row >> 2) & 1) << 1);
        offA[i] = (uint32_t)(min(row, M - 1 - m0) * (int)lda * 2 + chunk * 16);      // rows past M / N re-read the last valid row
        offW[i] = (uint32_t)(min(row, N - 1 - n0) * (int)ldw * 2 + chunk * 16);
    }
    const char* gA = reinterpret_cast<const char*>(A + (int64_t)m0 * lda + (int64_t)kbase * 32);
    const char* gW = reinterpret_cast<const char*>(W + (int64_t)n0 * ldw + (int64_t)kbase * 32);
    auto issue = [&](int s, int slot_bytes) {
        char* sa = smem + slot_bytes + wave * 2048;
        const char* a = gA + (int64_t)s * 64;
        const char* w = gW + (int64_t)s * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a + offA[i]),
                                             (__attribute__((address_space(3))) void*)(sa + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + offW[i]),
                                             (__attribute__((address_space(3))) void*)(sa + 16384 + i * 1024), 16, 0, 0);
        }
    };

    floatx4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int fo = ring_off(lane & 15, lane >> 4);
    const int constA = wm * 8192 + fo, constW = 16384 + wn * 4096 + fo;
    bf16x8 fa[8], fw[4];

    issue(0, 0); issue(1, RING_STAGE_BYTES); issue(2, 2 * RING_STAGE_BYTES); issue(3, 3 * RING_STAGE_BYTES);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");        // my pieces of stage 0 have landed
    __builtin_amdgcn_s_barrier();                            // stage 0 published
    if (ts) { ts[1] = wall_clock64(); ts[5] = (long long)__builtin_amdgcn_s_memtime(); }
    if (wm == 1) __builtin_amdgcn_s_barrier();               // trailing group starts half a stage later

    int slot_rd = 0, slot_wr = 4 * RING_STAGE_BYTES;         // ring slots (byte offsets) of stage s and of stage s + 4
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    typedef const __attribute__((address_space(3))) bf16x8* lds_fptr;
    const lds_cptr ring = (lds_cptr)smem;
    lds_cptr rdA = ring + constA, rdW = ring + constW;       // fragment-read bases of the stage about to be read
    // VAR 9 (diagnostic build, results unaffected): s_memtime stamps around the segments of ONE mid-loop stage, as in the kernel above
    unsigned long long stamp[8];
    const int probe = (VAR == 9 && g_dbg_ts) ? ns / 2 : -1;
#define STAMP(i) do { if (VAR == 9 && s == probe) { __builtin_amdgcn_sched_barrier(0); stamp[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
    // VAR 7 (diagnostic build): a time series — lane l of every wave keeps the shader clock at the start of stage l * series_stride
    // (one v_cndmask per stamp, no memory traffic inside the loop); written to g_dbg_ts[(block * 8 + wave) * 64 + l] after the loop
    uint32_t series = 0;
    const int series_stride = (VAR == 7 && g_dbg_ts) ? (ns + 63) / 64 : 0;
    int series_next = 0, series_lane = 0;
    auto stage = [&](int s, auto steady_c) {
        constexpr bool STEADY = decltype(steady_c)::value;
        if (VAR == 7 && series_stride && s == series_next) {
            { const uint32_t now = (uint32_t)__builtin_amdgcn_s_memtime(); series = lane == series_lane ? now : series; }
            series_next += series_stride; ++series_lane;
        }
        STAMP(0);
        // ---- LOAD phase (partner computes)
        {
            if (VAR == 2 && STEADY) { issue(s + 4, slot_wr); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int j = 0; j < 4; ++j) fw[j] = *(lds_fptr)(rdW + j * 16 * 64);
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = *(lds_fptr)(rdA + i * 16 * 64);
            __builtin_amdgcn_sched_barrier(0);
            STAMP(1);
            if (VAR == 9 && STEADY) {
                issue(s + 4, slot_wr);
                STAMP(2);
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                STAMP(3);
            } else if (STEADY && VAR == 1) {
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                issue(s + 4, slot_wr);
            } else if (STEADY) {
                if (VAR != 2) issue(s + 4, slot_wr);
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");    // retires my pieces of stage s+1; s+2 .. s+4 stay in flight
            } else {
                wait_vmcnt(4 * max(0, min(3, ns - 2 - s)));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            STAMP(4);
        }
        __builtin_amdgcn_s_barrier();
        STAMP(5);
        // ---- COMPUTE phase (partner loads)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        slot_wr = slot_rd;                                           // the slot just consumed is the next one refilled
        slot_rd = slot_rd == 4 * RING_STAGE_BYTES ? 0 : slot_rd + RING_STAGE_BYTES;
        rdA = ring + (constA + slot_rd);
        rdW = ring + (constW + slot_rd);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        STAMP(6);
        __builtin_amdgcn_s_barrier();
        STAMP(7);
    };
    int s = 0;
    for (; s + 4 < ns; ++s) stage(s, std::true_type{});
    for (; s < ns; ++s) stage(s, std::false_type{});
#undef STAMP
    if (VAR == 7 && series_stride) {
        { const uint32_t now = (uint32_t)__builtin_amdgcn_s_memtime(); series = lane == series_lane ? now : series; }     // end of the loop
        g_dbg_ts[((int64_t)blockIdx.x * 8 + wave) * 64 + lane] = lane <= series_lane ? (long long)series : -1;
    }
    if (VAR == 9 && probe >= 0 && lane == 0) {
        long long* o = g_dbg_ts + ((int64_t)blockIdx.x * 8 + wave) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (long long)stamp[i];
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();               // leading group: match the barrier count
    if (ts) { ts[2] = wall_clock64(); ts[6] = (long long)__builtin_amdgcn_s_memtime(); }
    if (SPLITK) {                                            // fp32 partial tile -> this split's workspace slice
        const int64_t np = (int64_t)tiles_n * 256;
        float* slice = reinterpret_cast<float*>(C) + (int64_t)blockIdx.y * ((int64_t)tiles_m * 256) * np;
        const int rl = m0 + wm * 128 + (lane & 15), c0 = n0 + wn * 64 + (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<floatx4*>(slice + (int64_t)(rl + i * 16) * np + c0 + j * 16) = acc[i][j];
        return;
    }
    epilogue_staged<256, 256, 8, 8, 4>(acc, ep, C, ldc, M, N, m0, n0, wm * 128, wn * 64, wave, lane, smem, ts);
    if (ts) ts[4] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// "quad64" kernel: the quad kernel on 64-deep K tiles, so that an LDS-DMA piece is 8 rows x 128 B — whole cache lines — instead
// of 16 rows x 64 B (measured on the quad kernel with a timing-only build: whole-line pieces are worth +10-14 %; the texture
// path handles a 64-lane request line by line).
//   * LDS = a ring of ten 16 KiB UNITS, a unit = 128 rows x 64 K of one operand (W rows 0-127, W rows 128-255, A rows 0-127,
//     A rows 128-255 of a K tile, in that order): unit h = 4 t + c lives in slot h mod 10.  Wave (wm, wn) reads exactly two
//     units per tile: A half wm and W half wn.  128-byte rows, 16-byte chunk c of row r at position c ^ (r & 7) (the swizzle is
//     applied to the DMA's per-lane SOURCE chunk; the destination is lane-linear).
//   * a K tile is two 32-deep steps of 64 MFMAs (same K order as every other kernel: bit-identical results).  Step (t, 0):
//     16 fragment reads of (t, second half) and the 8 pieces of W units of tile t + 2.  Step (t, 1): after 8 MFMAs the wave
//     retires its pieces of tile t + 1 (counted vmcnt(8): the W units of t + 2 stay in flight) and the workgroup meets — the ONE
//     barrier per 128 MFMAs: tile t + 1 is published and, since every wave passed its lgkmcnt(0) for the last reads of tile t,
//     tile t's four slots are free; then 16 fragment reads of (t + 1, first half), then the 8 pieces of the A units of tile
//     t + 2 into two of the freed slots (the other two take the W units of tile t + 3 one step later).
//   * a piece is issued at least two steps (~2 x 1024 MFMA cycles) before the barrier that needs it.
// ------------------------------------------------------------------------------------------------
#define Q64_UNIT 16384
template <int VAR>      // 0 production; timing-only builds: 1 no DMA inside the loop, 2 no fragment reads inside the loop
__global__ __launch_bounds__(256)
void gemm_bf16_quad64_k(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw,
                        void* __restrict__ C, int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n, GemmEpi ep, int group) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 10 units x 16 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tm, tn;
    tile_coords(blockIdx.x, tiles_m, tiles_n, tm, tn, group);
    const int m0 = tm * 256, n0 = tn * 256;
    const int nt = K / 64;                                           // >= 2, host-guaranteed

    const auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (int64_t)m0 * lda), 0, 0xFFFFFFFF, 0x00020000);
    const auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)n0 * ldw), 0, 0xFFFFFFFF, 0x00020000);
    // piece (half, q): rows half * 128 + 32 * wave + 8 q + lane / 8 of the tile; LDS position lane % 8 holds source chunk (lane % 8) ^ (row % 8)
    int offA[2][4], offW[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = h * 128 + wave * 32 + q * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ (row & 7);
            offA[h][q] = min(row, M - 1 - m0) * (int)lda * 2 + chunk * 16;
            offW[h][q] = min(row, N - 1 - n0) * (int)ldw * 2 + chunk * 16;
        }
    typedef __attribute__((address_space(3))) char* lds_ptr;
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    typedef const __attribute__((address_space(3))) bf16x8* lds_fptr;
    const lds_ptr ring_w = (lds_ptr)smem;
    const lds_cptr ring = (lds_cptr)smem;
    // the 8 pieces of this wave for the two W units (IS_A = false) or the two A units (true) of K tile t; u0 = slot of the first of the two units
    auto piece = [&](auto isa_c, auto p_c, int t, int u0) {
        constexpr bool IS_A = decltype(isa_c)::value;
        constexpr int p = decltype(p_c)::value, h = p >> 2, q = p & 3;
        int slot = u0 + h;
        slot = slot >= 10 ? slot - 10 : slot;
        const lds_ptr dst = ring_w + slot * Q64_UNIT + wave * 4096 + q * 1024;
        if (IS_A) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, dst, 16, offA[h][q], t * 128, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, dst, 16, offW[h][q], t * 128, 0, 0);
    };
    auto wrap = [](int u) { return u >= 10 ? u - 10 : u; };

    floatx4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // fragment (row i * 16 + lane % 16 of the unit, K chunk kk * 4 + lane / 16): the swizzle term is (lane % 16) % 8 = lane % 8 for every i
    const int fo0 = (lane & 15) * 128 + ((((lane >> 4)) ^ (lane & 7)) << 4);
    const int fo1 = (lane & 15) * 128 + ((((lane >> 4) + 4) ^ (lane & 7)) << 4);
    bf16x8 fa0[8], fw0[8], fa1[8], fw1[8];

    // prologue: K tiles 0 and 1 (units 0-7)
    static_for<0, 8>([&](auto pc) { piece(std::false_type{}, pc, 0, 0); });
    static_for<0, 8>([&](auto pc) { piece(std::true_type{}, pc, 0, 2); });
    static_for<0, 8>([&](auto pc) { piece(std::false_type{}, pc, 1, 4); });
    static_for<0, 8>([&](auto pc) { piece(std::true_type{}, pc, 1, 6); });
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");        // my pieces of tile 0 have landed
    __builtin_amdgcn_s_barrier();                            // tile 0 published
#pragma unroll
    for (int j = 0; j < 8; ++j) fw0[j] = *(lds_fptr)(ring + wn * Q64_UNIT + fo0 + j * 2048);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa0[i] = *(lds_fptr)(ring + (2 + wm) * Q64_UNIT + fo0 + i * 2048);

    int ub = 0;                                              // slot of unit 4 t (W rows 0-127 of the current tile)
    // One K tile.  STEADY (t + 2 < nt): every piece and every read exists — the body is one straight instruction stream, no
    // branch between the MFMAs (the first version tested `more1` / `more2` at run time around each piece and read: 32 scalar
    // branches per 128 MFMAs).  The last two tiles run the same body with the run-time tests.
    auto tile = [&](int t, auto steady_c) {
        constexpr bool STEADY = decltype(steady_c)::value;
        const bool more1 = STEADY || t + 1 < nt, more2 = STEADY || t + 2 < nt;
        const int ubn = wrap(ub + 4);                        // slot of unit 4 (t + 1)
        const int u8 = wrap(ub + 8);                         // slot of unit 4 (t + 2): W units of tile t + 2 (free since the barrier of tile t - 1)
        // ---- step (t, 0): MFMAs on set 0; reads of (t, second half) into set 1; pieces of the W units of tile t + 2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        {
            const lds_cptr pw = ring + wrap(ub + wn) * Q64_UNIT + fo1;
            const lds_cptr pa = ring + wrap(ub + 2 + wm) * Q64_UNIT + fo1;
            static_for<0, 64>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                constexpr int i = m >> 3, j = m & 7;
                if constexpr (m < 16) {
                    if (VAR != 2) {
                        if constexpr (m < 8) fw1[m] = *(lds_fptr)(pw + m * 2048);
                        else fa1[m - 8] = *(lds_fptr)(pa + (m - 8) * 2048);
                    }
                }
                if constexpr (m >= 16 && m < 40 && (m - 16) % 3 == 0) {
                    if (VAR != 1 && more2) piece(std::false_type{}, std::integral_constant<int, (m - 16) / 3>{}, t + 2, u8);
                }
                QUAD_MFMA(acc[i][j], fw0[j], fa0[i]);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- step (t, 1): MFMAs on set 1; rendezvous; reads of (t + 1, first half) into set 0; pieces of the A units of tile t + 2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        {
            const lds_cptr pw = ring + wrap(ubn + wn) * Q64_UNIT + fo0;
            const lds_cptr pa = ring + wrap(ubn + 2 + wm) * Q64_UNIT + fo0;
            const int u10 = ub;                              // slots of units 4 t, 4 t + 1 = units 4 (t + 2) + 2, + 3: freed by this step's barrier
            static_for<0, 64>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                constexpr int i = m >> 3, j = m & 7;
                if constexpr (m >= 8 && m < 24) {
                    constexpr int r = m - 8;
                    if (VAR != 2 && more1) {
                        if constexpr (r < 8) fw0[r] = *(lds_fptr)(pw + r * 2048);
                        else fa0[r - 8] = *(lds_fptr)(pa + (r - 8) * 2048);
                    }
                }
                if constexpr (m >= 24 && m < 48 && (m - 24) % 3 == 0) {
                    if (VAR != 1 && more2) piece(std::true_type{}, std::integral_constant<int, (m - 24) / 3>{}, t + 2, u10);
                }
                QUAD_MFMA(acc[i][j], fw1[j], fa1[i]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m == 7) {
                    // in flight, oldest first: W(t+1), A(t+1), W(t+2) [if it exists]: retire tile t + 1
                    if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
        ub = ubn;
    };
    int t = 0;
    for (; t + 2 < nt; ++t) tile(t, std::true_type{});
    for (; t < nt; ++t) tile(t, std::false_type{});
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");    // the last MFMAs' results (inline asm: no hazard tracking by the compiler)
    __syncthreads();                           // every wave is done with the ring before it becomes the output image
    epilogue_staged<256, 256, 4, 8, 8>(acc, ep, C, ldc, M, N, m0, n0, wm * 128, wn * 128, wave, lane, smem);
}

// ------------------------------------------------------------------------------------------------
// fp8 (OCP e4m3) operands: the SAME kernel in bytes — a ring stage is still 64 B per operand row (= 64 fp8 K-elements
// instead of 32 bf16), the DMA, swizzle and fragment reads are byte-identical; each 16-byte fragment feeds two
// v_mfma_f32_16x16x32_fp8_fp8 (its low and high 8 bytes: A and W use the same byte -> k assignment, and a dot product
// does not care in which order k is visited).  The operand stream, which bounds the bf16 kernel, halves per K; the fp8
// MFMA runs at the bf16 rate per K, so a stage (K = 64) is MFMA-bound at ~0.43 us per 32 K.  Per-row activation scales
// and per-output-channel weight scales are applied to the fp32 accumulators in phase A of the epilogue.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2)
void gemm_fp8_pingpong_k(const char* __restrict__ A, int64_t lda, const char* __restrict__ W, int64_t ldw,
                         void* __restrict__ C, int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n, GemmEpi ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [5 stages][A 16 KiB | W 16 KiB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    int tm, tn;
    tile_coords(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;
    const char* srcA[2];
    const char* srcW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 32 + i * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ (((row >> 2) & 1) << 1);
        srcA[i] = A + (int64_t)min(m0 + row, M - 1) * lda + chunk * 16;
        srcW[i] = W + (int64_t)min(n0 + row, N - 1) * ldw + chunk * 16;
    }
    const int ns = K / 64;                                   // >= 4 (host guarantees K >= 256)
    auto issue = [&](int s) {
        char* sa = smem + (s % RING_STAGES) * RING_STAGE_BYTES + wave * 32 * 64;
        char* sw = sa + 16384;
        const int64_t koff = (int64_t)s * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + koff),
                                             (__attribute__((address_space(3))) void*)(sa + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[i] + koff),
                                             (__attribute__((address_space(3))) void*)(sw + i * 1024), 16, 0, 0);
        }
    };
    floatx4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int fo = ring_off(lane & 15, lane >> 4);
    typedef __attribute__((ext_vector_type(2))) long long2_t;
    long2_t fa[8], fw[4];

    issue(0); issue(1); issue(2); issue(3);
    wait_vmcnt(12);
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();
    for (int s = 0; s < ns; ++s) {
        {
            const char* sa = smem + (s % RING_STAGES) * RING_STAGE_BYTES + (wm * 128) * 64 + fo;
            const char* sw = smem + (s % RING_STAGES) * RING_STAGE_BYTES + 16384 + (wn * 64) * 64 + fo;
#pragma unroll
            for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const long2_t*>(sw + j * 16 * 64);
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const long2_t*>(sa + i * 16 * 64);
            if (s + 4 < ns) issue(s + 4);
            wait_vmcnt(4 * max(0, min(3, ns - 2 - s)));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fw[j][0], fa[i][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fw[j][1], fa[i][1], acc[i][j], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    epilogue_staged<256, 256, 8, 8, 4, true>(acc, ep, C, ldc, M, N, m0, n0, wm * 128, wn * 64, wave, lane, smem);
}

// ------------------------------------------------------------------------------------------------
// "flow" kernel: the ping-pong main loop made persistent, with an ASYNCHRONOUS epilogue straight from the accumulators.
//
// What the staged epilogue costs on a short-K tile (K = 1280: 40 stages = 30 us): 2.6-4.4 us pipeline fill + 2-4 us
// phase A + 4.5-11 us phase B, none of it overlapped with MFMA work, and every CU reaches it at the same time, so
// the 128 KiB per CU arrive at HBM as one 33 MB burst.  Here, for the epilogue families that need nothing but the
// accumulators (plain / bias / activation / SwiGLU, bf16 out):
//   * after a tile's last stage the NEXT tile's first four K stages are issued into the (now free) ring at once;
//   * the epilogue is register-direct: bf16(acc + bias) [activation | SwiGLU pairing of the interleaved gate|up
//     accumulators, which sit in the same lane], two v_permlane16_swap per 16 x 32 block so that a lane owns 8
//     consecutive columns, one 16-byte buffer store (a wave-instruction writes 16 rows x 64 contiguous bytes; rows
//     past M fall outside the buffer descriptor and are dropped by the hardware, so every store instruction always
//     issues and the count below is exact); no LDS image, no barrier, no wait;
//   * the next main loop starts while those stores drain: vector-memory operations retire in issue order, so the
//     counted waits of the first three stages simply leave the NST stores (issued after stages 0-3, before stage 4)
//     in flight as well: s_waitcnt vmcnt(12 + NST) instead of vmcnt(12).  From stage 3 on the usual counts apply (the
//     stage-4 pieces were issued behind the stores: by then they have had ~3 stages + the fill to drain).
//   * the bias row of a wave (64 columns = 128 bytes) comes through the scalar cache (s_load, lgkmcnt): no vector
//     load that would either drain the pipeline or disturb the vmcnt arithmetic.
// Same tiles, same per-tile K order, same rounding points as gemm_bf16_pingpong_k + epilogue_staged: results are
// bit-identical (tests/test_ops_gpu.py kernels-agree test).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) uint32_t u32x16;

__device__ __forceinline__ void wait_vmcnt4(int n4) {              // waits vmcnt(4 * n4); n4 is wave-uniform, 0..7
    switch (n4) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    }
}

__device__ __forceinline__ uint32_t pack_bf2(float a, float b) { return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16); }

// rows (16-lane groups) 1 and 3 of `a` trade places with rows 0 and 2 of `b`: afterwards an even-row lane holds
// {its own a, its odd neighbour's a} and an odd-row lane {its even neighbour's b, its own b}
__device__ __forceinline__ void swap16(uint32_t& a, uint32_t& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}

template <int EPI>      // 0 plain/bias, 1 GELU(erf), 2 GELU(tanh), 3 ReLU, 4 SwiGLU (N/2 output columns), 5 bias + bf16 residual (may alias C)
__global__ __launch_bounds__(512, 2)
void gemm_bf16_flow_k(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw,
                      bf16_t* C, int ldc, int M, int N, int K, int tiles_m, int tiles_n,
                      const bf16_t* __restrict__ bias, int group, const bf16_t* res = nullptr, int ld_res = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [5 stages][A 16 KiB | W 16 KiB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = tiles_m * tiles_n;
    const int ns = K / 32;                                   // >= 4
    const int fo = ring_off(lane & 15, lane >> 4);
    constexpr int NST = (EPI == 4) ? 8 : 16;                 // epilogue store instructions per wave and tile
    const auto crs = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, (int)((int64_t)M * ldc * 2), 0x00020000);
    // EPI 5: the residual rows through a descriptor of their own (rows past M read as zero and are never stored)
    const auto rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 5 ? res : C), 0, (int)((int64_t)M * (EPI == 5 ? ld_res : ldc) * 2), 0x00020000);
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    typedef const __attribute__((address_space(3))) bf16x8* lds_fptr;
    const lds_cptr ring = (lds_cptr)smem;
    const int constA = wm * 8192 + fo, constW = 16384 + wn * 4096 + fo;

    // DMA sources (lean form, see gemm_bf16_lean_k): the K position advances with the pointers, which are stepped inside the
    // compute phase; a load phase carries no vector-ALU address arithmetic
    const bf16_t* srcA[2];
    const bf16_t* srcW[2];
    int m0 = 0, n0 = 0;
    auto set_tile = [&](int t) {
        int tm, tn;
        tile_coords(t, tiles_m, tiles_n, tm, tn, group);
        m0 = tm * 256; n0 = tn * 256;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 32 + i * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ (((row >> 2) & 1) << 1);
            srcA[i] = A + (int64_t)min(m0 + row, M - 1) * lda + chunk * 8;
            srcW[i] = W + (int64_t)min(n0 + row, N - 1) * ldw + chunk * 8;
        }
    };
    auto issue = [&](int slot_bytes) {                       // the next K stage of the current source pointers -> ring slot
        char* sa = smem + slot_bytes + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcA[i],
                                             (__attribute__((address_space(3))) void*)(sa + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcW[i],
                                             (__attribute__((address_space(3))) void*)(sa + 16384 + i * 1024), 16, 0, 0);
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) { srcA[i] += 32; srcW[i] += 32; }
    };
    auto fill4 = [&]() {
        issue(0); advance(); issue(RING_STAGE_BYTES); advance(); issue(2 * RING_STAGE_BYTES); advance(); issue(3 * RING_STAGE_BYTES); advance();
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    set_tile(tile);
    fill4();
    int extra4 = 0;                                          // (stores of the previous tile still queued behind stages 0-3) / 4
    for (;;) {
        const int cm0 = m0, cn0 = n0;                        // coordinates of the tile being computed
        floatx4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        bf16x8 fa[8], fw[4];
        wait_vmcnt4(3 + extra4);                             // my pieces of stage 0 have landed
        __builtin_amdgcn_s_barrier();                        // stage 0 published
        if (wm == 1) __builtin_amdgcn_s_barrier();           // trailing group starts half a stage later
        int slot_rd = 0, slot_wr = 4 * RING_STAGE_BYTES;
        lds_cptr rdA = ring + constA, rdW = ring + constW;
        // KIND 0: first three stages of a tile (the previous tile's stores are still queued behind stages 0-3: vmcnt(12 + NST));
        // KIND 1: steady state (issue + vmcnt(12), no data-dependent branch); KIND 2: last four stages (nothing left to issue)
        auto stage = [&](int s, auto kind_c) {
            constexpr int KIND = decltype(kind_c)::value;
            {
#pragma unroll
                for (int j = 0; j < 4; ++j) fw[j] = *(lds_fptr)(rdW + j * 16 * 64);
#pragma unroll
                for (int i = 0; i < 8; ++i) fa[i] = *(lds_fptr)(rdA + i * 16 * 64);
                __builtin_amdgcn_sched_barrier(0);
                if (KIND != 2) issue(slot_wr);
                // retire my pieces of stage s+1; stages 0-3 were issued BEFORE the previous tile's stores, stage 4 onwards behind them
                if (KIND == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if (KIND == 0) wait_vmcnt4(3 + extra4);
                else wait_vmcnt4(max(0, min(3, ns - 2 - s)) + (s <= 2 ? extra4 : 0));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            slot_wr = slot_rd;
            slot_rd = slot_rd == 4 * RING_STAGE_BYTES ? 0 : slot_rd + RING_STAGE_BYTES;
            rdA = ring + (constA + slot_rd);
            rdW = ring + (constW + slot_rd);
            if (KIND != 2) advance();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        };
        int s = 0;
        for (; s < 3 && s + 4 < ns; ++s) stage(s, std::integral_constant<int, 0>{});
        for (; s + 4 < ns; ++s) stage(s, std::integral_constant<int, 1>{});
        for (; s < ns; ++s) stage(s, std::integral_constant<int, 2>{});
        if (wm == 0) __builtin_amdgcn_s_barrier();           // leading group: match the barrier count
        // every wave retired its last fragment reads (lgkmcnt(0)) before that barrier: the whole ring is free

        const int next = tile + gridDim.x;
        const bool more = next < ntiles;
        const int colbase = cn0 + wn * 64;                   // wave-uniform; N % 64 == 0 -> a wave is all in or all out
        // EPI 5: the residual of the block's upper half (8 loads of 16 B per lane) goes out BEFORE the next tile's pieces — vector-memory
        // operations retire in issue order, so one counted wait, vmcnt(16), retires exactly those loads and leaves the 16 pieces in
        // flight.  The lower half's loads are issued into the same registers as the upper half's blocks are stored (32 registers for
        // the residual in all: 64 spill), and are retired with everything older by a vmcnt(0) between the halves.
        u32x4 rres[4][2];
        const uint32_t roff0 = EPI == 5 ? (uint32_t)(((cm0 + wm * 128 + (lane & 15)) * ld_res + colbase + ((lane >> 4) & 1) * 16 + (lane >> 5) * 8) * 2) : 0u;
        if (EPI == 5 && colbase < N) {
            static_for<0, 4>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                static_for<0, 2>([&](auto pc) {
                    constexpr int p = decltype(pc)::value;
                    rres[i][p] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff0 + (uint32_t)(i * 16 * ld_res * 2 + p * 64), 0, 0);
                });
            });
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) { set_tile(next); fill4(); }
        __builtin_amdgcn_sched_barrier(0);
        if (EPI == 5) {
            if (more) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- register-direct epilogue of tile (cm0, cn0)
        extra4 = 0;
        if (colbase < N) {
            const int fr = lane & 15, fq = lane >> 4;
            float bv[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[j][r] = 0.f;
            if (EPI != 4 && bias) {
                // the wave's 64 bias values through the scalar cache, 32 columns (16 dwords) at a time
                const bf16_t* bp = bias + colbase;
                static_for<0, 2>([&](auto hc) {
                    constexpr int h = decltype(hc)::value;
                    u32x16 sv;
                    asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(sv) : "s"(bp), "i"(h * 64) : "memory");
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int d = 0; d < 2; ++d) {        // lane needs dword j*8 + fq*2 + d of the wave's 32
                            const int b = jj * 8 + d;
                            const uint32_t w01 = (fq & 1) ? sv[b + 2] : sv[b];
                            const uint32_t w23 = (fq & 1) ? sv[b + 6] : sv[b + 4];
                            const uint32_t wv = (fq & 2) ? w23 : w01;
                            bv[2 * h + jj][2 * d] = __uint_as_float(wv << 16);
                            bv[2 * h + jj][2 * d + 1] = __uint_as_float(wv & 0xffff0000u);
                        }
                });
            }
            if (EPI != 4) {
                const uint32_t off0 = (uint32_t)(((cm0 + wm * 128 + fr) * ldc + colbase + (fq & 1) * 16 + (fq >> 1) * 8) * 2);
                static_for<0, 8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    static_for<0, 2>([&](auto pc) {
                        constexpr int p = decltype(pc)::value;
                        float y[2][4];
                        if constexpr (EPI == 1) {                       // erf GELU, two values per packed instruction
#pragma unroll
                            for (int h = 0; h < 2; ++h)
#pragma unroll
                                for (int r = 0; r < 4; r += 2) {
                                    const f32x2 g2 = gelu_erf_fast2(f32x2{rbf(acc[i][2 * p + h][r] + bv[2 * p + h][r]),
                                                                          rbf(acc[i][2 * p + h][r + 1] + bv[2 * p + h][r + 1])});
                                    y[h][r] = g2[0]; y[h][r + 1] = g2[1];
                                }
                        } else {
#pragma unroll
                            for (int h = 0; h < 2; ++h)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float v = acc[i][2 * p + h][r] + bv[2 * p + h][r];
                                    y[h][r] = (EPI == 0 || EPI == 5) ? v : act_apply(rbf(v), EPI);
                                }
                        }
                        uint32_t a0 = pack_bf2(y[0][0], y[0][1]), a1 = pack_bf2(y[0][2], y[0][3]);
                        uint32_t b0 = pack_bf2(y[1][0], y[1][1]), b1 = pack_bf2(y[1][2], y[1][3]);
                        swap16(a0, b0);
                        swap16(a1, b1);
                        u32x4 outv = u32x4{a0, a1, b0, b1};         // 8 consecutive columns of y0 = bf16(acc + bias)
                        if constexpr (EPI == 5) {                   // out = bf16(residual + y0), as epilogue_rows_res does from the LDS image
                            float q[8], t[8], z[8];
                            unpack8(rres[i & 3][p], q);
                            unpack8(outv, t);
#pragma unroll
                            for (int e = 0; e < 8; ++e) z[e] = q[e] + t[e];
                            outv = pack8(z);
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(outv, crs, off0 + (uint32_t)(i * 16 * ldc * 2 + p * 64), 0, 0);
                        if constexpr (EPI == 5 && i < 4)            // this block's residual registers are free: the same block of the lower half
                            rres[i][p] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff0 + (uint32_t)((i + 4) * 16 * ld_res * 2 + p * 64), 0, 0);
                    });
                    if constexpr (EPI == 5 && i == 3) {
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // lower-half residual landed (and, in order, all before it)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            } else {
                // packed column blocks of 16: acc[.][0] gate / acc[.][1] up of output block 2b, acc[.][2] / acc[.][3] of 2b+1
                const uint32_t off0 = (uint32_t)(((cm0 + wm * 128 + fr) * ldc + (colbase >> 1) + (fq & 1) * 16 + (fq >> 1) * 8) * 2);
                static_for<0, 8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    float y[2][4];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            y[h][r] = rbf(silu_fast(rbf(acc[i][2 * h][r]))) * rbf(acc[i][2 * h + 1][r]);
                    uint32_t a0 = pack_bf2(y[0][0], y[0][1]), a1 = pack_bf2(y[0][2], y[0][3]);
                    uint32_t b0 = pack_bf2(y[1][0], y[1][1]), b1 = pack_bf2(y[1][2], y[1][3]);
                    swap16(a0, b0);
                    swap16(a1, b1);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{a0, a1, b0, b1}, crs, off0 + (uint32_t)(i * 16 * ldc * 2), 0, 0);
                });
            }
            extra4 = EPI == 5 ? 2 : NST / 4;                  // EPI 5: only the lower half's 8 stores are still in flight (vmcnt(0) above)
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!more) break;
        tile = next;
    }
}

// ------------------------------------------------------------------------------------------------
// "flow64" kernel (round 3): the four-wave quad64 main loop made PERSISTENT, with the register-direct asynchronous epilogue of
// the flow kernel — and with the operand stream CONTINUOUS across output tiles.
//
// Why four waves on 64-deep K tiles: a wave owns 128 x 128 of the tile in 256 accumulator registers (the vendor library's
// geometry), reads a third less LDS per flop than the 128 x 64 waves of the ping-pong pair, and its LDS-DMA pieces are 8 rows x
// 128 B — whole cache lines.  With 32-deep stages a piece is 16 rows x 64 B: every 128-byte line is requested twice, half a line
// per stage, a stage apart (the 32 KiB vector L1 has long dropped it), which doubles the L2 -> CU requests of the operand
// stream; that request path is what the main loop of the 8-wave kernels runs into.  The quad64 main loop in
// its branch-free form measured +7 % over the lean kernel (geometric mean of the headline shapes, staged epilogue on both); what
// it still paid per tile — pipeline fill, staged epilogue, launch tail — is what this kernel removes:
//   * the K-tile stream never drains: the ring (ten 16 KiB units, four per K tile, see gemm_bf16_quad64_k) is addressed by a
//     running K-tile count over the workgroup's whole tile sequence, so during the LAST two K tiles of an output tile the pieces
//     issued are the FIRST two K tiles of the next one and the last step already reads the next tile's first fragments — the
//     barrier / counted-vmcnt protocol is the steady state's, unchanged, across the seam;
//   * a tile's first 64 MFMAs take the constant 0 as their C operand (no accumulator clearing);
//   * the epilogue runs between two K tiles of that stream, from the accumulators: bf16(acc + bias) [activation | SwiGLU
//     pairing], two v_permlane16_swap per 16 x 32 block so a lane owns 8 consecutive columns, one 16-byte buffer store per block
//     (rows past M fall outside the descriptor and are dropped: every store instruction issues, the count NST is exact); bias
//     through the scalar cache.  No LDS image, no barrier.  The stores retire in issue order behind the pieces already in flight,
//     so the one counted wait that would otherwise cover them — step (0, 1) of the next tile — leaves them in flight as well:
//     s_waitcnt vmcnt(8 + NST).
// Same tiles, same per-tile K order (32-deep MFMA steps, ascending), same rounding points as every other kernel: bit-identical
// results (tests/test_ops_gpu.py kernels-agree test).  Needs K % 64 == 0, K >= 256, N % 128 == 0 (a wave is all in or all out).
// ------------------------------------------------------------------------------------------------
template <int EPI>      // 0 plain/bias, 1 GELU(erf), 2 GELU(tanh), 3 ReLU, 4 SwiGLU (N/2 output columns), 5 bias + bf16 residual (may alias C)
__global__ __launch_bounds__(256)
void gemm_bf16_flow64_k(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw,
                        bf16_t* C, int ldc, int M, int N, int K, int tiles_m, int tiles_n, const bf16_t* __restrict__ bias, int group,
                        const bf16_t* res = nullptr, int ld_res = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 10 units x 16 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = tiles_m * tiles_n;
    const int nt = K / 64;                                           // >= 4, host-guaranteed
    constexpr int NST = (EPI == 4) ? 16 : 32;                        // epilogue store instructions per wave and tile
    // ... of which this many can still be in flight when the epilogue ends (EPI 5 waits for its residual loads block by block, and
    // vector-memory operations retire in order: only the stores issued after the last such wait remain)
    constexpr int NSTF = (EPI == 5) ? 8 : NST;
    const auto crs = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, (int)((int64_t)M * ldc * 2), 0x00020000);
    // EPI 5: the residual rows through a descriptor of their own (rows past M read as zero and are never stored)
    const auto rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 5 ? res : C), 0, (int)((int64_t)M * (EPI == 5 ? ld_res : ldc) * 2), 0x00020000);
    // Bias: each lane's 8 x 4 values (accumulator block j covers columns colbase + 16 j + 4 (lane / 16) ...) arrive by eight 8-byte
    // buffer loads issued ONE TILE AHEAD — in the prologue for the first tile, at the end of an epilogue for the next tile — so no
    // epilogue waits for them (through the scalar cache, as the 8-wave kernel does it, a tile paid four dependent scalar-load
    // latencies with nothing on the CU to hide them: -15 % on the ViT QKV shape).  The loads are inline asm: invisible to the
    // compiler's own wait insertion (which would drain the LDS-DMA pieces in flight at the first use), counted by hand like the
    // pieces.  No bias: a descriptor of zero records, every load returns 0 without touching memory, same counts.
    constexpr int NB = (EPI == 4) ? 0 : 8;                           // bias loads per wave and tile
    const auto brs = __builtin_amdgcn_make_buffer_rsrc((void*)(bias ? bias : C), 0, bias ? N * 2 : 0, 0x00020000);
    unsigned long long braw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define F64_BIAS1(J, VOFF) asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, 0 offen offset:%c3" : "=v"(braw[J]) : "v"(VOFF), "s"(brs), "i"((J) * 32) : "memory")
#define F64_LOAD_BIAS(N_TILE) do { if constexpr (EPI != 4) { const int voff_ = ((N_TILE) + wn * 128 + 4 * (lane >> 4)) * 2; \
        F64_BIAS1(0, voff_); F64_BIAS1(1, voff_); F64_BIAS1(2, voff_); F64_BIAS1(3, voff_); \
        F64_BIAS1(4, voff_); F64_BIAS1(5, voff_); F64_BIAS1(6, voff_); F64_BIAS1(7, voff_); } } while (0)
    typedef __attribute__((address_space(3))) char* lds_ptr;
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    typedef const __attribute__((address_space(3))) bf16x8* lds_fptr;
    const lds_ptr ring_w = (lds_ptr)smem;
    const lds_cptr ring = (lds_cptr)smem;
    auto wrap = [](int u) { return u >= 10 ? u - 10 : u; };

    // ONE set of operand sources: those of the output tile whose K tiles are being ISSUED (its corner in two buffer descriptors,
    // per-lane byte offsets of the wave's 8 + 8 pieces).  It moves on to the workgroup's next output tile two K tiles before the
    // multiplication does.  piece (half, q): rows half * 128 + 32 * wave + 8 q + lane / 8 of the tile; LDS position lane % 8 holds
    // source chunk (lane % 8) ^ (row % 8); rows past M / N re-read the last valid row (those outputs are never stored)
    int offA[2][4], offW[2][4];
    int mP = 0, nP = 0;
    auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0xFFFFFFFF, 0x00020000);
    auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0xFFFFFFFF, 0x00020000);
    auto set_sources = [&](int t) {
        int tm, tn;
        tile_coords(t, tiles_m, tiles_n, tm, tn, group);
        mP = tm * 256; nP = tn * 256;
        rA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (int64_t)mP * lda), 0, 0xFFFFFFFF, 0x00020000);
        rW = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)nP * ldw), 0, 0xFFFFFFFF, 0x00020000);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = h * 128 + wave * 32 + q * 8 + (lane >> 3);
                const int chunk = (lane & 7) ^ (row & 7);
                offA[h][q] = min(row, M - 1 - mP) * (int)lda * 2 + chunk * 16;
                offW[h][q] = min(row, N - 1 - nP) * (int)ldw * 2 + chunk * 16;
            }
    };
    // The ring as five PAIRS of units (32 KiB each: W rows 0-127 | W rows 128-255, or the two A units): K tile g has its W pair at
    // position 2 g mod 5 and its A pair at 2 g + 1 mod 5.  A pair is filled by the wave's 8 pieces at consecutive KiB of its own 4 KiB
    // of each unit: the LDS destination (M0) starts at pair + 4096 wave and steps by 1 KiB, + 13 KiB from unit 0 to unit 1.
    // (s_add_u32 writes SCC: declared, or the compiler keeps a loop condition in it across the statement.)
    // A piece is two instructions, as the vendor library issues it: the load, then the M0 step for the NEXT piece (so no wait state
    // sits between an M0 write and the load that uses it); M0 is set one MFMA before a pair's first piece.  The compiler has no LDS-DMA
    // of its own in this kernel, so nothing else writes M0 between these statements.
    const int lds_base = __builtin_amdgcn_readfirstlane((int)(uintptr_t)ring_w) + wave * 4096;
    auto nx = [](int pair) { const int n = pair + 32768; return n >= 163840 ? n - 163840 : n; };
#define F64_M0(ADDR) asm volatile("s_mov_b32 m0, %0" :: "s"(ADDR) : "memory")
#define F64_PIECE(VOFF, RSRC, KB, STEP) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, %3" :: "v"(VOFF), "s"(RSRC), "s"(KB), "i"(STEP) : "memory", "scc")
    // the p-th piece (0-7) of the pair being filled: W or A rows, this K tile's byte offset in a row
    auto piece_w = [&](auto p_c, int kb) {
        constexpr int p = decltype(p_c)::value;
        const int vo = offW[p >> 2][p & 3];                  // (locals: an asm operand alone does not capture in a generic lambda)
        const auto r = rW;
        F64_PIECE(vo, r, kb, (p == 3 ? 13312 : 1024));
    };
    auto piece_a = [&](auto p_c, int kb) {
        constexpr int p = decltype(p_c)::value;
        const int vo = offA[p >> 2][p & 3];
        const auto r = rA;
        F64_PIECE(vo, r, kb, (p == 3 ? 13312 : 1024));
    };

    // The 256 accumulators are NOT C++ values: accumulator (i, j) is a[4 (8 i + j) : 4 (8 i + j) + 3], named literally in the MFMA
    // and in the epilogue's reads.  As values under this loop (a conditional epilogue that reads and clears them inside the K-tile
    // loop) the register allocator kept part of them in arch VGPRs, moved them with v_accvgpr_write ahead of every MFMA and spilled
    // to scratch (scratch traffic would also break the counted vmcnt waits).  The clearing statement below declares all of
    // a[0:255] clobbered, which reserves that half of the register file; the compiler itself never touches it as long as its own
    // values fit the 256 arch VGPRs (checked by licv_gemm_flow_available: no private segment; tests: kernels agree bit for bit).
#include "gemm_acc256_clear.inc"
    // fragment (row i * 16 + lane % 16 of the unit, K chunk kk * 4 + lane / 16): the swizzle term is (lane % 16) % 8 = lane % 8 for every i
    const int fo0 = (lane & 15) * 128 + ((((lane >> 4)) ^ (lane & 7)) << 4);
    const int fo1 = (lane & 15) * 128 + ((((lane >> 4) + 4) ^ (lane & 7)) << 4);
    bf16x8 fa0[8], fw0[8], fa1[8], fw1[8];
    // accumulator (i, j) += W fragment j (x) A fragment i; M = 8 i + j is a compile-time constant at every call
#define F64_MFMA(M, w, a) asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(w), "v"(a), "i"(4 * (M)), "i"(4 * (M) + 3))
    // ... the first MFMA of an output tile onto accumulator (i, j): C = 0, so the epilogue does not have to clear the 256 registers it reads
#define F64_MFMA0(M, w, a) asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, 0" :: "v"(w), "v"(a), "i"(4 * (M)), "i"(4 * (M) + 3))
    // read accumulator (i, j) into four floats (the next tile's first MFMA onto it takes C = 0: nothing to clear)
    auto take = [&](auto m_c, float (&v)[4]) {
        constexpr int R = 4 * decltype(m_c)::value;
        asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
    };

    int tile = blockIdx.x;                                   // the output tile being multiplied
    if (tile >= ntiles) return;
    set_sources(tile);
    int mC = mP, nC = nP;                                    // its corner (the epilogue's addresses)

    F64_LOAD_BIAS(nC);                                       // older than every piece: retired by the first counted wait
    // prologue: K tiles 0 and 1 of the first output tile (pairs 0-3)
    F64_M0(lds_base);
    asm volatile("s_nop 0" ::: "memory");
    static_for<0, 8>([&](auto pc) { piece_w(pc, 0); });
    F64_M0(lds_base + 32768);
    asm volatile("s_nop 0" ::: "memory");
    static_for<0, 8>([&](auto pc) { piece_a(pc, 0); });
    F64_M0(lds_base + 65536);
    asm volatile("s_nop 0" ::: "memory");
    static_for<0, 8>([&](auto pc) { piece_w(pc, 128); });
    F64_M0(lds_base + 98304);
    asm volatile("s_nop 0" ::: "memory");
    static_for<0, 8>([&](auto pc) { piece_a(pc, 128); });
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");        // my pieces of K tile 0 have landed
    __builtin_amdgcn_s_barrier();                            // K tile 0 published
    const int rdW0 = fo0 + wn * Q64_UNIT, rdW1 = fo1 + wn * Q64_UNIT;      // fragment-read offsets inside a pair, both K halves
    const int rdA0 = fo0 + wm * Q64_UNIT, rdA1 = fo1 + wm * Q64_UNIT;
#pragma unroll
    for (int j = 0; j < 8; ++j) fw0[j] = *(lds_fptr)(ring + rdW0 + j * 2048);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa0[i] = *(lds_fptr)(ring + 32768 + rdA0 + i * 2048);

    int pW = 0;                                              // ring position (bytes) of the W pair of the K tile being multiplied
    int extra = 0;                                           // what this wave's last epilogue left queued: its stores (NSTF, if it was inside N) + the NB bias loads
    // One K tile of the workgroup's stream (g-th of the stream).  Nothing in it depends on where in an output tile it is except the
    // DATA of the pieces it issues (set_sources two K tiles before a seam; kb) and, for a tile's first K tile (FIRST), the counted wait
    // that has the previous epilogue's stores and bias loads in its queue.  After the last tile's K tile nt - 2 there is nothing left
    // to fetch: the same pieces are issued once more (K tiles 0 and 1 of the last tile again, into pairs that are free by the
    // protocol and never read), so every count stays what it is in the steady state.
    auto ktile = [&](auto first_c, int kb) {
        constexpr bool FIRST = decltype(first_c)::value;
        const int pA = nx(pW), pW1 = nx(pA), pA1 = nx(pW1), pW2 = nx(pA1);     // pairs of g (A), g + 1 (W, A), g + 2 (W; its A pair is pW)
        // ---- step 0: MFMAs on set 0; reads of (g, second half) into set 1; pieces of the W pair of K tile g + 2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        {
            const lds_cptr pw = ring + pW + rdW1;
            const lds_cptr pa = ring + pA + rdA1;
            static_for<0, 64>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                constexpr int i = m >> 3, j = m & 7;
                if constexpr (m < 16) {
                    if constexpr (m < 8) fw1[m] = *(lds_fptr)(pw + m * 2048);
                    else fa1[m - 8] = *(lds_fptr)(pa + (m - 8) * 2048);
                }
                if constexpr (m == 15) F64_M0(lds_base + pW2);
                if constexpr (m >= 16 && m < 40 && (m - 16) % 3 == 0) piece_w(std::integral_constant<int, (m - 16) / 3>{}, kb);
                if constexpr (FIRST) F64_MFMA0(m, fw0[j], fa0[i]); else F64_MFMA(m, fw0[j], fa0[i]);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- step 1: MFMAs on set 1; rendezvous; reads of (g + 1, first half) into set 0; pieces of the A pair of K tile g + 2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        {
            const lds_cptr pw = ring + pW1 + rdW0;
            const lds_cptr pa = ring + pA1 + rdA0;
            static_for<0, 64>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                constexpr int i = m >> 3, j = m & 7;
                if constexpr (m >= 8 && m < 24) {
                    constexpr int r = m - 8;
                    if constexpr (r < 8) fw0[r] = *(lds_fptr)(pw + r * 2048);
                    else fa0[r - 8] = *(lds_fptr)(pa + (r - 8) * 2048);
                }
                if constexpr (m == 23) F64_M0(lds_base + pW);        // the pair of K tile g's W units, freed by this step's barrier
                if constexpr (m >= 24 && m < 48 && (m - 24) % 3 == 0) piece_a(std::integral_constant<int, (m - 24) / 3>{}, kb);
                F64_MFMA(m, fw1[j], fa1[i]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m == 7) {
                    // in flight, oldest first: W(g+1), A(g+1), [the previous epilogue's stores and bias loads,] W(g+2): retire K tile g + 1
                    if constexpr (FIRST) {
                        if (extra == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        else if (extra == NB) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(8 + NB) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" :: "i"(8 + NSTF + NB) : "memory");
                    } else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
        pW = pW1;
    };
    for (;;) {
        ktile(std::true_type{}, 256);
        for (int t = 1; t + 2 < nt; ++t) ktile(std::false_type{}, (t + 2) * 128);
        if (tile + (int)gridDim.x < ntiles) set_sources(tile + gridDim.x);
        ktile(std::false_type{}, 0);
        ktile(std::false_type{}, 128);

        // ==== the output tile is complete: register-direct epilogue of tile (mC, nC); the next tile's K tiles 0 and 1 are in flight
        // or landed meanwhile, its first fragments are being read into set 0.  Every accumulator is cleared as it is read.
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");    // the last MFMAs' results (inline asm: no hazard tracking by the compiler)
        __builtin_amdgcn_sched_barrier(0);
        const int colbase = nC + wn * 128;                   // wave-uniform; N % 128 == 0 -> a wave is all in or all out
        const bool inside = colbase < N;
        if (inside) {                                        // one wave-uniform branch around the whole epilogue: every load / store below
                                                             // always issues, so the counted waits (mine and the compiler's) are exact
            const int fr = lane & 15, fq = lane >> 4;
            if constexpr (EPI != 4) {
                float bv[8][4];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    asm volatile("" : "+v"(braw[j]));            // loaded a tile ago (retired in order ahead of pieces long since waited for)
                    const uint32_t lo = (uint32_t)braw[j], hi = (uint32_t)(braw[j] >> 32);
                    bv[j][0] = __uint_as_float(lo << 16); bv[j][1] = __uint_as_float(lo & 0xffff0000u);
                    bv[j][2] = __uint_as_float(hi << 16); bv[j][3] = __uint_as_float(hi & 0xffff0000u);
                }
                const uint32_t off0 = (uint32_t)(((mC + wm * 128 + fr) * ldc + colbase + (fq & 1) * 16 + (fq >> 1) * 8) * 2);
                // EPI 5: out = bf16(residual + y0), y0 = bf16(acc + bias) (the rounding points of epilogue_rows_res).  The residual of
                // block b = 4 i + p (16 rows x 32 columns per wave-instruction, 16 B per lane, the addresses the store will use) is in
                // flight 8 blocks ahead, in 32 registers: issued at the start for blocks 0-7, then the load of block b + 8 right after
                // the store of block b.  One counted wait per block: the operations younger than the load of block b are the 7 loads
                // after it, or — from block 7 on — the 7 store / load pairs issued since (14), and at the end the stores alone.
                u32x4 rres[8];
                const uint32_t roff0 = EPI == 5 ? (uint32_t)(((mC + wm * 128 + fr) * ld_res + colbase + (fq & 1) * 16 + (fq >> 1) * 8) * 2) : 0u;
                if constexpr (EPI == 5) {
                    static_for<0, 8>([&](auto bc) {
                        constexpr int bb = decltype(bc)::value;
                        rres[bb] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff0 + (uint32_t)((bb >> 2) * 16 * ld_res * 2 + (bb & 3) * 64), 0, 0);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                }
                static_for<0, 8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    static_for<0, 4>([&](auto pc) {
                        constexpr int p = decltype(pc)::value;
                        float y[2][4], c[2][4];
                        take(std::integral_constant<int, 8 * i + 2 * p>{}, c[0]);
                        take(std::integral_constant<int, 8 * i + 2 * p + 1>{}, c[1]);
                        if constexpr (EPI == 1) {                       // erf GELU, two values per packed instruction
    #pragma unroll
                            for (int h = 0; h < 2; ++h)
    #pragma unroll
                                for (int r = 0; r < 4; r += 2) {
                                    const f32x2 g2 = gelu_erf_fast2(f32x2{rbf(c[h][r] + bv[2 * p + h][r]), rbf(c[h][r + 1] + bv[2 * p + h][r + 1])});
                                    y[h][r] = g2[0]; y[h][r + 1] = g2[1];
                                }
                        } else {
    #pragma unroll
                            for (int h = 0; h < 2; ++h)
    #pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float v = c[h][r] + bv[2 * p + h][r];
                                    y[h][r] = (EPI == 0 || EPI == 5) ? v : act_apply(rbf(v), EPI);
                                }
                        }
                        uint32_t a0 = pack_bf2(y[0][0], y[0][1]), a1 = pack_bf2(y[0][2], y[0][3]);
                        uint32_t b0 = pack_bf2(y[1][0], y[1][1]), b1 = pack_bf2(y[1][2], y[1][3]);
                        swap16(a0, b0);
                        swap16(a1, b1);
                        u32x4 outv = u32x4{a0, a1, b0, b1};         // 8 consecutive columns of y0 = bf16(acc + bias)
                        if constexpr (EPI == 5) {
                            constexpr int bb = 4 * i + p;
                            constexpr int younger = bb < 7 ? 7 + bb : (bb <= 24 ? 14 : 7 + (31 - bb));
                            __builtin_amdgcn_sched_barrier(0);
                            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(younger) : "memory");
                            __builtin_amdgcn_sched_barrier(0);
                            float q[8], tt[8], z[8];
                            unpack8(rres[bb & 7], q);
                            unpack8(outv, tt);
    #pragma unroll
                            for (int e = 0; e < 8; ++e) z[e] = q[e] + tt[e];
                            outv = pack8(z);
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(outv, crs, off0 + (uint32_t)(i * 16 * ldc * 2 + p * 64), 0, 0);
                        if constexpr (EPI == 5 && 4 * i + p + 8 < 32) {
                            constexpr int nb = 4 * i + p + 8;
                            __builtin_amdgcn_sched_barrier(0);
                            rres[nb & 7] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff0 + (uint32_t)((nb >> 2) * 16 * ld_res * 2 + (nb & 3) * 64), 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            } else {
                // packed column blocks of 16: acc[.][4q] gate / acc[.][4q+1] up of output block 2q, acc[.][4q+2] / acc[.][4q+3] of 2q+1
                const uint32_t off0 = (uint32_t)(((mC + wm * 128 + fr) * ldc + (colbase >> 1) + (fq & 1) * 16 + (fq >> 1) * 8) * 2);
                static_for<0, 8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    static_for<0, 2>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        float y[2][4], c[4][4];
                        take(std::integral_constant<int, 8 * i + 4 * q>{}, c[0]);
                        take(std::integral_constant<int, 8 * i + 4 * q + 1>{}, c[1]);
                        take(std::integral_constant<int, 8 * i + 4 * q + 2>{}, c[2]);
                        take(std::integral_constant<int, 8 * i + 4 * q + 3>{}, c[3]);
    #pragma unroll
                        for (int h = 0; h < 2; ++h)
    #pragma unroll
                            for (int r = 0; r < 4; ++r)
                                y[h][r] = rbf(silu_fast(rbf(c[2 * h][r]))) * rbf(c[2 * h + 1][r]);
                        uint32_t a0 = pack_bf2(y[0][0], y[0][1]), a1 = pack_bf2(y[0][2], y[0][3]);
                        uint32_t b0 = pack_bf2(y[1][0], y[1][1]), b1 = pack_bf2(y[1][2], y[1][3]);
                        swap16(a0, b0);
                        swap16(a1, b1);
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{a0, a1, b0, b1}, crs, off0 + (uint32_t)(i * 16 * ldc * 2 + q * 64), 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            }
        } else {
            // a wave past N (the last tile column when N % 256 == 128): nothing to store (its accumulators restart from C = 0 with the next tile)
        }
        asm volatile("s_nop 1" ::: "memory");                // accumulator writes (v_accvgpr_write) ahead of the next inline-asm MFMA
        __builtin_amdgcn_sched_barrier(0);
        if (tile + (int)gridDim.x >= ntiles) break;
        tile += gridDim.x;
        mC = mP; nC = nP;
        F64_LOAD_BIAS(nC);                                   // behind this epilogue's stores, ahead of the next K tile's pieces
        __builtin_amdgcn_sched_barrier(0);
        extra = (inside ? NSTF : 0) + NB;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the spare pieces issued after the last tile's K tile nt - 2 land in LDS: not after the workgroup is gone
#undef F64_MFMA
#undef F64_MFMA0
#undef F64_M0
#undef F64_PIECE
#undef F64_BIAS1
#undef F64_LOAD_BIAS
}
// ------------------------------------------------------------------------------------------------
// fp8 (OCP e4m3) operands on the 2x instruction: gemm_bf16_flow64_k in BYTES - the same ring, pieces, rendezvous, counted waits
// and register-direct epilogues; a K tile is 128 bytes a row = 128 fp8 elements - with v_mfma_f32_16x16x128_f8f6f4 (8 passes, 32
// cycles: 64 per wave and K tile where the bf16 kernel issues 128 of 16 cycles, so the operand stream per cycle is the same and
// the K extent per cycle doubles).  a_scale[m] * w_scale[n] multiply the fp32 accumulators in the epilogue, ahead of the bias.
// Needs K % 128 == 0, K >= 512, N % 128 == 0.
// ------------------------------------------------------------------------------------------------
template <int EPI>      // 0 plain/bias, 1 GELU(erf), 2 GELU(tanh), 3 ReLU, 4 SwiGLU (N/2 output columns), 5 bias + bf16 residual (may alias C)
__global__ __launch_bounds__(256)
void gemm_fp8_flow64_k(const char* __restrict__ A, int64_t lda, const char* __restrict__ W, int64_t ldw,
                       const float* __restrict__ a_scale, const float* __restrict__ w_scale,
                       bf16_t* C, int ldc, int M, int N, int K, int tiles_m, int tiles_n, const bf16_t* __restrict__ bias, int group,
                       const bf16_t* res = nullptr, int ld_res = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 10 units x 16 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = tiles_m * tiles_n;
    const int nt = K / 128;                                          // K tiles of 128 fp8 elements (128 bytes a row, as the bf16 kernel's); >= 4, host-guaranteed
    constexpr int NST = (EPI == 4) ? 16 : 32;                        // epilogue store instructions per wave and tile
    // ... of which this many can still be in flight when the epilogue ends (EPI 5 waits for its residual loads block by block, and
    // vector-memory operations retire in order: only the stores issued after the last such wait remain)
    constexpr int NSTF = (EPI == 5) ? 8 : NST;
    const auto crs = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, (int)((int64_t)M * ldc * 2), 0x00020000);
    // EPI 5: the residual rows through a descriptor of their own (rows past M read as zero and are never stored)
    const auto rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 5 ? res : C), 0, (int)((int64_t)M * (EPI == 5 ? ld_res : ldc) * 2), 0x00020000);
    // Bias: each lane's 8 x 4 values (accumulator block j covers columns colbase + 16 j + 4 (lane / 16) ...) arrive by eight 8-byte
    // buffer loads issued ONE TILE AHEAD — in the prologue for the first tile, at the end of an epilogue for the next tile — so no
    // epilogue waits for them (through the scalar cache, as the 8-wave kernel does it, a tile paid four dependent scalar-load
    // latencies with nothing on the CU to hide them: -15 % on the ViT QKV shape).  The loads are inline asm: invisible to the
    // compiler's own wait insertion (which would drain the LDS-DMA pieces in flight at the first use), counted by hand like the
    // pieces.  No bias: a descriptor of zero records, every load returns 0 without touching memory, same counts.
    // ... plus, one tile ahead in the same way, the scales of the fp8 operands: a_scale of the lane's 8 rows (eight 4-byte loads) and
    // w_scale of its 8 x 4 columns - those in COMPACT form: the 32 values depend on lane / 16 only, so lane (fq, fr) keeps elements
    // fr and fr + 16 of its group's list (element e = 4 j + r is column 16 j + 4 fq + r) and the epilogue fetches element e from lane
    // e % 16 of the lane's own row of 16 with one DPP move (row_newbcast).  Held unpacked, one tile ahead, the 32 registers did not
    // fit beside the 128 fragment registers: the compiler parked them in the accumulator file.
    constexpr int NB = (EPI == 4) ? 10 : 12;                         // scale (+ bias) loads per wave and tile
    const auto brs = __builtin_amdgcn_make_buffer_rsrc((void*)(bias ? bias : C), 0, bias ? N * 2 : 0, 0x00020000);
    // (bias in the compact form of the scales below: bf16 elements fr and fr + 16 of the lane group's 32 columns, one register once merged)
    unsigned bc0 = 0, bc1 = 0;
#define F8_BS1(KK, DST, VOFF) asm volatile("buffer_load_ushort %0, %1, %2, 0 offen offset:%c3" : "=v"(DST) : "v"(VOFF), "s"(brs), "i"((KK) * 128) : "memory")
    const auto wsr = __builtin_amdgcn_make_buffer_rsrc((void*)w_scale, 0, N * 4, 0x00020000);
    const auto asr = __builtin_amdgcn_make_buffer_rsrc((void*)a_scale, 0, M * 4, 0x00020000);     // rows past M: scale 0, rows never stored
    float swc[2];
    float sar[8];
#define F8_WS1(KK, VOFF) asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:%c3" : "=v"(swc[KK]) : "v"(VOFF), "s"(wsr), "i"((KK) * 256) : "memory")
#define F8_AS1(I, VOFF) asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:%c3" : "=v"(sar[I]) : "v"(VOFF), "s"(asr), "i"((I) * 64) : "memory")
    // element e = fr + 16 kk: column 16 (e / 4) + 4 fq + e % 4 = 64 kk + 16 (fr / 4) + 4 fq + fr % 4
#define F8_LOAD_SCALES(M_TILE, N_TILE) do { const int wv_ = ((N_TILE) + wn * 128 + 16 * ((lane & 15) >> 2) + 4 * (lane >> 4) + (lane & 3)) * 4; \
        const int av_ = ((M_TILE) + wm * 128 + (lane & 15)) * 4; \
        F8_WS1(0, wv_); F8_WS1(1, wv_); \
        if constexpr (EPI != 4) { const int bvo_ = wv_ >> 1; F8_BS1(0, bc0, bvo_); F8_BS1(1, bc1, bvo_); } \
        F8_AS1(0, av_); F8_AS1(1, av_); F8_AS1(2, av_); F8_AS1(3, av_); F8_AS1(4, av_); F8_AS1(5, av_); F8_AS1(6, av_); F8_AS1(7, av_); } while (0)
    // w_scale of column 16 j + 4 fq + r of the wave's 128 (E = 4 j + r): lane E % 16 of this lane's row holds it in swc[E / 16]
    auto swf = [&](auto e_c) -> float {
        constexpr int E = decltype(e_c)::value;
        float r;            // (volatile: as a builtin the 32 fetches were hoisted out of the row loop and held in 32 registers again)
        asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:%c2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(swc[E >> 4]), "i"(E & 15));
        return r;
    };
    typedef __attribute__((address_space(3))) char* lds_ptr;
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    typedef const __attribute__((address_space(3))) bf16x8* lds_fptr;
    const lds_ptr ring_w = (lds_ptr)smem;
    const lds_cptr ring = (lds_cptr)smem;
    auto wrap = [](int u) { return u >= 10 ? u - 10 : u; };

    // ONE set of operand sources: those of the output tile whose K tiles are being ISSUED (its corner in two buffer descriptors,
    // per-lane byte offsets of the wave's 8 + 8 pieces).  It moves on to the workgroup's next output tile two K tiles before the
    // multiplication does.  piece (half, q): rows half * 128 + 32 * wave + 8 q + lane / 8 of the tile; LDS position lane % 8 holds
    // source chunk (lane % 8) ^ (row % 8); rows past M / N re-read the last valid row (those outputs are never stored)
    // (fp8 form: the 16 per-piece offsets of the bf16 kernel are ONE register per operand - the lane's row within its wave's first
    //  8-row block and its swizzled chunk - plus a wave-uniform row-block term added when the piece is issued; rows past M / N fall
    //  outside the descriptor's range instead of being clamped: 14 registers that the fragments of the 128-deep MFMA need)
    int mP = 0, nP = 0;
    const int pchunk = (lane & 7) ^ ((lane >> 3) & 7);
    const int baseA = (wave * 32 + (lane >> 3)) * (int)lda + pchunk * 16;
    const int baseW = (wave * 32 + (lane >> 3)) * (int)ldw + pchunk * 16;
    auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0xFFFFFFFF, 0x00020000);
    auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0xFFFFFFFF, 0x00020000);
    auto set_sources = [&](int t) {
        int tm, tn;
        tile_coords(t, tiles_m, tiles_n, tm, tn, group);
        mP = tm * 256; nP = tn * 256;
        rA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (int64_t)mP * lda), 0, min(M - mP, 256) * (int)lda, 0x00020000);      // (lda, ldw: bytes)
        rW = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)nP * ldw), 0, min(N - nP, 256) * (int)ldw, 0x00020000);
    };
    // The ring as five PAIRS of units (32 KiB each: W rows 0-127 | W rows 128-255, or the two A units): K tile g has its W pair at
    // position 2 g mod 5 and its A pair at 2 g + 1 mod 5.  A pair is filled by the wave's 8 pieces at consecutive KiB of its own 4 KiB
    // of each unit: the LDS destination (M0) starts at pair + 4096 wave and steps by 1 KiB, + 13 KiB from unit 0 to unit 1.
    // (s_add_u32 writes SCC: declared, or the compiler keeps a loop condition in it across the statement.)
    // A piece is two instructions, as the vendor library issues it: the load, then the M0 step for the NEXT piece (so no wait state
    // sits between an M0 write and the load that uses it); M0 is set one MFMA before a pair's first piece.  The compiler has no LDS-DMA
    // of its own in this kernel, so nothing else writes M0 between these statements.
    const int lds_base = __builtin_amdgcn_readfirstlane((int)(uintptr_t)ring_w) + wave * 4096;
    auto nx = [](int pair) { const int n = pair + 32768; return n >= 163840 ? n - 163840 : n; };
#define F64_M0(ADDR) asm volatile("s_mov_b32 m0, %0" :: "s"(ADDR) : "memory")
#define F64_PIECE(VOFF, RSRC, KB, STEP) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, %3" :: "v"(VOFF), "s"(RSRC), "s"(KB), "i"(STEP) : "memory", "scc")
    // the p-th piece (0-7) of the pair being filled: W or A rows, this K tile's byte offset in a row
    auto piece_w = [&](auto p_c, int kb) {
        constexpr int p = decltype(p_c)::value;
        const int vo = baseW + ((p >> 2) * 128 + (p & 3) * 8) * (int)ldw;    // (locals: an asm operand alone does not capture in a generic lambda)
        const auto r = rW;
        F64_PIECE(vo, r, kb, (p == 3 ? 13312 : 1024));
    };
    auto piece_a = [&](auto p_c, int kb) {
        constexpr int p = decltype(p_c)::value;
        const int vo = baseA + ((p >> 2) * 128 + (p & 3) * 8) * (int)lda;
        const auto r = rA;
        F64_PIECE(vo, r, kb, (p == 3 ? 13312 : 1024));
    };

    // The 256 accumulators are NOT C++ values: accumulator (i, j) is a[4 (8 i + j) : 4 (8 i + j) + 3], named literally in the MFMA
    // and in the epilogue's reads.  As values under this loop (a conditional epilogue that reads and clears them inside the K-tile
    // loop) the register allocator kept part of them in arch VGPRs, moved them with v_accvgpr_write ahead of every MFMA and spilled
    // to scratch (scratch traffic would also break the counted vmcnt waits).  The clearing statement below declares all of
    // a[0:255] clobbered, which reserves that half of the register file; the compiler itself never touches it as long as its own
    // values fit the 256 arch VGPRs (checked by licv_gemm_flow_available: no private segment; tests: kernels agree bit for bit).
#include "gemm_acc256_clear.inc"
    // fragment (row i * 16 + lane % 16 of the unit, K chunk kk * 4 + lane / 16): the swizzle term is (lane % 16) % 8 = lane % 8 for every i
    const int fo0 = (lane & 15) * 128 + ((((lane >> 4)) ^ (lane & 7)) << 4);
    const int fo1 = (lane & 15) * 128 + ((((lane >> 4) + 4) ^ (lane & 7)) << 4);
    // a lane's operand of the 128-deep MFMA: 32 bytes = the 16-byte chunks (lane / 16) and (lane / 16) + 4 of its row - for A and W
    // alike, so both sides pair the same K elements (the sum does not care in which order k is visited)
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    i32x8 fa[8], fw[8];
    // accumulator (i, j) += W fragment j (x) A fragment i; M = 8 i + j is a compile-time constant at every call
#define F64_MFMA(M, w, a) asm volatile("v_mfma_f32_16x16x128_f8f6f4 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(w), "v"(a), "i"(4 * (M)), "i"(4 * (M) + 3))
#define F64_MFMA0(M, w, a) asm volatile("v_mfma_f32_16x16x128_f8f6f4 a[%c2:%c3], %0, %1, 0" :: "v"(w), "v"(a), "i"(4 * (M)), "i"(4 * (M) + 3))
    // read accumulator (i, j) into four floats (the next tile's first MFMA onto it takes C = 0: nothing to clear)
    auto take = [&](auto m_c, float (&v)[4]) {
        constexpr int R = 4 * decltype(m_c)::value;
        asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
    };

    int tile = blockIdx.x;                                   // the output tile being multiplied
    if (tile >= ntiles) return;
    set_sources(tile);
    int mC = mP, nC = nP;                                    // its corner (the epilogue's addresses)

    F8_LOAD_SCALES(mC, nC);
    // prologue: K tiles 0 and 1 of the first output tile (pairs 0-3)
    F64_M0(lds_base);
    asm volatile("s_nop 0" ::: "memory");
    static_for<0, 8>([&](auto pc) { piece_w(pc, 0); });
    F64_M0(lds_base + 32768);
    asm volatile("s_nop 0" ::: "memory");
    static_for<0, 8>([&](auto pc) { piece_a(pc, 0); });
    F64_M0(lds_base + 65536);
    asm volatile("s_nop 0" ::: "memory");
    static_for<0, 8>([&](auto pc) { piece_w(pc, 128); });
    F64_M0(lds_base + 98304);
    asm volatile("s_nop 0" ::: "memory");
    static_for<0, 8>([&](auto pc) { piece_a(pc, 128); });
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");        // my pieces of K tile 0 have landed
    __builtin_amdgcn_s_barrier();                            // K tile 0 published
    const int rdW0 = fo0 + wn * Q64_UNIT, rdW1 = fo1 + wn * Q64_UNIT;      // fragment-read offsets inside a pair, both K halves
    const int rdA0 = fo0 + wm * Q64_UNIT, rdA1 = fo1 + wm * Q64_UNIT;
    auto read_frag = [&](i32x8& f, lds_cptr p0, lds_cptr p1) {       // two ds_read_b128 into the halves of one 8-register operand
        const u32x4 lo = *(const __attribute__((address_space(3))) u32x4*)p0;
        const u32x4 hi = *(const __attribute__((address_space(3))) u32x4*)p1;
        f[0] = (int)lo.x; f[1] = (int)lo.y; f[2] = (int)lo.z; f[3] = (int)lo.w;
        f[4] = (int)hi.x; f[5] = (int)hi.y; f[6] = (int)hi.z; f[7] = (int)hi.w;
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) read_frag(fw[j], ring + rdW0 + j * 2048, ring + rdW1 + j * 2048);
#pragma unroll
    for (int i = 0; i < 8; ++i) read_frag(fa[i], ring + 32768 + rdA0 + i * 2048, ring + 32768 + rdA1 + i * 2048);

    int pW = 0;                                              // ring position (bytes) of the W pair of the K tile being multiplied
    int extra = 0;                                           // what this wave's last epilogue left queued: its stores (NSTF, if it was inside N) + the NB bias loads
    // One K tile of the workgroup's stream (g-th of the stream).  Nothing in it depends on where in an output tile it is except the
    // DATA of the pieces it issues (set_sources two K tiles before a seam; kb) and, for a tile's first K tile (FIRST), the counted wait
    // that has the previous epilogue's stores and bias loads in its queue.  After the last tile's K tile nt - 2 there is nothing left
    // to fetch: the same pieces are issued once more (K tiles 0 and 1 of the last tile again, into pairs that are free by the
    // protocol and never read), so every count stays what it is in the steady state.
    auto ktile = [&](auto first_c, int kb) {
        constexpr bool FIRST = decltype(first_c)::value;
        const int pA = nx(pW), pW1 = nx(pA), pA1 = nx(pW1), pW2 = nx(pA1);     // pairs of g (A), g + 1 (W, A), g + 2 (W; its A pair is pW)
        // The 64 MFMAs of K tile g (16 x 16 x 128, 32 cycles each: the cycles of the bf16 kernel's 128) run row by row, accumulator
        // (i, j) = W fragment j x A fragment i, on ONE set of fragments: a fragment is dead once its last MFMA has issued, and is
        // re-read for K tile g + 1 right then - A fragment i after row i (rows 0-3: after the rendezvous that publishes K tile g + 1),
        // W fragment j after MFMA (7, j), eight MFMAs (256 cycles) before K tile g + 1 needs it.
        // ---- step 0, rows 0-3: pieces of the W pair of K tile g + 2
        static_for<0, 32>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int i = m >> 3, j = m & 7;
            if constexpr (m == 0) F64_M0(lds_base + pW2);
            if constexpr (m >= 2 && m < 26 && (m - 2) % 3 == 0) piece_w(std::integral_constant<int, (m - 2) / 3>{}, kb);
            if constexpr (FIRST) F64_MFMA0(m, fw[j], fa[i]); else F64_MFMA(m, fw[j], fa[i]);
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---- step 1, rows 4-7: rendezvous (K tile g + 1 published, every wave done with the LDS of K tile g), re-reads, pieces of
        // the A pair of K tile g + 2
        {
            const lds_cptr pw0 = ring + pW1 + rdW0, pw1 = ring + pW1 + rdW1;
            const lds_cptr pa0 = ring + pA1 + rdA0, pa1 = ring + pA1 + rdA1;
            static_for<32, 64>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                constexpr int i = m >> 3, j = m & 7;
                if constexpr (m == 35) F64_M0(lds_base + pW);        // the pair of K tile g's W units, free since the rendezvous
                if constexpr (m >= 36 && m < 60 && (m - 36) % 3 == 0) piece_a(std::integral_constant<int, (m - 36) / 3>{}, kb);
                if constexpr (FIRST) F64_MFMA0(m, fw[j], fa[i]); else F64_MFMA(m, fw[j], fa[i]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m == 33) {
                    // in flight, oldest first: W(g+1), A(g+1), [the previous epilogue's stores and bias / scale loads,] W(g+2): retire K tile g + 1
                    if constexpr (FIRST) {
                        if (extra == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        else if (extra == NB) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(8 + NB) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" :: "i"(8 + NSTF + NB > 63 ? 63 : 8 + NSTF + NB) : "memory");     // (the counter holds 63: one operation stricter there)
                    } else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m >= 34 && m < 38) { read_frag(fa[m - 34], pa0 + (m - 34) * 2048, pa1 + (m - 34) * 2048); __builtin_amdgcn_sched_barrier(0); }   // rows 0-3, dead since step 0
                if constexpr (j == 7 && i < 7) { read_frag(fa[i], pa0 + i * 2048, pa1 + i * 2048); __builtin_amdgcn_sched_barrier(0); }                         // row i just ended
                if constexpr (i == 7) { read_frag(fw[j], pw0 + j * 2048, pw1 + j * 2048); __builtin_amdgcn_sched_barrier(0); }
                if constexpr (m == 63) { read_frag(fa[7], pa0 + 7 * 2048, pa1 + 7 * 2048); __builtin_amdgcn_sched_barrier(0); }
            });
        }
        pW = pW1;
    };
    for (;;) {
        ktile(std::true_type{}, 256);
        for (int t = 1; t + 2 < nt; ++t) ktile(std::false_type{}, (t + 2) * 128);
        if (tile + (int)gridDim.x < ntiles) set_sources(tile + gridDim.x);
        ktile(std::false_type{}, 0);
        ktile(std::false_type{}, 128);

        // ==== the output tile is complete: register-direct epilogue of tile (mC, nC); the next tile's K tiles 0 and 1 are in flight
        // or landed meanwhile, its first fragments are being read into set 0.  Every accumulator is cleared as it is read.
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");    // the last MFMAs' results (inline asm: no hazard tracking by the compiler)
        __builtin_amdgcn_sched_barrier(0);
        const int colbase = nC + wn * 128;                   // wave-uniform; N % 128 == 0 -> a wave is all in or all out
        const bool inside = colbase < N;
        if (inside) {                                        // one wave-uniform branch around the whole epilogue: every load / store below
                                                             // always issues, so the counted waits (mine and the compiler's) are exact
            const int fr = lane & 15, fq = lane >> 4;
            if constexpr (EPI != 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(sar[j]));   // loaded a tile ago (retired in order ahead of pieces long since waited for)
                asm volatile("" : "+v"(swc[0]), "+v"(swc[1]), "+v"(bc0), "+v"(bc1));
                const unsigned bcm = (bc0 & 0xffffu) | (bc1 << 16);
                // bias of column 16 j + 4 fq + r (E = 4 j + r): lane E % 16 of the row, half E / 16
                auto bvf = [&](auto e_c) -> float {
                    constexpr int E = decltype(e_c)::value;
                    unsigned r;
                    asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:%c2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(bcm), "i"(E & 15));
                    return __uint_as_float(E < 16 ? (r << 16) : (r & 0xffff0000u));
                };
                const uint32_t off0 = (uint32_t)(((mC + wm * 128 + fr) * ldc + colbase + (fq & 1) * 16 + (fq >> 1) * 8) * 2);
                // EPI 5: out = bf16(residual + y0), y0 = bf16(acc + bias) (the rounding points of epilogue_rows_res).  The residual of
                // block b = 4 i + p (16 rows x 32 columns per wave-instruction, 16 B per lane, the addresses the store will use) is in
                // flight 8 blocks ahead, in 32 registers: issued at the start for blocks 0-7, then the load of block b + 8 right after
                // the store of block b.  One counted wait per block: the operations younger than the load of block b are the 7 loads
                // after it, or — from block 7 on — the 7 store / load pairs issued since (14), and at the end the stores alone.
                u32x4 rres[8];
                const uint32_t roff0 = EPI == 5 ? (uint32_t)(((mC + wm * 128 + fr) * ld_res + colbase + (fq & 1) * 16 + (fq >> 1) * 8) * 2) : 0u;
                if constexpr (EPI == 5) {
                    static_for<0, 8>([&](auto bc) {
                        constexpr int bb = decltype(bc)::value;
                        rres[bb] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff0 + (uint32_t)((bb >> 2) * 16 * ld_res * 2 + (bb & 3) * 64), 0, 0);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                }
                static_for<0, 8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    static_for<0, 4>([&](auto pc) {
                        constexpr int p = decltype(pc)::value;
                        float y[2][4], c[2][4];
                        take(std::integral_constant<int, 8 * i + 2 * p>{}, c[0]);
                        take(std::integral_constant<int, 8 * i + 2 * p + 1>{}, c[1]);
                        float sc[2][4], bb[2][4];                       // a_scale[m] * w_scale[n] is applied as (acc * a) * w, like the 8-wave kernel
                        static_for<0, 8>([&](auto ec) {
                            constexpr int e8 = decltype(ec)::value;
                            sc[e8 >> 2][e8 & 3] = swf(std::integral_constant<int, 8 * p + e8>{});
                            bb[e8 >> 2][e8 & 3] = bvf(std::integral_constant<int, 8 * p + e8>{});
                        });
                        if constexpr (EPI == 1) {                       // erf GELU, two values per packed instruction
    #pragma unroll
                            for (int h = 0; h < 2; ++h)
    #pragma unroll
                                for (int r = 0; r < 4; r += 2) {
                                    const f32x2 g2 = gelu_erf_fast2(f32x2{rbf(c[h][r] * sar[i] * sc[h][r] + bb[h][r]), rbf(c[h][r + 1] * sar[i] * sc[h][r + 1] + bb[h][r + 1])});
                                    y[h][r] = g2[0]; y[h][r + 1] = g2[1];
                                }
                        } else {
    #pragma unroll
                            for (int h = 0; h < 2; ++h)
    #pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float v = c[h][r] * sar[i] * sc[h][r] + bb[h][r];     // C = (Aq . Wq^T) * a_scale[m] * w_scale[n] (+ bias)
                                    y[h][r] = (EPI == 0 || EPI == 5) ? v : act_apply(rbf(v), EPI);
                                }
                        }
                        uint32_t a0 = pack_bf2(y[0][0], y[0][1]), a1 = pack_bf2(y[0][2], y[0][3]);
                        uint32_t b0 = pack_bf2(y[1][0], y[1][1]), b1 = pack_bf2(y[1][2], y[1][3]);
                        swap16(a0, b0);
                        swap16(a1, b1);
                        u32x4 outv = u32x4{a0, a1, b0, b1};         // 8 consecutive columns of y0 = bf16(acc + bias)
                        if constexpr (EPI == 5) {
                            constexpr int bb = 4 * i + p;
                            constexpr int younger = bb < 7 ? 7 + bb : (bb <= 24 ? 14 : 7 + (31 - bb));
                            __builtin_amdgcn_sched_barrier(0);
                            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(younger) : "memory");
                            __builtin_amdgcn_sched_barrier(0);
                            float q[8], tt[8], z[8];
                            unpack8(rres[bb & 7], q);
                            unpack8(outv, tt);
    #pragma unroll
                            for (int e = 0; e < 8; ++e) z[e] = q[e] + tt[e];
                            outv = pack8(z);
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(outv, crs, off0 + (uint32_t)(i * 16 * ldc * 2 + p * 64), 0, 0);
                        if constexpr (EPI == 5 && 4 * i + p + 8 < 32) {
                            constexpr int nb = 4 * i + p + 8;
                            __builtin_amdgcn_sched_barrier(0);
                            rres[nb & 7] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff0 + (uint32_t)((nb >> 2) * 16 * ld_res * 2 + (nb & 3) * 64), 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            } else {
                // packed column blocks of 16: acc[.][4q] gate / acc[.][4q+1] up of output block 2q, acc[.][4q+2] / acc[.][4q+3] of 2q+1
                const uint32_t off0 = (uint32_t)(((mC + wm * 128 + fr) * ldc + (colbase >> 1) + (fq & 1) * 16 + (fq >> 1) * 8) * 2);
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(sar[j]));
                asm volatile("" : "+v"(swc[0]), "+v"(swc[1]));
                static_for<0, 8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    static_for<0, 2>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        float y[2][4], c[4][4];
                        take(std::integral_constant<int, 8 * i + 4 * q>{}, c[0]);
                        take(std::integral_constant<int, 8 * i + 4 * q + 1>{}, c[1]);
                        take(std::integral_constant<int, 8 * i + 4 * q + 2>{}, c[2]);
                        take(std::integral_constant<int, 8 * i + 4 * q + 3>{}, c[3]);
                        float sc[4][4];
                        static_for<0, 16>([&](auto ec) { constexpr int e16 = decltype(ec)::value; sc[e16 >> 2][e16 & 3] = swf(std::integral_constant<int, 16 * q + e16>{}); });
    #pragma unroll
                        for (int h = 0; h < 2; ++h)
    #pragma unroll
                            for (int r = 0; r < 4; ++r)
                                y[h][r] = rbf(silu_fast(rbf(c[2 * h][r] * sar[i] * sc[2 * h][r]))) * rbf(c[2 * h + 1][r] * sar[i] * sc[2 * h + 1][r]);
                        uint32_t a0 = pack_bf2(y[0][0], y[0][1]), a1 = pack_bf2(y[0][2], y[0][3]);
                        uint32_t b0 = pack_bf2(y[1][0], y[1][1]), b1 = pack_bf2(y[1][2], y[1][3]);
                        swap16(a0, b0);
                        swap16(a1, b1);
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{a0, a1, b0, b1}, crs, off0 + (uint32_t)(i * 16 * ldc * 2 + q * 64), 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            }
        } else {
            // a wave past N (the last tile column when N % 256 == 128): nothing to store (its accumulators restart from C = 0 with the next tile)
        }
        asm volatile("s_nop 1" ::: "memory");                // accumulator writes (v_accvgpr_write) ahead of the next inline-asm MFMA
        __builtin_amdgcn_sched_barrier(0);
        if (tile + (int)gridDim.x >= ntiles) break;
        tile += gridDim.x;
        mC = mP; nC = nP;
        F8_LOAD_SCALES(mC, nC);
        __builtin_amdgcn_sched_barrier(0);
        extra = (inside ? NSTF : 0) + NB;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the spare pieces issued after the last tile's K tile nt - 2 land in LDS: not after the workgroup is gone
#undef F64_MFMA
#undef F64_MFMA0
#undef F64_M0
#undef F64_PIECE
#undef F8_BS1
#undef F8_WS1
#undef F8_AS1
#undef F8_LOAD_SCALES
}

// ------------------------------------------------------------------------------------------------
// "mid" kernel (round 3): 128 x 128 tiles for the row counts between the weight-streaming kernel (M <= 32) and the 256-tile
// kernels (M >= 512) — the 32-token student of training, the prefill of hooked generate (M = B S = 256), the vision tower on a
// handful of images (M = 8 x 257) — and the producer of their split-K partials.  It is the quad64 main loop at half scale:
//   * 4 waves as 2 x 2, 64 x 64 per wave (16 accumulators), 64-deep K tiles;
//   * LDS = a ring of five PAIRS of 8 KiB units (a unit = 64 rows x 64 K of one operand, 128-byte rows, 16-byte chunk c of row r
//     at position c ^ (r & 7)): K tile g has its W pair at position 2 g mod 5, its A pair at 2 g + 1 mod 5; 80 KiB, so TWO
//     workgroups share a CU and cover each other's waits;
//   * LDS-DMA pieces of 8 rows x 128 B (whole cache lines), issued as `load; M0 step`, four per wave and pair, two K tiles ahead;
//     one barrier per K tile, counted vmcnt;
//   * staged epilogue (every epilogue family), or — SPLITK — fp32 partial tiles into [split][M_pad][N_pad] for
//     skinny_finalize_k (row-major, 4 columns per thread).
// Against the register-staged 128-tile kernel it replaces for K % 64 == 0: no VGPR round trip and no ds_write for the operands,
// two K tiles in flight instead of one, half the barriers.  Same K order and rounding points: bit-identical.
// ------------------------------------------------------------------------------------------------
#define MID_PAIR 16384
#define MID_LDS (5 * MID_PAIR)
#define MID_LDS_DEEP (9 * MID_PAIR)
// D = K tiles in flight ahead of the one being multiplied.  2: the ring of five pairs above (80 KiB, two workgroups per CU).  4: nine
// pairs (144 KiB, one workgroup per CU): the W pieces of K tile t + D are issued D - 0.5 K tiles before their rendezvous instead of
// 1.5 - an experiment (licv_gemm_experiment knob 10), bit-identical and no faster: see mid_deep().
// NT = 1: the W pieces (the once-read weight stream) carry the non-temporal cache policy (MI355X_MICROARCH.md, price list row nt-weights:
// issued -> landed -18 % for once-read bytes; the A pieces, re-read by every workgroup from L2, keep the default policy).  knob 11.
template <int SPLITK, int ABL = 0, int D = 2, int NT = 0>      // ABL (timing only, wrong results): 1 = the A pieces are never issued, 2 = the W pieces
__global__ __launch_bounds__(256, 2)      // (also for D = 4, one workgroup per CU by LDS: with 512 registers allowed the allocator moves the accumulators through AGPRs with copies around every loop)
void gemm_bf16_mid_k(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw,
                     void* __restrict__ C, int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n, GemmEpi ep, int per) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 5 pairs x 16 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tm, tn;
    tile_coords(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * 128, n0 = tn * 128;
    const int kt0 = SPLITK ? (int)blockIdx.y * per : 0;                       // first 64-deep K tile of this workgroup
    const int nt = SPLITK ? min(per, K / 64 - kt0) : K / 64;                  // >= 2, host-guaranteed
    typedef __attribute__((address_space(3))) char* lds_ptr;
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    typedef const __attribute__((address_space(3))) bf16x8* lds_fptr;
    const lds_ptr ring_w = (lds_ptr)smem;
    const lds_cptr ring = (lds_cptr)smem;

    // piece q (0-3) of a pair: rows 32 wave + 8 q + lane / 8 of the operand's 128; LDS position lane % 8 holds source chunk
    // (lane % 8) ^ (row % 8); rows past M / N re-read the last valid row (those outputs are never stored)
    const auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (int64_t)m0 * lda + (int64_t)kt0 * 64), 0, 0xFFFFFFFF, 0x00020000);
    const auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)n0 * ldw + (int64_t)kt0 * 64), 0, 0xFFFFFFFF, 0x00020000);
    int offA[4], offW[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wave * 32 + q * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        offA[q] = min(row, M - 1 - m0) * (int)lda * 2 + chunk * 16;
        offW[q] = min(row, N - 1 - n0) * (int)ldw * 2 + chunk * 16;
    }
    const int lds_base = __builtin_amdgcn_readfirstlane((int)(uintptr_t)ring_w) + wave * 4096;
    constexpr int RING = (2 * D + 1) * MID_PAIR;           // K tile g: W pair at 2 g mod (2 D + 1), A pair at 2 g + 1; A(t + D) takes W(t)'s pair
    auto nx = [](int pair) { const int n = pair + MID_PAIR; return n >= RING ? n - RING : n; };
    auto pv = [](int pair) { return pair == 0 ? RING - MID_PAIR : pair - MID_PAIR; };
#define MID_M0(ADDR) asm volatile("s_mov_b32 m0, %0" :: "s"(ADDR) : "memory")
#define MID_PIECE(VOFF, RSRC, KB) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, 0x400" :: "v"(VOFF), "s"(RSRC), "s"(KB) : "memory", "scc")
#define MID_PIECE_NT(VOFF, RSRC, KB) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen nt lds\n\ts_add_u32 m0, m0, 0x400" :: "v"(VOFF), "s"(RSRC), "s"(KB) : "memory", "scc")
#define MID_PIECE_W(VOFF, RSRC, KB) do { if constexpr (NT) MID_PIECE_NT(VOFF, RSRC, KB); else MID_PIECE(VOFF, RSRC, KB); } while (0)

    floatx4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    // fragment (row i * 16 + lane % 16 of the unit, K chunk kk * 4 + lane / 16): the swizzle term is (lane % 16) % 8 = lane % 8 for every i
    const int fo0 = (lane & 15) * 128 + ((((lane >> 4)) ^ (lane & 7)) << 4);
    const int fo1 = (lane & 15) * 128 + ((((lane >> 4) + 4) ^ (lane & 7)) << 4);
    const int rdW0 = fo0 + wn * 8192, rdW1 = fo1 + wn * 8192, rdA0 = fo0 + wm * 8192, rdA1 = fo1 + wm * 8192;
    bf16x8 fa0[4], fw0[4], fa1[4], fw1[4];

    // prologue: K tiles 0 .. D - 1 (pairs 0 .. 2 D - 1)
    static_for<0, D>([&](auto gc) {
        constexpr int gk = decltype(gc)::value;
        MID_M0(lds_base + 2 * gk * MID_PAIR);
        asm volatile("s_nop 0" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int vo = offW[q]; const auto r = rW; MID_PIECE_W(vo, r, gk * 128); }   // (locals: asm operands alone do not capture)
        MID_M0(lds_base + (2 * gk + 1) * MID_PAIR);
        asm volatile("s_nop 0" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int vo = offA[q]; const auto r = rA; MID_PIECE(vo, r, gk * 128); }
    });
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(8 * (D - 1)) : "memory");         // my pieces of K tile 0 have landed
    __builtin_amdgcn_s_barrier();                            // K tile 0 published
#pragma unroll
    for (int j = 0; j < 4; ++j) fw0[j] = *(lds_fptr)(ring + rdW0 + j * 2048);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa0[i] = *(lds_fptr)(ring + MID_PAIR + rdA0 + i * 2048);

    int pW = 0;                                              // ring position of the W pair of the K tile being multiplied
    // One K tile.  STEADY (t + D < nt): pieces of K tile t + D and reads of K tile t + 1 exist: one straight instruction stream.
    auto ktile = [&](int t, auto steady_c) {
        constexpr bool STEADY = decltype(steady_c)::value;
        const bool more1 = STEADY || t + 1 < nt, more2 = STEADY || t + D < nt;
        const int kb = (t + D) * 128;
        const int pA = nx(pW), pW1 = nx(pA), pA1 = nx(pW1), pW2 = pv(pW);      // pairs of t (A), t + 1 (W, A), t + D (W: the pair before pW; its A pair is pW)
        // ---- step 0: MFMAs on set 0; reads of (t, second half) into set 1; pieces of the W pair of K tile t + 2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        {
            const lds_cptr pw = ring + pW + rdW1;
            const lds_cptr pa = ring + pA + rdA1;
            if (more2) MID_M0(lds_base + pW2);
            static_for<0, 16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                constexpr int i = m >> 2, j = m & 3;
                if constexpr (m < 8) {
                    if constexpr (m < 4) fw1[m] = *(lds_fptr)(pw + m * 2048);
                    else fa1[m - 4] = *(lds_fptr)(pa + (m - 4) * 2048);
                }
                if constexpr (m >= 8 && m < 16 && (m & 1) == 0) {
                    if (more2 && !(ABL & 2)) MID_PIECE_W(offW[(m - 8) >> 1], rW, kb);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw0[j], fa0[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- step 1: MFMAs on set 1; rendezvous; reads of (t + 1, first half) into set 0; pieces of the A pair of K tile t + 2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        {
            const lds_cptr pw = ring + pW1 + rdW0;
            const lds_cptr pa = ring + pA1 + rdA0;
            static_for<0, 16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                constexpr int i = m >> 2, j = m & 3;
                if constexpr (m >= 2 && m < 10) {
                    constexpr int r = m - 2;
                    if (more1) {
                        if constexpr (r < 4) fw0[r] = *(lds_fptr)(pw + r * 2048);
                        else fa0[r - 4] = *(lds_fptr)(pa + (r - 4) * 2048);
                    }
                }
                if constexpr (m == 7) { if (more2) MID_M0(lds_base + pW); }      // the pair of K tile t's W units, freed by this step's barrier
                if constexpr (m >= 8 && m < 16 && (m & 1) == 0) {
                    if (more2 && !(ABL & 1)) MID_PIECE(offA[(m - 8) >> 1], rA, kb);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw1[j], fa1[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m == 1) {
                    // in flight, oldest first: W(t+1), A(t+1), W(t+2) [if it exists]: retire K tile t + 1
                    // in flight, oldest first: W, A of t + 1 .. t + D - 1, W(t + D): retire K tile t + 1 = leave 4 (2 D - 3) pieces
                    if (more2 && ABL != 2) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(4 * (2 * D - 3)) : "memory");   // (ABL 2: only A pieces are in flight here)
                    else if (D == 4 && nt - 1 - t >= 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // the tail: K tiles t + 1 .. nt - 1 in flight, whole
                    else if (D == 4 && nt - 1 - t == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
        pW = pW1;
    };
    int t = 0;
    for (; t + D < nt; ++t) ktile(t, std::true_type{});
    for (; t < nt; ++t) ktile(t, std::false_type{});
#undef MID_M0
#undef MID_PIECE
#undef MID_PIECE_NT
#undef MID_PIECE_W
    if (SPLITK) {                                            // fp32 partial tile -> this split's workspace slice ([split][M_pad][N_pad], pads of 128)
        const int64_t np = (int64_t)tiles_n * 128;
        float* slice = reinterpret_cast<float*>(C) + (int64_t)blockIdx.y * ((int64_t)tiles_m * 128) * np;
        const int rl = m0 + wm * 64 + (lane & 15), c0 = n0 + wn * 64 + (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<floatx4*>(slice + (int64_t)(rl + i * 16) * np + c0 + j * 16) = acc[i][j];
        return;
    }
    __syncthreads();                           // every wave is done with the ring before it becomes the output image
    epilogue_staged<128, 128, 4, 4, 4>(acc, ep, C, ldc, M, N, m0, n0, wm * 64, wn * 64, wave, lane, smem);
}

// ------------------------------------------------------------------------------------------------
// "tall" kernel (round 4): ONE 256-row tile of A against 128 columns of W, for 128 < M <= 256 — the text stack of the 32-token
// student (M = B S = 256) and the prefill of hooked generate.  Two things bound the mid kernel at these row counts (DESIGN.md
// section 5.2.2): its two 128-row tiles each stage and re-read the SAME 128 rows of W, and a wave that issues an LDS-DMA piece is held
// for 60-185 cycles per piece (MI355X_MICROARCH.md, per-instruction constants) during which its SIMD multiplies nothing — eight pieces
// per wave and K tile against 512 cycles of MFMA.  Here
//   * the four MULTIPLYING waves (0-3, one per SIMD) take 128 x 64 each (32 accumulators, pinned to AGPRs) and issue no vector-memory
//     instruction inside the K loop: fragment reads and MFMAs only;
//   * four LOADING waves (4-7, the second wave of each SIMD) issue every LDS-DMA piece, wait for them by counted vmcnt and meet the
//     multiplying waves at one barrier per K tile;
//   * LDS = a ring of three K tiles, each [W unit | A unit 0 | A unit 1] (a unit = 128 rows x 64 K, 128-byte rows, 16-byte chunk
//     c of row r at position c ^ (r & 7)): 144 KiB, one workgroup per CU; K tile t + 3 is issued right behind the barrier that
//     retires K tile t;
//   * staged epilogue (image by the multiplying waves, rows by all eight), or — SPLITK — fp32 partial tiles into
//     [split][256][N_pad] (the mid kernel's slice layout at M_pad = 256).
// Same K order and rounding points as the mid kernel: one pass is bit-identical to it, and so is every split-K slice.
// ------------------------------------------------------------------------------------------------
#define TALL_UNIT 16384
#define TALL_KT (3 * TALL_UNIT)
#define TALL_LDS (3 * TALL_KT)
#define TALL_TK 0.78       // us per K tile of a lone workgroup (the split-K plan's constant: 256 x 12288 x 4096 one pass 54.3 us, two splits 45.4; 22016: 59.9 / 84.7 - tools/tall_bench.py)
template <int SPLITK>
__global__ __launch_bounds__(512)
void gemm_bf16_tall_k(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw,
                      void* __restrict__ C, int64_t ldc, int M, int N, int K, int tiles_n, GemmEpi ep, int per) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 3 K tiles x 48 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = (int)blockIdx.x * 128;
    const int kt0 = SPLITK ? (int)blockIdx.y * per : 0;                       // first 64-deep K tile of this workgroup
    const int nt = SPLITK ? min(per, K / 64 - kt0) : K / 64;                  // >= 2, host-guaranteed
    typedef __attribute__((address_space(3))) char* lds_ptr;
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    typedef const __attribute__((address_space(3))) bf16x8* lds_fptr;
    auto nxt = [](int s) { return s == 2 * TALL_KT ? 0 : s + TALL_KT; };

    if (wave >= 4) {
        // ---- loading waves: wave 4 + p owns rows 32 p .. 32 p + 31 of every unit
        const int p = wave - 4;
        const auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (int64_t)kt0 * 64), 0, 0xFFFFFFFF, 0x00020000);
        const auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)n0 * ldw + (int64_t)kt0 * 64), 0, 0xFFFFFFFF, 0x00020000);
        // piece (unit, q): rows 32 p + 8 q + lane / 8 of the unit's 128; LDS position lane % 8 holds source chunk (lane % 8) ^ (row % 8);
        // rows past M / N re-read the last valid row (those outputs are never stored)
        int offA[2][4], offW[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = p * 32 + q * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ (row & 7);
            offW[q] = min(row, N - 1 - n0) * (int)ldw * 2 + chunk * 16;
            offA[0][q] = min(row, M - 1) * (int)lda * 2 + chunk * 16;
            offA[1][q] = min(row + 128, M - 1) * (int)lda * 2 + chunk * 16;
        }
        const lds_ptr ring_w = (lds_ptr)smem + p * 4096;
        auto issue = [&](int t, int slot) {                  // this wave's 12 pieces of K tile t: the W unit, then the two A units
#pragma unroll
            for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, ring_w + slot + q * 1024, 16, offW[q], t * 128, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, ring_w + slot + (1 + h) * TALL_UNIT + q * 1024, 16, offA[h][q], t * 128, 0, 0);
        };
        issue(0, 0);
        issue(1, TALL_KT);
        if (nt > 2) { issue(2, 2 * TALL_KT); asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // K tile 0 published
        int s = 0;                                           // ring slot of K tile t
        for (int t = 0; t < nt; ++t) {
            // in flight, oldest first: K tiles t + 1 and t + 2 [where they exist]: retire K tile t + 1
            if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // K tile t + 1 published; every multiplying wave is done with K tile t
            if (t + 3 < nt) issue(t + 3, s);
            s = nxt(s);
        }
    } else {
        // ---- multiplying waves
        const int wm = wave >> 1, wn = wave & 1;
        const lds_cptr ring = (lds_cptr)smem;
        floatx4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        // fragment (row i * 16 + lane % 16 of the wave's rows, K chunk kk * 4 + lane / 16): the swizzle term is (lane % 16) % 8 = lane % 8 for every i
        const int fo0 = (lane & 15) * 128 + ((((lane >> 4)) ^ (lane & 7)) << 4);
        const int fo1 = (lane & 15) * 128 + ((((lane >> 4) + 4) ^ (lane & 7)) << 4);
        const int rdW0 = fo0 + wn * 8192, rdW1 = fo1 + wn * 8192;
        const int rdA0 = fo0 + (1 + wm) * TALL_UNIT, rdA1 = fo1 + (1 + wm) * TALL_UNIT;
        bf16x8 fa0[8], fw0[4], fa1[8], fw1[4];

        __builtin_amdgcn_s_barrier();                        // K tile 0 published
#pragma unroll
        for (int j = 0; j < 4; ++j) fw0[j] = *(lds_fptr)(ring + rdW0 + j * 2048);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa0[i] = *(lds_fptr)(ring + rdA0 + i * 2048);

        int s = 0;                                           // ring slot of the K tile being multiplied
        // One K tile: ONE loop body for every K tile.  (The last one's step 1 reads the first half of a K tile that does not exist -
        // a slot nobody writes any more - into registers nobody uses.  With the last K tile as a second copy of the body the
        // register allocator moved accumulators between the two copies through arch VGPRs on the loop's exit edge, right behind
        // the loop's last inline-asm MFMAs, whose write-back it cannot see: stale reads.  The wait states that close the body keep
        // any such compiler-made accumulator access behind the loop safe.)
        auto ktile = [&]() {
            const int s1 = nxt(s);
            // ---- step 0: MFMAs on set 0; reads of (t, second half) into set 1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            {
                const lds_cptr pw = ring + s + rdW1;
                const lds_cptr pa = ring + s + rdA1;
                static_for<0, 32>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    constexpr int i = m >> 2, j = m & 3;
                    if constexpr (m < 12) {
                        if constexpr (m < 4) fw1[m] = *(lds_fptr)(pw + m * 2048);
                        else fa1[m - 4] = *(lds_fptr)(pa + (m - 4) * 2048);
                    }
                    QUAD_MFMA(acc[i][j], fw0[j], fa0[i]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            // ---- step 1: MFMAs on set 1; rendezvous; reads of (t + 1, first half) into set 0
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            {
                const lds_cptr pw = ring + s1 + rdW0;
                const lds_cptr pa = ring + s1 + rdA0;
                static_for<0, 32>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    constexpr int i = m >> 2, j = m & 3;
                    if constexpr (m >= 4 && m < 16) {
                        constexpr int r = m - 4;
                        if constexpr (r < 4) fw0[r] = *(lds_fptr)(pw + r * 2048);
                        else fa0[r - 4] = *(lds_fptr)(pa + (r - 4) * 2048);
                    }
                    QUAD_MFMA(acc[i][j], fw1[j], fa1[i]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (m == 3) {
                        __builtin_amdgcn_s_barrier();        // K tile t + 1 published; K tile t (read in full) handed back to the loading waves
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            }
            asm volatile("s_nop 7\n\ts_nop 4" ::: "memory");     // the last MFMAs' results (inline asm: no hazard tracking by the compiler)
            __builtin_amdgcn_sched_barrier(0);
            s = s1;
        };
        for (int t = 0; t < nt; ++t) ktile();
        if (SPLITK) {                                        // fp32 partial tile -> this split's workspace slice ([split][256][N_pad])
            const int64_t np = (int64_t)tiles_n * 128;
            float* slice = reinterpret_cast<float*>(C) + (int64_t)blockIdx.y * 256 * np;
            const int rl = wm * 128 + (lane & 15), c0 = n0 + wn * 64 + (lane >> 4) * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<floatx4*>(slice + (int64_t)(rl + i * 16) * np + c0 + j * 16) = acc[i][j];
        } else {
            // (no wave reads the ring after the last K tile's barrier, and no piece is in flight: it becomes the output image)
            epilogue_image<128, 8, 4>(acc, ep, M, N, 0, n0, wm * 128, wn * 64, lane, smem);
        }
    }
    if (SPLITK) return;
    __syncthreads();                                         // the image is whole
    epilogue_rows_inlined<256, 128, 8>(ep, C, ldc, M, N, 0, n0, wave, lane, smem);
}

// ------------------------------------------------------------------------------------------------
// 128 x 128 x 64, 4 waves, register staged (general shapes)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2)
void gemm_bf16_tile128_k(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw,
                         void* __restrict__ C, int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n, GemmEpi ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [2 stages][A 16 KiB | W 16 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tm, tn;
    tile_coords(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * 128, n0 = tn * 128;

    // buffer descriptors: out-of-range rows read as 0; K tail handled through the offset
    const int64_t a_bytes = (int64_t)(M - m0) * lda * 2, w_bytes = (int64_t)(N - n0) * ldw * 2;
    const int lim = 0x7ffffff0;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (int64_t)m0 * lda), 0, (int)(a_bytes < lim ? a_bytes : lim), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)n0 * ldw), 0, (int)(w_bytes < lim ? w_bytes : lim), 0x00020000);

    const int srow = tid >> 3, schunk = tid & 7;      // thread -> (row = tid/8 + 32*i, chunk = tid%8)
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
    auto store_tile = [&](int st) {
        char* sa = smem + st * 32768;
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
    epilogue_staged<128, 128, 4, 4, 4>(acc, ep, C, ldc, M, N, m0, n0, wm * 64, wn * 64, wave, lane, smem);
}

// ------------------------------------------------------------------------------------------------
// Split-K for skinny GEMMs (M <= 256: the student pass of training, decode steps of generate).  With one or two tile
// rows a 128 x 128 grid has 32-172 workgroups and each streams its whole weight slab alone: 134 us average on the
// student's shapes against a 7-36 us weight-bandwidth floor.  Here blockIdx.y cuts K into `splits` ranges; every
// workgroup writes its fp32 partial tile to its own slice of a caller-provided workspace ([split][M_pad][N_pad], no
// atomics -> bit-reproducible), and a second kernel sums the slices in a fixed order and runs the usual epilogue.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2)
void gemm_bf16_splitk_k(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw,
                        float* __restrict__ ws, int M, int N, int K, int tiles_m, int tiles_n, int kt_per_split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [2 stages][A 16 KiB | W 16 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int64_t a_bytes = (int64_t)(M - m0) * lda * 2, w_bytes = (int64_t)(N - n0) * ldw * 2;
    const int lim = 0x7ffffff0;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (int64_t)m0 * lda), 0, (int)(a_bytes < lim ? a_bytes : lim), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)n0 * ldw), 0, (int)(w_bytes < lim ? w_bytes : lim), 0x00020000);
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
    auto store_tile = [&](int st) {
        char* sa = smem + st * 32768;
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
    const int kt0 = blockIdx.y * kt_per_split, kt1 = min(nkt, kt0 + kt_per_split);
    if (kt0 < kt1) {
        load_tile(kt0);
        store_tile(0);
        __syncthreads();
        const int frow = lane & 15, fchunk = lane >> 4;
        for (int kt = kt0; kt < kt1; ++kt) {
            const int cur = (kt - kt0) & 1;
            if (kt + 1 < kt1) load_tile(kt + 1);
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
            if (kt + 1 < kt1) store_tile(cur ^ 1);
            __syncthreads();
        }
    }
    // partial tile -> this split's workspace slice, accumulator layout: lane owns 4 consecutive columns of 16 rows
    const int64_t np = (int64_t)tiles_n * 128;
    float* slice = ws + (int64_t)blockIdx.y * ((int64_t)tiles_m * 128) * np;
    const int rl = m0 + wm * 64 + (lane & 15), c0 = n0 + wn * 64 + (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<floatx4*>(slice + (int64_t)(rl + i * 16) * np + c0 + j * 16) = acc[i][j];
}

// ------------------------------------------------------------------------------------------------
// Weight-streaming GEMM for M <= 32 (decode steps of hooked generate: beams x questions rows, one token each).  Such a GEMM
// is a read of the weight matrix: 15.6 GB per decode step at Idefics-9B, a 3.5 ms floor at HBM rate; through the 128 x 128
// tile kernels (32-172 workgroups, three quarters of every A tile padding) it took ~10 ms.  Here:
//   * grid = ceil(N / 64) column blocks x `splits` K ranges (>= 512 workgroups); a wave owns 16 output columns;
//   * W goes global -> VGPR directly (read once, by one wave: no LDS round trip), 16 bytes per lane = one MFMA A-fragment,
//     eight loads in flight per wave before the first is consumed;
//   * the few activation rows of the K range are staged once per workgroup in LDS (row stride + 16 B: conflict-free
//     ds_read_b128 fragments), zero-padded to 16 / 32 rows;
//   * fp32 partials go to the caller's workspace as [split][32 rows][N padded to 128] and skinny_finalize_k adds them
//     in order and runs the usual epilogue (deterministic, no atomics).
// ------------------------------------------------------------------------------------------------
#define SKINNY_KR_MAX 1024          // K elements per split: 32 rows x 1024 x 2 B + padding = 66 KB of LDS

// Finalize of the skinny path: sum the fp32 slices in order and run the epilogue for <= 32 rows, four consecutive output
// columns per thread.  Same rounding points, in the same order, as epilogue_staged + epilogue_rows_generic (y = bf16(acc + bias);
// activation; SwiGLU pairing; row gate; gate scale; residual in the stream dtype) — the general finalize kernel walks
// 128 x 128 tiles through an LDS image, 17 us per call for 24 rows; this one is a few microseconds.
template <bool SC1>       // SC1: the slices were stored write-through by other workgroups of THIS launch and are read past this CU's L1
__device__ __forceinline__ void skinny_finalize_item(const float* __restrict__ ws, void* __restrict__ C, int64_t ldc, int N, int64_t np, int splits,
                                                     const GemmEpi& ep, int slice_rows, int m, int c) {      // row m, output columns c .. c + 3
    const int n_out = ep.swiglu ? N >> 1 : N;
    const int nv = min(4, n_out - c);
    const int64_t slice = (int64_t)slice_rows * np;
    const auto rsws = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws), 0, 0xFFFFFFFF, 0x00020000);
    auto sum4 = [&](int col) -> floatx4 {
        if constexpr (SC1) {
            const uint32_t o = (uint32_t)((int64_t)m * np + col) * 4u, st = (uint32_t)slice * 4u;
            u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsws, o, 0, 16);                               // aux 16 = sc1
            floatx4 v = *reinterpret_cast<floatx4*>(&r);
#pragma unroll 4
            for (int sp = 1; sp < splits; ++sp) {
                r = __builtin_amdgcn_raw_buffer_load_b128(rsws, o + (uint32_t)sp * st, 0, 16);
                v += *reinterpret_cast<floatx4*>(&r);                                                       // fixed order
            }
            return v;
        } else {
            const float* p = ws + (int64_t)m * np + col;
            floatx4 v = *reinterpret_cast<const floatx4*>(p);
#pragma unroll 4
            for (int sp = 1; sp < splits; ++sp) v += *reinterpret_cast<const floatx4*>(p + sp * slice);     // fixed order
            return v;
        }
    };
    float y[4];
    if (!ep.swiglu) {
        const floatx4 a = sum4(c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float b = (ep.bias && e < nv) ? bf2f(ep.bias[c + e]) : 0.f;
            y[e] = rbf(a[e] + b);
            if (ep.act) y[e] = rbf(act_apply(y[e], ep.act));
        }
    } else {
        const int pc = (c >> 4) * 32 + (c & 15);                    // packed gate columns; the matching up columns sit 16 further
        const floatx4 gsum = sum4(pc), usum = sum4(pc + 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = rbf(rbf(silu_fast(rbf(gsum[e]))) * rbf(usum[e]));
    }
    if (ep.row_gate && ep.row_gate[m] == 0.0f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = 0.f;
    }
    if (ep.use_scale) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = rbf(ep.scale * y[e]);
    }
    if (ep.residual) {
        if (ep.residual_dtype == LICV_F32) {
            const float* rp = reinterpret_cast<const float*>(ep.residual) + (int64_t)m * ep.ld_res + c;
            for (int e = 0; e < nv; ++e) y[e] = rp[e] + y[e];
        } else {
            const bf16_t* rp = reinterpret_cast<const bf16_t*>(ep.residual) + (int64_t)m * ep.ld_res + c;
            for (int e = 0; e < nv; ++e) y[e] = rbf(bf2f(rp[e]) + y[e]);
        }
    }
    if (ep.out_dtype == LICV_F32) {
        float* cp = reinterpret_cast<float*>(C) + (int64_t)m * ldc + c;
        if (nv == 4) *reinterpret_cast<floatx4*>(cp) = floatx4{y[0], y[1], y[2], y[3]};
        else for (int e = 0; e < nv; ++e) cp[e] = y[e];
    } else {
        bf16_t* cp = reinterpret_cast<bf16_t*>(C) + (int64_t)m * ldc + c;
        if (nv == 4) *reinterpret_cast<uint2*>(cp) = uint2{pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
        else for (int e = 0; e < nv; ++e) cp[e] = f2bf(y[e]);
    }
}

__global__ __launch_bounds__(256)
void skinny_finalize_k(const float* __restrict__ ws, void* __restrict__ C, int64_t ldc, int M, int N, int64_t np, int splits, GemmEpi ep,
                       int slice_rows) {              // rows a split's slice holds: 32 (weight-streaming kernel) or M padded to 128 (128-tile route)
    const int n_out = ep.swiglu ? N >> 1 : N;
    const int groups = (n_out + 3) >> 2;
    const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (item >= (int64_t)M * groups) return;
    const int m = (int)(item / groups), c = (int)(item - (int64_t)m * groups) * 4;
    skinny_finalize_item<false>(ws, C, ldc, N, np, splits, ep, slice_rows, m, c);
}

template <int MB, int NT = 0>       // 16-row blocks of A: 1 (M <= 16) or 2 (M <= 32); NT = 1: the weight stream is loaded non-temporal (knob 11)
__global__ __launch_bounds__(256, 3)       // three workgroups per CU (3 x 52 KB of LDS at M = 24): at most 168 registers
void gemm_bf16_skinny_k(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw, float* __restrict__ ws,
                        int M, int N, int K, int steps_per_split, int64_t np, unsigned* __restrict__ tickets, void* __restrict__ C, int64_t ldc, GemmEpi ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int nsteps = (K + 31) / 32;
    const int s0 = blockIdx.y * steps_per_split, s1 = min(nsteps, s0 + steps_per_split);
    const int ns = s1 - s0;
    const int kbase = s0 * 32, kr = ns * 32;
    const int xstr = kr * 2 + 16;                                    // LDS row stride in bytes
    const int n = blockIdx.x * 64 + wave * 16 + fr;
    // The weight stream is software-pipelined in two register sets of UN fragments: set (s + UN) is requested before set s is
    // multiplied, and the FIRST set before the activations are staged (it does not depend on them).  Every load is a BUFFER load
    // outside any branch; a fragment that does not exist (a step past this split's range, k >= K, a row >= N) gets an out-of-range
    // offset and comes back as zeros.  With `cond ? *p : 0` the compiler put each load in an exec-masked block of its own and its
    // waitcnt pass then drained the queue (s_waitcnt vmcnt(0)) in front of the first MFMA of every set - the set just requested
    // included, so only one set was ever in flight and its whole latency exposed: 2.7-3.0 TB/s cold, where the same access shape
    // with counted waits streams 5.5 TB/s (tools/stream_probe.py).
    constexpr int UN = 8;
    constexpr uint32_t OOB = 0xFFFFFFFFu;
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)blockIdx.x * 64 * ldw), 0, 0xFFFFFFFF, 0x00020000);
    const uint32_t wrow = (uint32_t)((wave * 16 + fr) * (int)ldw + kbase + fq * 8) * 2u;       // bytes from the workgroup's first row
    const bool row_ok = n < N;
    u32x4 wf[2][UN];
    auto loadw = [&](int s, u32x4 (&dst)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = kbase + (s + u) * 32 + fq * 8;
            const bool ok = (int)row_ok & (int)(s + u < ns) & (int)(k < K);
            dst[u] = __builtin_amdgcn_raw_buffer_load_b128(rsw, ok ? wrow + (uint32_t)(s + u) * 64u : OOB, 0, NT ? 2 : 0);     // aux 2 = nt
        }
    };
    // ---- activations of this K range -> LDS (zero rows past M, zero columns past K), and the first TWO weight sets.
    // Order of issue: the activation loads (L2 hits), then both weight sets, then the LDS writes.  Vector-memory operations retire in
    // order: with the weight set requested first (the version before) the activations' wait drained it - a whole HBM latency before the
    // first LDS write, and the second set was not requested until the barrier behind the writes.  Now the writes wait for the
    // activations alone (counted vmcnt, 16 weight loads left in flight) and two sets per wave are on their way while the image is built.
    const auto rsa = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0xFFFFFFFF, 0x00020000);
    const int chunks = kr / 8;
    constexpr int XU = 13;                                          // activation loads in flight per thread: one batch covers 25 rows (M = 24 + the zero row) x 1024 K; more would not fit 168 registers beside the two weight sets
    // rows 0 .. M - 1 and ONE zero row (index M) that every fragment row >= M reads: 25 rows instead of 32 at M = 24 is a third
    // workgroup per CU (3 x 52 KB of LDS), and 768 workgroups are one round instead of one and a half
    const int xrows = M + 1;
    const int nitem = xrows * chunks;
    u32x4 v[XU];
    // item c = row * chunks + ch of the image, c = tid + 256 j: (row, ch) walks by a fixed step - one division per thread, not one per item
    // (thirty-two 32-bit divisions were ~1000 VALU instructions at the head of every workgroup).  No branch around a load or a write: an
    // item past the image is a zero (out-of-range load) written into the zero row.
    const int q256 = 256 / chunks, r256 = 256 - q256 * chunks;
    int row_s = tid / chunks, ch_s = tid - row_s * chunks;          // first item of the batch being staged
    auto xload = [&]() {
        int row = row_s, ch = ch_s;
#pragma unroll
        for (int j = 0; j < XU; ++j) {
            const int k = kbase + ch * 8;
            const bool ok = (int)(row < M) & (int)(k < K);
            v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsa, ok ? (uint32_t)(row * (int)lda + k) * 2u : OOB, 0, 0);
            ch += r256; row += q256;
            if (ch >= chunks) { ch -= chunks; ++row; }
        }
    };
    auto xstore = [&]() {
        int row = row_s, ch = ch_s;
#pragma unroll
        for (int j = 0; j < XU; ++j) {
            *reinterpret_cast<u32x4*>(smem + min(row, M) * xstr + ch * 16) = v[j];
            ch += r256; row += q256;
            if (ch >= chunks) { ch -= chunks; ++row; }
        }
        row_s = row; ch_s = ch;
    };
    xload();
    loadw(0, wf[0]);
    loadw(UN, wf[1]);
    xstore();
    for (int c0 = 256 * XU; c0 < nitem; c0 += 256 * XU) { xload(); xstore(); }      // (M > 25 at 1024 K only)
    __syncthreads();
    const char* xp[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) xp[mb] = smem + min(mb * 16 + fr, M) * xstr + fq * 16;
    floatx4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = floatx4{0.f, 0.f, 0.f, 0.f};
    auto mul = [&](int s, const u32x4 (&src)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int so = s + u < ns ? (s + u) * 64 : 0;           // steps past the range carry zero weights: any in-range activations do
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xp[mb] + so);
                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&src[u]), xf, acc[mb], 0, 0, 0);
            }
        }
    };
    for (int s = 0; s < ns; s += 2 * UN) {
        mul(s, wf[0]);
        loadw(s + 2 * UN, wf[0]);                                   // (sets past the range: all lanes out of range, no traffic)
        mul(s + UN, wf[1]);
        loadw(s + 3 * UN, wf[1]);
    }
    // lane holds rows m = mb*16 + fr, columns n0 + fq*4 .. +3  (W was the A operand)
    float* slice = ws + (int64_t)blockIdx.y * 32 * np;
    const int64_t c0 = (int64_t)blockIdx.x * 64 + wave * 16 + fq * 4;
    if (!tickets) {                                                 // the caller runs skinny_finalize_k behind this launch
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
            *reinterpret_cast<floatx4*>(slice + (int64_t)(mb * 16 + fr) * np + c0) = acc[mb];
        if (MB == 1) *reinterpret_cast<floatx4*>(slice + (int64_t)(16 + fr) * np + c0) = floatx4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    // ---- in-launch reduction: the workgroup that draws the last ticket of its 64-column tile sums the splits' slabs (in split
    // order, as skinny_finalize_k does: same bits) and runs the epilogue - one launch less per projection of a decode step.
    // Hand-off in the write-through form of cdna_hip_programming.md 6/G16 (R1), placement-independent: every slab store carries sc1,
    // every storing wave drains its stores, the workgroup meets, one lane takes the ticket (relaxed, agent scope); the last
    // arriver reads every slab with sc1 loads (past its CU's L1).  The first version used plain stores and an agent-scope RELEASE
    // fence per workgroup: 768 L2 write-backs per launch, 63 us where the two-launch form took 32.  The flag travels through the
    // (now idle) activation image: no second LDS object.
    {
        const auto rss = __builtin_amdgcn_make_buffer_rsrc(slice, 0, 0xFFFFFFFF, 0x00020000);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(&acc[mb]);
            __builtin_amdgcn_raw_buffer_store_b128(v, rss, (uint32_t)((int64_t)(mb * 16 + fr) * np + c0) * 4u, 0, 16);
        }
        if (MB == 1) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rss, (uint32_t)((int64_t)(16 + fr) * np + c0) * 4u, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(&tickets[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == gridDim.y - 1 ? 1 : 0;
        if (last) __hip_atomic_store(&tickets[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // all tickets are zero again between launches
        *reinterpret_cast<volatile int*>(smem) = last;
    }
    __syncthreads();
    if (*reinterpret_cast<volatile int*>(smem) == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");          // (no instruction: keeps the compiler from moving the slab loads above the ticket)
    const int n_out = ep.swiglu ? N >> 1 : N;
    const int gpt = ep.swiglu ? 8 : 16;                             // 4-column groups of output per 64 weight rows (SwiGLU pairs them: 32 outputs)
    for (int item = tid; item < M * gpt; item += 256) {
        const int m = item / gpt, c = ((int)blockIdx.x * gpt + item % gpt) * 4;
        if (c < n_out) skinny_finalize_item<true>(ws, C, ldc, N, np, (int)gridDim.y, ep, 32, m, c);
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

// The experiments (csrc/lab/gemm_experiments.hip) are NOT part of this library: they are built into liblicv_hip_lab.so, which only
// tests and tools load, and which registers its three entry points here when it is loaded (licv_lab_register, include/licv_hip_lab.h).
// Without the lab library a licv_gemm_select() value that names one of its kernels is an error, never a silent fallback.
typedef int (*lab_launch_fn)(int which, const GemmArgs* g);
typedef int (*lab_knob_fn)(int knob, int value);
typedef int (*lab_ts_fn)(void* dev_buffer);
static lab_launch_fn g_lab_launch = nullptr;
static lab_knob_fn g_lab_knob = nullptr;
static lab_ts_fn g_lab_ts = nullptr;
extern "C" int licv_lab_register(void* launch, void* knob, void* timestamps) {
    g_lab_launch = (lab_launch_fn)launch; g_lab_knob = (lab_knob_fn)knob; g_lab_ts = (lab_ts_fn)timestamps;
    return LICV_OK;
}

// timing-only instrumentation (tools/gemm_phases.py, gemm_segments.py, gemm_series.py): a device buffer the diagnostic builds
// (gemm_bf16_lean_k<0, 7 | 8 | 9>, the experiments' ping-pong kernel) write their stamps to
extern "C" int licv_gemm_debug_timestamps(void* dev_buffer) {
    const int rc = set_dbg_ts(dev_buffer);
    return (rc != LICV_OK || !g_lab_ts) ? rc : g_lab_ts(dev_buffer);
}

static int g_fp8_flow64 = 1;       // knob 8: 0 = fp8 GEMMs stay on the 8-wave kernel with the 32-deep fp8 MFMA (A/B, tests)
static int g_skinny_inlaunch = 0;   // knob 7: 1 = the skinny kernel reduces over its splits inside its own launch (last workgroup of a tile).  Off: measured cold at
                                    // M = 24 it only moves the finalize's ~5 us into the producer's tail (12288 x 4096: 36.1 vs 32.3 us with the separate launch;
                                    // 4096 x 4096: 16.9 vs 18.3), see tools/stream_bench.py; kept as a tested alternative
static int g_pp_group = 0;      // experiment knob: tile-rows per XCD patch group (0 = heuristic)
static int g_splitk_enabled = 1;
static int g_big_tiles = 136;   // knob 6: fewest 256 x 256 tiles for which the 256-tile kernels are taken (see route_256)
static int g_force_splits = 0;  // knob 5 (A/B timing only): split count of the 128-tile route, 0 = the plan's own choice
static int g_mid_depth = 0;     // knob 10: 4 = the 128-tile mid kernel keeps FOUR K tiles in flight (nine-pair ring) wherever every range has 4 K tiles
// Measured (tools/mid_depth.py, M = 256, cold): bit-identical and within +-5 % of the five-pair ring on every shape with <= 256
// workgroups, 13-15 % slower where two workgroups per CU were possible (22016 / 32002 columns): the K-tile period of a lone
// workgroup is not the latency of its pieces.  Kept as an experiment, off by default.
static bool mid_deep(int64_t M, int64_t min_ktiles) {
    (void)M;
    return g_mid_depth == 4 && min_ktiles >= 4;
}
static int g_nt_weights = 0;   // knob 11: bit 0 = the 128-tile mid kernel's W pieces non-temporal, bit 1 = the skinny kernel's weight stream (A/B timing; bit-identical results)
static int g_tall = 1;         // knob 12: 1 = wide outputs at 128 < M <= 256 take the 256 x 128 tall kernel (one pass and split-K producer); 0 = the mid kernel (A/B)
static int g_force_kernel = 0;  // licv_gemm_select
static bool tall_ok(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw) {
    return M > 128 && M <= 256 && N >= 128 && K % 64 == 0 && K >= 128 && lda * 510 < (1ll << 31) && ldw * 510 < (1ll << 31);
}
// Where the tall kernel is the route (select 71: wherever it can run).  Measured cold at M = 256 (tools/tall_bench.py,
// profiles/r04_tall_bench_v2_loader_waves.txt, us tall | mid at each one's best split count): 22016 x 4096 SwiGLU 60.0 | 73.7, 32002 x 4096
// 78.0 | 86.1, 28672 x 4096 74.4 | 80.1, 12288 x 4096 45.4 | 49.1; but 8192 x 4096 34.4 | 33.2, 4096 x 4096 26.2 | 22.7, 4096 x 11008
// 44.0 | 42.5, 2048 x 4096 20.5 | 18.1: with few tile columns the 128-tile grid fills the chip with half the split-K slices.
static bool tall_route(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw) {
    if (!tall_ok(M, N, K, lda, ldw)) return false;
    if (g_force_kernel == 71) return true;
    return g_force_kernel == 0 && g_tall && N >= 10240;
}
static int g_mid_ablate = 0;    // knob 9 (timing only, WRONG RESULTS): the 128-tile mid kernel without its A pieces (1) / W pieces (2)
static int g_flow_default = 1;  // auto mode takes the flow kernels where they are eligible (knob 2 of licv_gemm_experiment; 0 = staged epilogues only)
// A/B timing knobs:
//   knob 0: (experiments' ping-pong kernel) per-XCD first-round start stagger, percent of an eighth of the estimated tile time
//   knob 1: tile-rows per XCD patch group (0 = the default 8)
//   knob 2: 0 = never take a flow kernel by default;  knob 4: 0 = licv_gemm_splitk_plan always answers "one pass" (the
//   batch-independence tests switch split-K off for every caller, the native layer runner included)
//   knob 7: 1 = the skinny kernel reduces over its splits in its own launch (default 0: a separate finalize launch)
//   knob 8: 0 = fp8 GEMMs never take the 4-wave kernel on the 128-deep MFMA
//   knob 9: timing-only ablation of the mid kernel's operand stream (1 = no A pieces, 2 = no W pieces; results are wrong)
//   knob 11: bit 0 / bit 1 = non-temporal weight loads in the mid / skinny kernel (default set below)
//   knob 12: 0 = row counts 129-256 stay on the 128-tile mid kernel instead of the 256 x 128 tall kernel
//   knob 10: 4 = the mid kernel keeps four K tiles in flight (nine-pair ring) wherever it fits; anything else = the five-pair ring
extern "C" int licv_gemm_experiment(int knob, int value) {
    if (knob == 0) return g_lab_knob ? g_lab_knob(0, value) : licv_set_error(LICV_E_UNSUPPORTED, "gemm_experiment: knob 0 belongs to liblicv_hip_lab.so, which is not loaded");
    else if (knob == 1) g_pp_group = value; else if (knob == 2) g_flow_default = value;
    else if (knob == 4) g_splitk_enabled = value;
    else if (knob == 5) g_force_splits = value;
    else if (knob == 6) g_big_tiles = value;
    else if (knob == 7) g_skinny_inlaunch = value;
    else if (knob == 8) g_fp8_flow64 = value;
    else if (knob == 9) g_mid_ablate = value;
    else if (knob == 10) g_mid_depth = value;
    else if (knob == 11) g_nt_weights = value;
    else if (knob == 12) g_tall = value;
    else return licv_set_error(LICV_E_BADARG, "gemm_experiment: unknown knob %d", knob);
    return LICV_OK;
}
static int g_num_cus = 256;        // persistent grid size (queried once)
// 0 auto; 1 tile128; 20 the 8-wave flow kernel where eligible (else lean); 22-27 lean variants; 40-42 quad64 variants (staged
// epilogue); 60 the 4-wave flow64 kernel where eligible; 70 the 128-tile mid kernel at any M; 71 the 256 x 128 tall kernel wherever it can run (129-256 rows); every other value names a kernel of
// gemm_experiments.hip
extern "C" int licv_gemm_select(int which) { g_force_kernel = which; return LICV_OK; }

// The flow kernel's counted s_waitcnt vmcnt(N) assume that the ONLY vector-memory operations a wave issues are its LDS-DMA
// pieces and its epilogue stores.  A register spill would add scratch loads/stores to that queue and silently break the
// count, so the kernel is used only if the code object reports no private segment for every instantiation.
static bool flow_scratch_free() {
    static int ok = -1;
    if (ok < 0) {
        ok = 1;
        const void* fns[6] = {(const void*)gemm_bf16_flow_k<0>, (const void*)gemm_bf16_flow_k<1>, (const void*)gemm_bf16_flow_k<2>,
                              (const void*)gemm_bf16_flow_k<3>, (const void*)gemm_bf16_flow_k<4>, (const void*)gemm_bf16_flow_k<5>};
        for (const void* f : fns) {
            hipFuncAttributes at;
            if (hipFuncGetAttributes(&at, f) != hipSuccess || at.localSizeBytes != 0) ok = 0;
        }
    }
    return ok == 1;
}

static bool flow64_scratch_free() {
    static int ok = -1;
    if (ok < 0) {
        ok = 1;
        const void* fns[6] = {(const void*)gemm_bf16_flow64_k<0>, (const void*)gemm_bf16_flow64_k<1>, (const void*)gemm_bf16_flow64_k<2>,
                              (const void*)gemm_bf16_flow64_k<3>, (const void*)gemm_bf16_flow64_k<4>, (const void*)gemm_bf16_flow64_k<5>};
        for (const void* f : fns) {
            hipFuncAttributes at;
            if (hipFuncGetAttributes(&at, f) != hipSuccess || at.localSizeBytes != 0) ok = 0;
            (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * Q64_UNIT);
        }
    }
    return ok == 1;
}

// per epilogue family: an instantiation with a private segment (spills would sit in the counted vmcnt queue) is not used
static bool fp8_flow64_scratch_free(int epi) {
    static int ok[6] = {-1, -1, -1, -1, -1, -1};
    if (ok[0] < 0) {
        const void* fns[6] = {(const void*)gemm_fp8_flow64_k<0>, (const void*)gemm_fp8_flow64_k<1>, (const void*)gemm_fp8_flow64_k<2>,
                              (const void*)gemm_fp8_flow64_k<3>, (const void*)gemm_fp8_flow64_k<4>, (const void*)gemm_fp8_flow64_k<5>};
        for (int i = 0; i < 6; ++i) {
            hipFuncAttributes at;
            ok[i] = (hipFuncGetAttributes(&at, fns[i]) == hipSuccess && at.localSizeBytes == 0) ? 1 : 0;
            (void)hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 10 * Q64_UNIT);
        }
    }
    return ok[epi] == 1;
}

// 1: the 8-wave flow kernel is usable; 2: the 4-wave flow64 kernel as well (bit 1); 4: its fp8 form (bit 2)
extern "C" int licv_gemm_flow_available(void) { return (flow_scratch_free() ? 1 : 0) | (flow64_scratch_free() ? 2 : 0) | (fp8_flow64_scratch_free(0) && fp8_flow64_scratch_free(4) ? 4 : 0); }

// Which tile size a dense GEMM takes.  The 256 x 256 kernels (flow64 / quad64) need enough tiles to fill the 256 CUs: below
// g_big_tiles of them (the vision tower on a few images: 2056 rows = 9 tile rows; Idefics2's 1-shot text stack) the 128 x 128
// route — the mid kernel, two workgroups per CU, split-K where the tiles are still too few — is faster.  knob 6 moves the bar.
// Measured with the flow64 kernel (tools/route_bench.py, N = 4096, K = 4096 / 1280, us 256-tile | 128-tile): 112 tiles 91 | 66, 128: 83 | 68,
// 144: 85 | 98, 192: 86 | 102, 256: 100 | 135 - one partial round of 256-tiles costs about the same whatever its fill, so from ~136
// tiles on it is ahead; 272: 162 | 147, 288: 162 | 149, 320: 161 | 153, 384: 165 | 185 - a second round that is less than a third
// full loses to the 128-tiles (two workgroups per CU even the tail out).
static bool route_256(int64_t M, int64_t N, int64_t K) {
    if (K % BK != 0 || K < 128 || M < 512 || N < 256) return false;
    const int64_t t = ((M + 255) / 256) * ((N + 255) / 256);
    if (t < g_big_tiles) return false;
    if (t <= 256 || t >= 768) return true;
    return 20 * t >= 13 * ((t + 255) / 256 * 256);
}

// The 256-tile kernels address C (and a residual) with 32-bit byte offsets from the tensor's base, which caps a launch at 2 GiB of
// output: SigLIP's fc1 at 264 images x 972 patches is 256608 x 4352 bf16 = 2.2 GB, and used to fall back to the 8-wave / staged kernels
// for that one projection (fp8: 2.9 ms against ~1.1 ms).  Taller outputs run as row blocks (multiples of 256 rows) instead - the same
// tiles, the same arithmetic per tile: bit-identical.  Returns the rows per block, or 0 when one launch does.
static int64_t rows_per_launch(int64_t M, int64_t ldc, int out_dtype, const licv_gemm_epilogue* e) {
    int64_t row_bytes = ldc * (out_dtype == LICV_F32 ? 4 : 2);
    if (e->residual) { const int64_t rb = e->ld_res * (e->residual_dtype == LICV_F32 ? 4 : 2); if (rb > row_bytes) row_bytes = rb; }
    if ((M + 256) * row_bytes < (1ll << 31)) return 0;
    const int64_t cap = (((1ll << 31) - 1) / row_bytes - 256) / 256 * 256;
    if (cap < 512) return 0;
    const int64_t blocks = (M + cap - 1) / cap;
    return ((M + blocks - 1) / blocks + 255) / 256 * 256;             // even blocks (<= cap: cap is itself a multiple of 256)
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
    if (const int64_t rows = rows_per_launch(M, ldc, e->out_dtype, e)) {
        for (int64_t m0 = 0; m0 < M; m0 += rows) {
            licv_gemm_epilogue eb = *e;
            if (e->residual) eb.residual = (const char*)e->residual + m0 * e->ld_res * (e->residual_dtype == LICV_F32 ? 4 : 2);
            if (e->row_gate) eb.row_gate = e->row_gate + m0;
            const int rc = licv_gemm_bf16((const char*)A + m0 * lda * 2, lda, W, ldw, (char*)C + m0 * ldc * (e->out_dtype == LICV_F32 ? 4 : 2), ldc,
                                          M - m0 < rows ? M - m0 : rows, N, K, &eb, stream);
            if (rc != LICV_OK) return rc;
        }
        return LICV_OK;
    }
    GemmEpi ep;
    ep.bias = (const bf16_t*)e->bias_bf16; ep.row_gate = e->row_gate; ep.residual = e->residual;
    ep.residual_dtype = e->residual_dtype; ep.ld_res = e->ld_res; ep.act = e->act; ep.swiglu = e->swiglu;
    ep.use_scale = e->use_scale; ep.scale = e->scale; ep.out_dtype = e->out_dtype;
    ep.a_scale = nullptr; ep.w_scale = nullptr;
    static bool attr_set = false;
    if (!attr_set) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            g_num_cus = cus;
        const int ring = RING_STAGES * RING_STAGE_BYTES;
        (void)hipFuncSetAttribute((const void*)gemm_bf16_tile128_k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_lean_k<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, ring);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_lean_k<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, ring);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_lean_k<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, ring);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_lean_k<0, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, ring);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_lean_k<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, ring);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_lean_k<0, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, ring);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_quad64_k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * Q64_UNIT);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_quad64_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * Q64_UNIT);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_quad64_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * Q64_UNIT);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_flow_k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, ring);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_flow_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ring);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_flow_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, ring);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_flow_k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, ring);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_flow_k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, ring);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_flow_k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, ring);
        attr_set = true;
    }
    const int fk = g_force_kernel;
    const bool can256 = (K % BK == 0) && K >= 128;
    bool big = route_256(M, N, K);
    const bool lean_ok = lda * 510 < (1ll << 31) && ldw * 510 < (1ll << 31);     // 32-bit lane offsets of the DMA sources
    const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)((N + 255) / 256);
    // tile-rows per XCD patch: 8 (a 32-CU XCD then works on an 8 x 4 patch); with <= 6 tile-columns an 8-row group is 40-48
    // tiles and the patch straddles two groups -> 2-row groups keep it compact (measured +6 % at N = 1280, K = 5120)
    const int pp_group = g_pp_group > 0 ? g_pp_group : (tiles_n <= 6 ? 2 : 8);
    // a kernel of the lab library (csrc/lab/gemm_experiments.hip), by number
    const bool product_sel = fk == 0 || fk == 1 || fk == 20 || (fk >= 22 && fk <= 27) || (fk >= 40 && fk <= 42) || fk == 60 || fk == 70 || fk == 71;
    if (!product_sel && K % BK == 0) {
        GemmArgs ga{A, lda, W, ldw, C, ldc, (int)M, (int)N, (int)K, ep, (hipStream_t)stream, g_pp_group, g_num_cus};
        if (!g_lab_launch) return licv_set_error(LICV_E_UNSUPPORTED, "gemm: licv_gemm_select(%d) names a kernel of liblicv_hip_lab.so, which is not loaded", fk);
        if (g_lab_launch(fk, &ga) == 1) { LICV_LAUNCH_CHECK(); return LICV_OK; }
    }
    // flow kernels: epilogues that need only the accumulators (and a bias row), bf16 out, whole waves in or out of N
    // ... or a bf16 residual (EPI 5: the ViT out / fc2 projections, usually in place) with no activation, gate or scale
    const bool flow_res = e->residual && e->residual_dtype == LICV_BF16 && !e->act && !e->swiglu && e->ld_res % 8 == 0 &&
                          (int64_t)(M + 256) * e->ld_res * 2 < (1ll << 31);
    const bool flow_ok = can256 && M >= 512 && N >= 256 && N % 64 == 0 && e->out_dtype == LICV_BF16 && (!e->residual || flow_res) && !e->row_gate &&
                         !e->use_scale && (int64_t)(M + 256) * ldc * 2 < (1ll << 31) && ldc % 8 == 0 && lean_ok &&
                         (!e->bias_bf16 || ((uintptr_t)e->bias_bf16 & 3) == 0);
    // ... except behind a bf16-residual epilogue, where the 128-tile kernel pays its staged epilogue (LDS image, residual rows read back)
    // per tile: SigLIP's out / fc2 projections at 16 x 972 patches (305 tiles) 74 | 65 us and 204 | 188 us (128-tile | 256-tile)
    if (!big && flow_res && can256 && M >= 512 && N >= 256 && N % 128 == 0 && K >= 256) {
        const int64_t t = ((M + 255) / 256) * ((N + 255) / 256);
        big = t > 256 && t < 768;
    }
    const bool use256 = (fk == 1 || fk == 70) ? false : (fk == 0 ? big : can256);
    const bool flow_auto = g_flow_default != 0;
    // flow64: the same epilogue families without the residual one, waves of 128 columns, at least four 64-deep K tiles
    const bool flow64_ok = flow_ok && N % 128 == 0 && K >= 256;
    if (use256 && flow64_ok && (fk == 60 || (fk == 0 && flow_auto)) && flow64_scratch_free()) {
        const dim3 grid(min(tiles_m * tiles_n, g_num_cus)), block(256);
#define FLOW64(E) gemm_bf16_flow64_k<E><<<grid, block, 10 * Q64_UNIT, (hipStream_t)stream>>>( \
            (const bf16_t*)A, lda, (const bf16_t*)W, ldw, (bf16_t*)C, (int)ldc, (int)M, (int)N, (int)K, tiles_m, tiles_n, ep.bias, pp_group)
        if (e->residual)
            gemm_bf16_flow64_k<5><<<grid, block, 10 * Q64_UNIT, (hipStream_t)stream>>>(
                (const bf16_t*)A, lda, (const bf16_t*)W, ldw, (bf16_t*)C, (int)ldc, (int)M, (int)N, (int)K, tiles_m, tiles_n, ep.bias, pp_group,
                (const bf16_t*)e->residual, (int)e->ld_res);
        else if (e->swiglu) FLOW64(4); else if (e->act == 1) FLOW64(1); else if (e->act == 2) FLOW64(2); else if (e->act == 3) FLOW64(3); else FLOW64(0);
#undef FLOW64
        LICV_LAUNCH_CHECK();
        return LICV_OK;
    }
    if (use256 && flow_ok && flow_scratch_free() && (fk == 20 || fk == 60 || (fk == 0 && flow_auto))) {
        const dim3 grid(min(tiles_m * tiles_n, g_num_cus)), block(512);
        const int ring = RING_STAGES * RING_STAGE_BYTES;
#define FLOW(E) gemm_bf16_flow_k<E><<<grid, block, ring, (hipStream_t)stream>>>( \
            (const bf16_t*)A, lda, (const bf16_t*)W, ldw, (bf16_t*)C, (int)ldc, (int)M, (int)N, (int)K, tiles_m, tiles_n, ep.bias, pp_group)
        if (e->residual)
            gemm_bf16_flow_k<5><<<grid, block, ring, (hipStream_t)stream>>>(
                (const bf16_t*)A, lda, (const bf16_t*)W, ldw, (bf16_t*)C, (int)ldc, (int)M, (int)N, (int)K, tiles_m, tiles_n, ep.bias, pp_group,
                (const bf16_t*)e->residual, (int)e->ld_res);
        else if (e->swiglu) FLOW(4); else if (e->act == 1) FLOW(1); else if (e->act == 2) FLOW(2); else if (e->act == 3) FLOW(3); else FLOW(0);
#undef FLOW
    } else if (use256 && lean_ok) {
        const dim3 grid(tiles_m * tiles_n), block(512);
        const int ring = RING_STAGES * RING_STAGE_BYTES;
#define QUAD64(V) gemm_bf16_quad64_k<V><<<grid, dim3(256), 10 * Q64_UNIT, (hipStream_t)stream>>>( \
            (const bf16_t*)A, lda, (const bf16_t*)W, ldw, C, ldc, (int)M, (int)N, (int)K, tiles_m, tiles_n, ep, pp_group)
#define LEAN(V) gemm_bf16_lean_k<0, V><<<grid, block, ring, (hipStream_t)stream>>>( \
            (const bf16_t*)A, lda, (const bf16_t*)W, ldw, C, ldc, (int)M, (int)N, (int)K, tiles_m, tiles_n, ep, pp_group)
        if (fk == 40) QUAD64(0); else if (fk == 41) QUAD64(1); else if (fk == 42) QUAD64(2);
        else if (fk == 22) LEAN(0); else if (fk == 23) LEAN(1); else if (fk == 24) LEAN(2); else if (fk == 25) LEAN(9); else if (fk == 26) LEAN(8);
        else if (fk == 27) LEAN(7);
        else QUAD64(0);            // the staged-epilogue default (fp32 output / residual, row gate, tanh-gate scale, N % 64 != 0): +7 % over lean
#undef QUAD64
#undef LEAN
    } else {
        const int t128m = (int)((M + 127) / 128), t128n = (int)((N + 127) / 128);
        // 129-256 rows: one 256-row tile per 128 columns (select 71: forced; 70 / knob 12 = 0: the mid kernel instead)
        if (tall_route(M, N, K, lda, ldw)) {
            static bool atall = false;
            if (!atall) { (void)hipFuncSetAttribute((const void*)gemm_bf16_tall_k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, TALL_LDS); atall = true; }
            gemm_bf16_tall_k<0><<<dim3(t128n), dim3(512), TALL_LDS, (hipStream_t)stream>>>(
                (const bf16_t*)A, lda, (const bf16_t*)W, ldw, C, ldc, (int)M, (int)N, (int)K, t128n, ep, 0);
        } else
        // the LDS-DMA 128-tile kernel where its K tiling applies (select 1 / 70: the register-staged general kernel / the mid kernel, forced)
        if (fk != 1 && K % 64 == 0 && K >= 128 && lean_ok) {
            static bool amid = false;
            if (!amid) { (void)hipFuncSetAttribute((const void*)gemm_bf16_mid_k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, MID_LDS); amid = true; }
            if (g_mid_ablate == 1 || g_mid_ablate == 2) {                 // knob 9: timing-only ablations of the operand stream
                if (g_mid_ablate == 1) { (void)hipFuncSetAttribute((const void*)gemm_bf16_mid_k<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, MID_LDS);
                    gemm_bf16_mid_k<0, 1><<<dim3(t128m * t128n), dim3(256), MID_LDS, (hipStream_t)stream>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, C, ldc, (int)M, (int)N, (int)K, t128m, t128n, ep, 0); }
                else { (void)hipFuncSetAttribute((const void*)gemm_bf16_mid_k<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, MID_LDS);
                    gemm_bf16_mid_k<0, 2><<<dim3(t128m * t128n), dim3(256), MID_LDS, (hipStream_t)stream>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, C, ldc, (int)M, (int)N, (int)K, t128m, t128n, ep, 0); }
            } else if (mid_deep(M, K / 64)) {
                static bool adeep = false;
                if (!adeep) { (void)hipFuncSetAttribute((const void*)gemm_bf16_mid_k<0, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, MID_LDS_DEEP); adeep = true; }
                gemm_bf16_mid_k<0, 0, 4><<<dim3(t128m * t128n), dim3(256), MID_LDS_DEEP, (hipStream_t)stream>>>(
                    (const bf16_t*)A, lda, (const bf16_t*)W, ldw, C, ldc, (int)M, (int)N, (int)K, t128m, t128n, ep, 0);
            } else
            if (g_nt_weights & 1) {
                static bool amidnt = false;
                if (!amidnt) { (void)hipFuncSetAttribute((const void*)gemm_bf16_mid_k<0, 0, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, MID_LDS); amidnt = true; }
                gemm_bf16_mid_k<0, 0, 2, 1><<<dim3(t128m * t128n), dim3(256), MID_LDS, (hipStream_t)stream>>>(
                (const bf16_t*)A, lda, (const bf16_t*)W, ldw, C, ldc, (int)M, (int)N, (int)K, t128m, t128n, ep, 0);
            } else
            gemm_bf16_mid_k<0><<<dim3(t128m * t128n), dim3(256), MID_LDS, (hipStream_t)stream>>>(
                (const bf16_t*)A, lda, (const bf16_t*)W, ldw, C, ldc, (int)M, (int)N, (int)K, t128m, t128n, ep, 0);
        } else
            gemm_bf16_tile128_k<<<dim3(t128m * t128n), dim3(256), 65536, (hipStream_t)stream>>>(
                (const bf16_t*)A, lda, (const bf16_t*)W, ldw, C, ldc, (int)M, (int)N, (int)K, t128m, t128n, ep);
    }
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

// Tickets of the skinny kernel's in-launch reduction: one counter per 64-column tile, all zero between launches (the last arriver
// of a tile resets it).  Launches on ONE stream are ordered and share a row of the pool; every stream gets a row of its own (the two
// batch slices of the engine run their projections concurrently), a 17th stream falls back to the separate finalize launch.
#define SKINNY_TICKET_ROWS 16
#define SKINNY_TICKET_TILES 1024
__device__ unsigned g_skinny_tickets[SKINNY_TICKET_ROWS][SKINNY_TICKET_TILES];
static unsigned* skinny_tickets_for(hipStream_t st, int tiles) {
    static std::mutex mu;
    static hipStream_t owner[SKINNY_TICKET_ROWS];
    static int used = 0;
    static unsigned* base = nullptr;
    if (tiles > SKINNY_TICKET_TILES) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!base && hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_skinny_tickets)) != hipSuccess) { base = nullptr; return nullptr; }
    for (int i = 0; i < used; ++i) if (owner[i] == st) return base + (size_t)i * SKINNY_TICKET_TILES;
    if (used == SKINNY_TICKET_ROWS) return nullptr;
    owner[used] = st;
    return base + (size_t)(used++) * SKINNY_TICKET_TILES;
}

// splits and workspace bytes the skinny-M path wants for (M, N, K); splits <= 1 means "use licv_gemm_bf16"
extern "C" int licv_gemm_splitk_plan(int64_t M, int64_t N, int64_t K, int* splits, int64_t* workspace_bytes) {
    LICV_CHECK_ARG(splits && workspace_bytes, "gemm_splitk_plan: null pointer");
    *splits = 1; *workspace_bytes = 0;
    if (!g_splitk_enabled) return LICV_OK;
    if (M > 0 && M <= 32 && N >= 256 && K >= 256 && K % 8 == 0) {    // weight-streaming kernel (gemm_bf16_skinny_k)
        const int64_t nblocks = (N + 63) / 64, nsteps = (K + 31) / 32;
        int64_t sp = (nsteps + SKINNY_KR_MAX / 32 - 1) / (SKINNY_KR_MAX / 32);
        const int64_t want = (512 + nblocks - 1) / nblocks;       // aim at >= 512 workgroups
        if (want > sp) sp = want;
        if (sp > nsteps / 4) sp = nsteps / 4 > 0 ? nsteps / 4 : 1;     // at least 4 K-steps per split
        const int64_t per = (nsteps + sp - 1) / sp;
        sp = (nsteps + per - 1) / per;
        if (sp >= 1 && per * 32 <= SKINNY_KR_MAX) {
            *splits = (int)(sp < 2 ? 2 : sp);                     // the split-K entry point wants >= 2; an empty extra range adds zeros
            *workspace_bytes = (int64_t)(*splits) * 32 * ((N + 127) / 128 * 128) * 4;
            return LICV_OK;
        }
    }
    if (M <= 0 || N < 128 || K % 8 != 0 || route_256(M, N, K)) return LICV_OK;
    // 128-tile route.  Two workgroups fit a CU (512 slots); a K tile is 64 deep.
    const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
    const int64_t nkt = (K + BK - 1) / BK;
    int64_t sp = 1;
    if (g_force_splits > 0) sp = g_force_splits;
    else if (K % BK == 0) {
        // The smallest estimated time wins: rounds of workgroups x K tiles per workgroup x time per K tile (the operand stream of a
        // CU is shared by its two workgroups: ~0.5 us per 32 KiB K tile alone, ~0.9 us each in pairs), plus — for sp > 1 — the fp32
        // partials written once and read once at ~3 TB/s and a second launch.  Constants from tools/mid_bench.py sweeps.
        double best = 1e30;
        const bool tall = tall_route(M, N, K, K, K);
        for (int64_t c : {1, 2, 3, 4, 6, 8, 12, 16}) {
            if (c > 1 && nkt / c < 4) break;                      // at least 4 K tiles per split
            if (tall) {
                // one 256 x 128 tile per 128 columns, one workgroup per CU: TALL_TK us per 48 KiB K tile
                const int64_t wgs = (N + 127) / 128 * c, per = (nkt + c - 1) / c;
                double t = (double)((wgs + 255) / 256) * (per * TALL_TK + 4.0);
                if (c > 1) t += (double)(c + 1) * M * N * 4.0 / 3.0e6 + 5.0;
                if (t < best) { best = t; sp = c; }
                continue;
            }
            const int64_t wgs = tiles * c, per = (nkt + c - 1) / c;
            const double rounds = (double)((wgs + 511) / 512);
            // (a lone workgroup's K tile: 0.5 us from warm sweeps, ~0.7 us cold, tools/split_sweep.py.  The cold figure is used where it was
            //  measured to pay and nothing else moves: the narrow outputs (N < 2048) of the vision tower on at most 8 images (M <= 2056: the
            //  student pass and the prefill of generate; the ViT's fc2 72 -> 52 us).  Used everywhere it splits the projections of a single
            //  32-shot question (perceiver at 2112 rows, text at 800) that its batch of 8 runs in one pass, and the question's logits then
            //  move away from the batch's by 0.063 relative L2 against the 0.05 bar of tests/test_fullsize_gpu.py P2 - measured twice.)
            const double tk = wgs <= 256 ? ((N < 2048 && M <= 2056) ? 0.7 : 0.5) : 0.9;
            double t = rounds * (per * tk + 4.0);
            if (c > 1) t += (double)(c + 1) * M * N * 4.0 / 3.0e6 + 5.0;
            if (t < best) { best = t; sp = c; }
        }
    } else if (M <= 256 && K >= 8192) {                           // ragged K: the register-staged producer, the round-1 rule
        sp = (512 + tiles - 1) / tiles;
        if (sp > nkt / 4) sp = nkt / 4;
        if (sp > 16) sp = 16;
    }
    if (sp < 2) return LICV_OK;
    *splits = (int)sp;
    *workspace_bytes = sp * ((M + 127) / 128 * 128) * ((N + 127) / 128 * 128) * 4;
    return LICV_OK;
}

// bytes of caller-provided scratch licv_gemm_bf16_splitk needs for (M, N, K): 0 when the one-pass kernels are used.  The library
// itself never allocates device memory; this is the only operator that wants a workspace, and the caller owns it.
extern "C" int64_t licv_workspace_size(int64_t M, int64_t N, int64_t K) {
    int sp = 1; int64_t nb = 0;
    if (licv_gemm_splitk_plan(M, N, K, &sp, &nb) != LICV_OK) return -1;
    return sp > 1 ? nb : 0;
}

// The split-K GEMM in two halves: the producer (fp32 slices into the workspace) and the finalize.  `finalize` false: the producer only —
// the caller's next row kernel sums the slices itself (licv_*_ws entry points: one launch less per projection of a decode step); the
// slice layout comes back through slice_elems / row_stride.
static int splitk_run(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                      int64_t M, int64_t N, int64_t K, const licv_gemm_epilogue* e, int splits,
                      void* workspace, int64_t workspace_bytes, void* stream, bool finalize, int64_t* slice_elems, int64_t* row_stride) {
    LICV_CHECK_ARG(A && W && C && e && workspace, "gemm_bf16_splitk: null pointer");
    LICV_CHECK_ARG(M > 0 && N > 0 && K > 0 && splits >= 2, "gemm_bf16_splitk: bad shape / splits");
    LICV_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && K % 8 == 0, "gemm_bf16_splitk: lda/ldw/K must be multiples of 8");
    LICV_CHECK_ARG(ldc % 4 == 0, "gemm_bf16_splitk: ldc (%lld) must be a multiple of 4", (long long)ldc);
    LICV_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
                   "gemm_bf16_splitk: pointers must be 16-byte aligned");
    LICV_CHECK_ARG(e->out_dtype == LICV_BF16 || e->out_dtype == LICV_F32, "gemm_bf16_splitk: bad out dtype");
    LICV_CHECK_ARG(e->act >= 0 && e->act <= 3, "gemm_bf16_splitk: bad activation %d", e->act);
    LICV_CHECK_ARG(!e->swiglu || (N % 32 == 0 && !e->bias_bf16 && !e->act), "gemm_bf16_splitk: swiglu needs N %% 32 == 0, no bias/act");
    LICV_CHECK_ARG(!e->residual || (e->ld_res % 4 == 0 && ((uintptr_t)e->residual & 15) == 0), "gemm_bf16_splitk: residual misaligned");
    if (M <= 32 && N >= 256 && K >= 256) {                     // weight-streaming kernel, slices of 32 rows
        const int64_t np = (N + 127) / 128 * 128;
        LICV_CHECK_ARG(workspace_bytes >= (int64_t)splits * 32 * np * 4, "gemm_bf16_splitk: workspace too small for the skinny path");
        const int nsteps = (int)((K + 31) / 32);
        const int per = (nsteps + splits - 1) / splits;
        LICV_CHECK_ARG(per * 32 <= SKINNY_KR_MAX, "gemm_bf16_splitk: %d splits leave more than %d K elements per split", splits, SKINNY_KR_MAX);
        GemmEpi eps;
        eps.bias = (const bf16_t*)e->bias_bf16; eps.row_gate = e->row_gate; eps.residual = e->residual;
        eps.residual_dtype = e->residual_dtype; eps.ld_res = e->ld_res; eps.act = e->act; eps.swiglu = e->swiglu;
        eps.use_scale = e->use_scale; eps.scale = e->scale; eps.out_dtype = e->out_dtype; eps.a_scale = nullptr; eps.w_scale = nullptr;
        static bool sattr = false;
        if (!sattr) {
            (void)hipFuncSetAttribute((const void*)gemm_bf16_skinny_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * (SKINNY_KR_MAX * 2 + 16));
            (void)hipFuncSetAttribute((const void*)gemm_bf16_skinny_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * (SKINNY_KR_MAX * 2 + 16));
                sattr = true;
        }
        hipStream_t sst = (hipStream_t)stream;
        const dim3 grid((unsigned)((N + 63) / 64), (unsigned)splits);
        const int mb = M <= 16 ? 1 : 2;
        const size_t lds = (size_t)(M + 1) * (per * 32 * 2 + 16);
        unsigned* tickets = (g_skinny_inlaunch && finalize) ? skinny_tickets_for(sst, (int)grid.x) : nullptr;
        if (slice_elems) { *slice_elems = 32 * np; *row_stride = np; }
        if (g_nt_weights & 2) {
            static bool sattr_nt = false;
            if (!sattr_nt) {
                (void)hipFuncSetAttribute((const void*)gemm_bf16_skinny_k<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * (SKINNY_KR_MAX * 2 + 16));
                (void)hipFuncSetAttribute((const void*)gemm_bf16_skinny_k<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * (SKINNY_KR_MAX * 2 + 16));
                sattr_nt = true;
            }
            if (mb == 1) gemm_bf16_skinny_k<1, 1><<<grid, 256, lds, sst>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (float*)workspace, (int)M, (int)N, (int)K, per, np, tickets, C, ldc, eps);
            else         gemm_bf16_skinny_k<2, 1><<<grid, 256, lds, sst>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (float*)workspace, (int)M, (int)N, (int)K, per, np, tickets, C, ldc, eps);
        } else {
        if (mb == 1) gemm_bf16_skinny_k<1><<<grid, 256, lds, sst>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (float*)workspace, (int)M, (int)N, (int)K, per, np, tickets, C, ldc, eps);
        else         gemm_bf16_skinny_k<2><<<grid, 256, lds, sst>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (float*)workspace, (int)M, (int)N, (int)K, per, np, tickets, C, ldc, eps);
        }
        if (!tickets && finalize) {
            const int n_out = e->swiglu ? (int)(N / 2) : (int)N;
            const int64_t items = (int64_t)M * ((n_out + 3) / 4);
            skinny_finalize_k<<<dim3((unsigned)((items + 255) / 256)), dim3(256), 0, sst>>>((const float*)workspace, C, ldc, (int)M, (int)N, np, splits, eps, 32);
        }
        LICV_LAUNCH_CHECK();
        return LICV_OK;
    }
    const int tiles_m = (int)((M + 127) / 128), tiles_n = (int)((N + 127) / 128);
    const int64_t mp = (int64_t)tiles_m * 128, npad = (int64_t)tiles_n * 128;
    const int64_t need = (int64_t)splits * mp * npad * 4;
    LICV_CHECK_ARG(workspace_bytes >= need, "gemm_bf16_splitk: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)need);
    const int nkt = (int)((K + BK - 1) / BK);
    const int per = (nkt + splits - 1) / splits;
    GemmEpi ep;
    ep.bias = (const bf16_t*)e->bias_bf16; ep.row_gate = e->row_gate; ep.residual = e->residual;
    ep.residual_dtype = e->residual_dtype; ep.ld_res = e->ld_res; ep.act = e->act; ep.swiglu = e->swiglu;
    ep.use_scale = e->use_scale; ep.scale = e->scale; ep.out_dtype = e->out_dtype;
    ep.a_scale = nullptr; ep.w_scale = nullptr;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_splitk_k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_mid_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, MID_LDS);
        attr = true;
    }
    hipStream_t st = (hipStream_t)stream;
    if (tall_route(M, N, K, lda, ldw) && nkt - (splits - 1) * per >= 2) {
        // 129-256 rows: the tall kernel as the producer (slices of 256 rows: the same layout)
        static bool atall = false;
        if (!atall) { (void)hipFuncSetAttribute((const void*)gemm_bf16_tall_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, TALL_LDS); atall = true; }
        gemm_bf16_tall_k<1><<<dim3(tiles_n, splits), dim3(512), TALL_LDS, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw,
            workspace, 0, (int)M, (int)N, (int)K, tiles_n, ep, per);
    } else
    if (g_force_kernel != 1 && K % 64 == 0 && nkt - (splits - 1) * per >= 2 && lda * 510 < (1ll << 31) && ldw * 510 < (1ll << 31)) {
        // the mid kernel as the producer (every split at least two K tiles deep)
        if (mid_deep(M, nkt - (splits - 1) * per)) {           // (the last range is the shortest)
            static bool adeep = false;
            if (!adeep) { (void)hipFuncSetAttribute((const void*)gemm_bf16_mid_k<1, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, MID_LDS_DEEP); adeep = true; }
            gemm_bf16_mid_k<1, 0, 4><<<dim3(tiles_m * tiles_n, splits), dim3(256), MID_LDS_DEEP, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw,
                workspace, 0, (int)M, (int)N, (int)K, tiles_m, tiles_n, ep, per);
        } else
        if (g_nt_weights & 1) {
            static bool asplnt = false;
            if (!asplnt) { (void)hipFuncSetAttribute((const void*)gemm_bf16_mid_k<1, 0, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, MID_LDS); asplnt = true; }
            gemm_bf16_mid_k<1, 0, 2, 1><<<dim3(tiles_m * tiles_n, splits), dim3(256), MID_LDS, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw,
            workspace, 0, (int)M, (int)N, (int)K, tiles_m, tiles_n, ep, per);
        } else
        gemm_bf16_mid_k<1><<<dim3(tiles_m * tiles_n, splits), dim3(256), MID_LDS, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw,
            workspace, 0, (int)M, (int)N, (int)K, tiles_m, tiles_n, ep, per);
    } else {
        gemm_bf16_splitk_k<<<dim3(tiles_m * tiles_n, splits), dim3(256), 65536, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw,
            (float*)workspace, (int)M, (int)N, (int)K, tiles_m, tiles_n, per);
    }
    // the row-major finalize (4 consecutive columns per thread: coalesced 16-byte reads of every slice, same rounding points as the
    // staged epilogue); the first version walked the partials in the accumulator layout through an LDS image: ~8 us per extra split
    if (slice_elems) { *slice_elems = mp * npad; *row_stride = npad; }
    if (finalize) {
        const int n_out = e->swiglu ? (int)(N / 2) : (int)N;
        const int64_t items = (int64_t)M * ((n_out + 3) / 4);
        skinny_finalize_k<<<dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st>>>((const float*)workspace, C, ldc, (int)M, (int)N, npad, splits, ep, (int)mp);
    }
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

extern "C" int licv_gemm_bf16_splitk(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                                     int64_t M, int64_t N, int64_t K, const licv_gemm_epilogue* e, int splits,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
    return splitk_run(A, lda, W, ldw, C, ldc, M, N, K, e, splits, workspace, workspace_bytes, stream, true, nullptr, nullptr);
}

// The producer half alone: `splits` fp32 slices of A W^T in the workspace, slice sp of row m at
// workspace + sp * slice_elems + m * row_stride (floats); summed in slice order and rounded to bf16 they are the plain GEMM's output.
extern "C" int licv_gemm_bf16_splitk_produce(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t N, int64_t K,
                                             int splits, void* workspace, int64_t workspace_bytes, int64_t* slice_elems,
                                             int64_t* row_stride, void* stream) {
    LICV_CHECK_ARG(slice_elems && row_stride, "gemm_bf16_splitk_produce: null pointer");
    licv_gemm_epilogue e;
    e.bias_bf16 = nullptr; e.row_gate = nullptr; e.residual = nullptr; e.residual_dtype = 0; e.ld_res = 0; e.act = 0; e.swiglu = 0;
    e.use_scale = 0; e.scale = 0.f; e.out_dtype = LICV_BF16;
    return splitk_run(A, lda, W, ldw, workspace, 4, M, N, K, &e, splits, workspace, workspace_bytes, stream, false, slice_elems, row_stride);
}

extern "C" int licv_gemm_fp8(const void* Aq, int64_t lda, const float* a_scale, const void* Wq, int64_t ldw, const float* w_scale,
                             void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const licv_gemm_epilogue* e, void* stream) {
    LICV_CHECK_ARG(Aq && Wq && a_scale && w_scale && C && e, "gemm_fp8: null pointer");
    LICV_CHECK_ARG(M > 0 && N > 0 && K >= 256 && K % 64 == 0, "gemm_fp8: needs K >= 256 and K %% 64 == 0 (M=%lld N=%lld K=%lld)", (long long)M, (long long)N, (long long)K);
    LICV_CHECK_ARG(lda >= K && ldw >= K && lda % 16 == 0 && ldw % 16 == 0, "gemm_fp8: leading dims (bytes) must be >= K and multiples of 16");
    LICV_CHECK_ARG(ldc % 4 == 0, "gemm_fp8: ldc (%lld) must be a multiple of 4", (long long)ldc);
    LICV_CHECK_ARG(((uintptr_t)Aq & 15) == 0 && ((uintptr_t)Wq & 15) == 0 && ((uintptr_t)C & 15) == 0, "gemm_fp8: pointers must be 16-byte aligned");
    LICV_CHECK_ARG(e->out_dtype == LICV_BF16 || e->out_dtype == LICV_F32, "gemm_fp8: bad out dtype");
    LICV_CHECK_ARG(e->act >= 0 && e->act <= 3, "gemm_fp8: bad activation %d", e->act);
    LICV_CHECK_ARG(!e->swiglu || (N % 32 == 0 && !e->bias_bf16 && !e->act), "gemm_fp8: swiglu needs N %% 32 == 0, no bias/act");
    LICV_CHECK_ARG(!e->residual || (e->ld_res % 4 == 0 && ((uintptr_t)e->residual & 15) == 0), "gemm_fp8: residual misaligned");
    LICV_CHECK_ARG(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "gemm_fp8: dimension too large");
    if (const int64_t rows = rows_per_launch(M, ldc, e->out_dtype, e)) {
        for (int64_t m0 = 0; m0 < M; m0 += rows) {
            licv_gemm_epilogue eb = *e;
            if (e->residual) eb.residual = (const char*)e->residual + m0 * e->ld_res * (e->residual_dtype == LICV_F32 ? 4 : 2);
            if (e->row_gate) eb.row_gate = e->row_gate + m0;
            const int rc = licv_gemm_fp8((const char*)Aq + m0 * lda, lda, a_scale + m0, Wq, ldw, w_scale, (char*)C + m0 * ldc * (e->out_dtype == LICV_F32 ? 4 : 2), ldc,
                                         M - m0 < rows ? M - m0 : rows, N, K, &eb, stream);
            if (rc != LICV_OK) return rc;
        }
        return LICV_OK;
    }
    GemmEpi ep;
    ep.bias = (const bf16_t*)e->bias_bf16; ep.row_gate = e->row_gate; ep.residual = e->residual;
    ep.residual_dtype = e->residual_dtype; ep.ld_res = e->ld_res; ep.act = e->act; ep.swiglu = e->swiglu;
    ep.use_scale = e->use_scale; ep.scale = e->scale; ep.out_dtype = e->out_dtype;
    ep.a_scale = a_scale; ep.w_scale = w_scale;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_fp8_pingpong_k, hipFuncAttributeMaxDynamicSharedMemorySize, RING_STAGES * RING_STAGE_BYTES); attr = true; }
    const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)((N + 255) / 256);
    // the 4-wave kernel on the 128-deep MFMA: the flow64 conditions in bytes (whole waves in or out of N, four 128-byte K tiles,
    // accumulator-only epilogues or a bf16 residual, 32-bit offsets), and no private segment in any instantiation
    const bool res_ok = e->residual && e->residual_dtype == LICV_BF16 && !e->act && !e->swiglu && e->ld_res % 8 == 0 &&
                        (int64_t)(M + 256) * e->ld_res * 2 < (1ll << 31);
    const bool f64 = g_fp8_flow64 && M >= 512 && N >= 256 && N % 128 == 0 && K % 128 == 0 && K >= 512 && e->out_dtype == LICV_BF16 &&
                     (!e->residual || res_ok) && !e->row_gate && !e->use_scale && (int64_t)(M + 256) * ldc * 2 < (1ll << 31) && ldc % 8 == 0 &&
                     lda * 510 < (1ll << 31) && ldw * 510 < (1ll << 31) && (!e->bias_bf16 || ((uintptr_t)e->bias_bf16 & 3) == 0) &&
                     ((uintptr_t)a_scale & 3) == 0 && ((uintptr_t)w_scale & 15) == 0 &&
                     fp8_flow64_scratch_free(e->residual ? 5 : (e->swiglu ? 4 : e->act));
    if (f64) {
        const int pp_group = g_pp_group > 0 ? g_pp_group : (tiles_n <= 6 ? 2 : 8);
        const dim3 grid(min(tiles_m * tiles_n, g_num_cus)), block(256);
#define F8FLOW(E) gemm_fp8_flow64_k<E><<<grid, block, 10 * Q64_UNIT, (hipStream_t)stream>>>( \
            (const char*)Aq, lda, (const char*)Wq, ldw, a_scale, w_scale, (bf16_t*)C, (int)ldc, (int)M, (int)N, (int)K, tiles_m, tiles_n, ep.bias, pp_group)
        if (e->residual)
            gemm_fp8_flow64_k<5><<<grid, block, 10 * Q64_UNIT, (hipStream_t)stream>>>(
                (const char*)Aq, lda, (const char*)Wq, ldw, a_scale, w_scale, (bf16_t*)C, (int)ldc, (int)M, (int)N, (int)K, tiles_m, tiles_n, ep.bias, pp_group,
                (const bf16_t*)e->residual, (int)e->ld_res);
        else if (e->swiglu) F8FLOW(4); else if (e->act == 1) F8FLOW(1); else if (e->act == 2) F8FLOW(2); else if (e->act == 3) F8FLOW(3); else F8FLOW(0);
#undef F8FLOW
        LICV_LAUNCH_CHECK();
        return LICV_OK;
    }
    gemm_fp8_pingpong_k<<<dim3(tiles_m * tiles_n), dim3(512), RING_STAGES * RING_STAGE_BYTES, (hipStream_t)stream>>>(
        (const char*)Aq, lda, (const char*)Wq, ldw, C, ldc, (int)M, (int)N, (int)K, tiles_m, tiles_n, ep);
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

