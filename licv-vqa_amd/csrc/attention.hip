// Tiled online-softmax attention for gfx950 (replaces hf eager_attention_forward,
// hf:idefics/modeling_idefics.py:450-470, vision.py:169-189, and the perceiver's einsum attention,
// hf:idefics/perceiver.py:150-166).  One kernel serves: causal self-attention with a key-padding mask,
// ViT / SigLIP bidirectional attention, the gated cross-attention with the per-token image mask, and the
// perceiver's latents -> [context, latents] attention; head dims 8..128, GQA, ragged Sq / Sk.
//
// Layout of the computation (wave = 64 lanes, 16x16x32 bf16 MFMA):
//   * workgroup = 4 waves = 64 query rows of one (batch, head); each wave owns 16 queries;
//   * K and V tiles of 64 keys are staged global -> VGPR -> LDS (issued before the MFMAs of the current
//     tile, written after them), zero-padded to the MFMA K granularity;
//   * S^T = K.Q^T (keys on the accumulator rows, the QUERY on lane&15), so every softmax statistic of a
//     query is lane-local up to two xor-shuffles (lanes l, l^16, l^32, l^48 share a query);
//   * P^T is then already in B-operand layout for  O^T = V^T.P^T : no LDS round trip for P; V^T
//     fragments come from the row-major V tile through the transposing LDS read ds_read_b64_tr_b16;
//   * O^T keeps the query on the lane too, so the running rescale is one multiply per register and the
//     epilogue stores 4 consecutive head-dim elements (8 bytes) per lane.
#include "common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define ATT_QB 64     // queries per workgroup
#define ATT_KB 64     // keys per tile

struct AttnP {
    const bf16_t* q; int64_t q_bs, q_rs;
    const bf16_t* k; const bf16_t* v; int64_t kv_bs, kv_rs;
    bf16_t* o;
    int B, Sq, Sk, nh, nkv, hd;
    float scale;
    int mask_mode;
    const int32_t* key_valid;
    const int32_t* img_mask; int n_img, img_len;
};

// LDS row stride (bytes) for a tile with `cols` bf16 columns: the smallest 16-byte multiple >= the row whose dword
// count is 8 mod 16 — brute-forced conflict-free for both access shapes used here (ds_read_b128 of 16 rows x
// 4 chunks, and ds_read_b64_tr_b16 of 8 rows x 4 column groups); e.g. 96 cols -> 224 B, 80 -> 160 B, 128 -> 288 B.
__host__ __device__ constexpr int lds_stride(int cols) {
    int dw = cols / 2;
    while (dw % 16 != 8) dw += 4;
    return dw * 4;
}

__device__ __forceinline__ bf16x4 lds_read_tr(const char* p) {
    typedef __attribute__((ext_vector_type(4))) short s4;
    s4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p));
    return *reinterpret_cast<bf16x4*>(&r);
}

// One 64-key tile for one 16-query sub-tile of a wave: S^T = K.Q^T, (mask), online softmax in the log2 domain,
// O^T += V^T.P^T.  NST (compile time) = 16-key sub-tiles that hold real keys, so the hot NST = 4 instance is one
// branch-free region the compiler can software-pipeline (a run-time `if (st < nst)` around each MFMA serialised
// every ds_read -> s_waitcnt -> v_mfma triple: 3800 cycles per tile instead of ~800).
template <int DPK, int DPV, int NST, typename MaskF>
__device__ __forceinline__ void attn_tile(const char* kt, const char* vt, const bf16x8 (&qf)[DPK / 32], floatx4 (&oacc)[DPV / 16],
                                          float& m_run, float& l_run, float sc, int g, int ql, bool need_mask, MaskF&& allowed) {
    constexpr int KSTR = lds_stride(DPK), VSTR = lds_stride(DPV);
    floatx4 s[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        if (st < NST) {
            s[st] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < DPK / 32; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt + (st * 16 + ql) * KSTR + (ks * 32 + g * 8) * 2);
                s[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[st], 0, 0, 0);
            }
        } else {
            s[st] = floatx4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
    }
    if (need_mask) {
#pragma unroll
        for (int st = 0; st < NST; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (!allowed(st, r)) s[st][r] = -INFINITY;
    }
    float tmax = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])), fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
    tmax = fmaxf(tmax, fmaxf(fmaxf(fmaxf(s[2][0], s[2][1]), fmaxf(s[2][2], s[2][3])), fmaxf(fmaxf(s[3][0], s[3][1]), fmaxf(s[3][2], s[3][3]))));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax * sc);               // running max of s*c  (c > 0 keeps the order)
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use); // m_run = -inf -> 0
    float psum = 0.f;
    bf16x8 pf[2];
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = (st < NST) ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[st][r], sc, -m_use)) : 0.f;
            psum += p;
            pf[st >> 1][(st & 1) * 4 + r] = (__bf16)p;
        }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DPV / 16; ++i) oacc[i] *= alpha;
    // O^T += V^T . P^T : k-step s2 covers sub-tiles 2*s2, 2*s2+1
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        if (2 * s2 < NST) {
#pragma unroll
            for (int dt = 0; dt < DPV / 16; ++dt) {
                const char* p0 = vt + ((2 * s2) * 16 + g * 4 + (ql >> 2)) * VSTR + (dt * 16 + (ql & 3) * 4) * 2;
                const bf16x4 lo = lds_read_tr(p0);
                const bf16x4 hi = lds_read_tr(p0 + 16 * VSTR);
                bf16x8 vf;
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[s2], oacc[dt], 0, 0, 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The key tiles of one 16-query sub-tile against K/V resident in LDS, hand-pipelined.  attn_tile above leaves the
// order of LDS reads to the compiler, which emits  ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma  twenty times per tile: every
// LDS latency sits on the wave's critical path (rocprofv3 on the ViT shape: waves parked 56 % of their cycles, MFMA busy 15 %,
// ~4000 cycles per tile for ~350 cycles of MFMA; profiles/r03_attn_pmc.txt).  Here every fragment of a phase is in registers
// before the phase starts:
//   QK   : the 4 x DPK/32 K fragments were read ahead of the previous tile's PV               (12 ds_read_b128 at 96)
//   V^T  : the 2 x DPV/16 x 2 transposing reads are issued before QK and land behind QK and the softmax    (20 at 80)
//   PV   : consumes them; the next tile's K reads are already in flight.
// Arithmetic and its order are those of attn_tile<.., 4> (same MFMA k-step order per accumulator, same sum order):
// results are bit-identical to it.  The row max travels over v_permlane{32,16}_swap (VALU) instead of ds_bpermute (an LDS
// round trip that would also drain the V^T reads in flight).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float max3_f32(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max_f32(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// max over the four lanes {ql, ql + 16, ql + 32, ql + 48}, result in all of them
__device__ __forceinline__ float max_over_groups(float x) {
    const unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // a[0] = low half twice, a[1] = high half twice
    const float h = max_f32(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned v = __float_as_uint(h);
    const auto b = __builtin_amdgcn_permlane16_swap(v, v, false, false);      // b[0] = even rows twice, b[1] = odd rows twice
    return max_f32(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// One tile of the pipelined walk: NST live 16-key sub-tiles (4 = a full tile), MASKED = keys past Sk are cut (ragged tile),
// NEXTK = read the K fragments of the tile at `kn` for the next call while this tile's softmax / PV run.
template <int DPK, int DPV, int NST, bool MASKED, bool NEXTK, typename MaskF>
__device__ __forceinline__ void attn_tile_pipe(bf16x8 (&kf)[4][DPK / 32], const char* vt, const char* kn, const bf16x8 (&qf)[DPK / 32],
                                               floatx4 (&oacc)[DPV / 16], float& m_run, float& l_run, float sc, MaskF&& allowed) {
    constexpr int VSTR = lds_stride(DPV), KSTR = lds_stride(DPK);
    constexpr int KS = DPK / 32, DT = DPV / 16, NS2 = (NST + 1) / 2;
    // V^T fragments of this tile: in flight across QK and the softmax
    bf16x4 vlo[NS2][DT], vhi[NS2][DT];
#pragma unroll
    for (int s2 = 0; s2 < NS2; ++s2)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            vlo[s2][dt] = lds_read_tr(vt + (s2 * 32) * VSTR + dt * 32);
            vhi[s2][dt] = lds_read_tr(vt + (s2 * 32 + 16) * VSTR + dt * 32);
        }
    __builtin_amdgcn_sched_barrier(0);
    // S^T = K.Q^T : k-step outer, so consecutive MFMAs write different accumulators
    floatx4 s[4];
#pragma unroll
    for (int st = 0; st < 4; ++st)
        s[st] = st < NST ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[st][0], qf[0], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0)
                         : floatx4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int ks = 1; ks < KS; ++ks)
#pragma unroll
        for (int st = 0; st < NST; ++st) s[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[st][ks], qf[ks], s[st], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // The max tree below is inline asm (fmaxf would first canonicalise each of the 16 MFMA outputs: 16 extra v_max), which the
    // compiler's hazard recogniser does not look into: an MFMA result read by VALU needs software wait states (no interlock;
    // without them the max read stale registers now and then - results 1 ulp off and different from run to run).
    asm volatile("s_nop 15");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MASKED) {
#pragma unroll
        for (int st = 0; st < NST; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (!allowed(st, r)) s[st][r] = -INFINITY;
    }
    // online softmax, log2 domain
    float tmax = max3_f32(max3_f32(s[0][0], s[0][1], s[0][2]), max3_f32(s[0][3], s[1][0], s[1][1]), max3_f32(s[1][2], s[1][3], s[2][0]));
    tmax = max3_f32(tmax, max3_f32(s[2][1], s[2][2], s[2][3]), max3_f32(s[3][0], s[3][1], max_f32(s[3][2], s[3][3])));
    tmax = max_over_groups(tmax);
    const float m_new = max_f32(m_run, tmax * sc);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
    float psum = 0.f;
    bf16x8 pf[2];
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = (st < NST) ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[st][r], sc, -m_use)) : 0.f;
            psum += p;
            pf[st >> 1][(st & 1) * 4 + r] = (__bf16)p;
        }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DT; ++i) oacc[i] *= alpha;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NEXTK) {
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[st][ks] = *reinterpret_cast<const bf16x8*>(kn + st * 16 * KSTR + ks * 64);
        __builtin_amdgcn_sched_barrier(0);
    }
    // O^T += V^T . P^T
#pragma unroll
    for (int s2 = 0; s2 < NS2; ++s2)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            bf16x8 vf;
            vf[0] = vlo[s2][dt][0]; vf[1] = vlo[s2][dt][1]; vf[2] = vlo[s2][dt][2]; vf[3] = vlo[s2][dt][3];
            vf[4] = vhi[s2][dt][0]; vf[5] = vhi[s2][dt][1]; vf[6] = vhi[s2][dt][2]; vf[7] = vhi[s2][dt][3];
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[s2], oacc[dt], 0, 0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
}

// All key tiles of one 16-query sub-tile: nfull full tiles, then the ragged one with `nst_last` live sub-tiles (0 = none).
// The K fragments of a tile are read ahead of the previous tile's PV; the LDS allocation covers the read-ahead of the (possibly
// ragged, possibly absent) tile after the last full one: rows past skp of the K region fall into the V region, never used.
template <int DPK, int DPV>
__device__ __forceinline__ void attn_unit_pipe(const char* sK, const char* sV, int nfull, int nst_last, int Sk, const bf16x8 (&qf)[DPK / 32],
                                               floatx4 (&oacc)[DPV / 16], float& m_run, float& l_run, float sc, int g, int ql) {
    constexpr int KSTR = lds_stride(DPK), VSTR = lds_stride(DPV);
    constexpr int KS = DPK / 32;
    const char* kp = sK + ql * KSTR + g * 16;                                                  // + (tile * 64 + st * 16) rows + ks * 64 bytes
    const char* vp = sV + (g * 4 + (ql >> 2)) * VSTR + (ql & 3) * 8;                           // + (tile * 64 + s2 * 32 [+ 16]) rows + dt * 32 bytes
    bf16x8 kf[4][KS];
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[st][ks] = *reinterpret_cast<const bf16x8*>(kp + st * 16 * KSTR + ks * 64);
    auto all = [](int, int) -> bool { return true; };
    for (int t = 0; t < nfull; ++t)
        attn_tile_pipe<DPK, DPV, 4, false, true>(kf, vp + t * ATT_KB * VSTR, kp + (t + 1) * ATT_KB * KSTR, qf, oacc, m_run, l_run, sc, all);
    const char* vt = vp + nfull * ATT_KB * VSTR;
    const int left = Sk - nfull * ATT_KB;
    auto in_range = [&](int st, int r) -> bool { return st * 16 + g * 4 + r < left; };
    if (nst_last == 1)      attn_tile_pipe<DPK, DPV, 1, true, false>(kf, vt, nullptr, qf, oacc, m_run, l_run, sc, in_range);
    else if (nst_last == 2) attn_tile_pipe<DPK, DPV, 2, true, false>(kf, vt, nullptr, qf, oacc, m_run, l_run, sc, in_range);
    else if (nst_last == 3) attn_tile_pipe<DPK, DPV, 3, true, false>(kf, vt, nullptr, qf, oacc, m_run, l_run, sc, in_range);
    else if (nst_last == 4) attn_tile_pipe<DPK, DPV, 4, true, false>(kf, vt, nullptr, qf, oacc, m_run, l_run, sc, in_range);
}

template <int DPK, int DPV, typename MaskF>
__device__ __forceinline__ void attn_tile_n(int nst, const char* kt, const char* vt, const bf16x8 (&qf)[DPK / 32],
                                            floatx4 (&oacc)[DPV / 16], float& m_run, float& l_run, float sc, int g, int ql,
                                            bool need_mask, MaskF&& allowed) {
    if (nst >= 4)      attn_tile<DPK, DPV, 4>(kt, vt, qf, oacc, m_run, l_run, sc, g, ql, need_mask, allowed);
    else if (nst == 3) attn_tile<DPK, DPV, 3>(kt, vt, qf, oacc, m_run, l_run, sc, g, ql, need_mask, allowed);
    else if (nst == 2) attn_tile<DPK, DPV, 2>(kt, vt, qf, oacc, m_run, l_run, sc, g, ql, need_mask, allowed);
    else               attn_tile<DPK, DPV, 1>(kt, vt, qf, oacc, m_run, l_run, sc, g, ql, need_mask, allowed);
}

// MODE = mask mode as a compile-time constant: the image-mask bookkeeping (run-time divisions, per-image bit sets) and the
// causal bounds otherwise sit in every instantiation's tile loop (1600 scalar instructions, SGPR spills to VGPR lanes).
template <int DPK, int DPV, int QT, int NW, int MODE, bool PIPE = false>   // NW waves per workgroup, QT 16-query sub-tiles per wave: QT*NW*16 queries
__global__ __launch_bounds__(NW * 64, 2)      // 2 waves per SIMD guaranteed (3 at head_dim 128 spills once MODE is a constant: 183 -> 251 us)
void attn_fwd_k(AttnP a) {                    // PIPE: full tiles through attn_tile_pipe (every fragment of a phase in registers before the phase)
    constexpr int NTHR = NW * 64;
    constexpr int SLAB = NW * 16;               // queries per slab (one sub-tile of every wave)
    constexpr int KSTR = lds_stride(DPK), VSTR = lds_stride(DPV);   // conflict-free row strides
    constexpr int KCH = DPK / 8, VCH = DPV / 8; // 16-byte chunks per row
    constexpr int KLD = (ATT_KB * KCH + NTHR - 1) / NTHR, VLD = (ATT_KB * VCH + NTHR - 1) / NTHR;
    constexpr int QBLK = SLAB * QT;
    __shared__ __attribute__((aligned(16))) char sK[ATT_KB * KSTR];
    __shared__ __attribute__((aligned(16))) char sV[ATT_KB * VSTR];
    __shared__ __attribute__((aligned(16))) int sValid[ATT_KB];
    __shared__ unsigned long long sImgUsed[NW];
    __shared__ int sAllValid;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, ql = lane & 15;
    const int qtiles = (a.Sq + QBLK - 1) / QBLK;
    int bid = blockIdx.x;
    int qt, pair;
    if (MODE == 1 && qtiles > 1) {
        // causal: a query tile's work grows with its index (qt + 1 key tiles).  Within chunks of 32 (batch, head) pairs the heavy
        // tiles are dispatched first and the light ones last, so the tail of the launch is made of short workgroups; a chunk's K/V
        // (32 pairs) still fits the L2s, which plain "heaviest first over all pairs" would give up
        constexpr int CP = 32;                              // 16 ... 256 pairs per chunk measured alike (129-137 us); 8: 157 us; plain order: 175-181 us
        const int npairs = a.B * a.nh, chunk = CP * qtiles;
        const int c0 = (bid / chunk) * CP, r = bid % chunk;
        const int csize = min(CP, npairs - c0);
        qt = qtiles - 1 - r / csize;
        pair = c0 + r % csize;
    } else {
        qt = bid % qtiles; pair = bid / qtiles;
    }
    const int head = pair % a.nh;
    const int b = pair / a.nh;
    const int kvh = head / (a.nh / a.nkv);
    const int q0 = qt * QBLK;
    const int coff = a.Sk - a.Sq;                          // causal offset (decode steps: Sq < Sk)
    const float sc = a.scale * 1.4426950408889634f;        // softmax runs on exp2

    // ---- Q fragments (B operand of S^T = K.Q^T): Q[qrow][ks*32 + 8g .. +7]; sub-tile qs covers rows
    // q0 + qs*64 + wave*16 .. +15 (this lane: + ql)
    bf16x8 qf[QT][DPK / 32];
#pragma unroll
    for (int qs = 0; qs < QT; ++qs) {
        const int qrow = q0 + qs * SLAB + wave * 16 + ql;
        const bf16_t* qp = a.q + (int64_t)b * a.q_bs + (int64_t)qrow * a.q_rs + (int64_t)head * a.hd;
#pragma unroll
        for (int ks = 0; ks < DPK / 32; ++ks) {
            const int d = ks * 32 + g * 8;
            u32x4 r = u32x4{0u, 0u, 0u, 0u};
            if (qrow < a.Sq && d < a.hd) r = *reinterpret_cast<const u32x4*>(qp + d);
            qf[qs][ks] = *reinterpret_cast<bf16x8*>(&r);
        }
    }

    int kend = a.Sk;
    if (MODE == 1) kend = min(a.Sk, min(q0 + QBLK, a.Sq) + coff);   // keys beyond the last query's diagonal
    const int ntiles = (kend + ATT_KB - 1) / ATT_KB;

    // image-mask mode with whole tiles inside one image: a tile is visited only if some query of this
    // workgroup attends that image (a token attends ONE image of ~33: >95 % of the key tiles drop out)
    const bool img_uniform = (MODE == 3) && (a.img_len % ATT_KB == 0);
    const bool img_skip = img_uniform && a.n_img <= 64;
    unsigned long long used = ~0ull;
    if (img_skip) {
        unsigned long long mine = 0;
#pragma unroll
        for (int qs = 0; qs < QT; ++qs) {
            const int qrow = q0 + qs * SLAB + wave * 16 + ql;
            const int32_t* imrow = a.img_mask + ((int64_t)b * a.Sq + qrow) * a.n_img;
            for (int n = 0; n < a.n_img; ++n) {
                const bool hit = qrow < a.Sq && imrow[n] != 0;
                if (__any(hit)) mine |= 1ull << n;
            }
        }
        if (lane == 0) sImgUsed[wave] = mine;
        __syncthreads();
        used = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) used |= sImgUsed[w];
    }
    auto tile_live = [&](int t) -> bool { return !img_skip || ((used >> ((t * ATT_KB) / a.img_len)) & 1ull); };
    auto next_tile = [&](int t) -> int { while (t < ntiles && !tile_live(t)) ++t; return t; };

    // The next tile's K / V (and key-valid flags) are BUFFER loads outside any branch; a chunk that does not exist (key >= Sk,
    // column >= hd, thread past the tile) gets an out-of-range offset and comes back as zeros.  With the loads inside `if`s the
    // compiler's waitcnt pass waited for them (vmcnt(0)) at the first LDS-dependent instruction of the tile being multiplied:
    // the prefetch was synchronous.  (When fits32 is false - a (batch, head) slice beyond 2 GB - the host does not launch this kernel.)
    const auto rsk = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.k + (int64_t)b * a.kv_bs + (int64_t)kvh * a.hd), 0, 0xFFFFFFFF, 0x00020000);
    const auto rsv = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.v + (int64_t)b * a.kv_bs + (int64_t)kvh * a.hd), 0, 0xFFFFFFFF, 0x00020000);
    const bool use_kv = a.key_valid && (MODE == 1 || MODE == 2);
    const auto rsf = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(use_kv ? a.key_valid + (int64_t)b * a.Sk : (const int32_t*)a.k), 0, 0xFFFFFFFF, 0x00020000);
    constexpr uint32_t OOB = 0xFFFFFFFFu;
    u32x4 rk[KLD], rv[VLD];
    int rvalid = 1;
    auto load_tile = [&](int t) {
        const int key0 = t * ATT_KB;
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            const int c = tid + i * NTHR;
            const int row = c / KCH, ch = c % KCH;
            const bool ok = (int)(c < ATT_KB * KCH) & (int)(key0 + row < a.Sk) & (int)(ch * 8 < a.hd);
            rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rsk, ok ? (uint32_t)((key0 + row) * (int)a.kv_rs + ch * 8) * 2u : OOB, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < VLD; ++i) {
            const int c = tid + i * NTHR;
            const int row = c / VCH, ch = c % VCH;
            const bool ok = (int)(c < ATT_KB * VCH) & (int)(key0 + row < a.Sk) & (int)(ch * 8 < a.hd);
            rv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsv, ok ? (uint32_t)((key0 + row) * (int)a.kv_rs + ch * 8) * 2u : OOB, 0, 0);
        }
        {
            const int key = key0 + tid;
            const bool in = (int)(tid < ATT_KB) & (int)(key < a.Sk);
            const int f = __builtin_amdgcn_raw_buffer_load_b32(rsf, (in && use_kv) ? (uint32_t)key * 4u : OOB, 0, 0);
            rvalid = in ? (use_kv ? (f != 0) : 1) : 0;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            const int c = tid + i * NTHR;
            if (c < ATT_KB * KCH) *reinterpret_cast<u32x4*>(sK + (c / KCH) * KSTR + (c % KCH) * 16) = rk[i];
        }
#pragma unroll
        for (int i = 0; i < VLD; ++i) {
            const int c = tid + i * NTHR;
            if (c < ATT_KB * VCH) *reinterpret_cast<u32x4*>(sV + (c / VCH) * VSTR + (c % VCH) * 16) = rv[i];
        }
        if (tid < ATT_KB) {
            sValid[tid] = rvalid;
            const bool all = __all(rvalid != 0);             // wave 0 holds all 64 keys of the tile
            if (tid == 0) sAllValid = all ? 1 : 0;
        }
    };

    floatx4 oacc[QT][DPV / 16];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qs = 0; qs < QT; ++qs) {
        m_run[qs] = -INFINITY; l_run[qs] = 0.f;
#pragma unroll
        for (int i = 0; i < DPV / 16; ++i) oacc[qs][i] = floatx4{0.f, 0.f, 0.f, 0.f};
    }

    int t = next_tile(0);
    if (t < ntiles) { load_tile(t); store_tile(); }
    __syncthreads();

    while (t < ntiles) {
        const int tn = next_tile(t + 1);
        if (tn < ntiles) load_tile(tn);
        const int key0 = t * ATT_KB;
        const int nst = min(4, (min(a.Sk, kend) - key0 + 15) >> 4);      // 16-key sub-tiles that hold real keys
        const bool tile_all_valid = sAllValid != 0;

#pragma unroll
        for (int qs = 0; qs < QT; ++qs) {
            const int qw0 = q0 + qs * SLAB + wave * 16;          // first query of this wave's sub-tile (wave-uniform)
            if (qw0 >= a.Sq) continue;                              // sub-tile past the last query: only helps loading
            if (MODE == 1 && key0 > qw0 + 15 + coff) continue;   // tile entirely in this sub-tile's future
            const int qrow = qw0 + ql;
            const bool qok = qrow < a.Sq;
            const int32_t* imrow = (MODE == 3 && qok) ? a.img_mask + ((int64_t)b * a.Sq + qrow) * a.n_img : nullptr;
            bool need_mask = (key0 + ATT_KB > a.Sk) || (qw0 + 16 > a.Sq) || !tile_all_valid;
            if (MODE == 1) need_mask = need_mask || (key0 + ATT_KB - 1 > qw0 + coff);
            if (MODE == 3) need_mask = true;
            const int img_ok_tile = (img_uniform && imrow) ? (imrow[key0 / a.img_len] != 0) : 0;
            int4 kvalid[4] = {int4{1, 1, 1, 1}, int4{1, 1, 1, 1}, int4{1, 1, 1, 1}, int4{1, 1, 1, 1}};
            if (need_mask) {
#pragma unroll
                for (int st = 0; st < 4; ++st) kvalid[st] = *reinterpret_cast<const int4*>(&sValid[st * 16 + g * 4]);
            }
            auto allowed = [&](int st, int r) -> bool {
                const int kl = st * 16 + g * 4 + r;
                const int key = key0 + kl;
                const int kv = r == 0 ? kvalid[st].x : (r == 1 ? kvalid[st].y : (r == 2 ? kvalid[st].z : kvalid[st].w));
                bool ok = qok && kv != 0;
                if (MODE == 1) ok = ok && (key <= qrow + coff);
                if (MODE == 3) {
                    if (img_uniform) ok = ok && img_ok_tile;
                    else ok = ok && imrow && (imrow[key / a.img_len] != 0);
                }
                return ok;
            };
            if (PIPE && nst >= 4) {
                bf16x8 kf[4][DPK / 32];
                const char* kp = sK + ql * KSTR + g * 16;
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int ks = 0; ks < DPK / 32; ++ks) kf[st][ks] = *reinterpret_cast<const bf16x8*>(kp + st * 16 * KSTR + ks * 64);
                const char* vp = sV + (g * 4 + (ql >> 2)) * VSTR + (ql & 3) * 8;
                if (need_mask) attn_tile_pipe<DPK, DPV, 4, true, false>(kf, vp, nullptr, qf[qs], oacc[qs], m_run[qs], l_run[qs], sc, allowed);
                else           attn_tile_pipe<DPK, DPV, 4, false, false>(kf, vp, nullptr, qf[qs], oacc[qs], m_run[qs], l_run[qs], sc, allowed);
            } else {
                attn_tile_n<DPK, DPV>(nst, sK, sV, qf[qs], oacc[qs], m_run[qs], l_run[qs], sc, g, ql, need_mask, allowed);
            }
        }
        __syncthreads();                                   // everyone done reading this tile
        if (tn < ntiles) store_tile();
        __syncthreads();
        t = tn;
    }

    // ---- epilogue: O[b, qrow, head*hd + d], lane holds d = dt*16 + 4g + r
#pragma unroll
    for (int qs = 0; qs < QT; ++qs) {
        const int qrow = q0 + qs * SLAB + wave * 16 + ql;
        float l = l_run[qs];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        if (qrow < a.Sq) {
            bf16_t* op = a.o + ((int64_t)b * a.Sq + qrow) * ((int64_t)a.nh * a.hd) + (int64_t)head * a.hd;
#pragma unroll
            for (int dt = 0; dt < DPV / 16; ++dt) {
                const int d = dt * 16 + g * 4;
                if (d < a.hd) {
                    uint2 u;
                    u.x = (uint32_t)f2bf(oacc[qs][dt][0] * inv) | ((uint32_t)f2bf(oacc[qs][dt][1] * inv) << 16);
                    u.y = (uint32_t)f2bf(oacc[qs][dt][2] * inv) | ((uint32_t)f2bf(oacc[qs][dt][3] * inv) << 16);
                    *reinterpret_cast<uint2*>(op + d) = u;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Resident-K/V variant for SHORT key sequences without a mask (ViT: 257 tokens, perceiver: 321 keys): the whole
// K and V of one (batch, head) are staged into LDS once (~100-140 KiB), after a single barrier the 8 waves walk
// their 16-query sub-tiles over all key tiles with no further global loads and no barriers, so the two waves
// of a SIMD drift apart and overlap MFMA with softmax VALU.  The tiled kernel above pays one global-load
// latency per 64-key tile per 64-query workgroup: 487 us per ViT layer vs the ~60 us of MFMA work in it.
// ------------------------------------------------------------------------------------------------
// -DLICV_ATTN_TRACE: lane 0 of every wave of workgroup 0 stamps s_memtime at the phase boundaries of its first items into
// the buffer given to licv_attn_debug_timestamps ([wave][128] long long); tools/attn_trace.py prints the deltas.
#ifdef LICV_ATTN_TRACE
static __device__ long long* g_attn_ts = nullptr;
#define ATTN_STAMP() do { if (ts && nev < 128) { ts[nev] = (long long)__builtin_readcyclecounter(); } ++nev; } while (0)
#else
#define ATTN_STAMP() do { } while (0)
#endif

template <int DPK, int DPV, int MAXI, bool PIPE>   // MAXI: 16-byte chunks per thread per operand (>= Sk * DPV/8 / 512); PIPE: hand-pipelined tiles
__global__ __launch_bounds__(512, 2)
void attn_resident_k(AttnP a, int skp, int n_items, int xcd_map) {
    constexpr int KSTR = lds_stride(DPK), VSTR = lds_stride(DPV);
    constexpr int KS = DPK / 32, DT = DPV / 16;
    extern __shared__ __attribute__((aligned(16))) char rsm[];
    char* sK = rsm;
    char* sV = rsm + skp * KSTR;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: descriptors stay in SGPRs)
    const int g = lane >> 4, ql = lane & 15;
    const float sc = a.scale * 1.4426950408889634f;
    const int nq = (a.Sq + 15) >> 4;                        // 16-query sub-tiles
    const int ntiles = (a.Sk + ATT_KB - 1) / ATT_KB;
    const int nfull = a.Sk / ATT_KB;                        // tiles with 64 real keys: no mask, 4 live sub-tiles
    const int nst_last = nfull < ntiles ? (a.Sk - nfull * ATT_KB + 15) >> 4 : 0;
    const int G = (int)gridDim.x;
#ifdef LICV_ATTN_TRACE
    long long* ts = (g_attn_ts && blockIdx.x == 0 && lane == 0) ? g_attn_ts + wave * 128 : nullptr;
    int nev = 0;
#endif

    // Persistent over (batch, head) items: K/V of item i+1 travel global -> registers while item i is computed.  Only the real
    // columns travel (DPV/8 16-byte chunks per row, the same (row, chunk) for K and V); the padding (K columns >= DPV, rows >= Sk)
    // is zeroed once below and never written again.
    //
    // Every global access is a BUFFER access outside any branch, lanes that have nothing to move get an out-of-range offset
    // (loads return zero, stores are dropped, no traffic).  With loads or stores inside exec-masked branches the compiler's waitcnt
    // pass fell back to s_waitcnt vmcnt(0) - in front of the QK MFMAs of every tile and at every hand-over of Q, i.e. each unit sat
    // out the full latency of whatever had just been issued (the next Q, the next item's K/V, the stores of the unit before).
    constexpr int HC = DPV / 8;
    constexpr uint32_t OOB = 0xFFFFFFFFu;
    const int nchunk = a.Sk * HC;
    for (int o = tid * 16; o < skp * (KSTR + VSTR); o += 512 * 16) *reinterpret_cast<u32x4*>(rsm + o) = u32x4{0u, 0u, 0u, 0u};
    auto rsrc = [](const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xFFFFFFFF, 0x00020000); };
    auto fetch = [&](int item, u32x4 (&rk)[MAXI], u32x4 (&rv)[MAXI]) {
        const bool item_ok = item < n_items;
        const int it = item_ok ? item : 0;
        const int hh = it % a.nh, bb = it / a.nh;
        const int kvh = hh / (a.nh / a.nkv);
        const auto rsk = rsrc(a.k + (int64_t)bb * a.kv_bs + (int64_t)kvh * a.hd);
        const auto rsv = rsrc(a.v + (int64_t)bb * a.kv_bs + (int64_t)kvh * a.hd);
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int c = tid + i * 512;
            const int row = c / HC, ch = c % HC;
            const bool ok = (int)item_ok & (int)(c < nchunk) & (int)(ch * 8 < a.hd);      // (& not &&: no branches around the loads)
            const uint32_t off = ok ? (uint32_t)(row * (int)a.kv_rs + ch * 8) * 2u : OOB;
            rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rsk, off, 0, 0);
            rv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsv, off, 0, 0);
        }
    };
    auto park = [&](const u32x4 (&rk)[MAXI], const u32x4 (&rv)[MAXI]) {
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int c = tid + i * 512;
            const int row = c / HC, ch = c % HC;
            if (c < nchunk && ch * 8 < a.hd) {
                *reinterpret_cast<u32x4*>(sK + row * KSTR + ch * 16) = rk[i];
                *reinterpret_cast<u32x4*>(sV + row * VSTR + ch * 16) = rv[i];
            }
        }
    };
    // Q fragments of the 16-query sub-tile `qsub` of `item` (zeros where there is no such row / column / item)
    auto load_q = [&](int item, int qsub, bf16x8 (&q)[KS]) {
        const bool item_ok = item < n_items;
        const int it = item_ok ? item : 0;
        const int hh = it % a.nh, bb = it / a.nh;
        const auto rsq = rsrc(a.q + (int64_t)bb * a.q_bs + (int64_t)hh * a.hd);
        const int qrow = qsub * 16 + ql;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d = ks * 32 + g * 8;
            const bool ok = (int)item_ok & (int)(qrow < a.Sq) & (int)(d < a.hd);
            const uint32_t off = ok ? (uint32_t)(qrow * (int)a.q_rs + d) * 2u : OOB;
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsq, off, 0, 0);
            q[ks] = *reinterpret_cast<const bf16x8*>(&r);
        }
    };

    // the heads of one image share cache lines (a head's 160-byte row piece straddles two 128-byte lines): workgroups of one XCD
    // (blockIdx % 8, its own L2) take CONSECUTIVE items, so both halves of a line are asked of the same L2 at about the same time
    int item = xcd_map ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    // A wave's units form ONE stream over all its items: (item, wave), (item, wave + 8), ..., (item + G, wave), ...; qf holds the
    // Q fragments of the unit about to run, qf_next those of the unit after it (loaded one unit ahead - across the item boundary
    // too: the first Q of an item is never requested behind that item's own K/V prefetch burst).
    bf16x8 qf[KS], qf_next[KS];
    {
        u32x4 rk[MAXI], rv[MAXI];
        fetch(item, rk, rv);
        load_q(item, wave, qf);
        const bool one = wave + 8 >= nq;
        load_q(one ? item + G : item, one ? wave : wave + 8, qf_next);
        __syncthreads();                        // the zero fill is complete before any real chunk lands on it
        if (item < n_items) park(rk, rv);
        __syncthreads();
    }
    for (; item < n_items; item += G) {
        const int head = item % a.nh;
        const int b = item / a.nh;
        const bool more = item + G < n_items;
        const auto rso = rsrc(a.o + ((int64_t)b * a.Sq * a.nh + head) * a.hd);
        ATTN_STAMP();                           // item start
        u32x4 rk[MAXI], rv[MAXI];               // (declared per item: nothing is carried from one item's prefetch into the next)
        fetch(item + G, rk, rv);
        int qsub = wave;                        // (nq >= 8: every wave has a unit; do-while so that the park's waits can count on one)
        do {
            ATTN_STAMP();                       // unit start
            floatx4 oacc[DT];
#pragma unroll
            for (int i = 0; i < DT; ++i) oacc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
            float m_run = -INFINITY, l_run = 0.f;
            if constexpr (PIPE) {
                attn_unit_pipe<DPK, DPV>(sK, sV, nfull, nst_last, a.Sk, qf, oacc, m_run, l_run, sc, g, ql);
            } else {
                auto never = [](int, int) -> bool { return true; };
                for (int t = 0; t < nfull; ++t)
                    attn_tile<DPK, DPV, 4>(sK + t * ATT_KB * KSTR, sV + t * ATT_KB * VSTR, qf, oacc, m_run, l_run, sc, g, ql, false, never);
                if (nst_last) {                     // ragged last tile: keys past Sk never contribute
                    const int key0 = nfull * ATT_KB;
                    auto allowed = [&](int st, int r) -> bool { return key0 + st * 16 + g * 4 + r < a.Sk; };
                    attn_tile_n<DPK, DPV>(nst_last, sK + key0 * KSTR, sV + key0 * VSTR, qf, oacc, m_run, l_run, sc, g, ql, true, allowed);
                }
            }
            ATTN_STAMP();                       // tiles done
            l_run += __shfl_xor(l_run, 16, 64);
            l_run += __shfl_xor(l_run, 32, 64);
            const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
            const int qrow = qsub * 16 + ql;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = dt * 16 + g * 4;
                typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
                u32x2 u;
                u.x = (uint32_t)f2bf(oacc[dt][0] * inv) | ((uint32_t)f2bf(oacc[dt][1] * inv) << 16);
                u.y = (uint32_t)f2bf(oacc[dt][2] * inv) | ((uint32_t)f2bf(oacc[dt][3] * inv) << 16);
                const bool ok = (int)(qrow < a.Sq) & (int)(d < a.hd);
                const uint32_t off = ok ? (uint32_t)(qrow * (a.nh * a.hd) + d) * 2u : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(u, rso, off, 0, 0);
            }
            // hand over the Q of the next unit in the stream, ask for the one after it
            int n2_q = qsub + 8, n2_item = item;                // the next unit in the stream ...
            if (n2_q >= nq) { n2_q = wave; n2_item += G; }
            n2_q += 8;                                          // ... and the one after it
            if (n2_q >= nq) { n2_q = wave; n2_item += G; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                // (a real move, here: as a plain assignment the copy is a phi, materialised at the loop's end behind the loads below,
                //  which then land in other registers and are copied once more - after s_waitcnt vmcnt(0) on loads just issued)
                u32x4 t = *reinterpret_cast<const u32x4*>(&qf_next[ks]), r;
                asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                             : "=&v"(r.x), "=&v"(r.y), "=&v"(r.z), "=&v"(r.w) : "v"(t.x), "v"(t.y), "v"(t.z), "v"(t.w));
                qf[ks] = *reinterpret_cast<const bf16x8*>(&r);
            }
            __builtin_amdgcn_sched_barrier(0);
            load_q(n2_item, n2_q, qf_next);
            qsub += 8;
        } while (qsub < nq);
        ATTN_STAMP();                           // units done
        __syncthreads();                        // every wave is done with this item's K/V
        ATTN_STAMP();                           // barrier passed
        if (more) park(rk, rv);
        ATTN_STAMP();                           // parked
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Long unmasked key sequences (SigLIP at 972+ patch tokens: K and V of a head do not fit LDS): the resident kernel's machinery on
// key CHUNKS.  An item = (batch, head, block of 128 queries: one 16-query sub-tile per wave, its (m, l, O) state in registers for
// the whole item); the keys go through LDS in chunks of CHUNK = 256 (four pipelined tiles per wave and chunk, one rendezvous pair
// per chunk instead of the tiled kernel's per 64 keys), the next chunk - of this item, or the first of the next one - travelling
// global -> registers meanwhile.  Against the tiled kernel: K / V are re-read per 128 queries instead of 64, a quarter of the
// barriers, every LDS fragment in registers before its MFMAs, and no global load in a branch.
// ------------------------------------------------------------------------------------------------
#define ATT_CHUNK 256
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {     // compile-time loop (register arrays indexed by constants only)
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int DPK, int DPV, int MAXI>   // MAXI: 16-byte chunks per thread per operand (>= ATT_CHUNK * DPV/8 / 512)
__global__ __launch_bounds__(512, 2)
void attn_chunked_k(AttnP a, int n_items, int qblocks) {
    constexpr int KSTR = lds_stride(DPK), VSTR = lds_stride(DPV);
    constexpr int KS = DPK / 32, DT = DPV / 16;
    constexpr int ROWS = ATT_CHUNK + 16;                    // (the transposing V read touches one 16-row group past the last real one)
    extern __shared__ __attribute__((aligned(16))) char rsm[];
    char* sK = rsm;
    char* sV = rsm + ROWS * KSTR;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, ql = lane & 15;
    const float sc = a.scale * 1.4426950408889634f;
    const int nch = (a.Sk + ATT_CHUNK - 1) / ATT_CHUNK;     // key chunks per item
    const int G = (int)gridDim.x;
    constexpr int HC = DPV / 8;
    constexpr uint32_t OOB = 0xFFFFFFFFu;
    for (int o = tid * 16; o < ROWS * (KSTR + VSTR); o += 512 * 16) *reinterpret_cast<u32x4*>(rsm + o) = u32x4{0u, 0u, 0u, 0u};
    auto rsrc = [](const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xFFFFFFFF, 0x00020000); };
    // item -> (batch, head, query block): the query blocks of one (batch, head) are consecutive items (they share its K / V in L2)
    // The prefetch of the next chunk is cut in four parts, part t issued ahead of tile t of the chunk being multiplied: issued in one
    // go at the head of the chunk, the 8 waves' 80 loads kept the CU's address unit busy for ~2.5 k cycles in which nothing else ran
    // (a quarter of the chunk).  Straight-line code: every part always issues (only a tile's arithmetic is skipped in a ragged chunk).
    auto fetch_part = [&](auto part, int item, int chunk, u32x4 (&rk)[MAXI], u32x4 (&rv)[MAXI]) {
        constexpr int P = decltype(part)::value;             // 0..3, or 4 = all
        constexpr int I0 = P == 4 ? 0 : P * MAXI / 4, I1 = P == 4 ? MAXI : (P + 1) * MAXI / 4;
        const bool item_ok = item < n_items;
        const int pair = (item_ok ? item : 0) / qblocks;
        const int hh = pair % a.nh, bb = pair / a.nh;
        const int kvh = hh / (a.nh / a.nkv);
        const auto rsk = rsrc(a.k + (int64_t)bb * a.kv_bs + (int64_t)kvh * a.hd);
        const auto rsv = rsrc(a.v + (int64_t)bb * a.kv_bs + (int64_t)kvh * a.hd);
        const int key0 = chunk * ATT_CHUNK;
#pragma unroll
        for (int i = I0; i < I1; ++i) {
            const int c = tid + i * 512;
            const int row = c / HC, ch = c % HC;
            const bool ok = (int)item_ok & (int)(row < ATT_CHUNK) & (int)(key0 + row < a.Sk) & (int)(ch * 8 < a.hd);
            const uint32_t off = ok ? (uint32_t)((key0 + row) * (int)a.kv_rs + ch * 8) * 2u : OOB;
            rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rsk, off, 0, 0);
            rv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsv, off, 0, 0);
        }
    };
    // (every row of the chunk is written: keys past Sk arrive as zeros, so a ragged last chunk leaves no stale rows behind)
    auto park = [&](const u32x4 (&rk)[MAXI], const u32x4 (&rv)[MAXI]) {
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int c = tid + i * 512;
            const int row = c / HC, ch = c % HC;
            if (row < ATT_CHUNK && ch * 8 < a.hd) {
                *reinterpret_cast<u32x4*>(sK + row * KSTR + ch * 16) = rk[i];
                *reinterpret_cast<u32x4*>(sV + row * VSTR + ch * 16) = rv[i];
            }
        }
    };
    auto load_q = [&](int item, bf16x8 (&q)[KS]) {
        const bool item_ok = item < n_items;
        const int it = item_ok ? item : 0;
        const int pair = it / qblocks, qb = it % qblocks;
        const int hh = pair % a.nh, bb = pair / a.nh;
        const auto rsq = rsrc(a.q + (int64_t)bb * a.q_bs + (int64_t)hh * a.hd);
        const int qrow = qb * 128 + wave * 16 + ql;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d = ks * 32 + g * 8;
            const bool ok = (int)item_ok & (int)(qrow < a.Sq) & (int)(d < a.hd);
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsq, ok ? (uint32_t)(qrow * (int)a.q_rs + d) * 2u : OOB, 0, 0);
            q[ks] = *reinterpret_cast<const bf16x8*>(&r);
        }
    };

    int item = (int)blockIdx.x;
    bf16x8 qf[KS], qf_next[KS];
    {
        u32x4 rk[MAXI], rv[MAXI];
        fetch_part(std::integral_constant<int, 4>{}, item, 0, rk, rv);
        load_q(item, qf);
        load_q(item + G, qf_next);
        __syncthreads();                        // the zero fill is complete before any real chunk lands on it
        if (item < n_items) park(rk, rv);
        __syncthreads();
    }
    for (; item < n_items; item += G) {
        const int pair = item / qblocks, qb = item % qblocks;
        const int head = pair % a.nh, b = pair / a.nh;
        const auto rso = rsrc(a.o + ((int64_t)b * a.Sq * a.nh + head) * a.hd);
        floatx4 oacc[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i) oacc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        float m_run = -INFINITY, l_run = 0.f;
        const bool wave_live = qb * 128 + wave * 16 < a.Sq;           // (the last query block of a head may not fill all 8 waves)
        int c = 0;
        do {
            const bool last = c + 1 >= nch;
            u32x4 rk[MAXI], rv[MAXI];
            const int nitem = last ? item + G : item, nchunk = last ? 0 : c + 1;
            const int left = min(ATT_CHUNK, a.Sk - c * ATT_CHUNK);     // real keys in this chunk
            const int nfull = wave_live ? left >> 6 : 0, nst_last = wave_live ? ((left & 63) + 15) >> 4 : 0;
            const char* kp = sK + ql * KSTR + g * 16;
            const char* vp = sV + (g * 4 + (ql >> 2)) * VSTR + (ql & 3) * 8;
            bf16x8 kf[4][KS];
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kf[st][ks] = *reinterpret_cast<const bf16x8*>(kp + st * 16 * KSTR + ks * 64);
            auto all = [](int, int) -> bool { return true; };
            static_for<0, 4>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                fetch_part(tc, nitem, nchunk, rk, rv);
                if (t < nfull)
                    attn_tile_pipe<DPK, DPV, 4, false, true>(kf, vp + t * ATT_KB * VSTR, kp + (t + 1) * ATT_KB * KSTR, qf, oacc, m_run, l_run, sc, all);
            });
            if (nst_last) {
                const char* vt = vp + nfull * ATT_KB * VSTR;
                const int rem = left - nfull * ATT_KB;
                auto in_range = [&](int st, int r) -> bool { return st * 16 + g * 4 + r < rem; };
                if (nst_last == 1)      attn_tile_pipe<DPK, DPV, 1, true, false>(kf, vt, nullptr, qf, oacc, m_run, l_run, sc, in_range);
                else if (nst_last == 2) attn_tile_pipe<DPK, DPV, 2, true, false>(kf, vt, nullptr, qf, oacc, m_run, l_run, sc, in_range);
                else if (nst_last == 3) attn_tile_pipe<DPK, DPV, 3, true, false>(kf, vt, nullptr, qf, oacc, m_run, l_run, sc, in_range);
                else                    attn_tile_pipe<DPK, DPV, 4, true, false>(kf, vt, nullptr, qf, oacc, m_run, l_run, sc, in_range);
            }
            ++c;
            __syncthreads();                    // every wave is done with this chunk
            if (!last || item + G < n_items) park(rk, rv);
            __syncthreads();
        } while (c < nch);
        l_run += __shfl_xor(l_run, 16, 64);
        l_run += __shfl_xor(l_run, 32, 64);
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        const int qrow = qb * 128 + wave * 16 + ql;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = dt * 16 + g * 4;
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
            u32x2 u;
            u.x = (uint32_t)f2bf(oacc[dt][0] * inv) | ((uint32_t)f2bf(oacc[dt][1] * inv) << 16);
            u.y = (uint32_t)f2bf(oacc[dt][2] * inv) | ((uint32_t)f2bf(oacc[dt][3] * inv) << 16);
            const bool ok = (int)(qrow < a.Sq) & (int)(d < a.hd);
            __builtin_amdgcn_raw_buffer_store_b64(u, rso, ok ? (uint32_t)(qrow * (a.nh * a.hd) + d) * 2u : OOB, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {       // (a real move: see attn_resident_k)
            u32x4 t = *reinterpret_cast<const u32x4*>(&qf_next[ks]), r;
            asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                         : "=&v"(r.x), "=&v"(r.y), "=&v"(r.z), "=&v"(r.w) : "v"(t.x), "v"(t.y), "v"(t.z), "v"(t.w));
            qf[ks] = *reinterpret_cast<const bf16x8*>(&r);
        }
        __builtin_amdgcn_sched_barrier(0);
        load_q(item + 2 * G, qf_next);
    }
}

#ifdef LICV_ATTN_TRACE
extern "C" int licv_attn_debug_timestamps(void* dev_buffer) {
    long long* p = (long long*)dev_buffer;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_ts), &p, sizeof(p)) == hipSuccess ? LICV_OK : LICV_E_HIP;
}
#endif

static int g_attn_force_tiled = 0;      // tests / A-B timing: bit 0 = never use the resident-K/V variant
static int g_attn_plain_items = 0;      //                     bit 1 = resident variant takes items in blockIdx order (no XCD grouping)
static int g_attn_tiled_no_pipe = 0;    //                     bit 3 = the TILED kernel at head dim 128 leaves the LDS read schedule of its full tiles to the compiler
static int g_attn_no_pipe = 0;          //                     bit 2 = resident variant leaves the LDS read schedule to the compiler (attn_tile)
extern "C" int licv_attn_select(int mode) { g_attn_force_tiled = mode & 1; g_attn_plain_items = (mode >> 1) & 1; g_attn_no_pipe = (mode >> 2) & 1; g_attn_tiled_no_pipe = (mode >> 3) & 1; return LICV_OK; }

extern "C" int licv_attn_fwd(const licv_attn_args* x, void* stream) {
    LICV_CHECK_ARG(x && x->q && x->k && x->v && x->o, "attn_fwd: null pointer");
    LICV_CHECK_ARG(x->B > 0 && x->Sq > 0 && x->Sk > 0 && x->n_heads > 0 && x->n_kv_heads > 0, "attn_fwd: bad shape");
    LICV_CHECK_ARG(x->n_heads % x->n_kv_heads == 0, "attn_fwd: n_heads must be a multiple of n_kv_heads");
    LICV_CHECK_ARG(x->head_dim % 8 == 0 && x->head_dim >= 8 && x->head_dim <= 128, "attn_fwd: head_dim %lld unsupported (multiple of 8, <= 128)", (long long)x->head_dim);
    LICV_CHECK_ARG(x->q_rs % 8 == 0 && x->kv_rs % 8 == 0 && x->q_bs % 8 == 0 && x->kv_bs % 8 == 0, "attn_fwd: strides must be multiples of 8 elements");
    LICV_CHECK_ARG(((uintptr_t)x->q & 15) == 0 && ((uintptr_t)x->k & 15) == 0 && ((uintptr_t)x->v & 15) == 0 && ((uintptr_t)x->o & 7) == 0, "attn_fwd: misaligned pointer");
    LICV_CHECK_ARG(x->mask_mode >= 0 && x->mask_mode <= 3, "attn_fwd: bad mask mode %d", x->mask_mode);
    LICV_CHECK_ARG(x->mask_mode != 3 || (x->img_mask && x->n_img > 0 && x->img_len > 0 && x->n_img * x->img_len >= x->Sk), "attn_fwd: image mask arguments inconsistent");
    LICV_CHECK_ARG(x->Sq < (1ll << 30) && x->Sk < (1ll << 30), "attn_fwd: sequence too long");
    AttnP p;
    p.q = (const bf16_t*)x->q; p.q_bs = x->q_bs; p.q_rs = x->q_rs;
    p.k = (const bf16_t*)x->k; p.v = (const bf16_t*)x->v; p.kv_bs = x->kv_bs; p.kv_rs = x->kv_rs;
    p.o = (bf16_t*)x->o;
    p.B = (int)x->B; p.Sq = (int)x->Sq; p.Sk = (int)x->Sk; p.nh = (int)x->n_heads; p.nkv = (int)x->n_kv_heads; p.hd = (int)x->head_dim;
    p.scale = x->scale; p.mask_mode = x->mask_mode; p.key_valid = x->key_valid;
    p.img_mask = x->img_mask; p.n_img = (int)x->n_img; p.img_len = (int)x->img_len;
    const int hd0 = (int)x->head_dim;
    // (Sq >= 128: with fewer than 8 sub-tiles of 16 queries some of the 8 waves idle — the perceiver's 64 latents run
    //  1.4x faster on the tiled kernel)
    if (x->mask_mode == 0 && !g_attn_force_tiled && hd0 > 32 && hd0 <= 96 && x->Sk >= 128 && x->Sq >= 128) {
        // resident-K/V variant: needs skp * (KSTR + VSTR) bytes of LDS
        const int skp = (int)((x->Sk + 15) / 16 * 16) + 16;       // the transposing V read touches one 16-row group past the last real one
        const int dpk = hd0 <= 64 ? 64 : 96, dpv = hd0 <= 64 ? 64 : (hd0 <= 80 ? 80 : 96);
        const int64_t lds = (int64_t)skp * (lds_stride(dpk) + lds_stride(dpv));
        const int64_t chunks = x->Sk * (dpv / 8);                 // 16-byte chunks per operand per item, 512 threads
        // (the kernel addresses one (batch, head) slice with 32-bit byte offsets from its base)
        const bool fits32 = x->Sq * x->q_rs * 2 < (1ll << 31) && x->Sk * x->kv_rs * 2 < (1ll << 31) && x->Sq * x->n_heads * x->head_dim * 2 < (1ll << 31);
        if (lds <= 160 * 1024 && chunks <= 10 * 512 && fits32) {
            const int n_items = (int)(x->B * x->n_heads);
            const dim3 rgrid((unsigned)(n_items < 256 ? n_items : 256)), rblock(512);
            hipStream_t rst = (hipStream_t)stream;
            const int mi = chunks <= 6 * 512 ? 6 : (chunks <= 8 * 512 ? 8 : 10);
            const int xcd_map = (rgrid.x % 8 == 0 && !g_attn_plain_items) ? 1 : 0;
#define RES_LAUNCH1(DK, DV, MI, PP) do { static bool attr_##DK##_##DV##_##MI##_##PP = false; \
                if (!attr_##DK##_##DV##_##MI##_##PP) { (void)hipFuncSetAttribute((const void*)attn_resident_k<DK, DV, MI, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_##DK##_##DV##_##MI##_##PP = true; } \
                attn_resident_k<DK, DV, MI, PP><<<rgrid, rblock, lds, rst>>>(p, skp, n_items, xcd_map); } while (0)
// (the pipelined walker holds 88 more fragment registers: with 8 or 10 prefetch chunks per operand it spills at 96 columns)
#define RES_LAUNCH(DK, DV, MI) do { if (g_attn_no_pipe || MI > (DK == 64 ? 8 : 6)) RES_LAUNCH1(DK, DV, MI, false); else RES_LAUNCH1(DK, DV, MI, true); } while (0)
            if (dpk == 64)      { if (mi == 6) RES_LAUNCH(64, 64, 6); else if (mi == 8) RES_LAUNCH(64, 64, 8); else RES_LAUNCH(64, 64, 10); }
            else if (dpv == 80) { if (mi == 6) RES_LAUNCH(96, 80, 6); else if (mi == 8) RES_LAUNCH(96, 80, 8); else RES_LAUNCH(96, 80, 10); }
            else                { if (mi == 6) RES_LAUNCH(96, 96, 6); else if (mi == 8) RES_LAUNCH(96, 96, 8); else RES_LAUNCH(96, 96, 10); }
#undef RES_LAUNCH
#undef RES_LAUNCH1
            LICV_LAUNCH_CHECK();
            return LICV_OK;
        }
    }
    if (x->mask_mode == 0 && !g_attn_force_tiled && hd0 > 32 && hd0 <= 96 && x->Sk >= 256 && x->Sq >= 128 &&
        x->Sq * x->q_rs * 2 < (1ll << 31) && x->Sk * x->kv_rs * 2 < (1ll << 31) && x->Sq * x->n_heads * x->head_dim * 2 < (1ll << 31)) {
        // keys too long for the resident variant: the chunked one (128 queries per item, 256 keys per chunk)
        const int dpk = hd0 <= 64 ? 64 : 96, dpv = hd0 <= 64 ? 64 : (hd0 <= 80 ? 80 : 96);
        const int64_t lds = (int64_t)(ATT_CHUNK + 16) * (lds_stride(dpk) + lds_stride(dpv));
        const int qblocks = (int)((x->Sq + 127) / 128);
        const int64_t items = x->B * x->n_heads * qblocks;
        if (items < (1ll << 30)) {
            const dim3 cgrid((unsigned)(items < 256 ? items : 256)), cblock(512);
            hipStream_t cst = (hipStream_t)stream;
#define CHK_LAUNCH(DK, DV, MI) do { static bool attr_c_##DK##_##DV = false; \
                if (!attr_c_##DK##_##DV) { (void)hipFuncSetAttribute((const void*)attn_chunked_k<DK, DV, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_c_##DK##_##DV = true; } \
                attn_chunked_k<DK, DV, MI><<<cgrid, cblock, lds, cst>>>(p, (int)items, qblocks); } while (0)
            if (dpk == 64)      CHK_LAUNCH(64, 64, 4);
            else if (dpv == 80) CHK_LAUNCH(96, 80, 5);
            else                CHK_LAUNCH(96, 96, 6);
#undef CHK_LAUNCH
            LICV_LAUNCH_CHECK();
            return LICV_OK;
        }
    }
    // 64 queries per 4-wave workgroup; the mask mode is a template constant (with it at run time the tile loop carried the
    // image-mask bookkeeping of mode 3 in every instantiation: 3700 lines of ISA, 1600 scalar instructions, SGPR spills;
    // as a constant the mode-0 loop is 350 lines: SigLIP 288 -> 217 us).  Measured and dropped (this kernel is bound by each wave's dependency chain
    // QK -> max -> exp2 -> PV and lives on waves in flight, not on loads or the LDS port):
    //  (i)   several 64-query slabs per workgroup sharing each fetched K/V tile: 745 vs 590 us on the ViT shape;
    //  (ii)  two 16-query sub-tiles per wave sharing every K / V^T fragment read (half the LDS bytes per flop):
    //        1.4-3x slower (240+ VGPRs, spills, half the waves in flight);
    //  (iii) issuing S^T(t+1) before softmax(t) with double-buffered tiles: 1.3-1.4x slower (184 VGPRs -> 2 waves per
    //        SIMD instead of 3, twice the LDS);
    //  (iv)  v_permlane16/32_swap instead of ds_bpermute for the 4-lane reductions: neutral;
    //  (v)   resident kernel, ViT (257 = 16 x 16 + 1 queries: one wave runs 3 sub-tiles, seven run 2): spreading the key tiles of
    //        the lone last sub-tile over the 8 waves and merging the partial (m, l, O) states through LDS: 367 -> 395 us;
    //  (vi)  skipping the accumulator rescale as a wave when alpha == 1 in every lane (bit-identical): ViT 362 -> 372 us, language
    //        self-attention 174 -> 170 us (tools/attn_bench.py): the branch costs what the 20 multiplies did;
    //  (vii) resident kernel with 12 / 16 waves per workgroup (3 / 4 per SIMD, 170 / 128 VGPRs): the hot tile loop is 104 VALU + 22
    //        MFMA + 32 LDS reads + 20 s_waitcnt per 64 keys and runs at ~47 % issue utilisation with 2 waves per SIMD, but the tile
    //        function needs ~190 registers beside the K/V prefetch: 88 / 158 spills, 362 -> 440 / 575 us.  More waves in flight
    //        need a tile function rebuilt for <= 120 registers (K / V^T fragments in halves), not a launch parameter.
    LICV_CHECK_ARG(x->Sk * x->kv_rs * 2 < (1ll << 32) && x->Sk * 4 < (1ll << 32), "attn_fwd: one (batch, head) slice of K / V spans more than 4 GB");
    const int hd = p.hd;
    const int64_t qtiles = (x->Sq + ATT_QB - 1) / ATT_QB;
    const int64_t nblk = x->B * x->n_heads * qtiles;
    LICV_CHECK_ARG(nblk < (1ll << 31), "attn_fwd: grid too large");
    const dim3 grid((unsigned)nblk), block(256);
    hipStream_t st = (hipStream_t)stream;
    // Full tiles through attn_tile_pipe (all K fragments, then all V^T fragments, in registers before the MFMAs that use them) where
    // it does not cost a wave per SIMD: head dim 128 runs at 2 waves per SIMD either way (177 -> 254 VGPRs): language self-attention
    // 138.5 -> 124.9 us (800 tokens), Mistral 2900 tokens 669 -> 546 us; at 96 / 80 it would go from 3 waves to 2: SigLIP 1659 -> 1939 us.
#define ATT_LAUNCH(DK, DV) do { switch (p.mask_mode) { \
        case 0:  if (DK == 128 && !g_attn_tiled_no_pipe) attn_fwd_k<DK, DV, 1, 4, 0, DK == 128><<<grid, block, 0, st>>>(p); else attn_fwd_k<DK, DV, 1, 4, 0><<<grid, block, 0, st>>>(p); break; \
        case 1:  if (DK == 128 && !g_attn_tiled_no_pipe) attn_fwd_k<DK, DV, 1, 4, 1, DK == 128><<<grid, block, 0, st>>>(p); else attn_fwd_k<DK, DV, 1, 4, 1><<<grid, block, 0, st>>>(p); break; \
        case 2:  if (DK == 128 && !g_attn_tiled_no_pipe) attn_fwd_k<DK, DV, 1, 4, 2, DK == 128><<<grid, block, 0, st>>>(p); else attn_fwd_k<DK, DV, 1, 4, 2><<<grid, block, 0, st>>>(p); break; \
        default: attn_fwd_k<DK, DV, 1, 4, 3><<<grid, block, 0, st>>>(p); break; } } while (0)
    if (hd <= 16)        ATT_LAUNCH(32, 16);
    else if (hd <= 32)   ATT_LAUNCH(32, 32);
    else if (hd <= 64)   ATT_LAUNCH(64, 64);
    else if (hd <= 80)   ATT_LAUNCH(96, 80);
    else if (hd <= 96)   ATT_LAUNCH(96, 96);
    else                 ATT_LAUNCH(128, 128);
#undef ATT_LAUNCH
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}
