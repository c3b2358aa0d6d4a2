// One decode step's self-attention block of hooked generate (ref:inference.py:300-321: B x num_beams rows, ONE new token each) as a
// single launch: the fused QKV projection's output (bf16 rows, or the fp32 split-K slices its producer left in the workspace) ->
// rotary on Q and K (hf:idefics/modeling_idefics.py:396-428, each bf16 op rounded) -> K | V appended to the row's KV cache ->
// attention of the new query over the cached keys (hf eager_attention_forward, :450-470) -> O.  It replaces three launches of a
// decode step per layer (rotary + append, the 64-query tiled attention kernel with one live query per workgroup, and - with
// kv_rows - the beam search's gather of the whole cache): the cache is never moved.  kv_rows[r][p] names the PHYSICAL cache row that
// holds position p of beam row r's history (licv_beam_step maintains it: a beam inherits its source beam's row indices and writes
// its own new token into its own row), so a reorder is an index update of B * beams * max_len ints.
//
// One 64-lane wave per (row, kv head); it serves the n_heads / n_kv_heads query heads of that group one after the other.  Sweeps of
// 64 keys, one key per lane: the lane requests its key's K row and V row together (16-byte loads; nothing waits for the softmax),
// takes the 128-dim dot with q from registers, and the sweep's softmax runs in the log2 domain with P rounded to bf16 before the PV
// product and the sum taken over the unrounded P - the arithmetic of csrc/attention.hip's tile function, online across sweeps; the V
// rows and probabilities then pass through LDS and every lane accumulates two head-dim columns over the keys in order.  Dependent
// memory round trips per call: (row table, positions, QKV slices) -> (cos / sin) -> (K and V rows).  The new token's own K / V
// never make a round trip through memory.
#include "common.h"

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_d;
#define DEC_VSTR 272                  // LDS stride of a V row (256 B + 16: conflict-free 16-byte row writes, 2-byte column reads)

struct DecodeP {
    const float* ws; int splits; int64_t slice, stride;       // split-K slices of the QKV projection (ws == nullptr: qkv16)
    const bf16_t* qkv16; int64_t ldq;
    const bf16_t* cosT; const bf16_t* sinT; const int64_t* pos; int64_t n_pos;
    bf16_t* cache; int64_t max_len;
    const int32_t* kv_rows; int64_t ld_rows;
    const int32_t* key_valid;
    bf16_t* out;
    int M, Sk, past, nh, nkv, hd;
    float scale;
};

// One element of the fused QKV row: the bf16 value, or bf16(slice 0 + slice 1 + ... in slice order), as skinny_finalize_k adds them.
// Every slice is requested before the first is used (clamped indices, selects instead of branches): a loop of load -> add would put
// one memory round trip per slice on the wave's critical path.
__device__ __forceinline__ float dec_in(const DecodeP& a, int64_t row, int64_t col) {
    if (!a.ws) return bf2f(a.qkv16[row * a.ldq + col]);
    const float* p = a.ws + row * a.stride + col;
    float v = 0.f;
    for (int s0 = 0; s0 < a.splits; s0 += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = p[(int64_t)min(s0 + u, a.splits - 1) * a.slice];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += (s0 + u < a.splits) ? t[u] : 0.f;
    }
    return rbf(v);
}

__global__ __launch_bounds__(64)
void decode_attn_k(DecodeP a) {
    extern __shared__ int srow_s[];                     // [Sk] ints: physical cache row of every key, -1 = masked
    __shared__ __attribute__((aligned(16))) bf16_t qs[128];
    const int lane = threadIdx.x;
    const int r = blockIdx.x / a.nkv, g = blockIdx.x % a.nkv;
    const int hd = a.hd, half = hd >> 1, rep = a.nh / a.nkv;
    const int64_t qd = (int64_t)a.nh * hd, kd = (int64_t)a.nkv * hd;
    const bool act = lane < half;
    const float sc = a.scale * 1.4426950408889634f;
    const int nch = hd >> 3;                            // 16-byte chunks per K / V row
    const int32_t* rows_r = a.kv_rows ? a.kv_rows + (int64_t)r * a.ld_rows : nullptr;
    const int32_t* valid_r = a.key_valid ? a.key_valid + (int64_t)r * a.Sk : nullptr;
    int* srow = srow_s;
    // ---- memory round trip 1: the position, this lane's row-table entry and mask for the FIRST sweep, and the raw elements of K, V and
    // the first query head (bf16 values or sums of split-K slices) - nothing here depends on anything else
    int64_t p = a.pos[r];
    const bool in1 = lane < a.Sk;
    const int pr1_raw = (in1 && rows_r) ? rows_r[lane] : r;
    const int vd1 = (in1 && valid_r) ? valid_r[lane] : 1;
    float klo = 0.f, khi = 0.f, v0 = 0.f, v1 = 0.f, qlo = 0.f, qhi = 0.f;
    if (act) {
        klo = dec_in(a, r, qd + (int64_t)g * hd + lane); khi = dec_in(a, r, qd + (int64_t)g * hd + lane + half);
        v0 = dec_in(a, r, qd + kd + (int64_t)g * hd + lane); v1 = dec_in(a, r, qd + kd + (int64_t)g * hd + lane + half);
        qlo = dec_in(a, r, (int64_t)(g * rep) * hd + lane); qhi = dec_in(a, r, (int64_t)(g * rep) * hd + lane + half);
    }
    for (int j = lane + 64; j < a.Sk; j += 64) {       // (longer histories: the rest of the table goes to LDS)
        const int pr = rows_r ? rows_r[j] : r;
        const bool ok = !valid_r || valid_r[j] != 0;
        srow[j] = ok ? pr : -1;
    }
    const int pr1 = (in1 && vd1 != 0) ? pr1_raw : -1;
    if (in1) srow[lane] = pr1;
    // ---- round trip 2: cos / sin of the position, and the K and V rows of the first sweep's keys (one key per lane) - they need the
    // position and the table entry only, not the rotary
    p = p < 0 ? 0 : (p >= a.n_pos ? a.n_pos - 1 : p);
    const float c = act ? bf2f(a.cosT[p * hd + lane]) : 0.f, s = act ? bf2f(a.sinT[p * hd + lane]) : 0.f;
    u32x4_d kv1[16], vv1[16];
    {
        const bf16_t* rowp = a.cache + ((int64_t)(pr1 >= 0 ? pr1 : r) * a.max_len + (in1 ? lane : 0)) * 2 * kd + (int64_t)g * hd;
        const u32x4_d* kp = reinterpret_cast<const u32x4_d*>(rowp);
        const u32x4_d* vp = reinterpret_cast<const u32x4_d*>(rowp + kd);
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) kv1[ch] = ch < nch ? kp[ch] : u32x4_d{0u, 0u, 0u, 0u};
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) vv1[ch] = ch < nch ? vp[ch] : u32x4_d{0u, 0u, 0u, 0u};
    }
    // ---- the new token's K (rotated) and V of this kv head: into the cache row of THIS beam row at position `past`, and kept in registers
    // (the loads above read position `past` of no row: a row's own key is taken from registers below)
    float k0 = 0.f, k1 = 0.f;
    bf16_t* crow = a.cache + ((int64_t)r * a.max_len + a.past) * 2 * kd + (int64_t)g * hd;
    if (act) {
        k0 = rbf(rbf(klo * c) + rbf(-khi * s));
        k1 = rbf(rbf(khi * c) + rbf(klo * s));
        crow[lane] = f2bf(k0); crow[lane + half] = f2bf(k1);
        crow[kd + lane] = f2bf(v0); crow[kd + lane + half] = f2bf(v1);
    }
    __shared__ __attribute__((aligned(16))) char vt[64 * DEC_VSTR];     // the chunk's V rows (bf16), one per key
    __shared__ float pch[64];                                            // ... and their probabilities
    for (int qh = 0; qh < rep; ++qh) {
        const int head = g * rep + qh;
        // ---- Q of this head, rotated; its score against the new key from registers
        float q0 = 0.f, q1 = 0.f;
        if (act) {
            const float lo = qh == 0 ? qlo : dec_in(a, r, (int64_t)head * hd + lane), hi = qh == 0 ? qhi : dec_in(a, r, (int64_t)head * hd + lane + half);
            q0 = rbf(rbf(lo * c) + rbf(-hi * s));
            q1 = rbf(rbf(hi * c) + rbf(lo * s));
            qs[lane] = f2bf(q0); qs[lane + half] = f2bf(q1);
        }
        const float s_new = wave_sum(q0 * k0 + q1 * k1);
        __syncthreads();
        u32x4_d qreg[16];
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) qreg[ch] = ch < nch ? reinterpret_cast<const u32x4_d*>(qs)[ch] : u32x4_d{0u, 0u, 0u, 0u};
        // ---- 64 keys per sweep, ONE key per lane: its K row and its V row are requested together (the V loads do not wait for the
        // softmax), scores -> online softmax across sweeps (the arithmetic of csrc/attention.hip's tiles) -> the V rows and the bf16
        // probabilities go through LDS and every lane then accumulates two head-dim columns over the sweep's keys in order
        float m_run = -INFINITY, l_run = 0.f, o0 = 0.f, o1 = 0.f;
        for (int j0 = 0; j0 < a.Sk; j0 += 64) {
            const int j = j0 + lane;
            const bool in = j < a.Sk;
            const int pr = in ? srow[j] : -1;
            const bool ok = pr >= 0, own = in && j == a.past;
            const bf16_t* rowp = a.cache + ((int64_t)(ok ? pr : r) * a.max_len + (in ? j : 0)) * 2 * kd + (int64_t)g * hd;
            const u32x4_d* kp = reinterpret_cast<const u32x4_d*>(rowp);
            const u32x4_d* vp = reinterpret_cast<const u32x4_d*>(rowp + kd);
            u32x4_d kv[16], vv[16];
            if (j0 == 0) {                                                        // requested in round trip 2, the same for every query head
#pragma unroll
                for (int ch = 0; ch < 16; ++ch) { kv[ch] = kv1[ch]; vv[ch] = vv1[ch]; }
            } else {
#pragma unroll
                for (int ch = 0; ch < 16; ++ch) kv[ch] = ch < nch ? kp[ch] : u32x4_d{0u, 0u, 0u, 0u};
#pragma unroll
                for (int ch = 0; ch < 16; ++ch) vv[ch] = ch < nch ? vp[ch] : u32x4_d{0u, 0u, 0u, 0u};
            }
            float acc = 0.f;
#pragma unroll
            for (int ch = 0; ch < 16; ++ch)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc = __builtin_fmaf(__uint_as_float(kv[ch][e] << 16), __uint_as_float(qreg[ch][e] << 16), acc);
                    acc = __builtin_fmaf(__uint_as_float(kv[ch][e] & 0xffff0000u), __uint_as_float(qreg[ch][e] & 0xffff0000u), acc);
                }
            float sj = own ? s_new : acc;
            if (!ok) sj = -INFINITY;
            const float cmax = wave_max(sj);
            const float m_new = fmaxf(m_run, cmax * sc);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);            // m_run = -inf -> 0
            const float pj = __builtin_amdgcn_exp2f(__builtin_fmaf(sj, sc, -m_use));
            l_run = l_run * alpha + wave_sum(pj);
            m_run = m_new;
            __syncthreads();                                                      // the previous sweep's readers are done with vt / pch
            pch[lane] = rbf(pj);
#pragma unroll
            for (int ch = 0; ch < 16; ++ch) if (ch < nch) *reinterpret_cast<u32x4_d*>(vt + lane * DEC_VSTR + ch * 16) = vv[ch];
            __syncthreads();
            o0 *= alpha; o1 *= alpha;
            if (act) {
                const int nk = min(64, a.Sk - j0);
                for (int jj = 0; jj < nk; ++jj) {
                    const float pv = pch[jj];
                    const bf16_t* vr = reinterpret_cast<const bf16_t*>(vt + jj * DEC_VSTR);
                    const bool ownj = (j0 + jj == a.past);                        // the new token's V comes from registers, not from the store just issued
                    const bool live = pv != 0.f;                                  // (a masked key contributes nothing, whatever its cache row holds)
                    o0 = __builtin_fmaf(pv, live ? (ownj ? v0 : bf2f(vr[lane])) : 0.f, o0);
                    o1 = __builtin_fmaf(pv, live ? (ownj ? v1 : bf2f(vr[lane + half])) : 0.f, o1);
                }
            }
        }
        if (act) {
            const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
            bf16_t* op = a.out + (int64_t)r * qd + (int64_t)head * hd;
            op[lane] = f2bf(o0 * inv); op[lane + half] = f2bf(o1 * inv);
        }
        __syncthreads();                               // qs / vt / pch are rewritten by the next query head
    }
}

extern "C" int licv_decode_attn(const licv_decode_attn_args* x, void* stream) {
    LICV_CHECK_ARG(x && (x->qkv_ws || x->qkv_bf16) && x->cos && x->sin && x->position_ids && x->cache && x->out, "decode_attn: null pointer");
    LICV_CHECK_ARG(x->M > 0 && x->past >= 0 && x->past < x->max_len, "decode_attn: bad rows / past (%lld of %lld)", (long long)x->past, (long long)x->max_len);
    LICV_CHECK_ARG(x->n_heads > 0 && x->n_kv_heads > 0 && x->n_heads % x->n_kv_heads == 0, "decode_attn: n_heads must be a multiple of n_kv_heads");
    LICV_CHECK_ARG(x->head_dim >= 16 && x->head_dim <= 128 && x->head_dim % 16 == 0, "decode_attn: head_dim %lld unsupported (multiple of 16, <= 128)", (long long)x->head_dim);
    LICV_CHECK_ARG(!x->qkv_ws || (x->splits >= 1 && x->slice_elems > 0 && x->row_stride > 0), "decode_attn: bad split-K workspace description");
    LICV_CHECK_ARG(x->qkv_ws || x->ldq >= (x->n_heads + 2 * x->n_kv_heads) * x->head_dim, "decode_attn: ldq smaller than the fused row");
    LICV_CHECK_ARG(((uintptr_t)x->cache & 15) == 0, "decode_attn: the cache must be 16-byte aligned");
    LICV_CHECK_ARG(!x->kv_rows || x->ld_kv_rows > x->past, "decode_attn: kv_rows rows shorter than the history");
    DecodeP p;
    p.ws = x->qkv_ws; p.splits = x->splits; p.slice = x->slice_elems; p.stride = x->row_stride;
    p.qkv16 = (const bf16_t*)x->qkv_bf16; p.ldq = x->ldq;
    p.cosT = (const bf16_t*)x->cos; p.sinT = (const bf16_t*)x->sin; p.pos = x->position_ids; p.n_pos = x->n_pos;
    p.cache = (bf16_t*)x->cache; p.max_len = x->max_len; p.kv_rows = x->kv_rows; p.ld_rows = x->ld_kv_rows; p.key_valid = x->key_valid;
    p.out = (bf16_t*)x->out;
    p.M = (int)x->M; p.Sk = (int)x->past + 1; p.past = (int)x->past; p.nh = (int)x->n_heads; p.nkv = (int)x->n_kv_heads; p.hd = (int)x->head_dim;
    p.scale = x->scale;
    const size_t lds = (size_t)p.Sk * sizeof(int);
    LICV_CHECK_ARG(lds <= 40 * 1024, "decode_attn: history of %d keys exceeds the kernel's LDS budget", p.Sk);
    decode_attn_k<<<dim3((unsigned)(x->M * x->n_kv_heads)), dim3(64), lds, (hipStream_t)stream>>>(p);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}
