// One decode step of the beam search of hooked generate (ref:inference.py:300-321 -> transformers GenerationMixin._beam_search, 5.x
// vectorised form, generation/utils.py:3077-3460) as ONE launch: log-softmax of every beam's logits, the top 2*num_beams
// continuations of every question over (num_beams x vocab), and the bookkeeping on the running / finished beam sets (the torch
// restatement of which was ~60 small ATen launches per step: licv/generation.py round 3).
//
// Two kernels on the stream, one C call.  beam_scan_k: BEAM_CHUNKS workgroups per question cut the vocabulary; each leaves, per beam,
// its chunk's (max, sum of exp) and its KEEP best logits with their token ids (within one beam the order of lp = ((x - max) - log(sum))
// + running_score is the order of x).  beam_finish_k, one workgroup per question: (1) combines the chunk statistics into each beam's
// log-sum-exp; (2)-(3) lp of the nb * BEAM_CHUNKS * KEEP surviving candidates in fp32, the order torch evaluates it in, and KEEP rounds
// of a block-wide arg-max; (4) lane 0 does the search bookkeeping of its question (a few dozen scalar operations) and leaves a copy
// plan in LDS; (5) all lanes copy the token rows of the new running / finished sets and the KV row table; the last workgroup to finish
// combines the per-question "keep going" flags.  Ties are broken towards the LOWER flat candidate index (beam * vocab + token)
// everywhere - a stable descending sort.  (The first version did everything in ONE workgroup per question: 8 workgroups reading
// 3 x 32002 logits three times took 278 us per step.)
#include "common.h"
#include <limits.h>

#define BEAM_THREADS 256
#define BEAM_MAX_NB 8
#define BEAM_CHUNKS 32                    // workgroups per question in the scan
#define BEAM_CHUNK_MAX 2048               // vocabulary entries per chunk the scan's LDS image holds (V <= 65536)

struct BeamCand { float v; int i; };
__device__ __forceinline__ bool cand_better(float av, int ai, float bv, int bi) { return av > bv || (av == bv && ai < bi); }

__device__ __forceinline__ float beam_block_reduce(float v, bool is_max, float* red) {
    v = is_max ? wave_max(v) : wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < BEAM_THREADS / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}

// scratch layout (floats / ints, per question b, beam k, chunk c): stats[((b*nb + k)*CHUNKS + c)*2 + {0: max, 1: sum of exp(x - max)}],
// then cand_x[((b*nb + k)*CHUNKS + c)*KEEP + j], then cand_i (token ids, int32) at the same index
__host__ __device__ inline int64_t beam_scratch_floats(int64_t B, int64_t nb) { return B * nb * BEAM_CHUNKS * (2 + 2 * 2 * nb); }

template <int KEEP>
__global__ __launch_bounds__(BEAM_THREADS)
void beam_scan_k(licv_beam_step_args a, float* __restrict__ stats, float* __restrict__ cand_x, int* __restrict__ cand_i) {
    constexpr int NB = KEEP / 2;
    __shared__ float sx[BEAM_CHUNK_MAX];
    __shared__ float red[BEAM_THREADS / 64];
    __shared__ float s_wv[BEAM_THREADS / 64]; __shared__ int s_wi[BEAM_THREADS / 64];
    const int b = blockIdx.x / BEAM_CHUNKS, c = blockIdx.x % BEAM_CHUNKS, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t V = a.V;
    const int ch = (int)((V + BEAM_CHUNKS - 1) / BEAM_CHUNKS);
    const int i0 = c * ch, n = max(0, min(ch, (int)V - i0));
    for (int beam = 0; beam < NB; ++beam) {
        const int64_t row = (int64_t)b * a.q_stride_rows + (int64_t)beam * a.beam_stride_rows;
        float mx = -INFINITY;
        for (int i = tid; i < n; i += BEAM_THREADS) {
            const float x = a.logits_dtype == LICV_F32 ? reinterpret_cast<const float*>(a.logits)[row * a.ld + i0 + i]
                                                       : bf2f(reinterpret_cast<const bf16_t*>(a.logits)[row * a.ld + i0 + i]);
            sx[i] = x;
            mx = fmaxf(mx, x);
        }
        mx = beam_block_reduce(mx, true, red);              // (its barriers also publish sx)
        float z = 0.f;
        for (int i = tid; i < n; i += BEAM_THREADS) z += expf(sx[i] - mx);
        z = beam_block_reduce(z, false, red);
        const int64_t slot = ((int64_t)b * NB + beam) * BEAM_CHUNKS + c;
        if (tid == 0) { stats[slot * 2] = mx; stats[slot * 2 + 1] = z; }
        if (a.suppress_eos && a.eos >= i0 && a.eos < i0 + n && tid == 0) sx[a.eos - i0] = -INFINITY;   // ranked as -inf, still part of the sum
        __syncthreads();
        // KEEP rounds of a block arg-max over the chunk image (ties: the lower token id); a taken entry becomes NaN-free "-inf, skipped"
        for (int r = 0; r < KEEP; ++r) {
            float bv = -INFINITY; int bi = INT_MAX;
            for (int i = tid; i < n; i += BEAM_THREADS) { const float x = sx[i]; if (x == x && cand_better(x, i, bv, bi)) { bv = x; bi = i; } }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (cand_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
            }
            __syncthreads();
            if (lane == 0) { s_wv[wave] = bv; s_wi[wave] = bi; }
            __syncthreads();
            float gv = s_wv[0]; int gi = s_wi[0];
            for (int w = 1; w < BEAM_THREADS / 64; ++w) if (cand_better(s_wv[w], s_wi[w], gv, gi)) { gv = s_wv[w]; gi = s_wi[w]; }
            if (tid == 0) {
                cand_x[slot * KEEP + r] = gv;
                cand_i[slot * KEEP + r] = gi == INT_MAX ? INT_MAX : i0 + gi;
                if (gi != INT_MAX) sx[gi] = __int_as_float(0x7fc00000);       // taken: NaN, skipped by `x == x`
            }
            __syncthreads();
        }
    }
}


template <int KEEP>
__global__ __launch_bounds__(BEAM_THREADS)
void beam_finish_k(licv_beam_step_args a, const float* __restrict__ stats, const float* __restrict__ cand_x, const int* __restrict__ cand_i) {
    constexpr int NB = KEEP / 2;
    constexpr int NC = NB * BEAM_CHUNKS * KEEP;                   // surviving candidates of a question
    __shared__ float s_lse[BEAM_MAX_NB], s_mx[BEAM_MAX_NB];
    __shared__ float s_wv[BEAM_THREADS / 64]; __shared__ int s_wi[BEAM_THREADS / 64];
    __shared__ int s_wp[BEAM_THREADS / 64];
    __shared__ float lpv[NC]; __shared__ int lpi[NC];
    __shared__ float top_lp[KEEP]; __shared__ int top_ix[KEEP];
    __shared__ int plan_run_src[NB], plan_run_tok[NB];            // new running row r <- running_in[src] with [cur] = tok
    __shared__ int plan_fin_src[NB], plan_fin_tok[NB];            // new finished row f <- finished_in[src] (tok < 0) or running_in[src] with [cur] = tok
    const int b = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t V = a.V;
    // ---- (1) per-beam log-softmax statistics from the chunk partials: max over chunks, sum of exp rescaled to it
    if (tid < NB) {
        const float* st = stats + ((int64_t)b * NB + tid) * BEAM_CHUNKS * 2;
        float mx = -INFINITY;
        for (int c = 0; c < BEAM_CHUNKS; ++c) mx = fmaxf(mx, st[c * 2]);
        float z = 0.f;
        for (int c = 0; c < BEAM_CHUNKS; ++c) z += st[c * 2 + 1] * expf(st[c * 2] - mx);     // (an empty chunk: exp(-inf) = 0 times 0)
        s_mx[tid] = mx; s_lse[tid] = logf(z);
    }
    __syncthreads();
    // ---- (2) lp of every surviving candidate, in torch's order of evaluation
    for (int i = tid; i < NC; i += BEAM_THREADS) {
        const int beam = i / (BEAM_CHUNKS * KEEP);
        const float x = cand_x[(int64_t)b * NC + i];
        const int tok = cand_i[(int64_t)b * NC + i];
        float v = ((x - s_mx[beam]) - s_lse[beam]) + a.run_scores_in[(int64_t)b * NB + beam];
        if (tok == INT_MAX) v = -INFINITY;
        lpv[i] = v;
        lpi[i] = tok == INT_MAX ? INT_MAX : (int)(beam * V + tok);
    }
    __syncthreads();
    // ---- (3) KEEP rounds of a block arg-max (ties: the lower flat index)
    for (int r = 0; r < KEEP; ++r) {
        float bv = -INFINITY; int bi = INT_MAX, bp = -1;
        for (int i = tid; i < NC; i += BEAM_THREADS) {
            const int fi = lpi[i];
            if (fi >= 0 && fi != INT_MAX && (bp < 0 || cand_better(lpv[i], fi, bv, bi))) { bv = lpv[i]; bi = fi; bp = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64), op = __shfl_xor(bp, o, 64);
            if (op >= 0 && (bp < 0 || cand_better(ov, oi, bv, bi))) { bv = ov; bi = oi; bp = op; }
        }
        __syncthreads();
        if (lane == 0) { s_wv[wave] = bv; s_wi[wave] = bi; s_wp[wave] = bp; }
        __syncthreads();
        float gv = s_wv[0]; int gi = s_wi[0], gp = s_wp[0];
        for (int w = 1; w < BEAM_THREADS / 64; ++w)
            if (s_wp[w] >= 0 && (gp < 0 || cand_better(s_wv[w], s_wi[w], gv, gi))) { gv = s_wv[w]; gi = s_wi[w]; gp = s_wp[w]; }
        __syncthreads();
        if (tid == 0) { top_lp[r] = gp >= 0 ? gv : -INFINITY; top_ix[r] = gp >= 0 ? gi : 0; if (gp >= 0) lpi[gp] = -1; }       // taken
        __syncthreads();                                   // ... before the next round scans the list
    }
    __syncthreads();
    // ---- (4) bookkeeping of this question (hf:generation/utils.py _beam_search: running beams, finished set, early-stop heuristic)
    const int64_t L = a.max_len, cur = a.cur, P = a.P;
    const int64_t* run_in = a.running_in + (int64_t)b * NB * L;
    const int64_t* fin_in = a.finished_in + (int64_t)b * NB * L;
    if (tid == 0) {
        int src[KEEP], tok[KEEP]; bool hits[KEEP]; float run_lp[KEEP];
        bool all_hits = true;
#pragma unroll
        for (int j = 0; j < KEEP; ++j) {
            src[j] = (int)(top_ix[j] / V); tok[j] = (int)(top_ix[j] % V);
            hits[j] = (cur + 1 >= L) || (a.eos >= 0 && tok[j] == a.eos);
            all_hits = all_hits && hits[j];
            run_lp[j] = top_lp[j] + (hits[j] ? 1.0f : 0.0f) * -1.0e9f;
        }
        // next running beams: the NB best continuations that did not just stop (stable selection)
        bool used[KEEP];
#pragma unroll
        for (int j = 0; j < KEEP; ++j) used[j] = false;
        float new_run0 = 0.f;
        for (int r = 0; r < NB; ++r) {
            int best = -1;
#pragma unroll
            for (int j = 0; j < KEEP; ++j) if (!used[j] && (best < 0 || run_lp[j] > run_lp[best])) best = j;
            used[best] = true;
            plan_run_src[r] = src[best]; plan_run_tok[r] = tok[best];
            a.run_scores_out[(int64_t)b * NB + r] = run_lp[best];
            a.beam_src_flat[(int64_t)b * NB + r] = (int64_t)b * NB + src[best];
            a.next_tokens[(int64_t)b * NB + r] = tok[best];
            if (r == 0) new_run0 = run_lp[best];
        }
        // finished set: only the top NB candidates may finalise
        bool all_fin_in = true;
        for (int f = 0; f < NB; ++f) all_fin_in = all_fin_in && a.is_fin_in[(int64_t)b * NB + f] != 0;
        const bool improve_in = a.improve_in[b] != 0;
        const float denom = (float)pow((double)(cur + 1 - P), (double)a.length_penalty);
        float m_sc[NB + KEEP]; bool m_fin[NB + KEEP]; int64_t m_len[NB + KEEP];
        for (int f = 0; f < NB; ++f) { m_sc[f] = a.fin_scores_in[(int64_t)b * NB + f]; m_fin[f] = a.is_fin_in[(int64_t)b * NB + f] != 0; m_len[f] = a.gen_len_in[(int64_t)b * NB + f]; }
#pragma unroll
        for (int j = 0; j < KEEP; ++j) {
            const bool just = hits[j] && j < NB;
            float s = top_lp[j] / denom;
            s = s + ((all_fin_in && a.early_stopping) ? 1.0f : 0.0f) * -1.0e9f;
            s = s + (improve_in ? 0.0f : 1.0f) * -1.0e9f;
            s = s + (just ? 0.0f : 1.0f) * -1.0e9f;
            m_sc[NB + j] = s; m_fin[NB + j] = just; m_len[NB + j] = cur + 1 - P;
        }
        bool mused[NB + KEEP];
        for (int j = 0; j < NB + KEEP; ++j) mused[j] = false;
        float new_fs[NB]; bool new_if[NB];
        bool all_fin_out = true;
        for (int f = 0; f < NB; ++f) {
            int best = -1;
            for (int j = 0; j < NB + KEEP; ++j) if (!mused[j] && (best < 0 || m_sc[j] > m_sc[best])) best = j;
            mused[best] = true;
            if (best < NB) { plan_fin_src[f] = best; plan_fin_tok[f] = -1; }
            else { plan_fin_src[f] = src[best - NB]; plan_fin_tok[f] = tok[best - NB]; }
            new_fs[f] = m_sc[best]; new_if[f] = m_fin[best];
            a.fin_scores_out[(int64_t)b * NB + f] = m_sc[best];
            a.is_fin_out[(int64_t)b * NB + f] = m_fin[best] ? 1 : 0;
            a.gen_len_out[(int64_t)b * NB + f] = m_len[best];
            all_fin_out = all_fin_out && m_fin[best];
        }
        // early-stop heuristic (early_stopping = False form): can the best running beam still beat the worst finished one?
        const float best_run = new_run0 / (float)pow((double)(cur + 1 - P), (double)a.length_penalty);
        float minfin = new_fs[0];
        for (int f = 1; f < NB; ++f) minfin = fminf(minfin, new_fs[f]);
        bool any_better = false;
        for (int f = 0; f < NB; ++f) any_better = any_better || (best_run > (new_if[f] ? minfin : -1.0e9f));
        const bool improve_out = improve_in && any_better;
        a.improve_out[b] = improve_out ? 1 : 0;
        // per-question contributions to the loop condition, combined by the last workgroup
        if (improve_out) atomicAdd(&a.sync[1], 1);
        if (all_hits) atomicAdd(&a.sync[2], 1);
        if (all_fin_out) atomicAdd(&a.sync[3], 1);
    }
    __syncthreads();
    // ---- (5) token rows of the new running / finished sets
    int64_t* run_out = a.running_out + (int64_t)b * NB * L;
    int64_t* fin_out = a.finished_out + (int64_t)b * NB * L;
    for (int r = 0; r < NB; ++r) {
        const int64_t* s = run_in + (int64_t)plan_run_src[r] * L;
        const int64_t t = plan_run_tok[r];
        for (int64_t c = tid; c < L; c += BEAM_THREADS) run_out[(int64_t)r * L + c] = (c == cur) ? t : s[c];
        const int64_t ft = plan_fin_tok[r];
        const int64_t* fs = (ft < 0 ? fin_in : run_in) + (int64_t)plan_fin_src[r] * L;
        for (int64_t c = tid; c < L; c += BEAM_THREADS) fin_out[(int64_t)r * L + c] = (ft >= 0 && c == cur) ? ft : fs[c];
    }
    // ---- the KV-cache row table of licv_decode_attn: a beam inherits its source beam's history rows and owns its next token's row
    if (a.kv_rows_in && a.kv_rows_out) {
        for (int r = 0; r < NB; ++r) {
            const int32_t* s = a.kv_rows_in + ((int64_t)b * NB + plan_run_src[r]) * a.kv_ld;
            int32_t* d = a.kv_rows_out + ((int64_t)b * NB + r) * a.kv_ld;
            for (int64_t c = tid; c < a.kv_ld; c += BEAM_THREADS) d[c] = (c == cur) ? (int32_t)(b * NB + r) : s[c];
        }
    }
    // ---- loop condition: the last workgroup to arrive combines the counters and clears them for the next step
    if (tid == 0) {
        __threadfence();
        const int arrived = atomicAdd(&a.sync[0], 1);
        if (arrived == (int)gridDim.x - 1) {
            const int n_improve = atomicExch(&a.sync[1], 0), n_hits = atomicExch(&a.sync[2], 0), n_fin = atomicExch(&a.sync[3], 0);
            atomicExch(&a.sync[0], 0);
            bool unfinished = n_improve > 0 && n_hits < (int)gridDim.x;
            if (a.early_stopping) unfinished = unfinished && n_fin < (int)gridDim.x;
            a.flags[0] = unfinished ? 1 : 0;
        }
    }
}

extern "C" int64_t licv_beam_step_scratch_bytes(int64_t B, int64_t nb) { return beam_scratch_floats(B, nb) * 4; }

extern "C" int licv_beam_step(const licv_beam_step_args* x, void* stream) {
    LICV_CHECK_ARG(x && x->logits && x->running_in && x->finished_in && x->run_scores_in && x->fin_scores_in && x->is_fin_in && x->improve_in &&
                   x->gen_len_in && x->running_out && x->finished_out && x->run_scores_out && x->fin_scores_out && x->is_fin_out &&
                   x->improve_out && x->gen_len_out && x->beam_src_flat && x->next_tokens && x->flags && x->sync && x->scratch, "beam_step: null pointer");
    LICV_CHECK_ARG(x->logits_dtype == LICV_BF16 || x->logits_dtype == LICV_F32, "beam_step: bad logits dtype");
    LICV_CHECK_ARG(x->nb >= 1 && x->nb <= BEAM_MAX_NB, "beam_step: num_beams %lld outside 1..%d", (long long)x->nb, BEAM_MAX_NB);
    LICV_CHECK_ARG(x->B >= 1 && x->V >= 2 * x->nb && x->ld >= x->V && x->nb * x->V < (1ll << 31), "beam_step: bad batch / vocabulary size");
    LICV_CHECK_ARG(x->V <= (int64_t)BEAM_CHUNKS * BEAM_CHUNK_MAX, "beam_step: vocabulary %lld above the scan's limit %d", (long long)x->V, BEAM_CHUNKS * BEAM_CHUNK_MAX);
    LICV_CHECK_ARG(x->scratch_bytes >= licv_beam_step_scratch_bytes(x->B, x->nb) && ((uintptr_t)x->scratch & 15) == 0,
                   "beam_step: scratch smaller than licv_beam_step_scratch_bytes(B, nb) or misaligned");
    LICV_CHECK_ARG(x->cur >= x->P && x->cur < x->max_len && x->P >= 1, "beam_step: position %lld outside [P, max_len)", (long long)x->cur);
    LICV_CHECK_ARG(x->running_in != x->running_out && x->finished_in != x->finished_out, "beam_step: the token rows need separate in / out buffers");
    LICV_CHECK_ARG((x->kv_rows_in == nullptr) == (x->kv_rows_out == nullptr) && (!x->kv_rows_in || (x->kv_rows_in != x->kv_rows_out && x->kv_ld > x->cur)),
                   "beam_step: the KV row table needs separate in / out buffers with rows longer than the current position");
    const int64_t slots = x->B * x->nb * BEAM_CHUNKS;
    float* stats = (float*)x->scratch;
    float* cand_x = stats + slots * 2;
    int* cand_i = (int*)(cand_x + slots * 2 * x->nb);
    const dim3 sgrid((unsigned)(x->B * BEAM_CHUNKS)), grid((unsigned)x->B), block(BEAM_THREADS);
    hipStream_t st = (hipStream_t)stream;
#define BEAM_LAUNCH(K) do { beam_scan_k<K><<<sgrid, block, 0, st>>>(*x, stats, cand_x, cand_i); beam_finish_k<K><<<grid, block, 0, st>>>(*x, stats, cand_x, cand_i); } while (0)
    switch (x->nb) {
        case 1: BEAM_LAUNCH(2); break;
        case 2: BEAM_LAUNCH(4); break;
        case 3: BEAM_LAUNCH(6); break;
        case 4: BEAM_LAUNCH(8); break;
        case 5: BEAM_LAUNCH(10); break;
        case 6: BEAM_LAUNCH(12); break;
        case 7: BEAM_LAUNCH(14); break;
        default: BEAM_LAUNCH(16); break;
    }
#undef BEAM_LAUNCH
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}
