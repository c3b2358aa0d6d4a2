// One decode step of the beam search of hooked generate (ref:inference.py:300-321 -> transformers GenerationMixin._beam_search, 5.x
// vectorised form, generation/utils.py:3077-3460) as ONE launch: log-softmax of every beam's logits, the top 2*num_beams
// continuations of every question over (num_beams x vocab), and the bookkeeping on the running / finished beam sets (the torch
// restatement of which was ~60 small ATen launches per step: licv/generation.py round 3).
//
// One workgroup per question.  Phases: (1) per beam: max and sum of exp over the vocabulary; (2) every lane keeps the best KEEP of
// the candidates it visits, lp = ((x - max) - log(sum)) + running_score in fp32, the order torch evaluates it in; (3) KEEP rounds of
// a block-wide arg-max over the lanes' list heads; (4) lane 0 does the search bookkeeping of its question (a few dozen scalar
// operations) and leaves a copy plan in LDS; (5) all lanes copy the token rows of the new running / finished sets; the last
// workgroup to finish combines the per-question "keep going" flags.  Ties are broken towards the LOWER flat candidate index
// (beam * vocab + token) everywhere - a stable descending sort.
#include "common.h"
#include <limits.h>

#define BEAM_THREADS 256
#define BEAM_MAX_NB 8

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

template <int KEEP>
__global__ __launch_bounds__(BEAM_THREADS)
void beam_step_k(licv_beam_step_args a) {
    constexpr int NB = KEEP / 2;
    __shared__ float red[BEAM_THREADS / 64];
    __shared__ float s_mx[BEAM_MAX_NB], s_lse[BEAM_MAX_NB];
    __shared__ float s_wv[BEAM_THREADS / 64]; __shared__ int s_wi[BEAM_THREADS / 64]; __shared__ int s_wt[BEAM_THREADS / 64];
    __shared__ float top_lp[KEEP]; __shared__ int top_ix[KEEP];
    __shared__ int plan_run_src[NB], plan_run_tok[NB];            // new running row r <- running_in[src] with [cur] = tok
    __shared__ int plan_fin_src[NB], plan_fin_tok[NB];            // new finished row f <- finished_in[src] (tok < 0) or running_in[src] with [cur] = tok
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t V = a.V;
    auto ldx = [&](int beam, int64_t i) -> float {
        const int64_t row = (int64_t)b * a.q_stride_rows + (int64_t)beam * a.beam_stride_rows;
        return a.logits_dtype == LICV_F32 ? reinterpret_cast<const float*>(a.logits)[row * a.ld + i]
                                          : bf2f(reinterpret_cast<const bf16_t*>(a.logits)[row * a.ld + i]);
    };
    // ---- (1) per-beam log-softmax statistics
    for (int beam = 0; beam < NB; ++beam) {
        float mx = -INFINITY;
        for (int64_t i = tid; i < V; i += BEAM_THREADS) mx = fmaxf(mx, ldx(beam, i));
        mx = beam_block_reduce(mx, true, red);
        float z = 0.f;
        for (int64_t i = tid; i < V; i += BEAM_THREADS) z += expf(ldx(beam, i) - mx);
        z = beam_block_reduce(z, false, red);
        if (tid == 0) { s_mx[beam] = mx; s_lse[beam] = logf(z); }
    }
    __syncthreads();
    // ---- (2) lane-local best KEEP, visited in increasing flat index (so `>` keeps the lower index on ties)
    float lv[KEEP]; int li[KEEP];
#pragma unroll
    for (int j = 0; j < KEEP; ++j) { lv[j] = -INFINITY; li[j] = INT_MAX; }
    for (int beam = 0; beam < NB; ++beam) {
        const float mx = s_mx[beam], lse = s_lse[beam], rs = a.run_scores_in[(int64_t)b * NB + beam];
        for (int64_t i = tid; i < V; i += BEAM_THREADS) {
            float v = ((ldx(beam, i) - mx) - lse);
            if (a.suppress_eos && i == a.eos) v = -INFINITY;
            v += rs;
            if (v > lv[KEEP - 1]) {
                lv[KEEP - 1] = v; li[KEEP - 1] = (int)(beam * V + i);
#pragma unroll
                for (int j = KEEP - 1; j > 0; --j) {
                    if (lv[j] > lv[j - 1]) { const float tv = lv[j]; lv[j] = lv[j - 1]; lv[j - 1] = tv; const int ti = li[j]; li[j] = li[j - 1]; li[j - 1] = ti; }
                }
            }
        }
    }
    // ---- (3) KEEP rounds of a block arg-max over the list heads
    int head = 0;
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = 0; r < KEEP; ++r) {
        float hv = -INFINITY; int hi = INT_MAX;
#pragma unroll
        for (int j = 0; j < KEEP; ++j) if (j == head) { hv = lv[j]; hi = li[j]; }
        float bv = hv; int bi = hi, bt = tid;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64), ot = __shfl_xor(bt, o, 64);
            if (cand_better(ov, oi, bv, bi)) { bv = ov; bi = oi; bt = ot; }
        }
        __syncthreads();
        if (lane == 0) { s_wv[wave] = bv; s_wi[wave] = bi; s_wt[wave] = bt; }
        __syncthreads();
        float gv = s_wv[0]; int gi = s_wi[0], gt = s_wt[0];
        for (int w = 1; w < BEAM_THREADS / 64; ++w) if (cand_better(s_wv[w], s_wi[w], gv, gi)) { gv = s_wv[w]; gi = s_wi[w]; gt = s_wt[w]; }
        if (tid == gt) ++head;
        if (tid == 0) { top_lp[r] = gv; top_ix[r] = gi; }
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

extern "C" int licv_beam_step(const licv_beam_step_args* x, void* stream) {
    LICV_CHECK_ARG(x && x->logits && x->running_in && x->finished_in && x->run_scores_in && x->fin_scores_in && x->is_fin_in && x->improve_in &&
                   x->gen_len_in && x->running_out && x->finished_out && x->run_scores_out && x->fin_scores_out && x->is_fin_out &&
                   x->improve_out && x->gen_len_out && x->beam_src_flat && x->next_tokens && x->flags && x->sync, "beam_step: null pointer");
    LICV_CHECK_ARG(x->logits_dtype == LICV_BF16 || x->logits_dtype == LICV_F32, "beam_step: bad logits dtype");
    LICV_CHECK_ARG(x->nb >= 1 && x->nb <= BEAM_MAX_NB, "beam_step: num_beams %lld outside 1..%d", (long long)x->nb, BEAM_MAX_NB);
    LICV_CHECK_ARG(x->B >= 1 && x->V >= 2 * x->nb && x->ld >= x->V && x->nb * x->V < (1ll << 31), "beam_step: bad batch / vocabulary size");
    LICV_CHECK_ARG(x->cur >= x->P && x->cur < x->max_len && x->P >= 1, "beam_step: position %lld outside [P, max_len)", (long long)x->cur);
    LICV_CHECK_ARG(x->running_in != x->running_out && x->finished_in != x->finished_out, "beam_step: the token rows need separate in / out buffers");
    LICV_CHECK_ARG((x->kv_rows_in == nullptr) == (x->kv_rows_out == nullptr) && (!x->kv_rows_in || (x->kv_rows_in != x->kv_rows_out && x->kv_ld > x->cur)),
                   "beam_step: the KV row table needs separate in / out buffers with rows longer than the current position");
    const dim3 grid((unsigned)x->B), block(BEAM_THREADS);
    hipStream_t st = (hipStream_t)stream;
    switch (x->nb) {
        case 1: beam_step_k<2><<<grid, block, 0, st>>>(*x); break;
        case 2: beam_step_k<4><<<grid, block, 0, st>>>(*x); break;
        case 3: beam_step_k<6><<<grid, block, 0, st>>>(*x); break;
        case 4: beam_step_k<8><<<grid, block, 0, st>>>(*x); break;
        case 5: beam_step_k<10><<<grid, block, 0, st>>>(*x); break;
        case 6: beam_step_k<12><<<grid, block, 0, st>>>(*x); break;
        case 7: beam_step_k<14><<<grid, block, 0, st>>>(*x); break;
        default: beam_step_k<16><<<grid, block, 0, st>>>(*x); break;
    }
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}
