// Native layer runner for the Idefics language stack (hooked forward given the image states): the same kernels, in the same
// order and with the same dispatch decisions as licv/idefics_engine.py's Python loop, issued from ONE C-ABI call.
//
// Why: a decode step of hooked generate (ref:inference.py:300-321: 3 beams x B rows, one token each) is ~400 kernel launches
// that stream 18 GB of weights (a ~4 ms floor at HBM rate) but cost ~25 us of interpreter + ctypes + allocator work EACH from
// Python: 12 ms per step, launch-bound.  From C++ a launch costs ~3-4 us, so the step becomes GPU-bound.  The same holds for
// the 32-token student / prefill shapes.  Results are bit-identical to the Python loop by construction (tests/test_runner_gpu.py).
//
// The caller owns every buffer (weights, inputs, scratch, KV caches, output); nothing is allocated or synchronised here.
#include "common.h"
#include <vector>

#define RUN(call) do { int rc__ = (call); if (rc__ != LICV_OK) return rc__; } while (0)

namespace {
// ------------------------------------------------------------------------------------------------
// Weight prefetch beside the decode chain (licv_runner_option 2).  A decode step streams 16 GB of weights through ~160 projections of
// 20 - 50 us; each pays ~10 us of fill, staging and tail during which HBM idles (~40 % of the step).  The weights are constants, so
// a SIDE stream may read the NEXT projection's matrix (default cache policy: it lands in the 256 MB Infinity Cache) while the
// current one runs; the projection then finds its operand on-die.  Throttle: the prefetch of projection k + 1 is queued behind the
// END of projection k - 1, and capped in bytes, so at most ~two matrices are in flight through the cache.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_pf;
__global__ __launch_bounds__(256)
void weight_prefetch_k(const u32x4_pf* __restrict__ p, int64_t n16, unsigned* sink) {
    u32x4_pf acc = {0u, 0u, 0u, 0u};
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * 8) {
        u32x4_pf v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int64_t j = i + u * stride; v[u] = p[j < n16 ? j : i]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    if (sink && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u) *sink = 1u;      // keeps the loads alive; sink is NULL at run time
}
struct Prefetch {
    int64_t cap_bytes = 0;                                // 0 = off
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> ev;
    std::vector<std::pair<const void*, int64_t>> plan;    // the call's projections in launch order: (weights, bytes)
    int k = 0;
    bool active = false;
};
Prefetch g_pf;
void pf_before(hipStream_t main) {
    Prefetch& f = g_pf;
    if (!f.active) return;
    const int k = f.k;
    if (k + 1 < (int)f.plan.size()) {
        if (k >= 1) (void)hipStreamWaitEvent(f.side, f.ev[k - 1], 0);
        const int64_t bytes = f.plan[k + 1].second < f.cap_bytes ? f.plan[k + 1].second : f.cap_bytes;
        weight_prefetch_k<<<64, 256, 0, f.side>>>((const u32x4_pf*)f.plan[k + 1].first, bytes / 16, nullptr);
    }
    (void)main;
}
void pf_after(hipStream_t main) {
    Prefetch& f = g_pf;
    if (!f.active) return;
    if (f.k < (int)f.ev.size()) (void)hipEventRecord(f.ev[f.k], main);
    ++f.k;
}

struct Ctx {
    const licv_idefics_text_weights* w;
    const licv_idefics_text_call* c;
    void* stream;
    int64_t M, H;
    void* h; int h_dt;                                    // residual stream: h16 (bf16) until the first hook, h32 (fp32) after
};

int linear(const Ctx& x, const void* A, int64_t M, const void* W, int64_t N, int64_t K, void* C, int64_t ldc, int out_dt,
           const void* residual = nullptr, int res_dt = 0, const float* row_gate = nullptr, const float* scale = nullptr, int swiglu = 0) {
    licv_gemm_epilogue ep;
    ep.bias_bf16 = nullptr; ep.row_gate = row_gate; ep.residual = residual; ep.residual_dtype = res_dt;
    ep.ld_res = swiglu ? N / 2 : N; ep.act = 0; ep.swiglu = swiglu; ep.use_scale = scale ? 1 : 0; ep.scale = scale ? *scale : 0.f;
    ep.out_dtype = out_dt;
    int splits = 1; int64_t ws = 0;
    RUN(licv_gemm_splitk_plan(M, N, K, &splits, &ws));    // the same decision licv.ops.linear takes
    pf_before((hipStream_t)x.stream);
    int rc;
    if (splits > 1) {
        if (ws > x.c->workspace_bytes) return licv_set_error(LICV_E_BADARG, "idefics_text_forward: workspace %lld B < %lld B needed by a %lld x %lld x %lld split-K GEMM",
                                                             (long long)x.c->workspace_bytes, (long long)ws, (long long)M, (long long)N, (long long)K);
        rc = licv_gemm_bf16_splitk(A, K, W, K, C, ldc, M, N, K, &ep, splits, x.c->workspace, x.c->workspace_bytes, x.stream);
    } else rc = licv_gemm_bf16(A, K, W, K, C, ldc, M, N, K, &ep, x.stream);
    pf_after((hipStream_t)x.stream);
    return rc;
}

// A projection whose bf16 output (a branch, or the fused Q|K|V rows) is consumed by a row kernel.  Where the plan splits K and the
// option allows it, only the producer half runs and the slices stay in the workspace for that row kernel to sum (`src.ws` set);
// otherwise the GEMM writes C as usual.  The workspace is free again once the consumer has run — before the next GEMM on the stream.
struct Slices { const float* ws = nullptr; int splits = 0; int64_t slice = 0, stride = 0; };
int g_sum_in_rows = 1;
int linear_to_rows(const Ctx& x, const void* A, int64_t M, const void* W, int64_t N, int64_t K, void* C, Slices* src) {
    *src = Slices{};
    int splits = 1; int64_t ws = 0;
    RUN(licv_gemm_splitk_plan(M, N, K, &splits, &ws));
    if (splits > 1 && g_sum_in_rows) {
        if (ws > x.c->workspace_bytes) return licv_set_error(LICV_E_BADARG, "idefics_text_forward: workspace %lld B < %lld B needed by a %lld x %lld x %lld split-K GEMM",
                                                             (long long)x.c->workspace_bytes, (long long)ws, (long long)M, (long long)N, (long long)K);
        pf_before((hipStream_t)x.stream);
        RUN(licv_gemm_bf16_splitk_produce(A, K, W, K, M, N, K, splits, x.c->workspace, x.c->workspace_bytes, &src->slice, &src->stride, x.stream));
        pf_after((hipStream_t)x.stream);
        src->ws = (const float*)x.c->workspace; src->splits = splits;
        return LICV_OK;
    }
    return linear(x, A, M, W, N, K, C, N, LICV_BF16);
}
}  // namespace

static int g_fold_residual = 1;
// option 0: fold the decoder layers' residual adds into the row kernels that follow them (default 1; 0 for A/B timing)
// option 1: at M < 512, split-K projections leave their slices for the row kernel behind them to sum (default 1; 0 for A/B timing)
// option 2: decode steps (M <= 32): MiB of the NEXT projection's weights a side stream reads ahead into the Infinity Cache (0 = off)
extern "C" int licv_runner_option(int option, int value) {
    if (option == 0) { g_fold_residual = value; return LICV_OK; }
    if (option == 1) { g_sum_in_rows = value; return LICV_OK; }
    if (option == 2) { g_pf.cap_bytes = (int64_t)(value < 0 ? 0 : value) << 20; return LICV_OK; }      // MiB of each next projection's weights to prefetch; 0 = off
    return licv_set_error(LICV_E_BADARG, "runner_option: unknown option %d", option);
}

extern "C" int licv_idefics_text_forward(const licv_idefics_text_weights* w, const licv_idefics_text_call* c, void* stream) {
    LICV_CHECK_ARG(w && c, "idefics_text_forward: null argument");
    LICV_CHECK_ARG(w->dec && w->embed && w->final_ln && w->lm_head && w->cos && w->sin, "idefics_text_forward: missing weights");
    LICV_CHECK_ARG(c->input_ids && c->key_valid && c->position_ids && c->image_states && c->img_mask && c->gate, "idefics_text_forward: missing inputs");
    LICV_CHECK_ARG(c->h16 && c->h32 && c->x && c->xn && c->q && c->qkv && c->o && c->act && c->logits, "idefics_text_forward: missing scratch / output buffers");
    LICV_CHECK_ARG(c->B > 0 && c->S > 0 && c->Sk >= c->S && c->Nk > 0 && c->n_img > 0, "idefics_text_forward: bad shape");
    LICV_CHECK_ARG(!c->kv_cache || (c->past + c->S == c->Sk && c->Sk <= c->cache_max_len), "idefics_text_forward: KV cache / mask lengths inconsistent");
    LICV_CHECK_ARG(c->kv_cache || c->Sk == c->S, "idefics_text_forward: without a KV cache the mask must span exactly the new tokens");
    LICV_CHECK_ARG(c->xkv_cached || c->xkv, "idefics_text_forward: no cross-attention K/V cache and no scratch to project into");
    const int64_t H = w->hidden, I = w->inter, nh = w->n_heads, hd = w->head_dim, E = w->img_dim;
    const int64_t B = c->B, S = c->S, M = B * S, Nk = c->Nk;
    const float att_scale = 1.0f / sqrtf((float)hd);
    hipStream_t st = (hipStream_t)stream;
    Ctx x{w, c, stream, M, H, c->h16, LICV_BF16};
    // the weight-prefetch plan of this call (decode steps only): the projections in the order the loop below launches them
    g_pf.active = false;
    if (g_pf.cap_bytes > 0 && M <= 32) {
        Prefetch& f = g_pf;
        f.plan.clear(); f.k = 0;
        for (int64_t l = 0; l < w->n_layers; ++l) {
            if (l % w->cross_interval == 0) {
                const licv_idefics_xattn_w& X = w->xat[l / w->cross_interval];
                f.plan.emplace_back(X.q_w, H * H * 2);
                if (!(c->xkv_cached && c->xkv_cached[l / w->cross_interval])) f.plan.emplace_back(X.kv_w, 2 * H * E * 2);
                f.plan.emplace_back(X.o_w, H * H * 2); f.plan.emplace_back(X.gu_w, 2 * I * H * 2); f.plan.emplace_back(X.down_w, H * I * 2);
            }
            const licv_idefics_dec_w& D = w->dec[l];
            f.plan.emplace_back(D.qkv_w, 3 * H * H * 2); f.plan.emplace_back(D.o_w, H * H * 2);
            f.plan.emplace_back(D.gu_w, 2 * I * H * 2); f.plan.emplace_back(D.down_w, H * I * 2);
        }
        f.plan.emplace_back(w->lm_head, w->vocab_total * H * 2);
        if (!f.side && hipStreamCreateWithFlags(&f.side, hipStreamNonBlocking) != hipSuccess) f.side = nullptr;
        while (f.side && f.ev.size() < f.plan.size()) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) break;
            f.ev.push_back(e);
        }
        f.active = f.side && f.ev.size() >= f.plan.size();
    }

    RUN(licv_embed_gather(c->input_ids, w->embed, w->embed_extra, c->h16, M, H, w->vocab, w->n_extra_vocab, stream));
    bool xn_valid = false;                                 // c->xn holds RMSNorm(h) for the next block (made by the fused hook kernel)
    bool pending = false; float pending_scale = 0.f;       // a cross layer's MLP branch waits in c->q (or as split-K slices) for the next norm to add it
    Slices pending_src;
    auto next_norm = [&](int64_t l) -> const void* {
        if (l + 1 >= w->n_layers) return w->final_ln;
        if ((l + 1) % w->cross_interval == 0) return w->xat[(l + 1) / w->cross_interval].in_ln;
        return w->dec[l + 1].in_ln;
    };
    for (int64_t l = 0; l < w->n_layers; ++l) {
        if (l % w->cross_interval == 0) {                  // gated cross-attention layer (hf:idefics/modeling_idefics.py:746-802)
            const int64_t j = l / w->cross_interval;
            const licv_idefics_xattn_w& X = w->xat[j];
            const void* xin = c->xn;
            if (!xn_valid) { RUN(licv_rmsnorm_fwd(x.h, x.h_dt, X.in_ln, c->x, M, H, 1, H, H, w->rms_eps, 0, stream)); xin = c->x; }
            xn_valid = false;
            RUN(linear(x, xin, M, X.q_w, H, H, c->q, H, LICV_BF16));
            const void* kv = c->xkv_cached ? c->xkv_cached[j] : nullptr;
            if (!kv) {
                void* dst = c->xkv_out ? c->xkv_out[j] : c->xkv;
                RUN(linear(x, c->image_states, B * Nk, X.kv_w, 2 * H, E, dst, 2 * H, LICV_BF16));
                if (X.kn_w) RUN(licv_rmsnorm_fwd(dst, LICV_BF16, X.kn_w, dst, B * Nk * nh, hd, nh, 2 * H, 2 * H, w->rms_eps, 0, stream));
                kv = dst;
            }
            if (X.qn_w) RUN(licv_rmsnorm_fwd(c->q, LICV_BF16, X.qn_w, c->q, M * nh, hd, nh, H, H, w->rms_eps, 0, stream));
            licv_attn_args a;
            a.q = c->q; a.q_bs = S * H; a.q_rs = H;
            a.k = kv; a.v = (const char*)kv + H * 2; a.kv_bs = Nk * 2 * H; a.kv_rs = 2 * H;
            a.o = c->o; a.B = B; a.Sq = S; a.Sk = Nk; a.n_heads = nh; a.n_kv_heads = nh; a.head_dim = hd;
            a.scale = att_scale; a.mask_mode = 3; a.key_valid = nullptr; a.img_mask = c->img_mask; a.n_img = c->n_img; a.img_len = w->img_len;
            RUN(licv_attn_fwd(&a, stream));
            if (g_fold_residual) {                         // both gated residual adds folded into the norms that follow (see below)
                Slices os;
                RUN(linear_to_rows(x, c->o, M, X.o_w, H, H, c->q, &os));
                if (os.ws) RUN(licv_add_rmsnorm_fwd_ws(x.h, x.h_dt, os.ws, os.splits, os.slice, os.stride, c->gate, 1, X.gate_attn, X.post_ln, c->x, M, H, w->rms_eps, 0, stream));
                else RUN(licv_add_rmsnorm_fwd(x.h, x.h_dt, c->q, c->gate, 1, X.gate_attn, X.post_ln, c->x, M, H, w->rms_eps, 0, stream));
                RUN(linear(x, c->x, M, X.gu_w, 2 * I, H, c->act, I, LICV_BF16, nullptr, 0, nullptr, nullptr, 1));
                RUN(linear_to_rows(x, c->act, M, X.down_w, H, I, c->q, &pending_src));
                pending_scale = X.gate_dense; pending = true;     // added by the decoder layer's input norm
            } else {
                RUN(linear(x, c->o, M, X.o_w, H, H, x.h, H, x.h_dt, x.h, x.h_dt, c->gate, &X.gate_attn));
                RUN(licv_rmsnorm_fwd(x.h, x.h_dt, X.post_ln, c->x, M, H, 1, H, H, w->rms_eps, 0, stream));
                RUN(linear(x, c->x, M, X.gu_w, 2 * I, H, c->act, I, LICV_BF16, nullptr, 0, nullptr, nullptr, 1));
                RUN(linear(x, c->act, M, X.down_w, H, I, x.h, H, x.h_dt, x.h, x.h_dt, nullptr, &X.gate_dense));
            }
        }
        const licv_idefics_dec_w& D = w->dec[l];           // decoder layer (hf:idefics/modeling_idefics.py:645-675), hooked on its output
        const void* xin = c->xn;
        if (pending) {                                     // the cross layer's MLP branch (in c->q): h += bf16(gate_dense * branch), then the input norm
            if (pending_src.ws) RUN(licv_add_rmsnorm_fwd_ws(x.h, x.h_dt, pending_src.ws, pending_src.splits, pending_src.slice, pending_src.stride, nullptr, 1, pending_scale,
                                                            D.in_ln, c->x, M, H, w->rms_eps, 0, stream));
            else RUN(licv_add_rmsnorm_fwd(x.h, x.h_dt, c->q, nullptr, 1, pending_scale, D.in_ln, c->x, M, H, w->rms_eps, 0, stream));
            xin = c->x; pending = false;
        } else if (!xn_valid) { RUN(licv_rmsnorm_fwd(x.h, x.h_dt, D.in_ln, c->x, M, H, 1, H, H, w->rms_eps, 0, stream)); xin = c->x; }
        xn_valid = false;
        Slices qs;                                         // decode steps: the QKV slices go straight into rotary + cache append
        if (c->kv_cache && M < 512) RUN(linear_to_rows(x, xin, M, D.qkv_w, 3 * H, H, c->qkv, &qs));
        else RUN(linear(x, xin, M, D.qkv_w, 3 * H, H, c->qkv, 3 * H, LICV_BF16));
        if (!c->kv_cache) RUN(licv_rotary_fwd(c->qkv, w->cos, w->sin, c->position_ids, M, nh, hd, 3 * H, H, 2, w->rope_len, stream));
        licv_attn_args a;
        a.q = c->qkv; a.q_bs = S * 3 * H; a.q_rs = 3 * H;
        a.o = c->o; a.B = B; a.Sq = S; a.n_heads = nh; a.n_kv_heads = nh; a.head_dim = hd; a.scale = att_scale; a.mask_mode = 1;
        a.key_valid = c->key_valid; a.img_mask = nullptr; a.n_img = 0; a.img_len = 0;
        if (c->kv_cache && S == 1) {
            // a decode step (one new token per row): rotary, the append to the row's cache and the attention over its history in ONE launch
            // (csrc/decode.hip); with c->kv_rows the history is read through the beam search's row table - the cache itself never moves
            licv_decode_attn_args d;
            d.qkv_ws = qs.ws; d.splits = qs.splits; d.slice_elems = qs.slice; d.row_stride = qs.stride;
            d.qkv_bf16 = qs.ws ? nullptr : c->qkv; d.ldq = 3 * H;
            d.cos = w->cos; d.sin = w->sin; d.position_ids = c->position_ids; d.n_pos = w->rope_len;
            d.cache = c->kv_cache[l]; d.max_len = c->cache_max_len; d.past = c->past;
            d.kv_rows = c->kv_rows; d.ld_kv_rows = c->ld_kv_rows; d.key_valid = c->key_valid;
            d.out = c->o; d.M = M; d.n_heads = nh; d.n_kv_heads = nh; d.head_dim = hd; d.scale = att_scale;
            RUN(licv_decode_attn(&d, stream));
        } else {
        if (!c->kv_cache) {
            a.k = (const char*)c->qkv + H * 2; a.v = (const char*)c->qkv + 2 * H * 2; a.kv_bs = S * 3 * H; a.kv_rs = 3 * H; a.Sk = S;
        } else {
            void* cache = c->kv_cache[l];
            // rotary (Q in place, K on its way into the cache) and the append in one launch
            if (qs.ws) RUN(licv_rotary_kv_append_ws(qs.ws, qs.splits, qs.slice, qs.stride, c->qkv, w->cos, w->sin, c->position_ids, B, S, nh, hd, w->rope_len,
                                                    cache, c->cache_max_len, c->past, stream));
            else RUN(licv_rotary_kv_append(c->qkv, w->cos, w->sin, c->position_ids, B, S, nh, hd, w->rope_len, cache, c->cache_max_len, c->past, stream));
            a.k = cache; a.v = (const char*)cache + H * 2; a.kv_bs = c->cache_max_len * 2 * H; a.kv_rs = 2 * H; a.Sk = c->Sk;
        }
        RUN(licv_attn_fwd(&a, stream));
        }
        const int slot = (c->hook_slot && c->icv) ? c->hook_slot[l] : -1;
        // Large batches (the 256-tile GEMMs): the two residual adds of the layer leave the GEMM epilogues — a read-modify-write of the
        // fp32 stream costs the o / down projections 16-28 % — and are folded into the row kernels that follow them (same sums, same
        // rounding: bit-identical): the projections write their bf16 branch (c->q is free here) through the register-direct epilogue.
        const bool fold = M >= 512 && g_fold_residual;
        if (fold) {
            RUN(linear(x, c->o, M, D.o_w, H, H, c->q, H, LICV_BF16));
            RUN(licv_add_rmsnorm_fwd(x.h, x.h_dt, c->q, nullptr, 0, 0.f, D.post_ln, c->x, M, H, w->rms_eps, 0, stream));
        } else {
            Slices os;
            RUN(linear_to_rows(x, c->o, M, D.o_w, H, H, c->q, &os));
            if (os.ws) RUN(licv_add_rmsnorm_fwd_ws(x.h, x.h_dt, os.ws, os.splits, os.slice, os.stride, nullptr, 0, 0.f, D.post_ln, c->x, M, H, w->rms_eps, 0, stream));
            else RUN(licv_add_rmsnorm_fwd(x.h, x.h_dt, c->q, nullptr, 0, 0.f, D.post_ln, c->x, M, H, w->rms_eps, 0, stream));
        }
        RUN(linear(x, c->x, M, D.gu_w, 2 * I, H, c->act, I, LICV_BF16, nullptr, 0, nullptr, nullptr, 1));
        if (fold && slot >= 0) {
            RUN(linear(x, c->act, M, D.down_w, H, I, c->q, H, LICV_BF16));
            RUN(licv_inject_renorm_pre_fwd(x.h, x.h_dt, c->q, c->icv + (int64_t)slot * H, c->alpha ? c->alpha + slot : nullptr, (float*)c->h32, M, H,
                                           next_norm(l), c->xn, w->rms_eps, stream));
            x.h = c->h32; x.h_dt = LICV_F32;
            xn_valid = true;
            continue;
        }
        if (slot >= 0) {                                   // the hook (ref:icv_src/icv_model/icv_intervention.py:61-86) fused with the next RMSNorm
            Slices ds;
            RUN(linear_to_rows(x, c->act, M, D.down_w, H, I, c->q, &ds));
            if (ds.ws) RUN(licv_inject_renorm_pre_fwd_ws(x.h, x.h_dt, ds.ws, ds.splits, ds.slice, ds.stride, c->icv + (int64_t)slot * H,
                                                         c->alpha ? c->alpha + slot : nullptr, (float*)c->h32, M, H, next_norm(l), c->xn, w->rms_eps, stream));
            else RUN(licv_inject_renorm_pre_fwd(x.h, x.h_dt, c->q, c->icv + (int64_t)slot * H, c->alpha ? c->alpha + slot : nullptr, (float*)c->h32, M, H,
                                                next_norm(l), c->xn, w->rms_eps, stream));
            x.h = c->h32; x.h_dt = LICV_F32;               // the fp32 ICV promotes the stream
            xn_valid = true;
        } else {
            RUN(linear(x, c->act, M, D.down_w, H, I, x.h, H, x.h_dt, x.h, x.h_dt));
        }
    }
    const void* xf = c->xn;
    if (!xn_valid) { RUN(licv_rmsnorm_fwd(x.h, x.h_dt, w->final_ln, c->x, M, H, 1, H, H, w->rms_eps, 0, stream)); xf = c->x; }
    int64_t rows = M;
    if (c->logits_rows && c->n_rows > 0) {
        LICV_CHECK_ARG(c->xsel, "idefics_text_forward: logits_rows given without the xsel scratch");
        RUN(licv_embed_gather(c->logits_rows, xf, nullptr, c->xsel, c->n_rows, H, M, 0, stream));
        xf = c->xsel; rows = c->n_rows;
    }
    const int rc_head = linear(x, xf, rows, w->lm_head, w->vocab_total, H, c->logits, c->ld_logits, LICV_BF16);
    g_pf.active = false;
    return rc_head;
}
