/*
 * licv_hip.h — C-ABI of liblicv_hip.so: the MI355X (gfx950) kernels of the L-ICV hot path.
 *
 * Boundary: the reference is pure Python; its FFI for this path would be a
 * ctypes binding.  Each entry point below names the reference/HF interface whose arithmetic it
 * replaces (ref: = ForJadeForest/LICV-VQA, hf: = transformers/models).  Conventions:
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless noted;
 *   - the caller owns all buffers; the library allocates nothing and never synchronises;
 *   - every launch goes on the `stream` argument (a hipStream_t passed as void*);
 *   - return 0 on success, a negative LICV_E_* otherwise; licv_last_error() gives the text
 *     (thread-local, host pointer);
 *   - dtype arguments use the LICV_* enums; leading dimensions (`ld*`) are in ELEMENTS.
 */
#ifndef LICV_HIP_H
#define LICV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an entry point is added, removed or changes its arguments; licv/_lib.py holds the same constant and refuses a
 * library that answers anything else.  1 = round 1; 2 = rounds 2-3 (fp8, split-K slices, runner, front-end, backward, image input);
 * 3 = round 4 (lab library split off; weight-streaming GEMM; beam scoring; decode-step fusion). */
#define LICV_ABI_VERSION 6

enum { LICV_BF16 = 0, LICV_F32 = 1 };
enum { LICV_OK = 0, LICV_E_BADARG = -1, LICV_E_UNSUPPORTED = -2, LICV_E_HIP = -3 };

int         licv_version(void);
const char* licv_last_error(void);

/* The one collective of the path (replaces the gradient all-reduce Lightning DDP / DeepSpeed ZeRO-2 run for the
 * reference, ref:config/trainer/ddp.yaml:5, ref:config/trainer/zero2.yaml:5): in-place sum (average != 0: mean) of n fp32 values - the
 * gradients of icv and alpha, 131 k floats - over the ranks of `nccl_comm`, an ncclComm_t of the RCCL ALREADY loaded and initialised in
 * the calling process, enqueued on `stream`.  The library opens no communicator and does not link RCCL: ncclAllReduce is looked up
 * (dlsym) at the first call; LICV_E_UNSUPPORTED when the process has no RCCL.  (The Python host keeps torch.distributed.all_reduce,
 * which owns its communicator: licv/trainer.py.) */
int licv_allreduce_small(void* nccl_comm, float* data, int64_t n, int average, void* stream);

/* ---- the hook: ref:icv_src/icv_model/icv_intervention.py:61-86 (intervention_function) ----
 * out[r,:] = (h[r,:]+v) / ||h[r,:]+v|| * ||h[r,:]||, fp32 maths, fp32 output (torch promotion of a
 * bf16/fp32 stream with the fp32 ICV).  v = alpha*icv[layer] when `alpha` != NULL (folds
 * ref:icv_src/icv_module.py:89-92), else v = icv_row as given.  `alpha` is a device pointer to one float.
 * If norm_w != NULL the RMSNorm that consumes the edited stream next (hf:idefics/modeling_idefics.py:342-350)
 * is fused: xn_out (bf16) = w * bf16(out * rsqrt(mean(out^2)+eps)).  out may alias h when h is fp32. */
int licv_inject_renorm_fwd(const void* h, int h_dtype, const float* icv_row, const float* alpha,
                           float* out, int64_t rows, int64_t hidden,
                           const void* norm_w, void* xn_out, float norm_eps, void* stream);

/* Idefics2 hook site = a residual BRANCH (`layers.N.mlp`, ref:config/lmm/idefics2-8B-base.yaml:8): the edit is applied to the
 * branch output and the stream becomes  out = residual + edit(branch)  (fp32).  norm_flavour selects the RMSNorm fused
 * on `out` (0 Idefics, 1 Mistral: single rounding on an fp32 stream). */
int licv_inject_renorm_add_fwd(const void* branch, int branch_dtype, const float* icv_row, const float* alpha,
                               const void* residual, int residual_dtype, float* out, int64_t rows, int64_t hidden,
                               const void* norm_w, void* xn_out, float norm_eps, int norm_flavour, void* stream);

/* The layer-output hook with the layer's LAST RESIDUAL ADD folded in (hf:idefics/modeling_idefics.py:760-763, `hidden_states =
 * residual + hidden_states` at the end of IdeficsDecoderLayer.forward, then ref:icv_src/icv_model/icv_intervention.py:62-84 on that
 * output): the tensor edited is h + branch in the stream's dtype (a bf16 stream rounds the sum to bf16, an fp32 stream adds in
 * fp32 — bit for bit what the down projection's residual epilogue wrote), so the down projection can leave its bf16 branch
 * through the register-direct GEMM epilogue.  Otherwise as licv_inject_renorm_fwd. */
int licv_inject_renorm_pre_fwd(const void* h, int h_dtype, const void* branch_bf16, const float* icv_row, const float* alpha,
                               float* out, int64_t rows, int64_t hidden,
                               const void* norm_w, void* xn_out, float norm_eps, void* stream);

/* backward of the hook for training (gradients reach icv and alpha only through here;
 * ref:icv_src/icv_module.py:97-98 runs the student pass with grad).  grad_h may be NULL.
 * grad_v_partial: (n_partials, hidden) fp32 workspace, fully overwritten; the caller sums over
 * dim 0.  n_partials = licv_inject_bwd_partials(rows). */
int64_t licv_inject_bwd_partials(int64_t rows);
int licv_inject_renorm_bwd(const void* h, int h_dtype, const float* icv_row, const float* alpha,
                           const float* grad_out, float* grad_h, float* grad_v_partial,
                           int64_t rows, int64_t hidden, void* stream);

/* ---- norms ----
 * Row r of the (rows x dim) problem lives at  x + (r / inner)*ld_outer + (r % inner)*dim  (same for out
 * with its own ld); inner=1, ld=dim is the dense case, inner=n_heads covers per-head q/k norms.
 * RMSNorm flavours: 0 = Idefics (hf:idefics/modeling_idefics.py:342-350: cast to bf16 THEN * weight),
 *                   1 = Mistral (hf:mistral/modeling_mistral.py:182-199: weight * cast-back).          */
int licv_rmsnorm_fwd(const void* x, int x_dtype, const void* w_bf16, void* out_bf16,
                     int64_t rows, int64_t dim, int64_t inner, int64_t ld_x, int64_t ld_out,
                     float eps, int flavour, void* stream);
/* h += branch in place (the stream's dtype: bf16 rounds the sum; hf:idefics/modeling_idefics.py:741-743, the residual add after
 * self-attention), then out_bf16 = RMSNorm(h) (the post-attention norm, :745): the o-projection's residual epilogue folded into the
 * norm that follows it.  Dense rows (ld = dim).  row_gate (rows fp32, may be NULL) and use_scale / scale reproduce the gated
 * cross-attention layer's epilogue (:789-791, :799-800): branch row -> 0 where the gate is 0, then bf16(scale * branch). */
int licv_add_rmsnorm_fwd(void* h, int h_dtype, const void* branch_bf16, const float* row_gate, int use_scale, float scale,
                         const void* w_bf16, void* out_bf16, int64_t rows, int64_t dim, float eps, int flavour, void* stream);
/* nn.LayerNorm on a bf16 tensor (hf:idefics/vision.py:286-299, perceiver.py:140-141,155-156): fp32
 * statistics, one rounding.  Input rows addressed as for RMSNorm; output row r goes to
 *   out + (r / inner)*ld_out + (r % inner)*dim + (out_group > 0 ? (r / out_group)*out_group_extra : 0)
 * which lets LN(context) and LN(latents) land directly inside the perceiver's concatenated K/V input. */
int licv_layernorm_fwd(const void* x_bf16, const void* w_bf16, const void* b_bf16, void* out_bf16,
                       int64_t rows, int64_t dim, int64_t inner, int64_t ld_x, int64_t ld_out,
                       int64_t out_group, int64_t out_group_extra, float eps, void* stream);

/* ---- rotary: hf:idefics/modeling_idefics.py:396-428 (rotate_half form, gathered by position_ids) ----
 * In place on `n_tensors` (1 or 2: q and k) head-major slices of a (rows x ld) bf16 buffer:
 * tensor t, head h, row r at x + r*ld + t*tensor_stride + h*head_dim.  cos/sin: (n_pos, head_dim) bf16. */
int licv_rotary_fwd(void* x_bf16, const void* cos_bf16, const void* sin_bf16, const int64_t* position_ids,
                    int64_t rows, int64_t n_heads, int64_t head_dim, int64_t ld, int64_t tensor_stride,
                    int n_tensors, int64_t n_pos, void* stream);
/* rotary + KV-cache append of one fused QKV projection in one launch (hooked generate, ref:inference.py:300-321 -> HF generate's
 * past_key_values): qkv is (batch * S, 3H) bf16; Q heads are rotated in place, the rotated K heads and V go to
 * cache[b, past + s, 0:H | H:2H] (cache: (batch, cache_max_len, 2H) bf16).  Same arithmetic as licv_rotary_fwd. */
int licv_rotary_kv_append(void* qkv_bf16, const void* cos_bf16, const void* sin_bf16, const int64_t* position_ids, int64_t batch, int64_t S,
                          int64_t n_heads, int64_t head_dim, int64_t n_pos, void* cache_bf16, int64_t cache_max_len, int64_t past, void* stream);

/* ---- dense layers: nn.Linear on MFMA (every F.linear under hf:idefics/ and hf:idefics2/) ----
 * C[M,N] = epilogue( A[M,K] (bf16, lda) x W[N,K]^T (bf16, ldw) ), fp32 accumulation.
 * Epilogue, in this order (each step rounds to bf16 as the unfused torch ops would):
 *   y = bf16(acc + bias[n])                      bias_bf16 may be NULL
 *   y = bf16(act(y))                             act: 0 none, 1 GELU(erf), 2 GELU(tanh), 3 ReLU
 *   swiglu != 0: columns come in (gate,up) 16-wide interleaved pairs (weights packed by
 *                licv_pack_gate_up) and y = bf16(bf16(silu(g)) * u); the output has N/2 columns
 *   row_gate (fp32 per row) != NULL: y = 0 where row_gate[m] == 0     (cross_attention_gate)
 *   scale != 0 flag `use_scale`: y = bf16(scale * y)                   (tanh(alpha) gates)
 *   residual != NULL: out = residual[m,n] + y   in the residual dtype (bf16 or fp32 stream)
 *   store to C as out_dtype (bf16 or fp32).  C may alias residual.
 * Requirements: K % 8 == 0, lda/ldw % 8 == 0, ldc % 4 == 0, 16-byte aligned base pointers. */
typedef struct {
    const void* bias_bf16;
    const float* row_gate;
    const void* residual;
    int   residual_dtype;
    int64_t ld_res;
    int   act;
    int   swiglu;
    int   use_scale;
    float scale;
    int   out_dtype;
} licv_gemm_epilogue;

int licv_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                   int64_t M, int64_t N, int64_t K, const licv_gemm_epilogue* ep, void* stream);
/* Skinny GEMMs (M <= 256: student pass, decode steps): K is cut into `splits` ranges, each workgroup writes its fp32 partial tile
 * to its own slice of a caller-provided workspace (no atomics: bit-reproducible), a second kernel sums the slices in order and
 * applies the epilogue.  licv_gemm_splitk_plan returns splits <= 1 when the plain kernel should be used. */
int licv_gemm_splitk_plan(int64_t M, int64_t N, int64_t K, int* splits, int64_t* workspace_bytes);
/* workspace bytes licv_gemm_bf16_splitk wants for (M, N, K); 0 = the one-pass kernels serve the shape; < 0 = bad arguments.
 * (SURVEY.md §8b "licv_workspace_size": the library allocates no device memory — every buffer, this one included, is the caller's.) */
int64_t licv_workspace_size(int64_t M, int64_t N, int64_t K);
int licv_gemm_bf16_splitk(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                          int64_t M, int64_t N, int64_t K, const licv_gemm_epilogue* e, int splits,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* The producer half alone, for a plain epilogue whose consumer is a row kernel (decode steps and the 32-token student pass of
 * ref:inference.py:300-321 / ref:icv_src/icv_module.py:97-98, where a launch costs as much as the kernel): `splits` fp32 slices of
 * A W^T are left in the workspace — slice sp of row m at workspace + sp * slice_elems + m * row_stride (floats) — and the consumer
 * (licv_add_rmsnorm_fwd_ws, licv_inject_renorm_pre_fwd_ws, licv_rotary_kv_append_ws) sums them in slice order and rounds to bf16
 * itself: bit for bit the finalize kernel's output, one launch less per projection. */
int licv_gemm_bf16_splitk_produce(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t N, int64_t K, int splits,
                                  void* workspace, int64_t workspace_bytes, int64_t* slice_elems, int64_t* row_stride, void* stream);
/* licv_add_rmsnorm_fwd / licv_inject_renorm_pre_fwd / licv_rotary_kv_append with the bf16 operand (branch, branch, qkv) replaced by
 * the split-K slices it would have been finalized from.  Bit-identical to finalize + the plain entry point (same per-lane sums).
 * licv_rotary_kv_append_ws writes the rotated Q rows into qkv_bf16[:, 0:H] (the attention kernel's Q operand) and nothing else there. */
int licv_add_rmsnorm_fwd_ws(void* h, int h_dtype, const float* ws, int splits, int64_t slice_elems, int64_t row_stride,
                            const float* row_gate, int use_scale, float scale, const void* w_bf16, void* out_bf16,
                            int64_t rows, int64_t dim, float eps, int flavour, void* stream);
int licv_inject_renorm_pre_fwd_ws(const void* h, int h_dtype, const float* ws, int splits, int64_t slice_elems, int64_t row_stride,
                                  const float* icv_row, const float* alpha, float* out, int64_t rows, int64_t hidden,
                                  const void* norm_w, void* xn_out, float norm_eps, void* stream);
int licv_rotary_kv_append_ws(const float* ws, int splits, int64_t slice_elems, int64_t row_stride, void* qkv_bf16,
                             const void* cos_bf16, const void* sin_bf16, const int64_t* position_ids, int64_t batch, int64_t S,
                             int64_t n_heads, int64_t head_dim, int64_t n_pos, void* cache_bf16, int64_t cache_max_len, int64_t past, void* stream);

/* ---- fp8 path (BASELINE configs[4]: "fp8 weights on CDNA4 MFMA"; no reference counterpart — parity bar in DESIGN.md) ----
 * q[r,k] = e4m3(x[r,k] / scale[r]), scale[r] = amax(x[r,:]) / 448 (OCP e4m3fn, round to nearest even).  Weights are quantised
 * once with the same kernel (rows = output channels); activations per call. */
int licv_quantize_rows_fp8(const void* x, int x_dtype, void* q_fp8, float* scale, int64_t rows, int64_t dim,
                           int64_t ld_x, int64_t ld_q, void* stream);
/* The row kernels that feed an fp8 GEMM write the fp8 image of their rows themselves (exactly licv_quantize_rows_fp8 of the bf16
 * rows they produce: e4m3 bytes (rows, dim) + one fp32 scale per row) - `out_bf16` may be NULL when nothing else reads the rows.
 * Rows are contiguous (ld = dim).  rmsnorm: as licv_rmsnorm_fwd; add_rmsnorm: as licv_add_rmsnorm_fwd without gate / scale;
 * inject_renorm_add: as licv_inject_renorm_add_fwd (xn_bf16 may be NULL); layernorm: as licv_layernorm_fwd, dim % 8 == 0. */
int licv_rmsnorm_fwd_q8(const void* x, int x_dtype, const void* w_bf16, void* out_bf16, void* q_fp8, float* q_scale, int64_t rows, int64_t dim,
                        float eps, int flavour, void* stream);
int licv_add_rmsnorm_fwd_q8(void* h, int h_dtype, const void* branch_bf16, const void* w_bf16, void* out_bf16, void* q_fp8, float* q_scale,
                            int64_t rows, int64_t dim, float eps, int flavour, void* stream);
int licv_inject_renorm_add_fwd_q8(const void* branch, int branch_dtype, const float* icv_row, const float* alpha,
                                  const void* residual, int residual_dtype, float* out_f32, int64_t rows, int64_t hidden,
                                  const void* norm_w_bf16, void* xn_bf16, void* q_fp8, float* q_scale, float norm_eps, int norm_flavour, void* stream);
int licv_layernorm_fwd_q8(const void* x_bf16, const void* w_bf16, const void* b_bf16, void* out_bf16, void* q_fp8, float* q_scale,
                          int64_t rows, int64_t dim, float eps, void* stream);
/* C = epilogue( (Aq . Wq^T) * a_scale[m] * w_scale[n] ), fp32 accumulate on v_mfma_f32_16x16x32_fp8_fp8; same epilogue
 * struct as licv_gemm_bf16.  Aq (M, K) and Wq (N, K) e4m3 bytes, K >= 256, K % 64 == 0, leading dims in bytes. */
int licv_gemm_fp8(const void* Aq, int64_t lda, const float* a_scale, const void* Wq, int64_t ldw, const float* w_scale,
                  void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const licv_gemm_epilogue* e, void* stream);

/* Kernel selection override for tests and A/B timing (process-wide).  0 = automatic: for M >= 512, N >= 256, K % 64 == 0 the
 * persistent "flow" kernel (register-direct asynchronous epilogue) where the epilogue allows it — plain / bias / activation /
 * SwiGLU / bias + bf16 residual, bf16 out, N % 64 == 0 — else the lean ping-pong kernel with the staged epilogue; the general
 * 128 x 128 kernel for everything smaller (M <= 32 with long K: licv_gemm_splitk).  1 = always the 128 x 128 kernel; 2 the round-1
 * single-barrier 256 x 256 kernel; 6 the round-1 ping-pong kernel; 8 persistent ping-pong; 20 flow wherever eligible; 21 pair kernel
 * (two K stages per phase); 22 lean ping-pong everywhere (23-27: its ordering / diagnostic builds); 30-37 four-wave kernel and its
 * timing builds; 40-42 four-wave kernel on 64-deep K tiles; 50 two workgroups per CU on 128 x 256 tiles; 60 the four-wave flow64 kernel
 * wherever eligible; 70 the 128-tile mid kernel at any M; 71 the 256 x 128 tall kernel wherever it can run (129-256 rows).  All full-result variants are
 * bit-identical (tests/test_ops_gpu.py).  Values that name a kernel of the lab library (2-8, 10-13, 21, 30-37, 50: csrc/lab/, built
 * into liblicv_hip_lab.so for tests and tools only) make licv_gemm_bf16 return LICV_E_UNSUPPORTED unless that library is loaded. */
int licv_gemm_select(int which);
/* Called by liblicv_hip_lab.so when it is loaded: registers the experiments' launcher / knob / timestamp entry points (or NULLs). */
int licv_lab_register(void* launch, void* knob, void* timestamps);
/* Knobs: 0 per-XCD start stagger of the round-1 ping-pong kernel in percent (off); 1 tile-rows per XCD patch (0 = heuristic);
 * 2 = 0: never take the flow kernel in auto mode; 4 = 0: licv_gemm_splitk_plan always answers "one pass" (split-K off for every
 * caller, the native layer runner included: the batch-independence tests compare bit for bit); 5 forced split-K count of the 128-tile
 * route (0 = the plan's choice); 6 fewest 256-tiles for the 256-tile kernels; 7 in-launch reduction of the skinny kernel; 8 = 0: fp8 GEMMs off the
 * 128-deep MFMA; 9 timing-only ablation of the mid kernel (wrong results); 10 = 4: four K tiles in flight in the mid kernel; 11 non-temporal
 * weight loads (bit 0 mid, bit 1 skinny); 12 = 0: wide outputs at 129-256 rows stay on the mid kernel instead of the tall kernel.  A/B timing
 * and tests only: every setting gives the same results except knob 9. */
int licv_gemm_experiment(int knob, int value);
/* 1 if the flow kernel may be dispatched (its code objects use no scratch memory: its counted waits rely on that), else 0 */
int licv_gemm_flow_available(void);
/* timing instrumentation: when non-NULL, wave 0 of every workgroup of the default kernel stores 5 wall_clock64() stamps
 * (start, pipeline filled, main loop done, output image in LDS, end) at dev_buffer[8 * blockIdx.x ...] (int64) */
int licv_gemm_debug_timestamps(void* dev_buffer);
/* (2I x K) gate/up weights -> the 16-row interleaved layout the swiglu epilogue expects. */
int licv_pack_gate_up(const void* gate_bf16, const void* up_bf16, void* packed_bf16,
                      int64_t inter, int64_t K, void* stream);

/* ---- attention (hf eager_attention_forward, hf:idefics/modeling_idefics.py:450-470; perceiver
 * hf:idefics/perceiver.py:150-166) as a tiled online-softmax kernel ----
 * Q: (B, Sq, n_heads, hd) at q + b*q_bs + s*q_rs + h*hd;  K,V likewise with (kv_bs, kv_rs) and
 * n_kv_heads (GQA: head h reads kv head h / (n_heads/n_kv_heads)).  O: (B, Sq, n_heads*hd) dense bf16.
 * mask_mode: 0 none; 1 causal with q offset (Sk - Sq) AND key_valid[b,key] != 0 (int32, may be NULL);
 *            2 key_valid only (bidirectional key-padding mask);
 *            3 image mask: allowed iff img_mask[b, q, key / img_len] != 0 (int32 (B,Sq,n_img)).
 * Rows with no allowed key produce zeros. */
typedef struct {
    const void* q; int64_t q_bs, q_rs;
    const void* k; const void* v; int64_t kv_bs, kv_rs;
    void* o;
    int64_t B, Sq, Sk, n_heads, n_kv_heads, head_dim;
    float scale;
    int   mask_mode;
    const int32_t* key_valid;
    const int32_t* img_mask; int64_t n_img, img_len;
} licv_attn_args;

int licv_attn_fwd(const licv_attn_args* a, void* stream);
/* tests / A-B timing: bit 0 = always the tiled kernel (never the resident-K/V variant used for short unmasked keys);
 * bit 1 = the resident variant walks its items in blockIdx order instead of grouping consecutive heads on one XCD;
 * bit 2 = the resident variant leaves its LDS read schedule to the compiler; bit 3 = so does the tiled kernel at head dim 128 */
int licv_attn_select(int mode);

/* ---- small data-movement kernels ---- */
/* hf:idefics/modeling_idefics.py:230-267 IdeficsDecoupledEmbedding (ids >= vocab -> additional table) */
int licv_embed_gather(const int64_t* ids, const void* table_bf16, const void* extra_bf16, void* out_bf16,
                      int64_t n_tokens, int64_t dim, int64_t vocab, int64_t n_extra, void* stream);
/* Conv2d(k=s=patch, no padding) as im2col: pixels (n_img,3,H,W) bf16 -> (n_img*gh*gw, ld_out) with the
 * first 3*patch*patch columns filled (channel-major, as conv weight.flatten(1)) and the rest zeroed. */
int licv_im2col_patches(const void* pix_bf16, void* out_bf16, int64_t n_img, int64_t height, int64_t width,
                        int64_t patch, int64_t ld_out, void* stream);
/* hf:idefics/vision.py:152-166 + pre_layrnorm :369: x = cat(cls, patches) + pos; emb_out = x (optional),
 * ln_out = LayerNorm(x).  patches: (n_img*n_patch, dim) bf16. */
int licv_vit_embed_ln(const void* patches_bf16, const void* cls_bf16, const void* pos_bf16,
                      const void* ln_w, const void* ln_b, void* out_bf16,
                      int64_t n_img, int64_t n_patch, int64_t dim, float eps, void* stream);
/* out[r, :] = src[r % period, :]  (latents.repeat, hf:idefics/perceiver.py:96) */
int licv_tile_rows(const void* src_bf16, void* out_bf16, int64_t rows, int64_t dim, int64_t period, void* stream);
/* out[idx[i], :] = src[i, :] (hf:idefics2/modeling_idefics2.py:789-815 inputs_merger) */
int licv_scatter_rows(const void* src_bf16, const int64_t* idx, void* out_bf16, int64_t n, int64_t dim, void* stream);
/* silu(g)*u for an un-fused (M, 2I) [gate | up] buffer (kept for tests / odd shapes). */
int licv_swiglu(const void* gu_bf16, void* out_bf16, int64_t rows, int64_t inter, void* stream);

/* ---- backward of the student pass (ref:icv_src/icv_module.py:97-98: hooked forward WITH grad; the LMM is frozen, so
 * only d loss / d hidden-state is propagated; dense-layer input grads reuse licv_gemm_bf16 on transposed weights) ---- */
/* RMSNorm backward: dx (+)= rs*(g - xhat*mean(g*xhat)); rows addressed as forward.  g = bf16(dy*w) wherever the forward multiplied
 * the weight into bf16 rows (flavour 0 = Idefics, hf:idefics/modeling_idefics.py:342-350: always; flavour 1 = Mistral,
 * hf:mistral/modeling_mistral.py:182-199: only on a bf16 stream - on the fp32 stream behind a hook the product and its gradient are
 * fp32, g = dy*w unrounded) */
int licv_rmsnorm_bwd(const void* x, int x_dtype, const void* w_bf16, const void* dy, int dy_dtype, void* dx, int dx_dtype,
                     int64_t rows, int64_t dim, int64_t inner, int64_t ld_x, int64_t ld_dy, int64_t ld_dx, float eps,
                     int accumulate, int flavour, void* stream);
/* The same with dy still in the fp32 split-K slices of the dgrad GEMM that produced it (licv_gemm_bf16_splitk_produce): dy[r, i] =
 * bf16(sum over the slices, in slice order) - bit for bit what the finalize launch writes and licv_rmsnorm_bwd reads back; rows of
 * >= 1024 elements, dy / x / dx dense (leading dimension = dim). */
int licv_rmsnorm_bwd_ws(const void* x, int x_dtype, const void* w_bf16, const float* ws, int splits, int64_t slice_elems,
                        int64_t row_stride, void* dx, int dx_dtype, int64_t rows, int64_t dim, float eps, int accumulate,
                        int flavour, void* stream);
/* SwiGLU backward on the unfused (rows, 2I) [gate | up] buffer */
int licv_swiglu_bwd(const void* gu_bf16, const void* dact_bf16, void* dgu_bf16, int64_t rows, int64_t inter, void* stream);
/* grad entering a residual branch: out = bf16(bf16(dh)*scale), rows with row_gate == 0 zeroed (row_gate may be NULL) */
int licv_branch_grad(const float* dh, void* out_bf16, int64_t rows, int64_t dim, float scale, int use_scale,
                     const float* row_gate, void* stream);
/* attention backward for short sequences (Sq*Sk <= 16384), same argument struct as the forward; dk/dv may be NULL;
 * dK/dV are written per QUERY head (n_heads*head_dim columns): with GQA reduce them with licv_head_group_sum */
int licv_attn_bwd_small(const licv_attn_args* a, const void* dout_bf16, void* dq_bf16, int64_t dq_bs, int64_t dq_rs,
                        void* dk_bf16, void* dv_bf16, int64_t dkv_bs, int64_t dkv_rs, void* stream);
/* A/B timing and tests of the backward kernels (results are bit-identical either way).  option 0 = 0: licv_attn_bwd_small keeps the
 * head's Q / K / V / dO rows in global memory (default 1: staged in LDS where they fit beside P and dS, 1024 lanes per (batch, head);
 * 2: staged on 256 lanes, timing only); option 1 = 0: licv_rmsnorm_bwd
 * on one wave per row at every row length (default 1: four waves per row from 1024 elements on). */
int licv_backward_option(int option, int value);
/* cross-entropy rows (the "hard" loss, ref:icv_src/icv_module.py:94-95,111-117; HF ForCausalLMLoss upcasts to fp32):
 * loss_rows[i] = logsumexp(logits[rows[i], :]) - logits[rows[i], labels[i]]  (may be NULL);
 * grad[(grad_rows ? grad_rows[i] : i), :] (+)= grad_coef * (softmax - onehot), bf16 (may be NULL). */
int licv_ce_rows(const void* logits, int dtype, const int64_t* rows, const int64_t* labels, int64_t n_rows, int64_t vocab,
                 int64_t ld, float* loss_rows, float grad_coef, const float* grad_coef_dev, void* grad_bf16, int64_t ld_grad,
                 const int64_t* grad_rows, int accumulate, void* stream);
/* backward of repeat_kv (GQA): out[r, g*hd + d] = sum over the `rep` query heads of group g of src[r, (g*rep+j)*hd + d] */
int licv_head_group_sum(const void* src_bf16, void* out_bf16, int64_t rows, int64_t n_groups, int64_t rep, int64_t head_dim,
                        int64_t ld_src, int64_t ld_out, void* stream);
/* d loss / d student logits for the masked-KL rows: (n_rows, ld_grad >= vocab) bf16, scaled by upstream * T^2 / n_rows
 * (times *upstream_dev when that device pointer is given: the autograd path's incoming gradient, read without a host sync;
 * licv_ce_rows' grad_coef_dev works the same way). */
int licv_kl_rows_bwd(const void* stu_logits, const void* tea_logits, int dtype, const int64_t* stu_rows, const int64_t* tea_rows,
                     int64_t n_rows, int64_t vocab, int64_t ld_stu, int64_t ld_tea, float temperature, float eps, float upstream,
                     const float* upstream_dev, void* grad_rows_bf16, int64_t ld_grad, void* stream);

/* ---- native layer runner: the Idefics language stack (32 decoder + 8 gated cross-attention layers, hooks, final norm, LM head;
 * hf:idefics/modeling_idefics.py:1052-1074,1179-1182) given the image states, in ONE call: the same kernels, order and dispatch as
 * the Python engine loop (bit-identical), issued from C++ so that launch-bound shapes (decode steps of hooked generate,
 * ref:inference.py:300-321; the 32-token student / prefill) are not paced by the interpreter.  All pointers are device pointers
 * unless marked HOST; the caller owns every buffer. ---- */
typedef struct { const void *in_ln, *qkv_w, *o_w, *post_ln, *gu_w, *down_w; } licv_idefics_dec_w;
typedef struct { const void *in_ln, *q_w, *kv_w, *o_w, *qn_w, *kn_w, *post_ln, *gu_w, *down_w; float gate_attn, gate_dense; } licv_idefics_xattn_w;
typedef struct {
    int64_t hidden, inter, n_heads, head_dim, n_layers, cross_interval;
    int64_t vocab, n_extra_vocab, vocab_total;         /* base vocabulary, additional embeddings, rows of the fused LM head */
    int64_t img_dim, img_len, rope_len;                /* width of the image states, latents per image, rows of cos/sin */
    float rms_eps;
    const void *embed, *embed_extra, *final_ln, *lm_head, *cos, *sin;
    const licv_idefics_dec_w* dec;                     /* HOST array [n_layers] */
    const licv_idefics_xattn_w* xat;                   /* HOST array [n_layers / cross_interval] */
} licv_idefics_text_weights;
typedef struct {
    const int64_t* input_ids;                          /* (B, S) */
    const int32_t* key_valid;                          /* (B, Sk) attention mask over past + new tokens */
    const int64_t* position_ids;                       /* (B*S) */
    const void* image_states;                          /* (B, Nk, img_dim) bf16 */
    const int32_t* img_mask;                           /* (B, S, n_img) */
    const float* gate;                                 /* (B*S) cross_attention_gate */
    int64_t B, S, Sk, Nk, n_img;
    const float* icv;                                  /* (n_hooked, hidden) fp32, or NULL: no intervention */
    const float* alpha;                                /* (n_hooked) fp32 folded into the hook, or NULL: icv is already scaled */
    const int32_t* hook_slot;                          /* HOST array [n_layers]: row of icv for that layer's output, -1 = not hooked */
    void* const* kv_cache;                             /* HOST array [n_layers] of (B, cache_max_len, 2*hidden) bf16, or NULL */
    int64_t cache_max_len, past;
    const int32_t* kv_rows; int64_t ld_kv_rows;        /* decode steps (S == 1): physical cache row of every history position (licv_decode_attn), or NULL */
    const void* const* xkv_cached;                     /* HOST array [n_x] of projected+normed cross-attention K|V (B, Nk, 2*hidden), or NULL */
    void* const* xkv_out;                              /* HOST array [n_x]: project INTO these (they become next step's xkv_cached), or NULL */
    const int64_t* logits_rows; int64_t n_rows;        /* logits only for these flat rows (n_rows = 0: all B*S rows) */
    void *h16, *h32, *x, *xn, *q, *qkv, *o, *act, *xkv, *xsel;   /* scratch: (M,H) bf16, (M,H) fp32, 3 x (M,H) bf16, (M,3H), (M,H), (M,I), (B*Nk,2H), (n_rows,H) */
    void* workspace; int64_t workspace_bytes;          /* split-K scratch (licv_workspace_size) */
    void* logits; int64_t ld_logits;                   /* (rows, ld_logits >= vocab_total) bf16 */
} licv_idefics_text_call;
/* Options of the runner: 0 = fold the decoder layers' two residual adds into the row kernels that follow (licv_add_rmsnorm_fwd,
 * licv_inject_renorm_pre_fwd), so the o / down projections take the register-direct GEMM epilogue (default 1; bit-identical either
 * way, 0 is for A/B timing).  1 = where the o / down / QKV projections run split-K (M < 512), let the row kernel behind each of
 * them sum the slices (the *_ws entry points) instead of a finalize launch (default 1; bit-identical). */
int licv_runner_option(int option, int value);
int licv_idefics_text_forward(const licv_idefics_text_weights* w, const licv_idefics_text_call* c, void* stream);

/* ---- one decode step's self-attention block (ref:inference.py:300-321: B x num_beams rows, one new token each; hf eager attention
 * hf:idefics/modeling_idefics.py:450-470, rotary :396-428, hf:mistral/modeling_mistral.py:122-179 for GQA) in ONE launch: fused QKV rows
 * (bf16, or the fp32 split-K slices of licv_gemm_bf16_splitk_produce) -> rotary on Q / K -> K | V appended to cache[r, past] -> attention
 * of the new query over positions 0..past -> out (M, n_heads * head_dim) bf16.  cache: (rows, max_len, 2 * n_kv_heads * head_dim) bf16,
 * K then V.  kv_rows (M, ld_kv_rows) int32, optional: the physical cache row holding position p of row r's history (beam search keeps
 * the cache in place and reorders this table, see licv_beam_step); NULL = every row reads its own cache row. */
typedef struct {
    const float* qkv_ws; int splits; int64_t slice_elems, row_stride;   /* split-K slices of the QKV projection, or NULL */
    const void* qkv_bf16; int64_t ldq;                                   /* (M, ldq) bf16 rows [Q | K | V] when qkv_ws == NULL */
    const void* cos; const void* sin; const int64_t* position_ids; int64_t n_pos;   /* (n_pos, head_dim) bf16 tables, (M) positions */
    void* cache; int64_t max_len, past;
    const int32_t* kv_rows; int64_t ld_kv_rows;
    const int32_t* key_valid;                                            /* (M, past + 1) attention mask, or NULL */
    void* out;
    int64_t M, n_heads, n_kv_heads, head_dim;
    float scale;
} licv_decode_attn_args;
int licv_decode_attn(const licv_decode_attn_args* a, void* stream);

/* ---- beam search bookkeeping of hooked generate (ref:inference.py:300-321 -> transformers GenerationMixin._beam_search, 5.x form,
 * generation/utils.py:3077-3460; ref:config/inference.yaml:26-30: 3 beams, 5 new tokens, length_penalty 0) ----
 * ONE call (two launches: a scan over vocabulary chunks, a finish per question) per decode step: log_softmax of every beam's logits,
 * top 2*nb of (nb x V) per question, running / finished set update, early-stop heuristic, and the loop condition.  State buffers are ping-ponged by the caller (in != out for the token rows). */
typedef struct {
    const void* logits; int logits_dtype;              /* rows of V logits, bf16 or fp32, row stride ld (elements) */
    int64_t ld;
    int64_t q_stride_rows, beam_stride_rows;           /* logits row of (question b, beam k) = b*q_stride_rows + k*beam_stride_rows
                                                          (nb, 1 in the loop; 1, 0 right after the prefill: all beams share the row) */
    int64_t B, nb, V, max_len, cur, P;                 /* cur = column being written, P = prompt length, max_len = P + max_new_tokens */
    int64_t eos;                                       /* -1 = no EOS token */
    int suppress_eos;                                  /* min_new_tokens not reached: EOS scores -inf */
    float length_penalty; int early_stopping;          /* early_stopping: 1 = True, 0 = False (the reference's setting) */
    const int64_t* running_in; const int64_t* finished_in;       /* (B, nb, max_len) token rows */
    const float* run_scores_in; const float* fin_scores_in;      /* (B, nb) */
    const uint8_t* is_fin_in; const uint8_t* improve_in;         /* (B, nb), (B) */
    const int64_t* gen_len_in;                                   /* (B, nb) generated length of each finished hypothesis */
    int64_t* running_out; int64_t* finished_out; float* run_scores_out; float* fin_scores_out; uint8_t* is_fin_out; uint8_t* improve_out;
    int64_t* gen_len_out;
    int64_t* beam_src_flat;                            /* (B*nb): row b*nb + source beam of every new running beam (the KV-cache reorder) */
    int64_t* next_tokens;                              /* (B*nb): the token each new running beam just appended (next step's input_ids) */
    int32_t* flags;                                    /* [0] = 1 while the search continues (written by the last workgroup) */
    int32_t* sync;                                     /* 4 int32, zero before the first call; the kernel leaves them zero */
    /* optional: the KV-cache row table of licv_decode_attn, (B*nb, kv_ld) int32.  Row r of the new table = row (source beam of r) of
     * the old one, then entry [cur] = r: the token a beam appends next lands in its OWN physical cache row.  NULL = not maintained. */
    const int32_t* kv_rows_in; int32_t* kv_rows_out; int64_t kv_ld;
    void* scratch; int64_t scratch_bytes;              /* >= licv_beam_step_scratch_bytes(B, nb), 16-byte aligned: the scan's per-chunk partials */
} licv_beam_step_args;
int64_t licv_beam_step_scratch_bytes(int64_t B, int64_t nb);
int licv_beam_step(const licv_beam_step_args* a, void* stream);

/* ---- device-side front-end: integer rules between the collator / processor and the first GEMM ---- */
/* Idefics image_attention_mask (B, S, n_images) int32 one-hot rows from input_ids (B, S) int64 by the incremental rule of
 * hf:idefics/processing_idefics.py:89-110 + :66-79 (ref:icv_src/icv_datamodule.py:80-124 gets it from processor.prepare_input). */
int licv_idefics_image_attention_mask(const int64_t* input_ids, int32_t* mask_out, int64_t B, int64_t S, int64_t n_images,
                                      int64_t image_token_id, int64_t eod_token_id, void* stream);
/* Idefics2: per image real flag (some pixel != 0), patch validity (n, gh*gw) int32 and NaViT position ids (n, gh*gw) int64
 * (hf:idefics2/modeling_idefics2.py:831-855, :136-170); boundaries = the module's fp32 arange(1/n_side, 1, 1/n_side);
 * pixel_attention_mask (n, H, W) bytes, may be NULL (= everything attended). */
int licv_idefics2_patch_front(const void* pixel_values_bf16, const void* pixel_attention_mask_u8, const float* boundaries,
                              int32_t* real_out, int32_t* patch_valid_out, int64_t* position_ids_out,
                              int64_t n_images, int64_t height, int64_t width, int64_t patch, int64_t n_side, void* stream);
/* Idefics2 inputs_merger (hf:idefics2/modeling_idefics2.py:789-815): the k-th <image> token (row-major) of input_ids (M) takes row k
 * of image_rows (n_image_rows, dim) in h (M, dim); rank_scratch: M int32; count_out (1 int32, optional) = number of <image> tokens. */
int licv_merge_image_rows(void* h_bf16, const int64_t* input_ids, const void* image_rows_bf16, int32_t* rank_scratch,
                          int32_t* count_out, int64_t M, int64_t dim, int64_t n_image_rows, int64_t image_token_id, void* stream);

/* Image input (ref:icv_src/icv_datamodule.py:80-124 -> processor.prepare_input -> hf:image_transforms.py rescale :118-122, normalize
 * :437; hf:idefics2 image processor's padding + pixel_attention_mask): src (n, H, W, 3) uint8 on the device -> dst (n, 3, H, W) bf16 =
 * bf16(((float)(u8 * rescale) - mean[c]) / std[c]); valid_hw (n, 2) int32 (optional): real height / width of each image inside the
 * padded H x W (pixels outside are 0, mask 0); mask (n, H, W) uint8 (optional).  mean3 / std3 are HOST pointers. */
int licv_preprocess_images(const void* src_u8, const int32_t* valid_hw, void* dst_bf16, void* mask_u8, int64_t n_images,
                           int64_t H, int64_t W, double rescale, const float* mean3, const float* std3, void* stream);

/* ---- loss + optimiser (ref:icv_src/icv_module.py:121-134, :171-209) ---- */
/* per-row KL(teacher||student) with eps inside the log, rows gathered by index; out_rows fp32 (n_rows). */
int licv_kl_rows_fwd(const void* stu_logits, const void* tea_logits, int dtype,
                     const int64_t* stu_rows, const int64_t* tea_rows, int64_t n_rows, int64_t vocab,
                     int64_t ld_stu, int64_t ld_tea, float temperature, float eps, float* out_rows, void* stream);
/* d/dT of the same per-row sum (ref:icv_src/icv_module.py:49-52: `temperature` is a Parameter, trainable with learnable_t);
 * the caller forms dL/dT = T^2 * mean(out_rows) + 2 T * mean(kl_rows).  out_rows fp32 (n_rows). */
int licv_kl_rows_dtemp(const void* stu_logits, const void* tea_logits, int dtype,
                       const int64_t* stu_rows, const int64_t* tea_rows, int64_t n_rows, int64_t vocab,
                       int64_t ld_stu, int64_t ld_tea, float temperature, float eps, float* out_rows, void* stream);
/* fused AdamW over a flat fp32 buffer; lr per element group given by a split index (alpha first). */
int licv_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_group0,
                    float lr0, float lr1, float beta1, float beta2, float eps, float weight_decay,
                    int64_t step, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LICV_HIP_H */
