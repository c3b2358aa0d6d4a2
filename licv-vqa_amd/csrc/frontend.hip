// Device-side front-end of the hot path: the integer rules that sit between the collator / processor and the
// first GEMM, written as HIP kernels so that a forward pass needs no host round trip (the first version prepared them with
// torch ops on the host side of the stream: patch_mask.cpu() -> bucketize -> upload, a boolean gather, nonzero()).
//   * Idefics  : image_attention_mask (B, S, N) from input_ids by the incremental rule of
//                hf:idefics/processing_idefics.py:89-110 (+ incremental_to_binary_attention_mask :66-79);
//   * Idefics2 : per image "is it a padding image" flag, patch validity mask and NaViT position ids
//                (hf:idefics2/modeling_idefics2.py:831-855, :136-170);
//   * Idefics2 : inputs_merger (hf:idefics2/modeling_idefics2.py:789-815) as rank-of-<image>-token + row copy, no nonzero().
// All of it is byte/integer work bounded by HBM bandwidth: coalesced 16-byte loads, one wave (or workgroup) per row / image.
#include <cstring>
#include "common.h"

// ------------------------------------------------------------------------------------------------
// Idefics: token t of row b sees the most recent <image> before or at t (index = number of <image> tokens so far - 1),
// except after an end-of-document token until the next <image>; tokens before the first image see none.  One wave per
// row walks the sequence in 64-token chunks; the running state (image count, last image position, last EOD position)
// is carried in scalars, the within-chunk prefix is two ballots and a few bit operations.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64)
void idefics_image_mask_k(const int64_t* __restrict__ ids, int32_t* __restrict__ out, int S, int n_img, int64_t image_tok, int64_t eod_tok) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t* row = ids + (int64_t)b * S;
    int32_t* orow = out + (int64_t)b * S * n_img;
    int count = -1, last_img = -1, last_eod = -1;
    const unsigned long long le = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull), lt = le >> 1;
    for (int c0 = 0; c0 < S; c0 += 64) {
        const int pos = c0 + lane;
        const int64_t tok = pos < S ? row[pos] : -1;
        const unsigned long long bi = __ballot(pos < S && tok == image_tok), be = __ballot(pos < S && tok == eod_tok);
        const unsigned long long bil = bi & le, bel = be & lt;
        const int cnt = count + __popcll(bil);
        const int li = bil ? c0 + 63 - __clzll(bil) : last_img;
        const int eb = bel ? c0 + 63 - __clzll(bel) : last_eod;        // last EOD strictly before this position
        const int idx = (eb > li) ? -1 : cnt;
        // one-hot rows of this chunk, written cooperatively: element e of the chunk's (64 x n_img) block
        const int nrows = min(64, S - c0), total = nrows * n_img;
        for (int e0 = 0; e0 < total; e0 += 64) {                      // uniform trip count: every lane takes part in the shuffle
            const int e = e0 + lane;
            const int r = min(e / n_img, nrows - 1), n = e - r * n_img;
            const int ridx = __shfl(idx, r, 64);
            if (e < total) orow[(int64_t)(c0 + r) * n_img + n] = (n == ridx) ? 1 : 0;
        }
        count += __popcll(bi);
        if (bi) last_img = c0 + 63 - __clzll(bi);
        if (be) last_eod = c0 + 63 - __clzll(be);
    }
}

extern "C" int licv_idefics_image_attention_mask(const int64_t* input_ids, int32_t* mask_out, int64_t B, int64_t S, int64_t n_images,
                                                 int64_t image_token_id, int64_t eod_token_id, void* stream) {
    LICV_CHECK_ARG(input_ids && mask_out, "idefics_image_attention_mask: null pointer");
    LICV_CHECK_ARG(B >= 0 && S > 0 && n_images > 0 && S < (1ll << 30) && n_images < (1ll << 20), "idefics_image_attention_mask: bad shape");
    if (B == 0) return LICV_OK;
    idefics_image_mask_k<<<(unsigned)B, 64, 0, (hipStream_t)stream>>>(input_ids, mask_out, (int)S, (int)n_images, image_token_id, eod_token_id);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

// ------------------------------------------------------------------------------------------------
// Idefics2 image front-end, one workgroup per image:
//   real[n]   = some pixel != 0                        (padding images are all-zero, hf:...:831-836)
//   valid[n,t]= every pixel of patch t is attended     (unfold + sum == P*P, :848-853)
//   pos[n,t]  = bucket(h) * n_side + bucket(w) for valid patches, else 0 (:150-170): fractional coordinate
//               i * (1 / nb) in fp32, clamped to 1 - 1e-6, rounded to bf16 (the pixel dtype), bucketize(right=True)
//               against the module's fp32 boundaries arange(1/n_side, 1, 1/n_side) — passed in, so they are the same
//               fp32 values torch.arange produces.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void idefics2_patch_front_k(const bf16_t* __restrict__ pix, const uint8_t* __restrict__ pam, const float* __restrict__ bounds,
                            int32_t* __restrict__ real, int32_t* __restrict__ valid, int64_t* __restrict__ pos,
                            int Hh, int Ww, int P, int n_side) {
    extern __shared__ unsigned char sm[];                // [gh*gw] patch validity, then [gh] + [gw] buckets (ints)
    const int n = blockIdx.x, tid = threadIdx.x;
    const int gh = Hh / P, gw = Ww / P, T = gh * gw;
    unsigned char* pv = sm;
    int* bkt = reinterpret_cast<int*>(sm + ((T + 3) & ~3));
    __shared__ int s_any, s_nbh, s_nbw;
    if (tid == 0) { s_any = 0; s_nbh = 0; s_nbw = 0; }
    __syncthreads();
    // (1) any non-zero pixel: 16-byte loads over 3*H*W bf16 (+-0 have no magnitude bits set; NaN counts as non-zero, as != does)
    {
        const int64_t total = (int64_t)3 * Hh * Ww;
        const bf16_t* p = pix + (int64_t)n * total;
        int any = 0;
        const int64_t nvec = ((uintptr_t)p & 15) == 0 ? total / 8 : 0;
        const uint4* pq = reinterpret_cast<const uint4*>(p);
        for (int64_t i = tid; i < nvec; i += 256) {
            const uint4 v = pq[i];
            any |= ((v.x | v.y | v.z | v.w) & 0x7fff7fffu) != 0;
        }
        for (int64_t i = nvec * 8 + tid; i < total; i += 256) any |= (p[i] & 0x7fff) != 0;
        if (__any(any) && (tid & 63) == 0) atomicOr(&s_any, 1);
    }
    // (2) patch validity
    const uint8_t* m = pam ? pam + (int64_t)n * Hh * Ww : nullptr;
    for (int t = tid; t < T; t += 256) {
        const int gy = t / gw, gx = t - gy * gw;
        int ok = 1;
        if (m) {
            for (int y = 0; y < P && ok; ++y) {
                const uint8_t* r = m + (int64_t)(gy * P + y) * Ww + gx * P;
                for (int x = 0; x < P; ++x) ok &= r[x] != 0;
            }
        }
        pv[t] = (unsigned char)ok;
    }
    __syncthreads();
    // (3) patches per column 0 / row 0 -> the fractional grid
    if (tid < gh && pv[tid * gw]) atomicAdd(&s_nbh, 1);
    if (tid >= 128 && tid - 128 < gw && pv[tid - 128]) atomicAdd(&s_nbw, 1);
    __syncthreads();
    auto bucket = [&](int i, int nb) -> int {
        const float inv = 1.0f / (float)nb;                          // torch: 1.0 / nb  (int -> fp32 division)
        float f = fminf((float)i * inv, 0.999999f);                   // clamp(max = 1 - 1e-6) in fp32
        if (!(f == f)) return 0;                                      // 0 * inf: only reachable for all-masked images (masked below)
        const float x = rbf(f);                                       // .to(bfloat16)
        int k = 0;
        for (int j = 0; j < n_side - 1; ++j) k += (bounds[j] <= x);   // bucketize(right=True): boundaries[k-1] <= x < boundaries[k]
        return k;
    };
    if (tid < gh) bkt[tid] = bucket(tid, s_nbh);
    if (tid >= 128 && tid - 128 < gw) bkt[gh + tid - 128] = bucket(tid - 128, s_nbw);
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
        const int gy = t / gw, gx = t - gy * gw;
        const int ok = pv[t];
        valid[(int64_t)n * T + t] = ok;
        pos[(int64_t)n * T + t] = ok ? (int64_t)bkt[gy] * n_side + bkt[gh + gx] : 0;
    }
    if (tid == 0) real[n] = s_any;
}

extern "C" int licv_idefics2_patch_front(const void* pixel_values_bf16, const void* pixel_attention_mask_u8, const float* boundaries,
                                         int32_t* real_out, int32_t* patch_valid_out, int64_t* position_ids_out,
                                         int64_t n_images, int64_t height, int64_t width, int64_t patch, int64_t n_side, void* stream) {
    LICV_CHECK_ARG(pixel_values_bf16 && boundaries && real_out && patch_valid_out && position_ids_out, "idefics2_patch_front: null pointer");
    LICV_CHECK_ARG(patch > 0 && height >= patch && width >= patch && n_side > 0, "idefics2_patch_front: bad geometry");
    LICV_CHECK_ARG(height / patch <= 128 && width / patch <= 128, "idefics2_patch_front: more than 128 patches per side");
    if (n_images <= 0) return LICV_OK;
    const int gh = (int)(height / patch), gw = (int)(width / patch);
    const size_t lds = (size_t)((gh * gw + 3) & ~3) + (size_t)(gh + gw) * sizeof(int);
    idefics2_patch_front_k<<<(unsigned)n_images, 256, lds, (hipStream_t)stream>>>(
        (const bf16_t*)pixel_values_bf16, (const uint8_t*)pixel_attention_mask_u8, boundaries, real_out, patch_valid_out, position_ids_out,
        (int)height, (int)width, (int)patch, (int)n_side);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

// ------------------------------------------------------------------------------------------------
// Idefics2 inputs_merger: the k-th <image> token of the flattened batch (row-major order, = masked_scatter order) receives
// row k of the image hidden states.  One workgroup ranks the tokens (chunked ballot scan), then every wave copies rows.
// `count_out[0]` receives the number of <image> tokens (the caller may compare it with the rows it has; never read here).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void image_token_rank_k(const int64_t* __restrict__ ids, int32_t* __restrict__ rank, int64_t M, int64_t image_tok, int32_t* __restrict__ count_out) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < M; c0 += 1024) {
        const int64_t p = c0 + tid;
        const bool is = p < M && ids[p] == image_tok;
        const unsigned long long bal = __ballot(is);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (p < M) rank[p] = is ? off + before : -1;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; base += t; }
        __syncthreads();
    }
    if (tid == 0 && count_out) count_out[0] = base;
}

__global__ __launch_bounds__(256)
void merge_rows_by_rank_k(bf16_t* __restrict__ h, const int32_t* __restrict__ rank, const bf16_t* __restrict__ src, int64_t M, int dim, int64_t n_src) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int r = rank[row];
    if (r < 0 || r >= n_src) return;                             // not an <image> token (or more tokens than rows: left untouched)
    const uint4* s = reinterpret_cast<const uint4*>(src + (int64_t)r * dim);
    uint4* d = reinterpret_cast<uint4*>(h + row * dim);
    for (int i = lane; i < dim / 8; i += 64) d[i] = s[i];
}

extern "C" int licv_merge_image_rows(void* h_bf16, const int64_t* input_ids, const void* image_rows_bf16, int32_t* rank_scratch,
                                     int32_t* count_out, int64_t M, int64_t dim, int64_t n_image_rows, int64_t image_token_id, void* stream) {
    LICV_CHECK_ARG(h_bf16 && input_ids && image_rows_bf16 && rank_scratch, "merge_image_rows: null pointer");
    LICV_CHECK_ARG(dim > 0 && dim % 8 == 0 && M < (1ll << 31), "merge_image_rows: dim must be a multiple of 8");
    LICV_CHECK_ARG((((uintptr_t)h_bf16 | (uintptr_t)image_rows_bf16) & 15) == 0, "merge_image_rows: pointers must be 16-byte aligned");
    if (M <= 0) return LICV_OK;
    hipStream_t st = (hipStream_t)stream;
    image_token_rank_k<<<1, 1024, 0, st>>>(input_ids, rank_scratch, M, image_token_id, count_out);
    merge_rows_by_rank_k<<<(unsigned)((M + 3) / 4), 256, 0, st>>>((bf16_t*)h_bf16, rank_scratch, (const bf16_t*)image_rows_bf16, M, (int)dim, n_image_rows);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}

// ------------------------------------------------------------------------------------------------
// Image input: uint8 HWC -> normalised bf16 CHW (+ Idefics2's pixel_attention_mask), what the HF image processors do to an image on
// the host before `processor.prepare_input` returns it (ref:icv_src/icv_datamodule.py:80-124; hf:image_transforms.py rescale :118-122
// = f32(f64(u8) * f64 scale), normalize :437 = (x - f32 mean) / f32 std, channels first; hf:idefics2 pads to the batch maximum with
// zeros and marks real pixels).  A byte has 256 values: the 3 x 256 table of final bf16 values is computed on the HOST with exactly
// those IEEE operations (so the result is the reference's float rounded once to bf16, bit for bit; the device's float division is
// not the host's), handed to the kernel by value (1.5 KB of kernel arguments), copied to LDS, and the pixel loop is a table
// look-up and a layout change: 3 bytes in, 3 bf16 planes out, consecutive lanes on consecutive pixels.
// ------------------------------------------------------------------------------------------------
struct ImageTable { bf16_t v[3][256]; };

__global__ __launch_bounds__(256)
void image_preprocess_k(const unsigned char* __restrict__ src, const int32_t* __restrict__ hw, bf16_t* __restrict__ dst,
                        unsigned char* __restrict__ mask, int64_t n, int H, int W, ImageTable tb) {
    __shared__ bf16_t table[3][256];
#pragma unroll
    for (int c = 0; c < 3; ++c) table[c][threadIdx.x] = tb.v[c][threadIdx.x];      // 256 threads: one byte value each
    __syncthreads();
    const int64_t plane = (int64_t)H * W;
    const int64_t img = blockIdx.y;
    const int h = hw ? hw[2 * img] : H, w = hw ? hw[2 * img + 1] : W;
    const unsigned char* s = src + img * plane * 3;
    bf16_t* d = dst + img * plane * 3;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < plane; p += (int64_t)gridDim.x * 256) {
        const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
        const bool real = y < h && x < w;
        const unsigned char r = s[3 * p], g = s[3 * p + 1], b = s[3 * p + 2];
        d[p] = real ? table[0][r] : (bf16_t)0;
        d[plane + p] = real ? table[1][g] : (bf16_t)0;
        d[2 * plane + p] = real ? table[2][b] : (bf16_t)0;
        if (mask) mask[img * plane + p] = real ? 1 : 0;
    }
}

extern "C" int licv_preprocess_images(const void* src_u8, const int32_t* valid_hw, void* dst_bf16, void* mask_u8, int64_t n_images,
                                      int64_t H, int64_t W, double rescale, const float* mean3, const float* std3, void* stream) {
    LICV_CHECK_ARG(src_u8 && dst_bf16 && mean3 && std3, "preprocess_images: null pointer");
    LICV_CHECK_ARG(n_images >= 0 && H > 0 && W > 0 && H * W < (1ll << 31), "preprocess_images: bad shape");
    LICV_CHECK_ARG(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "preprocess_images: zero std");
    if (n_images == 0) return LICV_OK;
    ImageTable tb;
    for (int v = 0; v < 256; ++v) {
        const volatile float x = (float)((double)v * rescale);        // hf:image_transforms.py:118-122 (volatile: one rounding per step,
        for (int c = 0; c < 3; ++c) {                                 //  no wider intermediate, whatever the host compiler's flags)
            const volatile float d = x - mean3[c];
            const volatile float q = d / std3[c];                     // :437
            uint32_t u; float qf = q; memcpy(&u, &qf, 4);
            tb.v[c][v] = (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);      // round to nearest even (finite values only)
        }
    }
    int64_t bx = (H * W + 255) / 256; bx = bx > 64 ? 64 : bx;
    image_preprocess_k<<<dim3((unsigned)bx, (unsigned)n_images), 256, 0, (hipStream_t)stream>>>(
        (const unsigned char*)src_u8, valid_hw, (bf16_t*)dst_bf16, (unsigned char*)mask_u8, n_images, (int)H, (int)W, tb);
    LICV_LAUNCH_CHECK();
    return LICV_OK;
}
