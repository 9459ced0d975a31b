/* C-ABI of libmars_hip.so: the MI355X (gfx950) kernels beneath the MARS
 * alignment-and-ranking path.
 *
 * The reference (a pure Python / PyTorch pipeline) has no FFI of its own; its seam is
 * the Python object API (mars.MARS.MARS, the four components, BackboneLoader).  Each
 * entry point below names the reference lines whose arithmetic it replaces.  The
 * Python shells in the package keep the reference signatures and call these through
 * ctypes with tensor.data_ptr() values.
 *
 * Conventions
 *   - every device buffer is allocated by the caller; the library owns no device memory apart from
 *     the 2 MiB static array of marship_vva_prior_static
 *   - all calls are asynchronous on `stream` (a hipStream_t passed as void*), no sync; the stream and
 *     every device pointer must belong to the device that is current in the calling thread
 *   - no state besides the three GEMM tuning hooks (variant, split, slots: process-wide relaxed atomics; see their
 *     comments) and the device-resident counters and scratch arrays named at their entry points
 *   - return 0 on success, negative on error; marship_last_error() gives the text
 *   - "TN GEMM": C[M,N] = A[M,K] * B[N,K]^T, both operands K-contiguous
 *   - dtype codes: 0 = float32, 1 = bfloat16
 */
#ifndef MARS_HIP_H
#define MARS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* marship_last_error(void);
int marship_version(void);

/* ---- dense contractions ------------------------------------------------------- */
typedef struct MarsGemm {
    const void* A; const void* B; void* C;
    int64_t lda, ldb, ldc;           /* in elements */
    int64_t sA0, sA1, sB0, sB1, sC0, sC1; /* batch strides (elements): outer, inner */
    int32_t M, N, K;                 /* K*elemsize must be a multiple of 128 bytes */
    int32_t nb0, nb1;                /* batch = nb0*nb1; outer index = z / nb1 */
    const float* bias;               /* [N] or NULL */
    const float* gamma;              /* [N] or NULL (LayerScale) */
    const float* residual;           /* f32 [M,N] or NULL, added last */
    int64_t ldr, sR0, sR1;
    float alpha;                     /* applied to the accumulator first */
    int32_t act;                     /* 0 none, 1 GELU(erf), 2 QuickGELU, 3 ReLU */
    int32_t in_dtype;                /* A and B: 0 f32, 1 bf16 */
    int32_t out_dtype;               /* C: 0 f32, 1 bf16 */
    /* A LayerNorm between two products, folded into both (all NULL: plain product; bf16 products, N a multiple
     * of 64, 16-byte aligned rows, one batch).  The f32 stream the LayerNorm reads is kept as two bf16 planes,
     * ln_copy = bf16(x) and ln_lo = bf16(x - ln_copy) (x = ln_copy + ln_lo to 16 mantissa bits), row stride
     * ldcopy; marship_row_stats makes them from f32 rows.  A product that adds to the stream sets ln_copy and
     * ln_lo: its residual is read from the planes (residual must be NULL), and
     *   with ln_part: the result goes back into the planes (C is not written) and ln_part receives, per row and
     *     run of 64 columns, the sum and the sum of squares of the result (marship_ln_finish -> mean, rstd);
     *   without ln_part: the result goes to C as f32 (the last product of a run). */
    void* ln_copy;
    int64_t ldcopy;
    float* ln_part;                  /* f32 [M, N/64, 2] */
    /* The product that would have read the normalised rows takes ln_copy as A, weights with the LayerNorm scale
     * multiplied into their columns, and
     * out = gamma*act(rstd[m]*(A.B^T - mean[m]*ln_colsum[n]) + bias[n])   (alpha must be 1, no residual): */
    const float* ln_stat;            /* f32 [M, 2]: mean and 1/sqrt(var + eps) of row m of the stream */
    const float* ln_colsum;          /* f32 [N]: sum over k of B[n, k] as stored (bf16 values, f32 sum) */
    void* ln_lo;                     /* bf16 [M, N], row stride ldcopy: see ln_copy */
} MarsGemm;
/* Every nn.Linear / conv-as-GEMM / similarity / refinement product on the path:
 * dinov2/layers/attention.py:51,61; mlp.py:35-41; patch_embed.py:76; clip/model.py:283-300,361;
 * alpha_clip/model.py:222,251,359-361; VisualVisualAlignmentModule.py:69,78-79;
 * PriorInformationRefinementModule.py:74-79. out = residual + gamma*act(alpha*A.B^T + bias). */
int marship_gemm(const MarsGemm* g, void* stream);
/* Tuning hook (no reference counterpart): selects the GEMM kernel variant, returns the previous
 * one. 0 = register-staged 128x128 tile, 1 = LDS-DMA 128x128, 2 = LDS-DMA 256x128,
 * 3 = LDS-DMA ring 256x256, 4 = chosen by shape (default), 5 = ring 256x128, 6 = ring 128x128,
 * 7 = ring 64x128, 8 = ring 256x256 with pair fills, 9 / 10 / 11 = 128x128 (eight slots), 64x128 (six slots),
 * 128x128 (four slots) with pair fills, 12 = 8 as persistent workgroups.  Process-wide: every later marship_gemm of every thread reads it (a
 * relaxed atomic: no data race, but no ordering with products other threads are launching) - set it at
 * start-up or while no other thread launches products.  The same holds for marship_set_gemm_split. */
int marship_set_gemm_variant(int variant);
/* The variant marship_gemm would run for this product (5, 6, 7, 8, 10, 11 or 12 under "chosen by shape"); used by
 * the benchmark to attribute launch times to kernels.  No reference counterpart. */
int marship_gemm_variant_for(const MarsGemm* g);
/* Tuning hook (no reference counterpart): under "chosen by shape" a tall unbatched product whose tiles would
 * leave most CUs idle in their last round runs as two launches, rows [0, r) on the large tile and rows [r, M)
 * as a product of their own; a bf16 product of between one and two rounds of 256 x 256 tiles whose row tiles fit
 * one round is cut the other way: whole columns of tiles that fill one round, then the remaining columns.
 * marship_set_gemm_split(0) keeps every product in one launch; returns the
 * previous setting.  marship_gemm_split_rows returns r (M when the product is not split by rows). */
int marship_set_gemm_split(int on);
int marship_gemm_split_rows(const MarsGemm* g);
/* Tuning hook (no reference counterpart): the number of workgroups a persistent launch (variant 12: every
 * workgroup walks the tiles b, b + slots, ... and fills the ring for its next tile under the epilogue of the one
 * in hand) uses; 0 = one per CU (default).  Rounded down to a multiple of 8 above 8.  Process-wide like the two
 * hooks above; returns the previous setting. */
int marship_set_gemm_slots(int slots);

/* ---- ViT glue ------------------------------------------------------------------ */
/* nn.LayerNorm over the last dim (dinov2 vision_transformer.py:97 eps 1e-6; clip/model.py:265-271
 * eps 1e-5, computed in f32).  x f32 [rows, D] with row stride x_stride; y out_dtype. */
int marship_layernorm(const float* x, int64_t x_stride, const float* w, const float* b, void* y,
                      int64_t y_stride, int rows, int D, float eps, int out_dtype, void* stream);
/* The same LayerNorm (clip/model.py:265-271, alpha_clip/model.py:254-258) folded into the products on either
 * side of it, see MarsGemm.ln_*: marship_ln_finish turns the row sums a stream product left in ln_part
 * ([rows, nparts, 2], nparts = N/64) into stat [rows, 2] = mean, 1/sqrt(var + eps) (var = E[x^2] - mean^2 in
 * f32); marship_row_stats computes the same statistics and the two bf16 planes straight from f32 rows, for the
 * first LayerNorm of a chain (copy = bf16(x), lo = bf16(x - copy), both [rows, D] with row stride ldcopy).
 * The folded form rounds x, not x - mean, to bf16: against a LayerNorm of the f32 rows it loses a factor of about
 * |mean| * rstd in accuracy on a row whose mean is large against its spread.  `watch` (may be null): one f32 that
 * both calls raise to the largest |mean| * rstd above `watch_limit` they meet (an atomic maximum; the caller zeroes
 * it and reads it when the stream has passed), for callers that want to fall back to stand-alone LayerNorms then. */
int marship_ln_finish(const float* part, int nparts, int rows, int D, float eps, float* stat, float* watch,
                      float watch_limit, void* stream);
int marship_row_stats(const float* x, int64_t x_stride, int rows, int D, float eps, void* copy, void* lo,
                      int64_t ldcopy, float* stat, float* watch, float watch_limit, void* stream);
/* Patch extraction for the strided conv patch embeddings (patch_embed.py:76; clip/model.py:361;
 * alpha_clip/model.py:359-361) with the input normalisation (x*prescale - mean[c])/std[c] fused
 * (backbone_loader.py:74-78,178-179,185).  img [B,C,H,W] (f32, or u8 when in_u8), out
 * [B*gh*gw, ldo] with column c*p*p+dy*p+dx+col_off; columns >= C*p*p+col_off up to ldo are not
 * touched. */
int marship_im2col(const void* img, int in_u8, int B, int C, int H, int W, int p, const float* mean,
                   const float* stdv, float prescale, void* out, int64_t ldo, int col_off,
                   int out_dtype, void* stream);
/* Text tokens: X[n*L + l, :] = table[ids[n*L + l], :] + pos[l, :]  (clip/model.py:488-489: token_embedding(text) +
 * positional_embedding).  table f32 [vocab, D], ids int32 [rows] (rows = n*L; an id outside the table is clamped
 * into it), pos f32 [L, D], X f32 [rows, D]. */
int marship_embed_tokens(const float* table, int vocab, const int32_t* ids, const float* pos, float* X,
                         int64_t rows, int L, int D, void* stream);
/* Token rows that do not come from the patch GEMM: cls(+pos[0]) and register tokens
 * (vision_transformer.py:230-241; clip/model.py:364-365). X f32 [B,T,D]. */
int marship_write_prefix_tokens(float* X, int B, int T, int D, const float* cls, const float* pos0,
                                const float* regs, int num_regs, void* stream);
/* Batched transpose of 2-byte or 4-byte elements: out[b][c][r] = in[b][r][c]. */
int marship_transpose(const void* in, int64_t ldi, int64_t sI0, int64_t sI1, void* out, int64_t ldo,
                      int64_t sO0, int64_t sO1, int rows, int cols, int nb0, int nb1, int elem_size,
                      void* stream);
/* Row softmax of attention scores (attention.py:57; alpha_clip/model.py:243; causal mask of
 * clip/model.py:471-477).  S f32 [nb, T, lds]; P out_dtype [nb, T, ldp], columns T..ldp-1 zeroed. */
int marship_softmax_rows(const float* S, int64_t lds, void* P, int64_t ldp, int nb, int T, int causal,
                         int out_dtype, void* stream);
/* acc[i,j] += scale * sum_h P[h, off+i, off+j]  (PriorInformationRefinementModule.py:31-45 and
 * nn.MultiheadAttention's head averaging, clip/model.py:295). P in_dtype [H, T, ldp]. */
int marship_attn_mean_accum(const void* P, int64_t ldp, int64_t head_stride, int H, int off, int Pn,
                            float scale, float* acc, int64_t ldacc, int in_dtype, void* stream);
/* SwiGLU gate: out[m*ldo + j] = silu(x[m, j]) * x[m, Hd + j]  (swiglu_ffn.py:29-33). */
int marship_swiglu(const void* x12, void* out, int64_t ldo, int M, int Hd, int dtype, void* stream);
/* dtype casts between f32 and bf16 over n contiguous elements. */
int marship_cast(const void* in, void* out, int64_t n, int in_dtype, int out_dtype, void* stream);

/* Fused attention for 64-wide heads on packed qkv [B*T, 3*D] (bf16):
 * O = softmax(q*scale k^T) v  (attention.py:54-60; alpha_clip/model.py:233-246), O bf16 [B*T, D];
 * lse f32 [B*H, T] receives log-sum-exp of the scaled scores (may be NULL).
 * score_scale_log2 says what the stored queries still lack: the softmax is taken over 2^(score_scale_log2 * q.k).
 * Raw queries: log2(e)/8 (the kernel multiplies its query fragments by it and rounds them to bf16 again).  Queries
 * the caller has already scaled by log2(e)/8 - e.g. in the epilogue of the qkv product, one rounding instead of
 * two - : exactly 1.0f, and the kernel uses them as they are.  The same parameter, with the same meaning, on the
 * two entry points below. */
int marship_flash_attn(const void* qkv, void* O, float* lse, int B, int T, int H, int causal,
                       float score_scale_log2, void* stream);
/* The same attention with q, k, v and the probabilities quantised to 8-bit floats (OCP e4m3, one scale per image,
 * head and operand) and both contractions on the fp8 MFMA; f32 softmax and accumulation, no mask, no lse.  The caller
 * brings the workspace: q8, k8 [B*H*Tp*64] bytes, vt8 [B*H*64*Tp] bytes, scales [B*H*6] f32 (factors, then maxima), Tp = T rounded up to 32.
 * For the configuration that names an fp8 attention path; the reference (attention.py:54-60; alpha_clip/model.py:233-246)
 * has no such mode. */
int marship_flash_attn_fp8(const void* qkv, void* O, int B, int T, int H, void* q8, void* k8, void* vt8, float* scales,
                           int Tp, float score_scale_log2, void* stream);
/* acc[i,j] += scale * sum_h exp(q_h[i].k_h[j]/8 - lse[h,i]) over the patch block starting at
 * token `off` of image 0 (bf16 qkv as above): the head/layer mean of the attention maps
 * (PriorInformationRefinementModule.py:31-45) without materialising them. */
int marship_flash_attn_mean(const void* qkv, const float* lse, int T, int H, int off, int Pn,
                            float scale, float* acc, int64_t ldacc, float score_scale_log2, void* stream);

/* ---- alignment / refinement ------------------------------------------------------ */
/* F.normalize(p=2, dim=1) on gathered rows (VisualVisualAlignmentModule.py:122-125;
 * FilteringMergingModule.py:179,200).  Row r of image b is x + b*x_bstride + (row_off+r)*x_stride. */
int marship_l2norm_rows(const float* x, int64_t x_bstride, int64_t x_stride, int row_off, int B,
                        int rows, int D, void* y, int64_t y_stride, int out_dtype, void* stream);
/* Foreground/background column statistics of S and the raw prior
 * vva = mean_fg*max_fg - mean_bg*max_bg, min-max scaled (VisualVisualAlignmentModule.py:76-102).
 * S f32 [R, ldS] (R = ns*P), fg u8 [R]; ws f32 workspace of 4*chunks*P floats; out f32 [P]. */
int marship_vva_prior(const float* S, int64_t ldS, const uint8_t* fg, int R, int P, float* ws,
                      int chunks, float* out, void* stream);
/* marship_vva_prior for callers that cannot allocate device memory or bring a workspace (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): no counterpart in the reference.
 * Same kernels and results as marship_vva_prior with chunks = 32; the partial sums live in a 2 MiB
 * file-scope device array of the library (one copy per GPU, held until the process ends), so P is limited
 * to 4096 query patches and a larger P is rejected with an error.  Calls no hipMalloc / hipFree.  The first
 * call on a device resolves the array's address there (make that call outside a capture).  All calls on one
 * GPU share the array: calls that overlap on one GPU (two streams, two threads) race; order them. */
int marship_vva_prior_static(const float* S, int64_t ldS, const uint8_t* fg, int R, int P, float* out,
                             void* stream);
/* (x - min) / (1e-7 + max - min) over n elements (single block). */
int marship_minmax(const float* x, float* y, int n, void* stream);
/* Box mask B of the thresholded prior (PriorInformationRefinementModule.py:53-65,91-122):
 * u8 = trunc(prior*255); keep u8 > int(thr*max(u8)); bounding box of every 8-connected component,
 * x1=min(x+w,W-1), y1=min(y+h,H-1); fill [y0:y1, x0:x1].  prior f32 [G,G], Bm f32 [G*G]. */
int marship_box_mask(const float* prior, int G, double threshold, float* Bm, void* stream);
/* D = A / colsum(A); D = D / rowsum(D)  (PriorInformationRefinementModule.py:70-71).
 * A f32 [P, ldA]; D f32 [P, ldD] with columns P..ldD-1 zeroed; colsum: f32 scratch of 33*P. */
int marship_sinkhorn1(const float* A, int64_t ldA, float* D, int64_t ldD, int P, float* colsum,
                      void* stream);
/* R = max(D, DDT) written both as R [P, ld] and R^T [P, ld], pads zeroed (line 74-75). */
int marship_max_and_transpose(const float* D, const float* DDT, int64_t ld, int P, float* R,
                              float* RT, void* stream);
/* out[i] = sum_j R2[i,j] * Bm[j] * prior[j]  (lines 81-87). */
int marship_masked_matvec(const float* R2, int64_t ld, const float* Bm, const float* prior, int P,
                          float* out, void* stream);

/* ---- resampling ------------------------------------------------------------------ */
/* Separable weighted resample with host-built tables (torch antialiased bicubic / bilinear:
 * VisualTextAlignmentModule.py:88-92; backbone_loader.py:182-186).  One pass along `axis`
 * (1 = width, 0 = height) of in [planes, H, W]: out[.., o, ..] = sum_k w[o,k] * in[.., xmin[o]+k, ..]
 * with `nout` output positions on that axis.  `quant_u8` applies v = trunc(v*255)/255 to the
 * input first (FilteringMergingModule.py:194); the result is (acc - shift)/scale.  Two calls
 * (width, then height) resample an image. */
int marship_resample_axis(const float* in, int planes, int H, int W, int axis, int nout,
                          const int32_t* xmin, const int32_t* xsize, const float* weights, int kmax,
                          int quant_u8, float shift, float scale, void* out, int out_dtype,
                          void* stream);
/* PIL (Pillow) 8-bit resample pass with 22-bit fixed-point coefficients
 * (backbone_loader.py:172-176 via PIL.Image.resize): out = clip8((2^21 + sum in*kk) >> 22). */
int marship_resample_axis_u8(const uint8_t* in, int planes, int H, int W, int axis, int nout,
                             const int32_t* xmin, const int32_t* xsize, const int32_t* kk, int kmax,
                             uint8_t* out, void* stream);
/* img f32 [C,H,W] in [0,1] -> u8 trunc(x*255) (torchvision ToPILImage on a float array). */
int marship_f32_to_u8(const float* in, uint8_t* out, int64_t n, void* stream);

/* ---- proposal pooling / scoring / merge / IoU ----------------------------------- */
/* adaptive_max_pool2d(mask, G) > 0 (FilteringMergingModule.py:73-81,104-107;
 * VisualVisualAlignmentModule.py:72-75). masks f32 [N,S,S] -> bits u8 [N, G*G]. */
int marship_pool_masks(const float* masks, int N, int S, int G, uint8_t* bits, void* stream);
/* Per-proposal patch statistics (FilteringMergingModule.py:108-110): cnt[n] = |m_n|,
 * sum_vva[n], sum_vta[n]; union_cnt = |OR_n m_n|. */
int marship_proposal_stats(const uint8_t* bits, int N, int P, const float* vva, const float* vta,
                           int32_t* cnt, float* sum_vva, float* sum_vta, int32_t* union_cnt,
                           void* stream);
/* AlphaCLIP score: s[n] = <f_n/|f_n|, t/|t|> (FilteringMergingModule.py:97,179,200). */
int marship_cosine_scores(const float* feats, int64_t ldf, int N, int E, const float* text,
                          float* scores, void* stream);
/* Scores, ranking and keep decision (FilteringMergingModule.py:118-138,213-217) in the number types of
 * the reference's numpy expressions: EMD min-max and the mean of four in float64, the AlphaCLIP min-max
 * in float32.  order[k] = index of the k-th best proposal (stable), keep[n] in {0,1}.  A proposal whose
 * score is not finite (emd NaN: no patches) ranks after every finite one and is not kept. */
int marship_rank(const double* emd, const float* aclip, const int32_t* cnt, const float* sum_vva,
                 const float* sum_vta, const int32_t* union_cnt, int N, double alpha,
                 double static_thr, double dynamic_thr, double* score, int32_t* order,
                 uint8_t* keep, void* stream);
/* merged = (sum of kept masks > 0) (FilteringMergingModule.py:219). */
int marship_merge_masks(const float* masks, const uint8_t* keep, int N, int64_t HW, float* merged,
                        void* stream);
/* Proposal hand-off (main_MARS.py:62 reads one [N, H, W] tensor per episode): the same masks as one bit per
 * pixel, words[mask][j] bit i = masks[mask][32 j + i] > 0 with ceil(HW / 32) words per mask, and back to {0,1}
 * floats.  No counterpart in the reference (it stores floats). */
int marship_pack_masks(const float* masks, int n, int64_t HW, uint32_t* words, void* stream);
int marship_unpack_masks(const uint32_t* words, int n, int64_t HW, float* masks, void* stream);
/* Intersection / union areas (mars/utils/evaluation.py:11-39): counts[0..3] += {inter0, inter1,
 * union0, union1} over pixels whose ignore flag is 0. */
int marship_iou_counts(const float* pred, const float* gt, const float* ignore, int64_t HW,
                       float* counts, void* stream);

/* Matched points inside each mask (matcher/Matcher.py:1167-1174,1200-1201): points are (x, y) in
 * pixels, truncated and clipped to the image as the reference does; counts[k] = number of the K
 * points whose pixel of mask k is non-zero.  masks [n, H, W] f32, points [K, 2] f32, counts int32 [n]. */
int marship_points_in_masks(const float* masks, int n, int H, int W, const float* points, int K,
                            int32_t* counts, void* stream);
/* Prompt-similarity table (mars/ClipPipeline.py:165-168): out[i, p] = scale * cos(feats[i], text[p]);
 * feats [N, E] (row stride ldf), text [P, E], out f32 [N, P]. */
int marship_cosine_table(const float* feats, int64_t ldf, int N, int E, const float* text, int P,
                         float scale, float* out, void* stream);

/* ---- Segment-Anything proposal stage --------------------------------------------- */
/* Row gather: dst[r, dst_col : dst_col+ncols] = idx[r] >= 0 ? src[idx[r], src_col : src_col+ncols] : 0.
 * Window partition / un-partition with zero padding (segment_anything/modeling/image_encoder.py:243-289),
 * 3x3 neighbourhoods of the neck convolution (:96-102) and the pixel shuffle of the transposed
 * convolutions (mask_decoder.py:53-59) are all instances.  dtype 0 = f32, 1 = bf16 (same on both sides). */
int marship_gather_rows(const void* src, int64_t ld_src, int src_col, const int32_t* idx, void* dst,
                        int64_t ld_dst, int dst_col, int64_t rows, int ncols, int dtype, void* stream);
/* Attention of the mask decoder's two-way transformer (segment_anything/modeling/transformer.py:218-240):
 * out[b, i, h*hd + d] = sum_j softmax_j(scale * q[b, i, h, :] . k[b, j, h, :]) v[b, j, h, d] on UNPADDED heads of 16 or 32
 * dims, for products where one side is a handful of tokens: Tk <= 16 (image -> tokens, tokens -> tokens; any Tq) or
 * Tq <= 16 (tokens -> image; any Tk).  q [B*Tq, ldq], k / v [B*Tk, ld], out [B*Tq, ldo], head h at column h*hd; H a power of
 * two; rows 16-byte aligned; f32 accumulation; dtype 0 = f32, 1 = bf16 (all four tensors).  `shared` bit 0: q holds one
 * set of Tq rows that every prompt uses (the image-side queries of the first layer, where the image stream is still the
 * same for every prompt); bit 1: the same for k and v (Tk rows). */
int marship_attn_small(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* out, int64_t ldo, int B, int H, int hd, int Tq, int Tk, float scale, int shared, int dtype,
                       void* stream);
/* Decomposed relative-position terms (image_encoder.py:325-361): q is read from the packed qkv
 * buffer (row b*T+t, column head*hdp+d, d < hd); Rh [gh, Hk, hd] and Rw [gw, Wk, hd] are the
 * get_rel_pos tables; rel_h [B*H, T, Hk], rel_w [B*H, T, Wk] f32. */
int marship_relpos_bias(const void* qkv, int64_t ld, int B, int T, int H, int hd, int hdp, int gw,
                        int Hk, int Wk, const float* Rh, const float* Rw, float* rel_h, float* rel_w,
                        int dtype, void* stream);
/* The same with the queries of token row r = b*T + t read from row in_rows[r] of qkv, or from pad_row for a negative
 * entry (as marship_sam_attn_rows): bf16 queries, grids and tables up to 16 a side, head dims a multiple of 8 up to 128. */
int marship_relpos_bias_rows(const void* qkv, int64_t ld, int B, int T, int H, int hd, int hdp, int gw,
                             int Hk, int Wk, const float* Rh, const float* Rw, float* rel_h, float* rel_w,
                             const int32_t* in_rows, const void* pad_row, int dtype, void* stream);
/* The attention of an encoder block in one kernel, without the score matrix (image_encoder.py:224-240 with the
 * bias of :357-359): O[b*T+t, head*hdp+d] = sum_k softmax_k(scale q.k + rel_h[.., k / Wk] + rel_w[.., k % Wk]) v[k, d]
 * for bf16 qkv rows (row stride ld; q, k, v of head h at columns h*hdp, (H+h)*hdp, (2H+h)*hdp), hdp 64 or 128
 * (zero-padded head dims give zero outputs) or 80 (the ViT-H head as it is), rel_h [nb*H, T, Hk], rel_w [nb*H, T, Wk] as marship_relpos_bias
 * leaves them.  Replaces the score product, marship_softmax_bias_rows and the P.V product for bf16 towers.
 * T <= 65536, Wk <= 65535, scale > 0. */
int marship_sam_attn(const void* qkv, int64_t ld, void* O, int64_t ldo, const float* rel_h, const float* rel_w,
                     int nb, int T, int H, int hdp, int Hk, int Wk, float scale, void* stream);
/* The same for windows that are not materialised (window_partition and window_unpartition,
 * image_encoder.py:176-185,243-289, inside the kernel): qkv holds the image's rows; token row r = b*T + t of the windows
 * is row in_rows[r] of qkv (i32 on the device), or - for a negative entry, a window's padding token - pad_row, one row in
 * the layout of a qkv row that holds what the projection makes of a zero input (its bias); the result of token row r is
 * written to row out_rows[r] of O, or dropped for a negative entry.  Either map may be null (rows in place).  The caller
 * answers for entries inside qkv and O. */
int marship_sam_attn_rows(const void* qkv, int64_t ld, void* O, int64_t ldo, const float* rel_h, const float* rel_w,
                          int nb, int T, int H, int hdp, int Hk, int Wk, float scale, const int32_t* in_rows,
                          const void* pad_row, const int32_t* out_rows, void* stream);
/* P[row, k] = softmax_k(S[row, k] + rel_h[row, k / Wk] + rel_w[row, k % Wk]) (image_encoder.py:357-359,236;
 * without the bias terms: transformer.py:231-233), zeros in columns T..ldp-1.  Rows up to 16384 long. */
int marship_softmax_bias_rows(const float* S, int64_t lds, void* P, int64_t ldp, int64_t nrows, int T,
                              const float* rel_h, const float* rel_w, int Hk, int Wk, int out_dtype,
                              void* stream);
/* Random-Fourier position code (prompt_encoder.py:185-192): coords [n, 2] in [0,1], G [2, F],
 * out[i] = [sin(2 pi (2c-1).G), cos(...)] (2F values, row stride ldo).  With labels (int32 [n], may be
 * NULL) the point-prompt rule of :86-90 is applied from table [3, 2F] = {not_a_point, negative,
 * positive}: label -1 -> table[0]; 0 -> code + table[1]; 1 -> code + table[2]. */
int marship_fourier_pe(const float* coords, const float* G, int F, const int32_t* labels,
                       const float* table, float* out, int64_t ldo, int64_t n, void* stream);
/* out[r, :] = a[r, :] + b[r % b_rows, :] (b may be NULL), f32 in, f32 or bf16 out: the "queries + pe"
 * operands of the two-way transformer (transformer.py:158-178). */
int marship_add_cast(const float* a, const float* b, int64_t b_rows, void* out, int64_t rows, int D,
                     int out_dtype, void* stream);
/* Elementwise activation with a type change (act 0 none, 1 GELU(erf), 3 ReLU): mask_decoder.py:56. */
int marship_act_cast(const float* in, void* out, int64_t n, int act, int out_dtype, void* stream);
/* One pass over the rows of the decoder's image-side stream (transformer.py:151-183: norm4 of a layer, then the
 * `keys + key_pe` and `keys` operands of the next attention): y = LayerNorm(x) (w, b non-null; stored as f32 in `y` if
 * that is non-null, which may be x itself) or y = x; c0 = y + add[r % add_rows] and c1 = y in out_dtype (either may be
 * null).  x f32 [rows, D] contiguous, D a multiple of 4 up to 2048. */
int marship_stream_copies(const float* x, const float* w, const float* b, float eps, float* y, const float* add,
                          int64_t add_rows, void* c0, void* c1, int64_t rows, int D, int out_dtype, void* stream);
/* LayerNorm over rows of D <= 256 channels, then an activation (0 none, 1 erf GELU), f32 -> out_dtype with row stride
 * ldo: LayerNorm2d + GELU of the output upscaling (mask_decoder.py:53-59, common.py:31-43). */
int marship_layernorm_act_rows(const float* x, const float* w, const float* b, float eps, void* out, int64_t ldo,
                               int64_t rows, int D, int act, int out_dtype, void* stream);
/* The output upscaling at the shipped decoder width (256 -> 64 -> 32 channels), bf16 operands, without the f32 tensors
 * between its stages (mask_decoder.py:53-59,136-143).
 * marship_sam_upscale_ln_gelu: ConvTranspose2d(256, 64, 2, 2) + LayerNorm2d(eps) + GELU.  x bf16 [prompts*G*G, ldx] token-major;
 *   w0 bf16 [32][256][8]: entry (kc, r) holds W[r][8kc .. 8kc+7] with W[(dy, dx, c)][k] = weight[k, c, dy, dx]; prm f32 [3][64] =
 *   convolution bias, LayerNorm weight, LayerNorm bias; out bf16 [prompts*4*G*G, ldo], pixels row-major on the 2G x 2G grid.
 * marship_sam_upscale_masks: ConvTranspose2d(64, 32, 2, 2) + GELU + (hyper_in @ upscaled).  u = the result above
 *   (G2 = 2G, a multiple of 32); w3 bf16 [(dy, dx, c) = 128][64]; bias f32 [32]; hyper f32 [prompts*4, ldh]; masks f32
 *   [prompts, 4, 2*G2, 2*G2].  GELU in both is the erf form evaluated to 1.5e-7 absolute. */
int marship_sam_upscale_ln_gelu(const void* x, int64_t ldx, const void* w0, const float* prm, float eps, void* out,
                                int64_t ldo, int64_t prompts, int G, int x_tiled, void* stream);
int marship_sam_upscale_masks(const void* u, int64_t ldu, const void* w3, const float* bias, const float* hyper,
                              int64_t ldh, float* masks, int64_t prompts, int G2, void* stream);
/* The image -> token half of a two-way transformer layer at the shipped width (transformer.py:176-183, 218-240: 256 channels,
 * eight heads of 16 dims, at most eight prompt tokens; bf16 operands) as one pass over the image-side stream:
 *   q = (keys + key_pe) Wq^T + bq;  o = attention(q, k, v) per head with scale 1/4;  y = LayerNorm(resid + o Wo^T + bo) (norm4);
 *   c1 = bf16(y), and y itself as f32 when `y` is not null.
 * key_pe is the same for every prompt, so its share of q comes from a table: qbias f32 [Ni, 128] = key_pe Wq^T + bq, and
 * qin is the bf16 copy of keys alone.
 * marship_sam_i2t_kv_operands lays the prompts' projected tokens k, v (bf16 [prompts*Tk, ld]) out as the matrix-instruction
 * operands the pass reads: kv_ops bf16 [prompts][8][64][2][8].
 * marship_sam_i2t_block: qin bf16 [prompts*Ni, ldq] (or [Ni, ldq] with q_shared: every prompt's stream is still the same);
 * wq bf16 [32][128][8]: entry (kc, r) = Wq[r][8kc .. 8kc+7]; wo bf16 [16][256][8]: entry (2h + g, r)[e] = Wo[r][16h + 8(e/4) + 4g + e%4];
 * prm f32 = bo [256], LayerNorm weight [256], bias [256]; resid f32 [prompts*Ni, 256] (or [Ni, 256] with resid_shared);
 * Ni a multiple of 32.  qin / c1 and resid / y may be the same buffers.
 * `tiled` (bit 0: qin, bit 1: resid, bit 2: y and c1): the tensor is stored in the order this kernel's lanes hold it instead of
 * row-major: for tile t of 32 tokens, lane = 32 g + n: bf16: 16-byte entry (16 t + ks) 64 + lane = channels 16 ks + 8 g .. + 7 of
 * token 32 t + n; f32: entry ((8 t + mb) 4 + i) 64 + lane = channels 32 mb + 8 i + 4 g .. + 3.  Every load and store is then 1 KB
 * contiguous a wave.  marship_sam_t2i_block and marship_sam_upscale_ln_gelu read the bf16 form with x_tiled = 1. */
int marship_sam_i2t_kv_operands(const void* k, int64_t ldk, const void* v, int64_t ldv, int prompts, int Tk, void* kv_ops,
                                void* stream);
int marship_sam_i2t_block(const void* qin, int64_t ldq, int q_shared, const void* wq, const void* wo, const float* qbias,
                          const float* prm, const void* kv_ops, int Tk, const float* resid, int resid_shared, int prompts,
                          int Ni, float eps, float* y, void* c1, int tiled, void* stream);
/* The token -> image attention of a two-way transformer layer at the shipped width (transformer.py:160-166, 192-197, 218-240:
 * 256 channels, eight heads of 16 dims, at most eight prompt tokens; bf16 operands) with the key and value projections of the
 * image-side stream made inside the kernel:  k = (x + key_pe) Wk^T + bk, v = x Wv^T + bv, out = softmax(q k^T / 4) v per head.
 * marship_sam_t2i_q_operands lays the prompts' projected queries q (bf16 [prompts*Tq, ldq]) out as the matrix-instruction
 * operands the pass reads: q_ops bf16 [prompts][8][64][8].
 * marship_sam_t2i_block: x bf16 [prompts*Ni, ldx] (the bf16 copy of keys); wk, wv bf16 [32][128][8]: entry (kc, r) = W[r][8kc .. 8kc+7];
 * kpe_ops bf16 [Ni/32][8][64][8]: key_pe Wk^T in operand order: entry (tile, 2 pair + s, lane (token n, half g))[e] =
 * (key_pe Wk^T)[32 tile + n][16 (2 pair + s) + 8 (e/4) + 4 g + e%4] (bk may be included or not: a term that is the same for every
 * image token cancels in the softmax); bv f32 [128]; out bf16 [prompts*Tq, ldo] (columns 0 .. 127).  Ni a multiple of 32.
 * x_shared: x is [Ni, ldx], one stream for every prompt (the first layer without mask prompts). */
int marship_sam_t2i_q_operands(const void* q, int64_t ldq, int prompts, int Tq, void* q_ops, void* stream);
int marship_sam_t2i_block(const void* x, int64_t ldx, const void* wk, const void* wv, const void* kpe_ops, const float* bv,
                          const void* q_ops, int prompts, int Ni, int Tq, void* out, int64_t ldo, int x_tiled, int x_shared,
                          void* stream);
/* masks[b, m, p] = sum_c hyper[b, m, c] * up[b*P + p, c], four masks a prompt, C <= 64 channels (mask_decoder.py:141-143:
 * hyper_in @ upscaled_embedding); hyper f32 [B*4, ldh], up f32 [B*P, ldu], masks f32 [B, 4, P]. */
int marship_hyper_product(const float* hyper, int64_t ldh, const float* up, int64_t ldu, float* masks, int B,
                          int64_t P, int C, void* stream);
/* Per mask of logits [n, H, W] (utils/amg.py:155-175, 310-353): out[k] = {#(> thr+off), #(> thr-off),
 * x0, y0, x1, y1 of (> thr)} (int32 x 6; the box of an empty mask is 0,0,0,0). */
int marship_mask_stats(const float* masks, int n, int H, int W, float thr, float off, int32_t* out,
                       void* stream);
/* One axis of a composed resampling as a banded table: output o reads inputs lo[o] .. lo[o] + size[o] - 1 with weights
 * w[o * kmax + k] (device pointers). */
typedef struct MarsAxisTable { const int32_t* lo; const int32_t* size; const float* w; int32_t kmax; } MarsAxisTable;
/* Sam.postprocess_masks (segment_anything/modeling/sam.py:134-162) fused with what the automatic mask generator does
 * with its result (automatic_mask_generator.py:417-441, utils/amg.py:155-175): the image-size logits are evaluated from the
 * low-resolution ones [.., L, L] a band of `band` output rows at a time (first along x with tx [W entries], then along y
 * with ty [H entries], the order and sums of marship_resample_axis) and never stored.  Plane p of the input starts at
 * low + (p / nm) * stride_b + (p % nm) * stride_c.  max_rows = the most source rows any band reads (the caller computes it
 * from ty; max_rows * (L + W) floats must fit in 64 KB of LDS).
 * marship_resample_mask_stats: stats[p] of marship_mask_stats for planes 0 .. n-1; with `bits` (f32 [n, H, W], may be NULL) also
 *   what marship_resample_threshold writes for every plane, in the same pass (for callers whose filters keep every mask).
 * marship_resample_threshold: out[i] f32 [H, W] = (logits of plane sel[i] (or i when sel is NULL) > thr) as 0 / 1. */
int marship_resample_mask_stats(const float* low, int64_t stride_b, int64_t stride_c, int nm, int n, int L, int H, int W,
                                const MarsAxisTable* tx, const MarsAxisTable* ty, int band, int max_rows, float thr,
                                float off, int32_t* stats, float* bits, void* stream);
int marship_resample_threshold(const float* low, int64_t stride_b, int64_t stride_c, int nm, const int32_t* sel, int nsel,
                               int L, int H, int W, const MarsAxisTable* tx, const MarsAxisTable* ty, int band,
                               int max_rows, float thr, float* out, void* stream);
/* Dense embedding of mask prompts (segment_anything/modeling/prompt_encoder.py:51-58, 102-105): masks f32
 * [B, 1, 4G, 4G] -> out f32 [B, G*G, C] (token-major).  w1 [m4,1,2,2], w2 [m,m4,2,2], w3 [C,m] with their biases,
 * g / h = LayerNorm2d weight / bias. */
int marship_mask_embed(const float* masks, int B, int G, int m4, int m, int C, const float* w1, const float* b1,
                       const float* g1, const float* h1, const float* w2, const float* b2, const float* g2,
                       const float* h2, const float* w3, const float* b3, float eps, float* out, void* stream);
/* out = in > thr ? 1 : 0 (automatic_mask_generator.py:440). */
int marship_threshold_masks(const float* in, float thr, float* out, int64_t n, void* stream);
/* Greedy box NMS for one category (torchvision batched_nms as called at automatic_mask_generator.py:370-376):
 * host pointers; keep receives the surviving indices by decreasing score. */
int marshost_box_nms(const float* boxes, const float* scores, int n, float iou_threshold, int32_t* keep,
                     int32_t* nkeep);
/* segment_anything/utils/amg.py:274-298 remove_small_regions (there: cv2.connectedComponentsWithStats, 8-connected),
 * in place on a {0,1} byte mask [h, w].  holes != 0 fills background components smaller than area_thresh, holes == 0
 * removes foreground components smaller than it (the largest stays if all are).  *changed = 1 iff any region was. */
int marshost_remove_small_regions(uint8_t* mask, int h, int w, float area_thresh, int holes, int32_t* changed);
/* scipy.optimize.linear_sum_assignment as the prompt-sampling stage calls it (matcher/Matcher.py:449-450,
 * 471-472, 364-365, 376): w is a host f32 [n, m] matrix with row stride ld; row_ind / col_ind receive
 * min(n, m) pairs ordered by row.  Returns the pair count, negative on error (non-finite entries). */
int marshost_lap(const float* w, int64_t ld, int n, int m, int maximize, int64_t* row_ind, int64_t* col_ind);

/* ---- GradCAM backward pieces ----------------------------------------------------- */
/* LayerNorm backward w.r.t. the input: dx = add + LN'(x)[dy*w].  dy row r is dy + r*dy_stride
 * (dy_stride 0 broadcasts one row); rows < skip_rows get dy = 0. */
int marship_layernorm_bwd(const float* x, const float* w, const float* dy, int64_t dy_stride,
                          float dy_scale, int skip_rows, const float* add, float* dx, int rows, int D,
                          float eps, void* stream);
/* d/dz [z*sigmoid(1.702 z)] applied to dh (clip/model.py:274-276). */
int marship_quickgelu_bwd(const float* dh, const float* pre, float* out, int64_t n, void* stream);
/* dS = P * (dP - rowsum(dP*P)) * scale, per row (softmax backward). */
int marship_softmax_bwd(const float* P, const float* dP, float* dS, int64_t ld, int nb, int T,
                        float scale, void* stream);
/* out[c] = mean over rows r0..r1-1 of x[r, c]. */
int marship_col_mean(const float* x, int64_t ld, int r0, int r1, int D, float* out, void* stream);
/* Head of forward_last_layer and its gradient (clip/model.py:506-521): from f = mean patch token
 * of ln_post, logits = softmax(exp(logit_scale) * norm(f@proj) . norm(text)^T); df = d logits[0]/df.
 * proj f32 [D,E], text f32 [2,E]. */
int marship_cam_head(const float* f, const float* proj, const float* text, int D, int E,
                     float logit_scale_exp, float* logits, float* df, void* stream);
/* cam[t] = relu(sum_c w[c]*act[t,c]) for rows r0.. (grad_cam.py:16-23; base_cam.py:55-60,154). */
int marship_cam_rows(const float* act, int64_t ld, const float* w, int r0, int rows, int D, float* cam,
                     void* stream);

/* ---- host routines --------------------------------------------------------------- */
/* Exact EMD scores 1 - emd2(1/n, 1/m, C[fg rows][:, mask cols]) with C = (1-S)/2
 * (FilteringMergingModule.py:142-169; matcher/Matcher.py:1188-1193; POT 0.9.4 ot.emd2).
 * S host f32 [R, ldS] (the similarity matrix, or the cost matrix itself when is_cost != 0);
 * fg u8 [R]; bits u8 [N, P]; out f64 [N].  Network simplex, one problem per worker thread. */
int marshost_emd_scores(const float* S, int64_t ldS, int R, int P, const uint8_t* fg,
                        const uint8_t* bits, int N, int is_cost, double* out, int threads);
/* Same solver on an explicit cost matrix (f64 [n, m]). */
double marshost_emd2_uniform(const double* C, int n, int m);

#ifdef __cplusplus
}
#endif
#endif /* MARS_HIP_H */
