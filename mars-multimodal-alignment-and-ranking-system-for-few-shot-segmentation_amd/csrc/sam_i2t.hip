// The image -> token half of a layer of the mask decoder's two-way transformer as one pass over the image-side stream
// (segment_anything/modeling/transformer.py:176-183 with the Attention of :218-240, 256 channels, eight heads of 16
// dims after the downsampling by two, at most eight prompt tokens, bf16 operands):
//
//     q    = (keys + key_pe) Wq^T + bq                         [tokens, 128]
//     o    = softmax(q_h k_h^T / 4) v_h   per head h           [tokens, 128]      (k, v: the prompt's projected tokens)
//     keys = LayerNorm(keys + o Wo^T + bo)                      [tokens, 256]  f32, and its bf16 copy for the next products.
//
// key_pe is the same for every prompt and the projection is linear: q = keys Wq^T + (key_pe Wq^T + bq), and the second
// term is a table of [image tokens, 128] made once.  So the pass reads the bf16 copy of keys alone, and no copy of
// keys + key_pe is ever made (the token -> image attention of the next layer takes its keys the same way).
//
// As separate launches this is a projection, an attention kernel, a projection with residual and a normalisation pass:
// the 4.3 GB f32 stream of 1024 prompts is read and written twice and 4 GB of q and o go through memory in between.
// Here a wave owns 32 tokens from the first load to the last store:
//
//   * Wq and Wo (64 KB each as bf16) live in LDS for the life of the workgroup, in runs of eight K values, run-major, so
//     a wave reads an A operand of v_mfma_f32_32x32x16_bf16 as 64 consecutive 16-byte entries.
//   * Every product has the tokens on the N side (lane = token) and the feature index on the M side.  The result of one
//     product is then, register for register, the B operand of the next once the K index of that next product is
//     permuted to the order in which a lane holds its results: slot e of half-wave g  <->  dim 8 (e / 4) + 4 g + e % 4
//     of a 16-dim head.  Wo is stored with its K index in that order, and so are the prompt's keys; nothing is
//     transposed or exchanged between the four products (q, scores, o, output).
//   * The prompt's k (as A operand: key on M) and v^T (as A operand: dim on M, key on K) are prepared once per prompt
//     by i2t_kv_operands_kernel in exactly the register layout, 32 bytes a lane and head.
//   * Scores of a token and head are four registers in each half-wave (keys 0-3 and 4-7): the softmax needs one
//     v_permlane32_swap for the maximum and one for the sum.  LayerNorm over the 256 channels is 128 registers of a
//     lane and of its partner: again one swap per statistic.
#include "common.h"
#include "../../include/mars_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v2i32 __attribute__((ext_vector_type(2)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

constexpr int C = 256, W = 128, HEADS = 8, HD = 16;
constexpr int LDS_WQ = W * C * 2, LDS_WO = C * W * 2;        // bytes
constexpr int NPRM = 3 * C;                                   // bo, LayerNorm weight, LayerNorm bias

__device__ __forceinline__ float halves_sum(float a) {
    const unsigned u = __builtin_bit_cast(unsigned, a);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float halves_max(float a) {
    const unsigned u = __builtin_bit_cast(unsigned, a);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ int head_dim_of_slot(int half, int e) { return 8 * (e >> 2) + 4 * half + (e & 3); }

// ops[b][h][lane] = { k operand, v^T operand } (two bf16x8): k: lane (key m, half g) holds k[b, m, 16 h + dim(g, e)], zero
// for m >= Tk; v^T: lane (dim m < 16, half g) holds v[b, 4 g + e, 16 h + m] in slots e < 4 (zero past Tk), zero elsewhere.
__global__ __launch_bounds__(512) void i2t_kv_operands_kernel(const __bf16* __restrict__ k, int64_t ldk, const __bf16* __restrict__ v, int64_t ldv,
                                                              int Tk, bf16x8* __restrict__ ops) {
    const int b = blockIdx.x, h = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 31, half = lane >> 5;
    bf16x8 ka, va;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ka[e] = m < Tk ? k[((int64_t)b * Tk + m) * ldk + h * HD + head_dim_of_slot(half, e)] : (__bf16)0.f;
        const int key = 4 * half + e;
        va[e] = (e < 4 && m < HD && key < Tk) ? v[((int64_t)b * Tk + key) * ldv + h * HD + m] : (__bf16)0.f;
    }
    bf16x8* o = ops + (((int64_t)b * HEADS + h) * 64 + lane) * 2;
    o[0] = ka;
    o[1] = va;
}

// qin bf16 [q_rows, ldq] (q_rows = tokens, or Ni when every prompt shares it); wq bf16 [32][128][8] and wo bf16 [16][256][8]
// in runs of eight K, run-major (wo with its K index in slot order); qbias f32 [Ni, 128] = key_pe Wq^T + bq; prm f32
// [3 * 256]; kv as above; resid f32 [tokens or Ni, 256]; y f32 [tokens, 256] or null; c1 = bf16(y).  Eight waves a workgroup,
// wave w of workgroup g owns the tiles [(8 g + w) per, (8 g + w + 1) per) of 32 tokens.
// qin / c1 and resid / y may be the same buffers: a tile reads its rows before it writes them and no other tile touches them.
// All loads of a tile come before its first store: a wait for a load behind stores would also wait for those stores.
// `tiled` (bit 0: qin, bit 1: resid, bit 2: y and c1): that tensor is not row-major but in the order the lanes of this
// kernel hold it - for tile t of 32 tokens, bf16: 16-byte entry (16 t + ks) 64 + lane = the eight channels 16 ks + 8 half of
// token 32 t + n; f32: entry ((8 t + mb) 4 + i) 64 + lane = the four channels 32 mb + 8 i + 4 half.  Row-major, a store of
// this kernel is 32 bytes a token row (partial lines; the f32 stream went out at under 3 TB/s); tiled, every load and
// store is 1 KB contiguous a wave.  The consumers that read the stream with the same lane mapping (this kernel in the
// next layer, the token -> image block, the first upscaling kernel) take the tiled form; a tiled qin / resid is never
// shared between prompts.
__global__ __launch_bounds__(512) void i2t_block_kernel(const __bf16* qin, int64_t ldq, int q_shared, const __bf16* __restrict__ wq,
                                                        const __bf16* __restrict__ wo, const float* __restrict__ qbias, const float* __restrict__ prm,
                                                        const bf16x8* __restrict__ kv, int Tk, const float* resid, int resid_shared, int Ni,
                                                        float eps, float* y, __bf16* c1, int64_t ntiles, int64_t per, int tiled) {
    __shared__ __align__(16) unsigned char smem[LDS_WQ + LDS_WO + NPRM * 4];
    bf16x8* sWq = (bf16x8*)smem;                        // [32][128]
    bf16x8* sWo = (bf16x8*)(smem + LDS_WQ);             // [16][256]
    float* sP = (float*)(smem + LDS_WQ + LDS_WO);       // bo [256], ln w [256], ln b [256]
    for (int i = threadIdx.x; i < LDS_WQ / 16; i += 512) sWq[i] = ((const bf16x8*)wq)[i];
    for (int i = threadIdx.x; i < LDS_WO / 16; i += 512) sWo[i] = ((const bf16x8*)wo)[i];
    for (int i = threadIdx.x; i < NPRM; i += 512) sP[i] = prm[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, half = lane >> 5;
    // keys 4 half + j past Tk never win the maximum and weigh nothing
    float kmask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kmask[j] = 4 * half + j < Tk ? 0.f : -INFINITY;
    const int64_t t0 = ((int64_t)blockIdx.x * 8 + wave) * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
#pragma unroll 1
    for (int64_t t = t0; t < t1; ++t) {
        // Everything read from LDS is the same for every tile; left to itself the compiler reads all of it (a few hundred
        // registers' worth) in front of the loop and spills it.  An offset of zero that passes through an opaque statement
        // once a tile keeps the reads inside the loop (an offset, not the pointer: the reads must stay LDS reads).
        int opaque = 0;
        asm volatile("" : "+v"(opaque));
        const bf16x8 *tWq = sWq + opaque, *tWo = sWo + opaque;
        const float* tP = sP + opaque;
        const float *sBo = tP, *sLw = tP + C, *sLb = tP + 2 * C;
        const int64_t tok = t * 32 + n;
        const int64_t b = tok / Ni;
        const int pl = (int)(tok - b * Ni);
        // ---- q^T = Wq x^T: four blocks of 32 dims = two heads each
        bf16x8 ob[HEADS];                                // per head: first q, then o, as B operands
        {
            const __bf16* xp = qin + (q_shared ? (int64_t)pl : tok) * ldq + half * 8;
            const float* qb = qbias + (int64_t)pl * W + half * 4;
            bf16x8 xb[C / 16];
#pragma unroll
            for (int ks = 0; ks < C / 16; ++ks)
                xb[ks] = (tiled & 1) ? ((const bf16x8*)qin)[(t * 16 + ks) * 64 + lane] : *(const bf16x8*)(xp + ks * 16);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                f32x16 acc = (f32x16)(0.f);
                const bf16x8* wp = tWq + half * W + mb * 32 + n;
#pragma unroll
                for (int ks = 0; ks < C / 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[ks * 2 * W], xb[ks], acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 bq = *(const f32x4*)(qb + mb * 32 + i * 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j) ob[2 * mb + (i >> 1)][4 * (i & 1) + j] = (__bf16)(acc[4 * i + j] + bq[j]);
                }
            }
        }
        // ---- per head: scores against the prompt's keys, softmax, o^T = v^T p^T
        const bf16x8* kvp = kv + ((int64_t)b * HEADS * 64 + lane) * 2;
#pragma unroll
        for (int h = 0; h < HEADS; ++h) {
            const bf16x8 ka = kvp[h * 128], va = kvp[h * 128 + 1];
            const f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, ob[h], (f32x16)(0.f), 0, 0, 0);
            // registers 0..3: keys 4 half + j of this lane's token; scores in units of log 2 (1/4 = 1/sqrt(16))
            float sc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[j] = fmaf(s[j], 0.25f * 1.44269504088896340736f, kmask[j]);
            const float mx = halves_max(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])));
            float l = 0.f;
            bf16x8 pb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p = __builtin_amdgcn_exp2f(sc[j] - mx);
                l += p;
                pb[j] = (__bf16)p;
                pb[4 + j] = (__bf16)0.f;
            }
            const float inv = __builtin_amdgcn_rcpf(halves_sum(l));
            const f32x16 o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb, (f32x16)(0.f), 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 8; ++e) ob[h][e] = (__bf16)(o[e] * inv);          // registers 0..7: dims 8 i + 4 half + j
        }
        // ---- keys += o Wo^T + bo, eight blocks of 32 channels; LayerNorm over the 256
        f32x16 acc[C / 32];
#pragma unroll
        for (int mb = 0; mb < C / 32; ++mb) {
            acc[mb] = (f32x16)(0.f);
            const bf16x8* wp = tWo + half * C + mb * 32 + n;
#pragma unroll
            for (int h = 0; h < HEADS; ++h) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[h * 2 * C], ob[h], acc[mb], 0, 0, 0);
        }
        const float* rp = resid + (resid_shared ? (int64_t)pl : tok) * C + half * 4;
        float sum = 0.f;
#pragma unroll
        for (int mb = 0; mb < C / 32; ++mb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 r = (tiled & 2) ? ((const f32x4*)resid)[((t * 8 + mb) * 4 + i) * 64 + lane] : *(const f32x4*)(rp + mb * 32 + i * 8);
                const f32x4 bo = *(const f32x4*)(sBo + mb * 32 + i * 8 + half * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[mb][4 * i + j] += r[j] + bo[j];
                    sum += acc[mb][4 * i + j];
                }
            }
        const float mean = halves_sum(sum) * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int mb = 0; mb < C / 32; ++mb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[mb][e] -= mean;
                sq = fmaf(acc[mb][e], acc[mb][e], sq);
            }
        const float rstd = 1.0f / sqrtf(halves_sum(sq) * (1.0f / C) + eps);
        float* yp = y ? y + tok * C + half * 4 : nullptr;
#pragma unroll
        for (int mb = 0; mb < C / 32; ++mb)
#pragma unroll
            for (int rpair = 0; rpair < 2; ++rpair) {
                bf16x4 a1[2];                            // [u]: group i = 2 rpair + u of eight channels
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = 2 * rpair + u, col = mb * 32 + i * 8;
                    const f32x4 lw = *(const f32x4*)(sLw + col + half * 4), lb = *(const f32x4*)(sLb + col + half * 4);
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = fmaf(acc[mb][4 * i + j] * rstd, lw[j], lb[j]);
                        a1[u][j] = (__bf16)v[j];
                    }
                    if (yp) {
                        if (tiled & 4) ((f32x4*)y)[((t * 8 + mb) * 4 + i) * 64 + lane] = v;
                        else *(f32x4*)(yp + col) = v;
                    }
                }
                // the half-waves trade halves: the lower lane ends with the eight channels of group 2 rpair, the upper with 2 rpair + 1
                const v2i32 ai = __builtin_bit_cast(v2i32, a1[0]), bi = __builtin_bit_cast(v2i32, a1[1]);
                const auto s0 = __builtin_amdgcn_permlane32_swap((unsigned)ai[0], (unsigned)bi[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap((unsigned)ai[1], (unsigned)bi[1], false, false);
                const v4i32 o = {(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
                if (tiled & 4) ((v4i32*)c1)[(t * 16 + 2 * mb + rpair) * 64 + lane] = o;
                else *(v4i32*)(c1 + tok * C + mb * 32 + 8 * (2 * rpair + half)) = o;
            }
    }
}

}  // namespace

extern "C" int marship_sam_i2t_kv_operands(const void* k, int64_t ldk, const void* v, int64_t ldv, int prompts, int Tk, void* kv_ops,
                                           void* stream) {
    MARS_CHECK(k && v && kv_ops && prompts > 0 && Tk > 0 && Tk <= 8, "sam_i2t_kv_operands: %d prompts of %d tokens (1 .. 8)", prompts, Tk);
    MARS_CHECK(ldk >= W && ldv >= W && ((uintptr_t)kv_ops % 16) == 0, "sam_i2t_kv_operands: rows of 128 values, 16-byte aligned result");
    hipLaunchKernelGGL(i2t_kv_operands_kernel, dim3((unsigned)prompts), dim3(512), 0, (hipStream_t)stream, (const __bf16*)k, ldk, (const __bf16*)v, ldv,
                       Tk, (bf16x8*)kv_ops);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_sam_i2t_block(const void* qin, int64_t ldq, int q_shared, const void* wq, const void* wo, const float* qbias,
                                     const float* prm, const void* kv_ops, int Tk, const float* resid, int resid_shared, int prompts, int Ni,
                                     float eps, float* y, void* c1, int tiled, void* stream) {
    MARS_CHECK(qin && wq && wo && qbias && prm && kv_ops && resid && c1, "sam_i2t_block: null pointer");
    MARS_CHECK(prompts > 0 && Ni > 0 && Ni % 32 == 0 && Tk > 0 && Tk <= 8, "sam_i2t_block: %d prompts, %d image tokens (a multiple of 32), %d prompt tokens (1 .. 8)",
               prompts, Ni, Tk);
    MARS_CHECK(ldq >= C && ldq % 8 == 0, "sam_i2t_block: row stride %lld", (long long)ldq);
    MARS_CHECK(tiled >= 0 && tiled <= 7 && !((tiled & 1) && (q_shared || ldq != C)) && !((tiled & 2) && resid_shared),
               "sam_i2t_block: tiled %d (a tiled input is per prompt, 256 values a token)", tiled);
    MARS_CHECK((((uintptr_t)qin | (uintptr_t)wq | (uintptr_t)wo | (uintptr_t)prm | (uintptr_t)kv_ops | (uintptr_t)resid | (uintptr_t)qbias | (uintptr_t)y |
                 (uintptr_t)c1) % 16) == 0, "sam_i2t_block: 16-byte alignment");
    static const int cus = [] { int dev = 0, c = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256; return c; }();
    const int64_t ntiles = (int64_t)prompts * Ni / 32;
    int64_t waves = (int64_t)cus * 8;
    if (waves > ntiles) waves = ntiles;
    const int64_t per = (ntiles + waves - 1) / waves;
    const unsigned grid = (unsigned)((ntiles + per * 8 - 1) / (per * 8));
    hipLaunchKernelGGL(i2t_block_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, (const __bf16*)qin, ldq, q_shared, (const __bf16*)wq,
                       (const __bf16*)wo, qbias, prm, (const bf16x8*)kv_ops, Tk, resid, resid_shared, Ni, eps, y, (__bf16*)c1, ntiles, per, tiled);
    MARS_LAUNCH_CHECK();
    return 0;
}
