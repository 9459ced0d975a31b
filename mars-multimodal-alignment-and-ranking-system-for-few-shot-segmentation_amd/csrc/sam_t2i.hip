// The token -> image attention of the mask decoder's two-way transformer (segment_anything/modeling/transformer.py:
// 160-166 and 192-197 with the Attention of :218-240; 256 channels, eight heads of 16 dims, at most eight prompt tokens,
// bf16 operands) with the key and value projections of the image-side stream made on the fly:
//
//     k = (keys + key_pe) Wk^T + bk,   v = keys Wv^T + bv              [image tokens, 128] each, per prompt
//     o = softmax(q_h k_h^T / 4) v_h   per head h                       [prompt tokens, 128]
//
// As separate launches the two projections write 2 GB for 1024 prompts and the attention kernel reads them back.  Here
// a wave walks 32 image tokens at a time, keeps the bf16 stream tile as matrix-instruction operands in registers and
// never stores k or v:
//
//   * key_pe enters through linearity: q.(Wk (x + pe) + bk) = q.(Wk x) + q.(Wk pe) + q.bk.  The last term is the same for
//     every image token and cancels in the softmax; the middle one is one more pair of matrix instructions against a
//     table (Wk pe, bf16, already in operand order) that belongs to the model.
//   * Two heads share one 32-column score tile: columns 0..15 are the queries of the even head, 16..31 of the odd one,
//     the K index runs over the 32 dims of both, and the query operand is zero where head and column do not match.
//   * k^T = Wk x^T has the image tokens on the N side: its accumulators are, in place, the A operand (token on M) of the
//     score product S^T = k q^T.  v = x Wv^T has the tokens on the M side: its accumulators are the A operand (dim on M,
//     token on K) of o^T += v^T p^T, and the probabilities, which a lane holds for one (head, query) column and sixteen
//     tokens, are its B operand as they stand.  Nothing is transposed.
//   * Online softmax with one state per lane (a column = one head and query), maxima and sums of the two half-waves
//     meeting through v_permlane32_swap.  Two waves share a prompt (half of its image tokens each) and merge their
//     states through LDS at the end.
#include "common.h"
#include "../../include/mars_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C = 256, W = 128, HD = 16;
constexpr int LDS_W = W * C * 2;                 // bytes of one projection weight as bf16
constexpr float LOG2E = 1.44269504088896340736f;

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

// qops[b][pair][s][lane] (bf16x8): the query operand of the score product of heads (2 pair, 2 pair + 1), K step s = the
// dims of head 2 pair + s.  Lane (column n, half g): column n < 16 is query n of the even head, n >= 16 query n - 16 of the
// odd head; slot e holds q[b, query, 16 head + dim(g, e)] / 4 where the column's head is the K step's head, zero elsewhere
// (and for queries past Tq).
__global__ __launch_bounds__(512) void t2i_q_operands_kernel(const __bf16* __restrict__ q, int64_t ldq, int Tq, bf16x8* __restrict__ qops) {
    const int b = blockIdx.x, ps = threadIdx.x >> 6, pair = ps >> 1, s = ps & 1, lane = threadIdx.x & 63, n = lane & 31, half = lane >> 5;
    const int query = n & 15, odd = n >> 4;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        v[e] = (odd == s && query < Tq) ? (__bf16)(0.25f * (float)q[((int64_t)b * Tq + query) * ldq + (2 * pair + s) * HD + head_dim_of_slot(half, e)])
                                        : (__bf16)0.f;
    qops[((int64_t)b * 8 + ps) * 64 + lane] = v;
}

// x bf16 [prompts * Ni, ldx]; wk, wv bf16 [32][128][8] (runs of eight K, run-major); kpe bf16x8 [Ni / 32][8][64]: (Wk key_pe) of a
// tile as the A operands of the score product ([pair][s] like qops, lane = token); bv f32 [128]; out bf16 [prompts * Tq, ldo].
// Workgroup g: prompts 4 g .. 4 g + 3, waves 2 i and 2 i + 1 the two halves of the image tokens of prompt 4 g + i.
__global__ __launch_bounds__(512) void t2i_block_kernel(const __bf16* __restrict__ x, int64_t ldx, const __bf16* __restrict__ wk,
                                                        const __bf16* __restrict__ wv, const bf16x8* __restrict__ kpe, const float* __restrict__ bv,
                                                        const bf16x8* __restrict__ qops, int prompts, int Ni, int Tq, __bf16* __restrict__ out,
                                                        int64_t ldo, int x_tiled, int x_shared) {
    __shared__ __align__(16) unsigned char smem[2 * LDS_W];
    bf16x8* sWk = (bf16x8*)smem;                       // [32][128]
    bf16x8* sWv = (bf16x8*)(smem + LDS_W);             // [32][128]
    for (int i = threadIdx.x; i < LDS_W / 16; i += 512) {
        sWk[i] = ((const bf16x8*)wk)[i];
        sWv[i] = ((const bf16x8*)wv)[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, half = lane >> 5;
    const int64_t b = (int64_t)blockIdx.x * 4 + (wave >> 1);
    const int ntile = Ni / 32, split = (ntile + 1) / 2;
    const int ta = (wave & 1) ? split : 0, tb = (wave & 1) ? ntile : split;
    const bool live = b < prompts;

    f32x16 o[4];                                        // o^T of a head pair: [dim of both heads][column]
    float m[4], l[4];                                   // per column (this lane's), in units of log 2; l: this half-wave's share
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        o[p] = (f32x16)(0.f);
        m[p] = -INFINITY;
        l[p] = 0.f;
    }
    if (live) {
        const bf16x8* qp = qops + (b * 8) * 64 + lane;
#pragma unroll 1
        for (int t = ta; t < tb; ++t) {
            // (an offset of zero through an opaque statement once a tile: the LDS reads below must not be hoisted out of the loop)
            int opaque = 0;
            asm volatile("" : "+v"(opaque));
            const bf16x8 *tWk = sWk + opaque, *tWv = sWv + opaque;
            const __bf16* xp = x + ((x_shared ? 0 : b * Ni) + (int64_t)t * 32 + n) * ldx + half * 8;      // (x_shared: one stream for every prompt)
            bf16x8 xb[C / 16];
#pragma unroll
            for (int ks = 0; ks < C / 16; ++ks)         // (x_tiled: the order sam_i2t_block leaves it in, 1 KB a load)
                xb[ks] = x_tiled ? ((const bf16x8*)x)[((b * ntile + t) * 16 + ks) * 64 + lane] : *(const bf16x8*)(xp + ks * 16);
            const bf16x8* kpt = kpe + ((int64_t)t * 8) * 64 + lane;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // k^T of the pair's 32 dims x 32 tokens; registers 8 s .. 8 s + 7 are head 2 p + s in slot order
                f32x16 kt = (f32x16)(0.f);
                const bf16x8* wp = tWk + half * W + p * 32 + n;
#pragma unroll
                for (int ks = 0; ks < C / 16; ++ks) kt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[ks * 2 * W], xb[ks], kt, 0, 0, 0);
                f32x16 st = (f32x16)(0.f);                 // S^T[token][column]: this lane's column, tokens 8 i + 4 half + j
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 ka;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ka[e] = (__bf16)kt[8 * s + e];
                    const bf16x8 qb = qp[(2 * p + s) * 64];
                    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qb, st, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kpt[(2 * p + s) * 64], qb, st, 0, 0, 0);
                }
                float mx = st[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
                mx = halves_max(mx) * LOG2E;
                const float mn = fmaxf(m[p], mx);
                const float alpha = __builtin_amdgcn_exp2f(m[p] - mn);      // (first tile: m = -inf, alpha = 0, o = 0)
                m[p] = mn;
                float ls = 0.f;
                bf16x8 pb[2];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(fmaf(st[r], LOG2E, -mn));
                    ls += pr;
                    pb[r >> 3][r & 7] = (__bf16)pr;
                }
                l[p] = fmaf(l[p], alpha, ls);
                if (!__all(alpha == 1.f)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[p][r] *= alpha;
                }
                // v of the 32 tokens x the pair's 32 dims; registers 8 ks .. 8 ks + 7 are the tokens of K step ks
                f32x16 vt = (f32x16)(0.f);
                const bf16x8* vp = tWv + half * W + p * 32 + n;
#pragma unroll
                for (int ks = 0; ks < C / 16; ++ks) vt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb[ks], vp[ks * 2 * W], vt, 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 va;
#pragma unroll
                    for (int e = 0; e < 8; ++e) va[e] = (__bf16)vt[8 * ks + e];
                    o[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb[ks], o[p], 0, 0, 0);
                }
            }
        }
    }
    // ---- the two waves of a prompt merge through LDS (the weights are no longer needed), the even wave finishes
    __syncthreads();
    float* sS = (float*)smem;                           // [4 prompts][4 pairs][64 lanes][10]: m, l, eight o values
    const int odd = n >> 4;                             // this lane's column belongs to head 2 p + odd: its o registers 8 odd .. 8 odd + 7
#pragma unroll
    for (int p = 0; p < 4; ++p) l[p] = halves_sum(l[p]);
    if (wave & 1) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float* d = sS + ((((wave >> 1) * 4 + p) * 64) + lane) * 10;
            d[0] = m[p];
            d[1] = l[p];
#pragma unroll
            for (int e = 0; e < 8; ++e) d[2 + e] = odd ? o[p][8 + e] : o[p][e];
        }
    }
    __syncthreads();
    if ((wave & 1) || !live) return;
    const int query = n & 15;
    if (query >= Tq) return;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float* d = sS + ((((wave >> 1) * 4 + p) * 64) + lane) * 10;
        const float m1 = d[0], l1 = d[1];
        const float mn = fmaxf(m[p], m1);
        const float a0 = m[p] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m[p] - mn), a1 = m1 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m1 - mn);
        const float inv = 1.0f / (l[p] * a0 + l1 * a1);
        const int head = 2 * p + odd;
        __bf16* op = out + (b * Tq + query) * ldo + head * HD;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x4 r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * i + j;
                const float mine = odd ? o[p][8 + e] : o[p][e];
                r[j] = (__bf16)((mine * a0 + d[2 + e] * a1) * inv + bv[head * HD + 8 * i + 4 * half + j]);
            }
            *(bf16x4*)(op + 8 * i + 4 * half) = r;
        }
    }
}

}  // namespace

extern "C" int marship_sam_t2i_q_operands(const void* q, int64_t ldq, int prompts, int Tq, void* q_ops, void* stream) {
    MARS_CHECK(q && q_ops && prompts > 0 && Tq > 0 && Tq <= 8, "sam_t2i_q_operands: %d prompts of %d tokens (1 .. 8)", prompts, Tq);
    MARS_CHECK(ldq >= W && ((uintptr_t)q_ops % 16) == 0, "sam_t2i_q_operands: rows of 128 values, 16-byte aligned result");
    hipLaunchKernelGGL(t2i_q_operands_kernel, dim3((unsigned)prompts), dim3(512), 0, (hipStream_t)stream, (const __bf16*)q, ldq, Tq, (bf16x8*)q_ops);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_sam_t2i_block(const void* x, int64_t ldx, const void* wk, const void* wv, const void* kpe_ops, const float* bv,
                                     const void* q_ops, int prompts, int Ni, int Tq, void* out, int64_t ldo, int x_tiled, int x_shared, void* stream) {
    MARS_CHECK(x && wk && wv && kpe_ops && bv && q_ops && out, "sam_t2i_block: null pointer");
    MARS_CHECK(prompts > 0 && Ni > 0 && Ni % 32 == 0 && Tq > 0 && Tq <= 8, "sam_t2i_block: %d prompts, %d image tokens (a multiple of 32), %d prompt tokens (1 .. 8)",
               prompts, Ni, Tq);
    MARS_CHECK(ldx >= C && ldx % 8 == 0 && ldo >= W && ldo % 4 == 0, "sam_t2i_block: row strides %lld / %lld", (long long)ldx, (long long)ldo);
    MARS_CHECK(!x_tiled || (ldx == C && !x_shared), "sam_t2i_block: a tiled x has 256 values a token and is per prompt");
    MARS_CHECK((((uintptr_t)x | (uintptr_t)wk | (uintptr_t)wv | (uintptr_t)kpe_ops | (uintptr_t)q_ops) % 16) == 0 && ((uintptr_t)out % 8) == 0 &&
                   ((uintptr_t)bv % 4) == 0, "sam_t2i_block: alignment");
    hipLaunchKernelGGL(t2i_block_kernel, dim3((unsigned)((prompts + 3) / 4)), dim3(512), 0, (hipStream_t)stream, (const __bf16*)x, ldx, (const __bf16*)wk,
                       (const __bf16*)wv, (const bf16x8*)kpe_ops, bv, (const bf16x8*)q_ops, prompts, Ni, Tq, (__bf16*)out, ldo, x_tiled, x_shared);
    MARS_LAUNCH_CHECK();
    return 0;
}
