// Attention with 8-bit (OCP e4m3) operands for 64-wide heads: softmax(q k^T / 8) v with q, k, v and the
// probabilities quantised to fp8 and both contractions on v_mfma_f32_16x16x32_fp8_fp8 (f32 accumulation, f32
// softmax).  Same arithmetic as dinov2/layers/attention.py:54-60 and alpha_clip/model.py:233-246 up to the
// quantisation; the configuration that asks for it trades precision, not speed: with 64-wide heads the kernel is
// bound by the exponentials, as the bf16 one is.
//
// Three launches: `amax_qkv_kernel` finds the largest magnitude of every (image, head) slice of q, k and v;
// `quant_qkv_fp8_kernel` scales the slices by 448 / amax and writes q8 / k8 [B*H, Tp, 64] and v transposed, vt8 [B*H, 64, Tp] (rows beyond T are zero, Tp a multiple of 32);
// `flash_attn_fp8_kernel` gives every wave 64 queries and walks the keys 32 at a time.  The scores are formed
// transposed (S^T = K Q^T: keys are MFMA rows, a query is a lane column), so a lane holds, for its query, keys
// 4g..4g+3 of both 16-key halves of the tile - exactly the eight contraction slots the second product
// O^T = V^T P^T expects from its B operand when V^T supplies the same keys in the same order.
#include "common.h"
#include "../../include/mars_hip.h"

namespace {

constexpr float FP8_MAX = 448.0f;
// probabilities are stored as 256 p (<= 256 < 448): keeps small ones out of the subnormals

// maximum / sum over the four lanes that share lane % 16 (the four key groups of one query), on the gfx950 row and
// half-wave swaps instead of two trips through the LDS crossbar
// (written out: the compiler folds the builtin's pair of results into one when both operands hold the same value;
// the two wait states cover the swap's read of registers a VALU instruction has just written)
__device__ __forceinline__ void swap_rows16(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap_halves32(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float group4_max(float x) {
    float a = x, b = x;
    swap_rows16(a, b);                        // a = rows [0, 0, 2, 2], b = rows [1, 1, 3, 3]
    a = fmaxf(a, b);
    b = a;
    swap_halves32(a, b);                      // a = halves [lo, lo], b = [hi, hi]
    return fmaxf(a, b);
}
__device__ __forceinline__ float group4_sum(float x) {
    float a = x, b = x;
    swap_rows16(a, b);
    a = a + b;
    b = a;
    swap_halves32(a, b);
    return a + b;
}

__device__ __forceinline__ float clamp8(float v) { return fminf(fmaxf(v, -FP8_MAX), FP8_MAX); }

// four floats -> four fp8 bytes (little end first)
__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp8(a), clamp8(b), w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp8(c), clamp8(d), w, true);
    return (uint32_t)w;
}

// eight floats -> eight fp8 bytes
__device__ __forceinline__ uint2 pack8_fp8(const bf16x8& v, float inv) {
    return make_uint2(pack4_fp8((float)v[0] * inv, (float)v[1] * inv, (float)v[2] * inv, (float)v[3] * inv),
                      pack4_fp8((float)v[4] * inv, (float)v[5] * inv, (float)v[6] * inv, (float)v[7] * inv));
}

// Pass 1: largest magnitude of every (image, operand, head) slice.  A workgroup takes 32 token rows of one image and
// reads them whole (a thread keeps its 16-byte columns and walks down the rows, so every load instruction covers
// contiguous memory); the eight threads that share a head join their maxima and one of them adds the result with an
// integer atomic maximum on the bit pattern (non-negative floats order like integers: exact, order independent).
// amax: [B][3][H] as bit patterns, zeroed by the caller.
__global__ __launch_bounds__(256) void amax_qkv_kernel(const __bf16* __restrict__ qkv, int T, int H, unsigned* __restrict__ amax) {
    const int b = blockIdx.y, r0 = blockIdx.x * 32, r1 = min(r0 + 32, T);
    const int nch = 3 * H * 8;                                           // 16-byte chunks per row
    const int64_t ld = 3LL * H * 64;
    const __bf16* base = qkv + (int64_t)b * T * ld;
    for (int c = threadIdx.x; c < nch; c += 256) {
        float m = 0.f;
#pragma unroll 4
        for (int r = r0; r < r1; ++r) {
            const bf16x8 v = *(const bf16x8*)(base + (int64_t)r * ld + 8 * c);
#pragma unroll
            for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf((float)v[j]));
        }
        m = fmaxf(m, __shfl_xor(m, 1, 64));
        m = fmaxf(m, __shfl_xor(m, 2, 64));
        m = fmaxf(m, __shfl_xor(m, 4, 64));
        if ((c & 7) == 0) atomicMax(amax + (int64_t)b * 3 * H + (c >> 3), __builtin_bit_cast(unsigned, m));
    }
}

// Pass 2: one workgroup per (image, head) and 32 tokens: q8, k8 rows (a thread packs eight dims of one token) and the
// transposed vt8 (a thread packs eight tokens of one dim).  Rows beyond T are zero.
__global__ __launch_bounds__(256) void quant_qkv_fp8_kernel(const __bf16* __restrict__ qkv, int T, int Tp, int H,
                                                            const unsigned* __restrict__ amax, uint8_t* __restrict__ q8,
                                                            uint8_t* __restrict__ k8, uint8_t* __restrict__ vt8,
                                                            float* __restrict__ scales) {
    const int bh = blockIdx.y, b = bh / H, h = bh % H, k0 = blockIdx.x * 32;
    const int64_t ld = 3LL * H * 64;
    const __bf16* base = qkv + (int64_t)b * T * ld + h * 64;
    float inv[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const float m = fmaxf(__builtin_bit_cast(float, amax[(int64_t)b * 3 * H + s * H + h]), 1e-20f);
        inv[s] = FP8_MAX / m;
        if (blockIdx.x == 0 && threadIdx.x == 0) scales[bh * 3 + s] = m / FP8_MAX;          // dequantisation factor
    }
    {
        const int t = k0 + (threadIdx.x >> 3), d8 = (threadIdx.x & 7) * 8;
        uint2 wq = make_uint2(0, 0), wk = make_uint2(0, 0);
        if (t < T) {
            wq = pack8_fp8(*(const bf16x8*)(base + (int64_t)t * ld + d8), inv[0]);
            wk = pack8_fp8(*(const bf16x8*)(base + (int64_t)t * ld + H * 64 + d8), inv[1]);
        }
        *(uint2*)(q8 + ((int64_t)bh * Tp + t) * 64 + d8) = wq;
        *(uint2*)(k8 + ((int64_t)bh * Tp + t) * 64 + d8) = wk;
    }
    {
        const int d = threadIdx.x & 63, t8 = k0 + (threadIdx.x >> 6) * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = t8 + j < T ? (float)base[(int64_t)(t8 + j) * ld + 2 * H * 64 + d] * inv[2] : 0.f;
        *(uint2*)(vt8 + ((int64_t)bh * 64 + d) * Tp + t8) = make_uint2(pack4_fp8(v[0], v[1], v[2], v[3]), pack4_fp8(v[4], v[5], v[6], v[7]));
    }
}

// One wave: QB blocks of 16 queries (the key and value fragments of a tile serve all of them), keys 32 at a time,
// the fragments of the next tile requested before the current one is worked on.
constexpr int QB = 4;
struct KvFrag { long k[2][2]; long v[4]; };

__device__ __forceinline__ void load_kv(KvFrag& f, const uint8_t* __restrict__ kb, const uint8_t* __restrict__ vb, int Tp,
                                        int k0, int i, int g) {
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        const uint8_t* kr = kb + (int64_t)(k0 + 16 * hb + i) * 64 + 8 * g;
        f.k[hb][0] = *(const long*)kr;
        f.k[hb][1] = *(const long*)(kr + 32);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint8_t* vr = vb + (int64_t)(16 * d + i) * Tp + k0 + 4 * g;
        const uint32_t vlo = *(const uint32_t*)vr, vhi = *(const uint32_t*)(vr + 16);
        f.v[d] = (long)(((uint64_t)vhi << 32) | vlo);
    }
}

__global__ __launch_bounds__(256) void flash_attn_fp8_kernel(const uint8_t* __restrict__ q8, const uint8_t* __restrict__ k8,
                                                             const uint8_t* __restrict__ vt8, const float* __restrict__ scales,
                                                             int T, int Tp, int H, __bf16* __restrict__ O, float score_scale_log2) {
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.x * (64 * QB) + wave * (16 * QB);
    if (q0 >= T) return;                                  // whole wave beyond the sequence (no barriers in this kernel)
    const float sq = scales[bh * 3 + 0], sk = scales[bh * 3 + 1], sv = scales[bh * 3 + 2];
    const float sc = sq * sk * score_scale_log2;                            // scores in the exp2 domain
    const uint8_t* qb = q8 + (int64_t)bh * Tp * 64;
    const uint8_t* kb = k8 + (int64_t)bh * Tp * 64;
    const uint8_t* vb = vt8 + (int64_t)bh * 64 * Tp;

    // B operand of S^T = K Q^T: column = query, contraction slots = dims 8g..8g+7 (+32 for the second half)
    long qf[QB][2];
#pragma unroll
    for (int c = 0; c < QB; ++c) {
        const int qrow = min(q0 + 16 * c + i, Tp - 1);
        qf[c][0] = *(const long*)(qb + (int64_t)qrow * 64 + 8 * g);
        qf[c][1] = *(const long*)(qb + (int64_t)qrow * 64 + 32 + 8 * g);
    }
    f32x4 acc[QB][4];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int c = 0; c < QB; ++c) {
        m_run[c] = -INFINITY;
        l_run[c] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    KvFrag cur, nxt;
    load_kv(cur, kb, vb, Tp, 0, i, g);
    for (int k0 = 0; k0 < Tp; k0 += 32) {
        if (k0 + 32 < Tp) load_kv(nxt, kb, vb, Tp, k0 + 32, i, g);
        const bool edge = k0 + 32 > T;                     // only the last tile holds keys beyond T
#pragma unroll
        for (int c = 0; c < QB; ++c) {
            // scores of 32 keys: two 16-key halves, each two MFMAs over the 64 dims; this lane gets keys
            // k0 + 16 hb + 4 g + r of query q0 + 16 c + i
            // raw accumulators z; the score in the exp2 domain is z * sc (sc > 0, so maxima can be taken on z)
            float z8[8];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                z = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(cur.k[hb][0], qf[c][0], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(cur.k[hb][1], qf[c][1], z, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = z[r];
                    if (edge && k0 + 16 * hb + 4 * g + r >= T) v = -INFINITY;
                    z8[4 * hb + r] = v;
                }
            }
            float mloc = fmaxf(fmaxf(fmaxf(z8[0], z8[1]), fmaxf(z8[2], z8[3])), fmaxf(fmaxf(z8[4], z8[5]), fmaxf(z8[6], z8[7])));
            mloc = group4_max(mloc);
            const float m_new = fmaxf(m_run[c], mloc * sc);     // finite: every tile holds at least one key below T
            if (__any(m_new != m_run[c])) {                     // the running maximum moved for some query of the block
                const float alpha = __builtin_amdgcn_exp2f(m_run[c] - m_new);
                l_run[c] *= alpha;
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[c][d] *= alpha;
                m_run[c] = m_new;
            }
            // probabilities as 256 p = exp2(z sc - (m - 8)): one fused multiply-add and one exponential per score
            const float off = 8.0f - m_new;
            float p[8], psum = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                p[j] = __builtin_amdgcn_exp2f(fmaf(z8[j], sc, off));
                psum += p[j];
            }
            l_run[c] += psum;                               // 256 x the partial row sum over this lane's keys
            // B operand of O^T = V^T P^T: slots 0..3 = keys 4g..4g+3 of the first half, 4..7 = the same of the second
            // (256 p <= 256 < 448: no clamp needed)
            int wlo = 0, whi = 0;
            wlo = __builtin_amdgcn_cvt_pk_fp8_f32(p[0], p[1], wlo, false);
            wlo = __builtin_amdgcn_cvt_pk_fp8_f32(p[2], p[3], wlo, true);
            whi = __builtin_amdgcn_cvt_pk_fp8_f32(p[4], p[5], whi, false);
            whi = __builtin_amdgcn_cvt_pk_fp8_f32(p[6], p[7], whi, true);
            const long pf = (long)(((uint64_t)(uint32_t)whi << 32) | (uint32_t)wlo);
#pragma unroll
            for (int d = 0; d < 4; ++d) acc[c][d] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(cur.v[d], pf, acc[c][d], 0, 0, 0);
        }
        cur = nxt;
    }
#pragma unroll
    for (int c = 0; c < QB; ++c) {
        const float l = group4_sum(l_run[c]);
        const int q = q0 + 16 * c + i;
        if (q < T) {
            const float f = sv / l;                            // l and the accumulators both carry the factor 256
            __bf16* op = O + ((int64_t)b * T + q) * (H * 64) + h * 64 + 4 * g;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (__bf16)(acc[c][d][r] * f);
                *(bf16x4*)(op + 16 * d) = o;
            }
        }
    }
}

}  // namespace

extern "C" int marship_flash_attn_fp8(const void* qkv, void* O, int B, int T, int H, void* q8, void* k8, void* vt8,
                                      float* scales, int Tp, float score_scale_log2, void* stream) {
    MARS_CHECK(qkv && O && q8 && k8 && vt8 && scales, "flash_attn_fp8: null pointer");
    MARS_CHECK(score_scale_log2 > 0.f && score_scale_log2 < 1e6f, "flash_attn_fp8: score scale must be positive and finite");
    MARS_CHECK(B > 0 && T > 0 && H > 0 && Tp >= T && Tp % 32 == 0 && Tp - T < 32, "flash_attn_fp8: bad shape B=%d T=%d H=%d Tp=%d", B, T, H, Tp);
    MARS_CHECK(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)O % 8) == 0 && ((uintptr_t)q8 % 8) == 0 && ((uintptr_t)k8 % 8) == 0 &&
               ((uintptr_t)vt8 % 4) == 0, "flash_attn_fp8: operands must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    unsigned* amax = (unsigned*)(scales + (int64_t)B * H * 3);          // second half of the scales workspace
    MARS_CHECK(hipMemsetAsync(amax, 0, sizeof(unsigned) * (size_t)B * H * 3, s) == hipSuccess, "flash_attn_fp8: memset failed");
    hipLaunchKernelGGL(amax_qkv_kernel, dim3((T + 31) / 32, B), dim3(256), 0, s, (const __bf16*)qkv, T, H, amax);
    MARS_LAUNCH_CHECK();
    hipLaunchKernelGGL(quant_qkv_fp8_kernel, dim3(Tp / 32, B * H), dim3(256), 0, s, (const __bf16*)qkv, T, Tp, H, amax, (uint8_t*)q8,
                       (uint8_t*)k8, (uint8_t*)vt8, scales);
    MARS_LAUNCH_CHECK();
    hipLaunchKernelGGL(flash_attn_fp8_kernel, dim3((T + 64 * QB - 1) / (64 * QB), B * H), dim3(256), 0, s, (const uint8_t*)q8, (const uint8_t*)k8,
                       (const uint8_t*)vt8, scales, T, Tp, H, (__bf16*)O, score_scale_log2);
    MARS_LAUNCH_CHECK();
    return 0;
}
