// Attention with 8-bit (OCP e4m3) operands for 64-wide heads: softmax(q k^T / 8) v with q, k, v and the
// probabilities quantised to fp8 and both contractions on v_mfma_f32_16x16x32_fp8_fp8 (f32 accumulation, f32
// softmax).  Same arithmetic as dinov2/layers/attention.py:54-60 and alpha_clip/model.py:233-246 up to the
// quantisation; the configuration that asks for it trades precision, not speed: with 64-wide heads the kernel is
// bound by the exponentials, as the bf16 one is.
//
// Two launches: `quant_qkv_fp8_kernel` scales every (image, head) slice of q, k and v by 448 / amax of the slice and
// writes q8 / k8 [B*H, Tp, 64] and v transposed, vt8 [B*H, 64, Tp] (rows beyond T are zero, Tp a multiple of 32);
// `flash_attn_fp8_kernel` gives every wave 16 queries and walks the keys 32 at a time.  The scores are formed
// transposed (S^T = K Q^T: keys are MFMA rows, a query is a lane column), so a lane holds, for its query, keys
// 4g..4g+3 of both 16-key halves of the tile - exactly the eight contraction slots the second product
// O^T = V^T P^T expects from its B operand when V^T supplies the same keys in the same order.
#include "common.h"
#include "../../include/mars_hip.h"

namespace {

constexpr float FP8_MAX = 448.0f;
constexpr float P_SCALE = 256.0f;            // probabilities are stored as 256 p: keeps small ones out of the subnormals

__device__ __forceinline__ float clamp8(float v) { return fminf(fmaxf(v, -FP8_MAX), FP8_MAX); }

// four floats -> four fp8 bytes (little end first)
__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp8(a), clamp8(b), w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp8(c), clamp8(d), w, true);
    return (uint32_t)w;
}

__global__ __launch_bounds__(256) void quant_qkv_fp8_kernel(const __bf16* __restrict__ qkv, int T, int Tp, int H,
                                                            uint8_t* __restrict__ q8, uint8_t* __restrict__ k8,
                                                            uint8_t* __restrict__ vt8, float* __restrict__ scales) {
    __shared__ float sm[16];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int64_t ld = 3LL * H * 64;
    const __bf16* base = qkv + (int64_t)b * T * ld + h * 64;
    float inv[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        float m = 0.f;
        for (int e = threadIdx.x; e < T * 16; e += blockDim.x) {
            const int t = e >> 4, d4 = (e & 15) * 4;
            const bf16x4 v = *(const bf16x4*)(base + (int64_t)t * ld + s * H * 64 + d4);
            m = fmaxf(m, fmaxf(fmaxf(fabsf((float)v[0]), fabsf((float)v[1])), fmaxf(fabsf((float)v[2]), fabsf((float)v[3]))));
        }
        m = block_reduce_max(m, sm);
        m = fmaxf(m, 1e-20f);
        inv[s] = FP8_MAX / m;
        if (threadIdx.x == 0) scales[bh * 3 + s] = m / FP8_MAX;          // dequantisation factor
    }
    // q8, k8: [Tp, 64] rows, four dims per thread; rows beyond T are zero
    for (int e = threadIdx.x; e < Tp * 16; e += blockDim.x) {
        const int t = e >> 4, d4 = (e & 15) * 4;
        uint32_t wq = 0, wk = 0;
        if (t < T) {
            const bf16x4 q = *(const bf16x4*)(base + (int64_t)t * ld + d4);
            const bf16x4 k = *(const bf16x4*)(base + (int64_t)t * ld + H * 64 + d4);
            wq = pack4_fp8((float)q[0] * inv[0], (float)q[1] * inv[0], (float)q[2] * inv[0], (float)q[3] * inv[0]);
            wk = pack4_fp8((float)k[0] * inv[1], (float)k[1] * inv[1], (float)k[2] * inv[1], (float)k[3] * inv[1]);
        }
        *(uint32_t*)(q8 + ((int64_t)bh * Tp + t) * 64 + d4) = wq;
        *(uint32_t*)(k8 + ((int64_t)bh * Tp + t) * 64 + d4) = wk;
    }
    // vt8: [64, Tp], four keys per thread (the dim runs fastest over the threads so the reads of a row coalesce)
    for (int e = threadIdx.x; e < (Tp / 4) * 64; e += blockDim.x) {
        const int d = e & 63, t4 = (e >> 6) * 4;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = t4 + j < T ? (float)base[(int64_t)(t4 + j) * ld + 2 * H * 64 + d] * inv[2] : 0.f;
        *(uint32_t*)(vt8 + ((int64_t)bh * 64 + d) * Tp + t4) = pack4_fp8(v[0], v[1], v[2], v[3]);
    }
}

__global__ __launch_bounds__(256) void flash_attn_fp8_kernel(const uint8_t* __restrict__ q8, const uint8_t* __restrict__ k8,
                                                             const uint8_t* __restrict__ vt8, const float* __restrict__ scales,
                                                             int T, int Tp, int H, __bf16* __restrict__ O) {
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.x * 64 + wave * 16;
    if (q0 >= T) return;                                  // whole wave beyond the sequence (no barriers in this kernel)
    const float sq = scales[bh * 3 + 0], sk = scales[bh * 3 + 1], sv = scales[bh * 3 + 2];
    const float sc = sq * sk * 0.125f * 1.44269504088896340736f;         // scores in the exp2 domain
    const uint8_t* qb = q8 + (int64_t)bh * Tp * 64;
    const uint8_t* kb = k8 + (int64_t)bh * Tp * 64;
    const uint8_t* vb = vt8 + (int64_t)bh * 64 * Tp;

    // B operand of S^T = K Q^T: column = query q0 + i, contraction slots = dims 8g..8g+7 (+32 for the second half)
    const int qrow = min(q0 + i, Tp - 1);
    const long qf0 = *(const long*)(qb + (int64_t)qrow * 64 + 8 * g);
    const long qf1 = *(const long*)(qb + (int64_t)qrow * 64 + 32 + 8 * g);

    f32x4 acc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) acc[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = 0; k0 < Tp; k0 += 32) {
        // scores of 32 keys: two 16-key halves, each two MFMAs over the 64 dims
        f32x4 s[2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const uint8_t* kr = kb + (int64_t)(k0 + 16 * hb + i) * 64 + 8 * g;
            const long kf0 = *(const long*)kr, kf1 = *(const long*)(kr + 32);
            f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
            z = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(kf0, qf0, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(kf1, qf1, z, 0, 0, 0);
            s[hb] = z;
        }
        // this lane: keys k0 + 16 hb + 4 g + r of query q0 + i
        float mloc = -INFINITY;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + 16 * hb + 4 * g + r;
                const float v = key < T ? s[hb][r] * sc : -INFINITY;
                s[hb][r] = v;
                mloc = fmaxf(mloc, v);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);            // finite: every tile holds at least one key below T
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float p[8], psum = 0.f;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = exp2f(s[hb][r] - m_new);
                p[4 * hb + r] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;                      // partial over this lane's keys; joined over g at the end
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[d] *= alpha;
        // B operand of O^T = V^T P^T: slots 0..3 = keys 4g..4g+3 of the first half, 4..7 = the same of the second
        const uint32_t plo = pack4_fp8(p[0] * P_SCALE, p[1] * P_SCALE, p[2] * P_SCALE, p[3] * P_SCALE);
        const uint32_t phi = pack4_fp8(p[4] * P_SCALE, p[5] * P_SCALE, p[6] * P_SCALE, p[7] * P_SCALE);
        const long pf = (long)(((uint64_t)phi << 32) | plo);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint8_t* vr = vb + (int64_t)(16 * d + i) * Tp + k0 + 4 * g;
            const uint32_t vlo = *(const uint32_t*)vr, vhi = *(const uint32_t*)(vr + 16);
            const long vf = (long)(((uint64_t)vhi << 32) | vlo);
            acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(vf, pf, acc[d], 0, 0, 0);
        }
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const int q = q0 + i;
    if (q < T) {
        const float f = sv / (P_SCALE * l_run);
        __bf16* op = O + ((int64_t)b * T + q) * (H * 64) + h * 64 + 4 * g;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (__bf16)(acc[d][r] * f);
            *(bf16x4*)(op + 16 * d) = o;
        }
    }
}

}  // namespace

extern "C" int marship_flash_attn_fp8(const void* qkv, void* O, int B, int T, int H, void* q8, void* k8, void* vt8,
                                      float* scales, int Tp, void* stream) {
    MARS_CHECK(qkv && O && q8 && k8 && vt8 && scales, "flash_attn_fp8: null pointer");
    MARS_CHECK(B > 0 && T > 0 && H > 0 && Tp >= T && Tp % 32 == 0 && Tp - T < 32, "flash_attn_fp8: bad shape B=%d T=%d H=%d Tp=%d", B, T, H, Tp);
    MARS_CHECK(((uintptr_t)qkv % 8) == 0 && ((uintptr_t)O % 8) == 0 && ((uintptr_t)q8 % 8) == 0 && ((uintptr_t)k8 % 8) == 0 &&
               ((uintptr_t)vt8 % 4) == 0, "flash_attn_fp8: operands must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(quant_qkv_fp8_kernel, dim3(B * H), dim3(256), 0, s, (const __bf16*)qkv, T, Tp, H, (uint8_t*)q8, (uint8_t*)k8,
                       (uint8_t*)vt8, scales);
    MARS_LAUNCH_CHECK();
    hipLaunchKernelGGL(flash_attn_fp8_kernel, dim3((T + 63) / 64, B * H), dim3(256), 0, s, (const uint8_t*)q8, (const uint8_t*)k8,
                       (const uint8_t*)vt8, scales, T, Tp, H, (__bf16*)O);
    MARS_LAUNCH_CHECK();
    return 0;
}
