// Output upscaling of the mask decoder (segment_anything/modeling/mask_decoder.py:53-59,136-143) at the shipped width
// (256 transformer channels -> 64 -> 32), bf16 operands, as two kernels that keep every intermediate f32 tensor out of
// HBM:
//
//   upscale_ln_gelu_kernel   ConvTranspose2d(256 -> 64, k 2, s 2) + LayerNorm2d + GELU.  A transposed convolution with
//       kernel = stride = 2 is one matrix product a token: out[(2y+dy, 2x+dx), :] = W[dy, dx] x[(y, x), :] + b.  The
//       whole weight (four 64 x 256 matrices, 128 KB as bf16) lives in LDS for the life of a workgroup; a wave owns 32
//       tokens at a time, holds their 256 channels as matrix-instruction operands in registers and runs the four
//       positions over them.  Output channels sit on the M side of v_mfma_f32_32x32x16_bf16, tokens on the N side,
//       so the 64 channels a LayerNorm averages over are 32 accumulator registers of a lane and of its partner in the
//       other half-wave: the statistics cost one v_permlane32_swap each.
//   upscale_masks_kernel     ConvTranspose2d(64 -> 32, k 2, s 2) + GELU + the product with the four hypernetwork
//       vectors of the prompt: masks[b, m, pixel] = sum_c hyper[b, m, c] * gelu(W[dy, dx] u[(y, x), :] + b)[c].  The
//       weight (128 x 64) is held in registers as A operands; the 32-channel result of a pixel never leaves the
//       accumulators: it is reduced against the four vectors in place and only the four mask values are written.
//
// Both are bound by the vector ALU work of the erf GELU (one evaluation per element of tensors of 1 G and 2 G
// elements at 1024 prompts), which is why it is evaluated with the five-term rational form below rather than erff.
#include "common.h"
#include "../../include/mars_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v2i32 __attribute__((ext_vector_type(2)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

// (sum over both half-waves of a, in the lower half-wave; of b, in the upper)
__device__ __forceinline__ float halves_sum2(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float halves_sum(float a) { return halves_sum2(a, a); }

// 0.5 v (1 + erf(v / sqrt 2)), erf by Abramowitz & Stegun 7.1.26 (absolute error below 1.5e-7, i.e. f32 rounding
// level for the activations here): erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z), z >= 0; the result
// is v - v e / 2 for v >= 0 and -|v| e / 2 below.  Two values at a time: the multiplies and multiply-adds are
// v_pk_*_f32 (two values an issue slot); only the reciprocal and the exponential stay one value an instruction.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) {
    const f32x2 a = __builtin_elementwise_abs(v), z = a * 0.70710678118654752440f;
    const f32x2 d = z * 0.3275911f + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 p = t * 1.061405429f + -1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t + -0.284496736f;
    p = p * t + 0.254829592f;
    const f32x2 x = z * z * -1.44269504088896340736f;
    const f32x2 ex = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
    const f32x2 e = p * t * ex;
    return (a * -0.5f) * e + __builtin_elementwise_max(v, (f32x2)(0.f));
}

constexpr int UP0_K = 256, UP0_C = 64;          // first transposed convolution: channels in, channels out
constexpr int UP0_LDS_W = 4 * UP0_C * UP0_K * 2; // bytes of its weight as bf16

// x bf16 [tokens, ldx] (tokens = prompts * G * G, token-major), w0 bf16 [32][256][8]: 8-element runs of K, run-major
// (entry (kc, r) = W[r][8 kc .. 8 kc + 7], r = (dy, dx, channel)), prm f32 [3][64]: convolution bias, LayerNorm weight and
// bias.  out bf16 [prompts * 4 G G, ldo] in the upscaled row-major pixel order.  Persistent: eight waves a workgroup,
// wave w of workgroup g walks the 32-token tiles 8 g + w, 8 g + w + 8 gridDim, ...
__global__ __launch_bounds__(512) void upscale_ln_gelu_kernel(const __bf16* __restrict__ x, int64_t ldx, const __bf16* __restrict__ w0,
                                                              const float* __restrict__ prm, float eps, __bf16* __restrict__ out, int64_t ldo,
                                                              int G, int64_t ntiles, int x_tiled) {
    __shared__ __align__(16) unsigned char smem[UP0_LDS_W + 3 * UP0_C * 4];
    bf16x8* sW = (bf16x8*)smem;
    float* sP = (float*)(smem + UP0_LDS_W);
    for (int i = threadIdx.x; i < UP0_LDS_W / 16; i += 512) sW[i] = ((const bf16x8*)w0)[i];
    if (threadIdx.x < 3 * UP0_C) sP[threadIdx.x] = prm[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, half = lane >> 5;
    const int Ni = G * G;
    for (int64_t t = (int64_t)blockIdx.x * 8 + wave; t < ntiles; t += (int64_t)gridDim.x * 8) {
        const int64_t tok = t * 32 + n;
        const __bf16* xp = x + tok * ldx + half * 8;
        bf16x8 xb[UP0_K / 16];
#pragma unroll
        for (int ks = 0; ks < UP0_K / 16; ++ks)       // (x_tiled: the order sam_i2t_block leaves it in, 1 KB a load)
            xb[ks] = x_tiled ? ((const bf16x8*)x)[(t * 16 + ks) * 64 + lane] : *(const bf16x8*)(xp + ks * 16);
        const int64_t b = tok / Ni;
        const int pl = (int)(tok - b * Ni), y = pl / G, xx = pl - y * G;
#pragma unroll 1
        for (int pos = 0; pos < 4; ++pos) {
            f32x16 acc[2];
            acc[0] = (f32x16)(0.f);
            acc[1] = (f32x16)(0.f);
            const bf16x8* wp = sW + half * 256 + pos * UP0_C + n;
#pragma unroll
            for (int ks = 0; ks < UP0_K / 16; ++ks) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[ks * 512], xb[ks], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[ks * 512 + 32], xb[ks], acc[1], 0, 0, 0);
            }
            // this lane: channels 32 mb + 8 i + 4 half + j of token n; the partner lane holds the other 32
            float s = 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 cb = *(const f32x4*)(sP + mb * 32 + i * 8 + half * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[mb][4 * i + j] += cb[j];
                        s += acc[mb][4 * i + j];
                    }
                }
            const float mean = halves_sum(s) * (1.0f / UP0_C);
            float q = 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    acc[mb][e] -= mean;
                    q = fmaf(acc[mb][e], acc[mb][e], q);
                }
            const float rstd = 1.0f / sqrtf(halves_sum(q) * (1.0f / UP0_C) + eps);
            const int dy = pos >> 1, dx = pos & 1;
            __bf16* orow = out + ((b * 4 * Ni) + (int64_t)(2 * y + dy) * (2 * G) + 2 * xx + dx) * ldo;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    // groups 2 rp and 2 rp + 1 of eight channels: the half-waves trade halves so that each lane stores 16 bytes
                    bf16x4 g0, g1;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int i = 2 * rp + u;
                        const f32x4 lw = *(const f32x4*)(sP + UP0_C + mb * 32 + i * 8 + half * 4);
                        const f32x4 lb = *(const f32x4*)(sP + 2 * UP0_C + mb * 32 + i * 8 + half * 4);
#pragma unroll
                        for (int j = 0; j < 4; j += 2) {
                            const f32x2 r = gelu_erf2((f32x2){acc[mb][4 * i + j], acc[mb][4 * i + j + 1]} * rstd * (f32x2){lw[j], lw[j + 1]} +
                                                      (f32x2){lb[j], lb[j + 1]});
                            if (u == 0) { g0[j] = (__bf16)r[0]; g0[j + 1] = (__bf16)r[1]; }
                            else { g1[j] = (__bf16)r[0]; g1[j + 1] = (__bf16)r[1]; }
                        }
                    }
                    const v2i32 ai = __builtin_bit_cast(v2i32, g0), bi = __builtin_bit_cast(v2i32, g1);
                    const auto s0 = __builtin_amdgcn_permlane32_swap((unsigned)ai[0], (unsigned)bi[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap((unsigned)ai[1], (unsigned)bi[1], false, false);
                    const v4i32 o = {(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
                    *(v4i32*)(orow + mb * 32 + 8 * (2 * rp + half)) = o;
                }
        }
    }
}

constexpr int UP3_K = 64, UP3_C = 32;

// u bf16 [prompts * P1, ldu] (P1 = G2 * G2 pixels a prompt, row-major), w3 bf16 [128][64] (row = (dy, dx, channel)),
// bias f32 [32], hyper f32 [prompts * 4, ldh], masks f32 [prompts, 4, 2 G2, 2 G2].  grid (P1 / (32 * 4 * TILES), prompts);
// wave w of workgroup g owns tiles (4 g + w) TILES ... of 32 consecutive pixels (a tile never crosses a pixel row:
// G2 is a multiple of 32).
template <int TILES>
__global__ __launch_bounds__(256) void upscale_masks_kernel(const __bf16* __restrict__ u, int64_t ldu, const __bf16* __restrict__ w3,
                                                            const float* __restrict__ bias, const float* __restrict__ hyper, int64_t ldh,
                                                            float* __restrict__ masks, int G2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, half = lane >> 5;
    const int64_t b = blockIdx.y, P1 = (int64_t)G2 * G2;
    bf16x8 wa[4][UP3_K / 16];
#pragma unroll
    for (int pos = 0; pos < 4; ++pos)
#pragma unroll
        for (int ks = 0; ks < UP3_K / 16; ++ks) wa[pos][ks] = *(const bf16x8*)(w3 + (pos * UP3_C + n) * UP3_K + ks * 16 + half * 8);
    // channels 8 i + 4 half + j of the bias and of the prompt's four vectors
    f32x4 cb[4], hy[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cb[i] = *(const f32x4*)(bias + 8 * i + 4 * half);
#pragma unroll
        for (int m = 0; m < 4; ++m) hy[m][i] = *(const f32x4*)(hyper + (b * 4 + m) * ldh + 8 * i + 4 * half);
    }
    const int W4 = 2 * G2;
#pragma unroll 1
    for (int tt = 0; tt < TILES; ++tt) {
        const int pix0 = ((blockIdx.x * 4 + wave) * TILES + tt) * 32;
        const __bf16* up = u + (b * P1 + pix0 + n) * ldu + half * 8;
        bf16x8 xb[UP3_K / 16];
#pragma unroll
        for (int ks = 0; ks < UP3_K / 16; ++ks) xb[ks] = *(const bf16x8*)(up + ks * 16);
        const int Y = pix0 / G2, X = pix0 - Y * G2 + n;
        float val[4][2];                         // [position][m - 2 half]
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            f32x16 acc = (f32x16)(0.f);
#pragma unroll
            for (int ks = 0; ks < UP3_K / 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[pos][ks], xb[ks], acc, 0, 0, 0);
            f32x2 part[4] = {(f32x2)(0.f), (f32x2)(0.f), (f32x2)(0.f), (f32x2)(0.f)};      // (even, odd channels: summed at the end)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const f32x2 g = gelu_erf2((f32x2){acc[4 * i + j] + cb[i][j], acc[4 * i + j + 1] + cb[i][j + 1]});
#pragma unroll
                    for (int m = 0; m < 4; ++m) part[m] = g * (f32x2){hy[m][i][j], hy[m][i][j + 1]} + part[m];
                }
            val[pos][0] = halves_sum2(part[0][0] + part[0][1], part[2][0] + part[2][1]);      // lower half-wave: mask 0, upper: mask 2
            val[pos][1] = halves_sum2(part[1][0] + part[1][1], part[3][0] + part[3][1]);      //                       1           3
        }
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                float* mo = masks + ((b * 4 + 2 * half + mm) * W4 + 2 * Y + dy) * (int64_t)W4 + 2 * X;
                *(float2*)mo = make_float2(val[2 * dy][mm], val[2 * dy + 1][mm]);
            }
    }
}

}  // namespace

extern "C" int marship_sam_upscale_ln_gelu(const void* x, int64_t ldx, const void* w0, const float* prm, float eps, void* out, int64_t ldo,
                                           int64_t prompts, int G, int x_tiled, void* stream) {
    MARS_CHECK(x && w0 && prm && out && prompts > 0 && G > 0, "sam_upscale_ln_gelu: bad arguments");
    const int64_t tokens = prompts * G * G;
    MARS_CHECK(tokens % 32 == 0, "sam_upscale_ln_gelu: %lld tokens (a multiple of 32)", (long long)tokens);
    MARS_CHECK(ldx >= UP0_K && ldx % 8 == 0 && ldo >= UP0_C && ldo % 8 == 0 && (((uintptr_t)x | (uintptr_t)w0 | (uintptr_t)out | (uintptr_t)prm) % 16) == 0,
               "sam_upscale_ln_gelu: rows must be 16-byte aligned and hold 256 / 64 channels");
    MARS_CHECK(!x_tiled || ldx == UP0_K, "sam_upscale_ln_gelu: a tiled x has 256 values a token");
    static const int cus = [] { int dev = 0, c = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256; return c; }();
    const int64_t ntiles = tokens / 32;
    const int64_t want = (ntiles + 7) / 8;
    const unsigned grid = (unsigned)(want < cus ? want : cus);
    hipLaunchKernelGGL(upscale_ln_gelu_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, (const __bf16*)x, ldx, (const __bf16*)w0, prm, eps,
                       (__bf16*)out, ldo, G, ntiles, x_tiled);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_sam_upscale_masks(const void* u, int64_t ldu, const void* w3, const float* bias, const float* hyper, int64_t ldh,
                                         float* masks, int64_t prompts, int G2, void* stream) {
    MARS_CHECK(u && w3 && bias && hyper && masks && prompts > 0 && prompts <= 65535 && G2 > 0, "sam_upscale_masks: bad arguments");
    MARS_CHECK(G2 % 32 == 0, "sam_upscale_masks: %d pixels a row (a multiple of 32)", G2);
    MARS_CHECK(ldu >= UP3_K && ldu % 8 == 0 && ldh >= UP3_C && ldh % 4 == 0 &&
                   (((uintptr_t)u | (uintptr_t)w3 | (uintptr_t)bias | (uintptr_t)hyper) % 16) == 0 && ((uintptr_t)masks % 8) == 0,
               "sam_upscale_masks: rows must be 16-byte aligned and hold 64 / 32 channels");
    const int64_t tiles = (int64_t)G2 * G2 / 32;      // a prompt
    hipStream_t s = (hipStream_t)stream;
    if (tiles % 16 == 0)
        hipLaunchKernelGGL(upscale_masks_kernel<4>, dim3((unsigned)(tiles / 16), (unsigned)prompts), dim3(256), 0, s, (const __bf16*)u, ldu,
                           (const __bf16*)w3, bias, hyper, ldh, masks, G2);
    else if (tiles % 4 == 0)
        hipLaunchKernelGGL(upscale_masks_kernel<1>, dim3((unsigned)(tiles / 4), (unsigned)prompts), dim3(256), 0, s, (const __bf16*)u, ldu,
                           (const __bf16*)w3, bias, hyper, ldh, masks, G2);
    else {
        MARS_CHECK(false, "sam_upscale_masks: %lld tiles a prompt (a multiple of 4)", (long long)tiles);
    }
    MARS_LAUNCH_CHECK();
    return 0;
}
