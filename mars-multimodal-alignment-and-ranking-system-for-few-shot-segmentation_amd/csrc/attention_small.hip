// Attention for the mask decoder's two-way transformer (segment_anything/modeling/transformer.py:218-240): eight heads of
// 16 or 32 dims, and one side of every product is a handful of prompt tokens while the other is the 4096 image tokens
// (or the tokens again).  A matrix-instruction kernel would pad the head dims to 64 and the token count to a tile:
// four to nine times the bytes and flops of the real work, which is why the padded form spent its time in softmax,
// transpose and batched-product launches over those pads.  Both forms here are vector-ALU kernels on unpadded rows:
//
//   few keys    (Tk <= 16; image -> tokens, tokens -> tokens): a thread owns one (query row, head).  K and V of the
//               prompt sit in LDS as f32; scores, softmax and P V stay in registers.  The eight heads of a row are
//               eight neighbouring lanes, so a wave reads and writes whole rows.
//   few queries (Tq <= 8 a pass; tokens -> image): a workgroup owns one prompt; a thread owns (key lane, head) and
//               walks the keys (the next key's rows in flight while the current one is used) with an online softmax
//               state per query in registers (queries in LDS); the key lanes
//               that share a head are combined by shuffles and one pass through LDS.
//
// f32 accumulation and f32 exponentials in both (expf for few keys; few queries keeps its scores in units of log 2 and
// uses v_exp_f32), operands and results f32 or bf16.
#include "common.h"
#include "../../include/mars_hip.h"

namespace {

template <typename T, int HD>
__device__ __forceinline__ void load_row(const T* __restrict__ p, float (&x)[HD]) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < HD / 8; ++i) {
            const bf16x8 v = *(const bf16x8*)(p + 8 * i);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[8 * i + e] = (float)v[e];
        }
    } else {
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
            const f32x4 v = *(const f32x4*)(p + 4 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[4 * i + e] = v[e];
        }
    }
}

template <typename T, int HD>
__device__ __forceinline__ void store_row(T* __restrict__ p, const float (&x)[HD]) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < HD / 8; ++i) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)x[8 * i + e];
            *(bf16x8*)(p + 8 * i) = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) *(f32x4*)(p + 4 * i) = (f32x4){x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]};
    }
}

// few keys: grid (ceil(Tq / (ROWS_IT * 256 / H)), B); LDS: K then V of the prompt as f32 [Tk][H * HD]
template <typename T, int HD, int TKMAX, int ROWS_IT>
__global__ __launch_bounds__(256) void attn_few_keys_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk,
                                                            const T* __restrict__ v, int64_t ldv, T* __restrict__ out, int64_t ldo, int H,
                                                            int Tq, int Tk, float scale, int qb, int kb) {
    extern __shared__ float sm[];
    const int W = H * HD, b = blockIdx.y, tid = threadIdx.x;          // qb / kb: rows a prompt of q / of k and v (0: one set for all)
    float* sK = sm;
    float* sV = sm + TKMAX * W;
    for (int i = tid; i < Tk * W; i += 256) {
        const int j = i / W, c = i - j * W;
        sK[i] = (float)k[((int64_t)b * kb + j) * ldk + c];
        sV[i] = (float)v[((int64_t)b * kb + j) * ldv + c];
    }
    __syncthreads();
    const int h = tid % H;
    // ROWS_IT passes of 256 / H rows a workgroup: the LDS copy above is paid once for all of them
    for (int it = 0; it < ROWS_IT; ++it) {
        const int row = (blockIdx.x * ROWS_IT + it) * (256 / H) + tid / H;
        if (row >= Tq) return;
        float qv[HD];
        load_row<T, HD>(q + ((int64_t)b * qb + row) * ldq + h * HD, qv);
        float s[TKMAX];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < TKMAX; ++j) {
            s[j] = -INFINITY;
            if (j < Tk) {
                const float* kr = sK + j * W + h * HD;
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < HD; ++e) d = fmaf(qv[e], kr[e], d);
                s[j] = d * scale;
                m = fmaxf(m, s[j]);
            }
        }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < TKMAX; ++j) {
            s[j] = j < Tk ? expf(s[j] - m) : 0.f;
            l += s[j];
        }
        float o[HD];
#pragma unroll
        for (int e = 0; e < HD; ++e) o[e] = 0.f;
#pragma unroll
        for (int j = 0; j < TKMAX; ++j)
            if (j < Tk) {
                const float* vr = sV + j * W + h * HD;
#pragma unroll
                for (int e = 0; e < HD; ++e) o[e] = fmaf(s[j], vr[e], o[e]);
            }
        const float inv = 1.0f / l;
#pragma unroll
        for (int e = 0; e < HD; ++e) o[e] *= inv;
        store_row<T, HD>(out + ((int64_t)b * Tq + row) * ldo + h * HD, o);
    }
}

// few queries: grid (B); queries q0 .. q0 + nq - 1 (nq <= TQ) of every prompt.  LDS: the queries as f32 [TQ][H * HD],
// then the per-wave partial states [4][H][TQ][HD + 2].
template <typename T, int HD, int TQ>
__global__ __launch_bounds__(256) void attn_few_queries_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk,
                                                               const T* __restrict__ v, int64_t ldv, T* __restrict__ out, int64_t ldo, int H,
                                                               int Tq, int Tk, float scale, int q0, int nq, int qb, int kb) {
    extern __shared__ float sm[];
    const int W = H * HD, b = blockIdx.x, tid = threadIdx.x;
    float* sQ = sm;                                   // [TQ][W]
    float* sP = sm + TQ * W;                          // [4][H][TQ][HD + 2]
    // scores are kept in units of log 2 (the scale carries log2 e), so every exponential is one v_exp_f32
    for (int i = tid; i < nq * W; i += 256) {
        const int j = i / W, c = i - j * W;
        sQ[i] = (float)q[((int64_t)b * qb + q0 + j) * ldq + c] * (scale * 1.44269504088896340736f);
    }
    __syncthreads();
    const int h = tid % H, kl = tid / H, nkl = 256 / H;          // key lanes: keys kl, kl + nkl, ...
    float m[TQ], l[TQ], acc[TQ][HD];
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
        m[j] = -INFINITY;
        l[j] = 0.f;
#pragma unroll
        for (int e = 0; e < HD; ++e) acc[j][e] = 0.f;
    }
    // the rows of the next key are requested before the current one is used
    typedef T raw_t __attribute__((ext_vector_type(HD)));
    const T* kp = k + ((int64_t)b * kb + kl) * ldk + h * HD;
    const T* vp = v + ((int64_t)b * kb + kl) * ldv + h * HD;
    raw_t kr = {}, vr = {};
    if (kl < Tk) { kr = *(const raw_t*)kp; vr = *(const raw_t*)vp; }
    for (int key = kl; key < Tk; key += nkl) {
        float kv[HD], vv[HD];
#pragma unroll
        for (int e = 0; e < HD; ++e) { kv[e] = (float)kr[e]; vv[e] = (float)vr[e]; }
        if (key + nkl < Tk) {
            kp += (int64_t)nkl * ldk;
            vp += (int64_t)nkl * ldv;
            kr = *(const raw_t*)kp;
            vr = *(const raw_t*)vp;
        }
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            if (j < nq) {
                const float* qr = sQ + j * W + h * HD;
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < HD; ++e) s = fmaf(qr[e], kv[e], s);
                const float mn = fmaxf(m[j], s);
                const float a = __builtin_amdgcn_exp2f(m[j] - mn), p = __builtin_amdgcn_exp2f(s - mn);       // (m = -inf at the first key: a = 0)
                l[j] = l[j] * a + p;
#pragma unroll
                for (int e = 0; e < HD; ++e) acc[j][e] = fmaf(acc[j][e], a, p * vv[e]);
                m[j] = mn;
            }
        }
    }
    // the key lanes of a head inside the wave: lanes that differ in the bits above log2(H)
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
        if (j < nq) {
            for (int off = H; off < 64; off <<= 1) {
                const float mo = __shfl_xor(m[j], off), lo = __shfl_xor(l[j], off);
                const float mn = fmaxf(m[j], mo);
                const float a = m[j] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m[j] - mn), bsc = mo == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mo - mn);
                l[j] = l[j] * a + lo * bsc;
#pragma unroll
                for (int e = 0; e < HD; ++e) acc[j][e] = acc[j][e] * a + __shfl_xor(acc[j][e], off) * bsc;
                m[j] = mn;
            }
            if (lane < H) {
                float* dst = sP + (((size_t)wave * H + h) * TQ + j) * (HD + 2);
                dst[0] = m[j];
                dst[1] = l[j];
#pragma unroll
                for (int e = 0; e < HD; ++e) dst[2 + e] = acc[j][e];
            }
        }
    }
    __syncthreads();
    // one thread per (query, head, dim): the four waves' partials
    for (int i = tid; i < nq * W; i += 256) {
        const int j = i / W, c = i - j * W, hh = c / HD, e = c - hh * HD;
        float mn = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) mn = fmaxf(mn, sP[(((size_t)w * H + hh) * TQ + j) * (HD + 2)]);
        float lt = 0.f, at = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float* src = sP + (((size_t)w * H + hh) * TQ + j) * (HD + 2);
            const float a = src[0] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(src[0] - mn);
            lt += src[1] * a;
            at += src[2 + e] * a;
        }
        out[((int64_t)b * Tq + q0 + j) * ldo + c] = (T)(at / lt);
    }
}

template <typename T, int HD>
int launch_small(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out, int64_t ldo, int B, int H,
                 int Tq, int Tk, float scale, int shared, hipStream_t s) {
    const int W = H * HD, qb = (shared & 1) ? 0 : Tq, kb = (shared & 2) ? 0 : Tk;
    if (Tk <= 16) {
        const int rpp = 256 / H;                          // rows a pass
        const dim3 block(256);
#define MARS_FEW_KEYS(TKM, IT)                                                                                                                \
    hipLaunchKernelGGL((attn_few_keys_kernel<T, HD, TKM, IT>), dim3((unsigned)((Tq + rpp * IT - 1) / (rpp * IT)), (unsigned)B), block,             \
                       2 * TKM * W * sizeof(float), s, (const T*)q, ldq, (const T*)k, ldk, (const T*)v, ldv, (T*)out, ldo, H, Tq, Tk, scale, qb, kb)
        if (Tq >= 8 * rpp) {
            if (Tk <= 8) MARS_FEW_KEYS(8, 8); else MARS_FEW_KEYS(16, 8);
        } else {
            if (Tk <= 8) MARS_FEW_KEYS(8, 1); else MARS_FEW_KEYS(16, 1);
        }
#undef MARS_FEW_KEYS
        MARS_LAUNCH_CHECK();
        return 0;
    }
    constexpr int TQ = HD == 16 ? 8 : 4;                 // queries a pass: their states live in registers
    const size_t shm = ((size_t)TQ * W + (size_t)4 * H * TQ * (HD + 2)) * sizeof(float);
    for (int q0 = 0; q0 < Tq; q0 += TQ) {
        const int nq = Tq - q0 < TQ ? Tq - q0 : TQ;
        hipLaunchKernelGGL((attn_few_queries_kernel<T, HD, TQ>), dim3((unsigned)B), dim3(256), shm, s, (const T*)q, ldq, (const T*)k, ldk,
                           (const T*)v, ldv, (T*)out, ldo, H, Tq, Tk, scale, q0, nq, qb, kb);
        MARS_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace

extern "C" int marship_attn_small(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out,
                                  int64_t ldo, int B, int H, int hd, int Tq, int Tk, float scale, int shared, int dtype, void* stream) {
    MARS_CHECK(q && k && v && out, "attn_small: null pointer");
    MARS_CHECK(B > 0 && B <= 65535 && Tq > 0 && Tk > 0, "attn_small: bad shape B=%d Tq=%d Tk=%d", B, Tq, Tk);
    MARS_CHECK(H == 1 || H == 2 || H == 4 || H == 8 || H == 16 || H == 32 || H == 64, "attn_small: %d heads (a power of two up to 64)", H);
    MARS_CHECK(hd == 16 || hd == 32, "attn_small: head dim %d (16 or 32)", hd);
    MARS_CHECK(Tk <= 16 || Tq <= 16, "attn_small: one side must have at most 16 tokens (Tq=%d, Tk=%d)", Tq, Tk);
    MARS_CHECK(dtype == 0 || dtype == 1, "attn_small: dtype %d", dtype);
    MARS_CHECK(shared >= 0 && shared <= 3, "attn_small: shared %d (bit 0: q, bit 1: k and v)", shared);
    const int W = H * hd, es = dtype == 1 ? 2 : 4;
    MARS_CHECK(ldq >= W && ldk >= W && ldv >= W && ldo >= W, "attn_small: a row stride is smaller than heads x head dim");
    MARS_CHECK((ldq * es) % 16 == 0 && (ldk * es) % 16 == 0 && (ldv * es) % 16 == 0 && (ldo * es) % 16 == 0 &&
               ((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16 == 0, "attn_small: rows must be 16-byte aligned");
    MARS_CHECK((size_t)2 * 16 * W * sizeof(float) <= 64 * 1024, "attn_small: %d x %d is too wide for the LDS copy of K and V", H, hd);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 1) return hd == 16 ? launch_small<__bf16, 16>(q, ldq, k, ldk, v, ldv, out, ldo, B, H, Tq, Tk, scale, shared, s)
                                    : launch_small<__bf16, 32>(q, ldq, k, ldk, v, ldv, out, ldo, B, H, Tq, Tk, scale, shared, s);
    return hd == 16 ? launch_small<float, 16>(q, ldq, k, ldk, v, ldv, out, ldo, B, H, Tq, Tk, scale, shared, s)
                    : launch_small<float, 32>(q, ldq, k, ldk, v, ldv, out, ldo, B, H, Tq, Tk, scale, shared, s);
}
