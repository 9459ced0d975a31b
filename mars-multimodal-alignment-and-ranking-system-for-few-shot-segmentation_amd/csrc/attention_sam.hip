// Fused attention with the decomposed relative-position bias of the SAM image encoder
// (segment_anything/modeling/image_encoder.py:224-240, 325-361): softmax(scale q.k + rel_h[q, kh] + rel_w[q, kw]) v
// for packed projections with head dims padded to 64 or 128, without a score matrix in memory.
//
// Same orientation as the 64-wide kernel of attention.hip: a wave owns 16-query blocks and computes S^T = K Q^T
// with v_mfma_f32_16x16x32_bf16, so a query is a lane column (query = lane & 15), its 64 keys of a tile sit in 16
// accumulator registers (keys 16 f + 4 g + j for lane group g), the softmax statistics are per-lane scalars and
// the probabilities are already the B operand of O^T += V^T P^T; V^T fragments come from the row-major V tile
// through ds_read_b64_tr_b16.  The bias is added to the f32 scores in registers: rel_h / rel_w hold one row of
// Hk / Wk values per query, and a key k belongs to (k / Wk, k % Wk).  When Wk is the tile width (the global
// blocks of a 1024-pixel image: 64 x 64 tokens), a tile is one key row: one rel_h value per query and tile and
// four 16-byte loads of rel_w.  Classic online softmax in the exp2 domain (the bias terms are scaled by log2(e)
// on the way in, the accumulator by scale * log2(e)).
#include <type_traits>
#include "common.h"
#include "../../include/mars_hip.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

constexpr int KT = 64;                  // keys per tile
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ float vmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float rows_max(float x) {       // over lanes l, l^16, l^32, l^48
    x = fmaxf(x, __shfl_xor(x, 16));
    return fmaxf(x, __shfl_xor(x, 32));
}
__device__ __forceinline__ bf16x8 pack_bf16x8(f32x4 a, f32x4 b) {
    bf16x8 r;
    r[0] = (__bf16)a[0]; r[1] = (__bf16)a[1]; r[2] = (__bf16)a[2]; r[3] = (__bf16)a[3];
    r[4] = (__bf16)b[0]; r[5] = (__bf16)b[1]; r[6] = (__bf16)b[2]; r[7] = (__bf16)b[3];
    return r;
}

// HD: head dim as stored (64 or 128: padded; 80: the ViT-H head as it is, 160-byte rows - the matrix instruction's third
// K step is half zeros in registers, and the LDS rows are laid out linearly without the bank swizzle, which needs a
// power-of-two row); QB: 16-query blocks per wave; four waves per workgroup.  K / V tiles of 64 keys
// are double buffered and filled by LDS-DMA; a tile's last fragments are in registers before the fill of the next
// tile starts (hipcc would otherwise drain the DMAs in front of every LDS read that follows one).
// ALLKV (64 < T <= 224: the 14 x 14 windows): one workgroup of NW waves per (image, head); all of K and V is brought into
// LDS by one burst of fills and one barrier, then every wave walks its 16-query blocks over the resident tiles.  With
// four 64-query workgroups per (window, head), each re-filling the same 196 keys tile by tile behind a barrier, the
// windowed blocks spent their time waiting: four dependent fill latencies for a few hundred matrix instructions.
template <int HD, int QB, bool ROWTILE, int NW = 4, bool ALLKV = false, bool ROWMAP = false>
__global__ __launch_bounds__(NW * 64, (ALLKV && HD <= 80) ? 4 : 2) void sam_attn_kernel(const __bf16* __restrict__ qkv, int64_t ld, __bf16* __restrict__ O,
                                                          int64_t ldo, const float* __restrict__ rel_h,
                                                          const float* __restrict__ rel_w, int T, int H, int Hk, int Wk,
                                                          float c2, unsigned wk_magic, const int32_t* __restrict__ out_rows,
                                                          const int32_t* __restrict__ in_rows, const __bf16* __restrict__ pad_row) {
    constexpr int ROWB = HD * 2, KS = (HD + 31) / 32, DM = HD / 16, QPB = NW * QB * 16;
    constexpr bool LINEAR = (1024 % ROWB) != 0;    // rows that do not tile a 1-KiB fill piece: no swizzle
    static_assert(!ALLKV || QB == 1, "the resident form walks one 16-query block a wave at a time");
    constexpr int PPO = KT * ROWB / 1024;          // 1-KiB DMA pieces per operand tile
    constexpr int LPR = ROWB / 16;                 // 16-byte chunks (lanes of a fill) per key row
    static_assert(KT * ROWB % 1024 == 0, "an operand tile is a whole number of fill pieces");
    // [buf][K|V] tiles of TB bytes; the resident form: three whole tiles and the first 32 rows of a fourth (T <= 224), K and V each,
    // 70 KB at 80 dims: two workgroups fit a CU
    constexpr int TB = KT * ROWB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[ALLKV ? 7 * TB : 4 * TB];
    auto tile = [&](int buf, int opnd) -> unsigned char* { return lds + buf * 2 * TB + opnd * ((ALLKV && buf == 3) ? TB / 2 : TB); };
    // The resident form adds the bias on the matrix pipe when a query's two bias rows fit one K step (Hk + Wk <= 32):
    // bias[key][q] = sum_i E[key][i] R[i][q] with R[., q] = (rel_h[q][0 .. Hk), rel_w[q][0 .. Wk), 0 ...) and E[key] the
    // indicator of the two entries key = (kh, kw) picks.  A lane makes its eight entries of E from the key's (kh, kw)
    // with a few integer instructions; R is the query block's B operand, as three bf16 terms whose sum is the f32 value
    // (8 + 8 + 8 mantissa bits).  That replaces 32 gathered loads per query block and tile, which the texture path
    // serves a few lanes a clock: they were most of a block's time whatever the occupancy.
    const bool bias_mm = ALLKV && Hk + Wk <= 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r16 = lane & 15;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    int q0 = blockIdx.x * QPB + wave * (QB * 16);
    const int64_t row_b = ld * 2;                  // bytes per token row of qkv
    const unsigned char* base = (const unsigned char*)qkv + (int64_t)b * T * row_b + (int64_t)h * ROWB;
    const int64_t opnd_b = (int64_t)H * ROWB;      // bytes from q to k and from k to v inside a row
    // token row t of this image (window): its row of qkv, or - with a row map - the row the map names, or the row of
    // padding values (what the projection makes of a zero input: its bias) for a negative entry
    auto token_row = [&](int t) -> const unsigned char* {
        if (!ROWMAP || !in_rows) return base + (int64_t)t * row_b;
        const int r = in_rows[(int64_t)b * T + t];
        return r >= 0 ? (const unsigned char*)qkv + (int64_t)r * row_b + (int64_t)h * ROWB : (const unsigned char*)pad_row + (int64_t)h * ROWB;
    };
    const bool wave_active = q0 < T;               // trailing waves of the last workgroup only help with the fills

    bf16x8 qf[QB][KS];
    const float* hrow[QB];                         // this lane's rows of the bias terms
    const float* wrow[QB];
    bf16x8 rb[ALLKV ? 3 : 1];                      // (bias_mm) R of the query block: entries 8 g .. 8 g + 7, in units of c2
    auto load_queries = [&]() {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int qrow = min(q0 + qb * 16 + r16, T - 1);
            const unsigned char* qp = token_row(qrow);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                qf[qb][ks] = bf16x8{};
                if (4 * ks + g < LPR) qf[qb][ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(qp + 64 * ks + 16 * g));
            }
            const int64_t qr = (int64_t)bh * T + qrow;
            hrow[qb] = rel_h + qr * Hk;
            wrow[qb] = rel_w + qr * Wk;
        }
        if constexpr (ALLKV) {
            if (bias_mm) {
                const float unit = LOG2E / c2;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int i = 8 * g + e;
                    float v = 0.f;
                    if (i < Hk) v = hrow[0][i];
                    else if (i < Hk + Wk) v = wrow[0][i - Hk];
                    v *= unit;
                    const __bf16 hi = (__bf16)v;
                    const float r1 = v - (float)hi;
                    const __bf16 mid = (__bf16)r1;
                    rb[0][e] = hi;
                    rb[1][e] = mid;
                    rb[2][e] = (__bf16)(r1 - (float)mid);
                }
            }
        }
    };
    load_queries();

    auto issue_tile = [&](int t, int buf) {
#pragma unroll
        for (int p = wave; p < 2 * PPO; p += NW) {       // fill pieces of 1 KiB: 64 consecutive chunks of an operand tile
            const int opnd = p / PPO, pp = p % PPO;
            if (ALLKV && buf == 3 && pp >= PPO / 2) continue;         // (the fourth resident tile holds 32 rows)
            const int c = pp * 64 + lane, row = c / LPR, slot = c - row * LPR;
            const int chunk = LINEAR ? slot : slot ^ ((row >> 1) & 7);
            const int key = min(t * KT + row, T - 1);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(token_row(key) + (1 + opnd) * opnd_b + chunk * 16),
                                             (lds_ptr_t)(tile(buf, opnd) + pp * 1024), 16, 0, 0);
        }
    };

    f32x4 o[QB][DM + 1];        // un-normalised output, d blocks of 16; [DM]: the running sum of the probabilities
    float mrun[QB];
    auto reset_state = [&]() {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            mrun[qb] = -INFINITY;
#pragma unroll
            for (int m = 0; m <= DM; ++m) o[qb][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    reset_state();
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    const int nt = (T + KT - 1) / KT;

    bf16x8 kf[4][KS], vf[DM][2];
    const int ksw = (r16 >> 1) & 7;                                   // swizzle term of K rows 16 f + r16
    const int vq = r16 >> 2, vp = r16 & 3, vsw = (2 * g + (vq >> 1)) & 7;    // of V rows 32 s + 4 g + vq (+ 16)

    auto read_k = [&](int buf) {
        const unsigned char* kt = tile(buf, 0);
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
            {
                kf[f][ks] = bf16x8{};
                if (4 * ks + g < LPR)
                    kf[f][ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(kt + (16 * f + r16) * ROWB + ((LINEAR ? 4 * ks + g : (4 * ks + g) ^ ksw) << 4)));
            }
    };
    // (the resident form reads a fragment when it is needed: with everything of a tile in registers at once a wave needs
    // more than the 128 registers that let two workgroups share a CU)
    auto read_k_frag = [&](int buf, int f, bf16x8 (&dst)[KS]) {
        const unsigned char* kt = tile(buf, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            dst[ks] = bf16x8{};
            if (4 * ks + g < LPR)
                dst[ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(kt + (16 * f + r16) * ROWB + ((LINEAR ? 4 * ks + g : (4 * ks + g) ^ ksw) << 4)));
        }
    };
    auto read_v_half = [&](int buf, int sh, bf16x8 (&dst)[DM]) {
        const unsigned char* vt = tile(buf, 1);
#pragma unroll
        for (int m = 0; m < DM; ++m) {
            const int off = (4 * g + vq) * ROWB + ((LINEAR ? 2 * m + (vp >> 1) : (2 * m + (vp >> 1)) ^ vsw) << 4) + 8 * (vp & 1);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vt + off + sh * 32 * ROWB));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vt + off + sh * 32 * ROWB + 16 * ROWB));
            const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            dst[m] = __builtin_bit_cast(bf16x8, r);
        }
    };
    auto read_v = [&](int buf) {
        const unsigned char* vt = tile(buf, 1);
#pragma unroll
        for (int m = 0; m < DM; ++m) {
            // V^T[d = 16 m + 4 vp ..][keys]: byte column 32 m + 8 vp of rows 32 s + 4 g + vq and + 16
            const int off = (4 * g + vq) * ROWB + ((LINEAR ? 2 * m + (vp >> 1) : (2 * m + (vp >> 1)) ^ vsw) << 4) + 8 * (vp & 1);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vt + off + s * 32 * ROWB));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vt + off + s * 32 * ROWB + 16 * ROWB));
                const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                vf[m][s] = __builtin_bit_cast(bf16x8, r);
            }
        }
    };

    bf16x8 pr[QB][2];           // the tile's probabilities, as the B operands of the two k-steps of P.V
    // scores, bias, running maximum, probabilities (and the rescale of the accumulators when the maximum moved)
    auto scores = [&](int t, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const int nfrag = MASKED ? min(4, (T - t * KT + 15) >> 4) : 4;      // fragments with a key < T
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            f32x4 s[4];
            float hval = 0.f;
            if constexpr (ROWTILE) hval = hrow[qb][t] * LOG2E;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (!MASKED || f < nfrag) {
                    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if constexpr (ALLKV) read_k_frag(t, f, kf[f & 1]);           // (tile t lives in buffer t)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ALLKV ? (f & 1) : f][ks], qf[qb][ks], acc, 0, 0, 0);
                    if constexpr (ROWTILE) {
                        const f32x4 wv = *(const f32x4*)(wrow[qb] + 16 * f + 4 * g);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = fmaf(acc[j], c2, fmaf(wv[j], LOG2E, hval));
                    } else if (ALLKV && bias_mm) {
                        // E of key 64 t + 16 f + r16, entries 8 g .. 8 g + 7: bit e of m is set where the entry is one
                        const int key = min(t * KT + 16 * f + r16, T - 1);
                        const int kh = Wk == 1 ? key : (int)__umulhi((unsigned)key, wk_magic), kw = key - kh * Wk;
                        const unsigned ia = (unsigned)(kh - 8 * g), ib = (unsigned)(Hk + kw - 8 * g);
                        const unsigned m = (ia < 8u ? 1u << ia : 0u) | (ib < 8u ? 1u << ib : 0u);
                        v4u32 ew;
#pragma unroll
                        for (int d = 0; d < 4; ++d) ew[d] = ((m >> (2 * d)) & 1u) * 0x3f80u + ((m >> (2 * d + 1)) & 1u) * 0x3f800000u;
                        const bf16x8 ef = __builtin_bit_cast(bf16x8, ew);
#pragma unroll
                        for (int i = 0; i < (ALLKV ? 3 : 1); ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ef, rb[i], acc, 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] *= c2;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // key / Wk as a multiply: wk_magic = ceil(2^32 / Wk), exact while key Wk < 2^32 (the host checks T)
                            const int key = min(t * KT + 16 * f + 4 * g + j, T - 1);
                            const int kh = Wk == 1 ? key : (int)__umulhi((unsigned)key, wk_magic), kw = key - kh * Wk;
                            acc[j] = fmaf(acc[j], c2, (hrow[qb][kh] + wrow[qb][kw]) * LOG2E);
                        }
                    }
                    if constexpr (MASKED) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (t * KT + 16 * f + 4 * g + j >= T) acc[j] = -INFINITY;
                    }
                    s[f] = acc;
                } else {
                    s[f] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
            }
            float mx = vmax3(s[0][0], s[0][1], s[0][2]);
            mx = vmax3(mx, s[0][3], s[1][0]);
            mx = vmax3(mx, s[1][1], s[1][2]);
            mx = vmax3(mx, s[1][3], s[2][0]);
            mx = vmax3(mx, s[2][1], s[2][2]);
            mx = vmax3(mx, s[2][3], s[3][0]);
            mx = vmax3(mx, s[3][1], s[3][2]);
            mx = fmaxf(mx, s[3][3]);
            mx = rows_max(mx);
            // every tile holds a key < T for every query (the first tile always does), so the maximum is finite
            const float mnew = fmaxf(mrun[qb], mx);
            const float alpha = __builtin_amdgcn_exp2f(mrun[qb] - mnew);
            mrun[qb] = mnew;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) s[f][j] = __builtin_amdgcn_exp2f(s[f][j] - mnew);
            if (!__all(alpha == 1.f)) {
#pragma unroll
                for (int m = 0; m <= DM; ++m) { o[qb][m][0] *= alpha; o[qb][m][1] *= alpha; o[qb][m][2] *= alpha; o[qb][m][3] *= alpha; }
            }
            pr[qb][0] = pack_bf16x8(s[0], s[1]);
            pr[qb][1] = pack_bf16x8(s[2], s[3]);
        }
    };
    auto accumulate = [&](int t) {
        const bool second = (T - t * KT) > 32;                   // keys beyond the first 32 of the tile
        if constexpr (ALLKV) {
            bf16x8 vh[DM];
            read_v_half(t, 0, vh);
#pragma unroll
            for (int m = 0; m < DM; ++m) o[0][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh[m], pr[0][0], o[0][m], 0, 0, 0);
            o[0][DM] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pr[0][0], o[0][DM], 0, 0, 0);
            if (second) {
                read_v_half(t, 1, vh);
#pragma unroll
                for (int m = 0; m < DM; ++m) o[0][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh[m], pr[0][1], o[0][m], 0, 0, 0);
                o[0][DM] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pr[0][1], o[0][DM], 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
            for (int m = 0; m < DM; ++m) o[qb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[m][0], pr[qb][0], o[qb][m], 0, 0, 0);
            o[qb][DM] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pr[qb][0], o[qb][DM], 0, 0, 0);
            if (second) {
#pragma unroll
                for (int m = 0; m < DM; ++m) o[qb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[m][1], pr[qb][1], o[qb][m], 0, 0, 0);
                o[qb][DM] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pr[qb][1], o[qb][DM], 0, 0, 0);
            }
        }
    };

    auto store_queries = [&]() {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int myq = q0 + qb * 16 + r16;
            if (myq < T) {
                const float inv = 1.0f / o[qb][DM][0];
                int64_t orow = (int64_t)b * T + myq;
                if constexpr (ROWMAP) {
                    if (out_rows) orow = out_rows[orow];        // (negative: not written)
                    if (orow < 0) continue;
                }
                __bf16* op = O + orow * ldo + h * HD;
#pragma unroll
                for (int m = 0; m < DM; ++m) {
                    bf16x4 v;
                    v[0] = (__bf16)(o[qb][m][0] * inv); v[1] = (__bf16)(o[qb][m][1] * inv);
                    v[2] = (__bf16)(o[qb][m][2] * inv); v[3] = (__bf16)(o[qb][m][3] * inv);
                    *(bf16x4*)(op + 16 * m + 4 * g) = v;
                }
            }
        }
    };

    if constexpr (ALLKV) {
        if constexpr (ROWMAP) {
            // the source rows of every fill piece of this wave first (independent loads), then the fills
            constexpr int PW = (2 * PPO + NW - 1) / NW;
            const unsigned char* src[4][PW];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < PW; ++i) {
                    const int p = wave + i * NW, pp = p % PPO;
                    const int c = pp * 64 + lane, row = c / LPR;
                    src[t][i] = base;
                    if (t < nt && p < 2 * PPO) src[t][i] = token_row(min(t * KT + row, T - 1));
                }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < PW; ++i) {
                    const int p = wave + i * NW, opnd = p / PPO, pp = p % PPO;
                    if (t >= nt || p >= 2 * PPO || (t == 3 && pp >= PPO / 2)) continue;
                    const int c = pp * 64 + lane, row = c / LPR, slot = c - row * LPR;
                    const int chunk = LINEAR ? slot : slot ^ ((row >> 1) & 7);
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[t][i] + (1 + opnd) * opnd_b + chunk * 16), (lds_ptr_t)(tile(t, opnd) + pp * 1024), 16, 0, 0);
                }
        } else {
            for (int t = 0; t < nt; ++t) issue_tile(t, t);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // (q0 of the first pass is this wave's block of workgroup 0: the grid has one workgroup per (image, head))
        for (; q0 < T; q0 += NW * 16) {
            if (q0 != wave * 16) { load_queries(); reset_state(); }
            for (int t = 0; t < nt; ++t) {
                if (t * KT + KT > T) scores(t, std::true_type{});
                else scores(t, std::false_type{});
                accumulate(t);
            }
            store_queries();
        }
        return;
    }
    issue_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // per tile: K fragments -> scores and probabilities; V fragments (in the registers the K fragments no longer
    // need); only then the fill of the next tile, which lands while P.V runs; wait for it and meet at the barrier
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (wave_active) {
            read_k(buf);
            if (t * KT + KT > T) scores(t, std::true_type{});
            else scores(t, std::false_type{});
            read_v(buf);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nt) issue_tile(t + 1, buf ^ 1);
        if (wave_active) accumulate(t);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    store_queries();
}

template <int HD, int QB>
int launch(const void* qkv, int64_t ld, void* O, int64_t ldo, const float* rel_h, const float* rel_w, int nb, int T,
           int H, int Hk, int Wk, float c2, const int32_t* out_rows, const int32_t* in_rows, const __bf16* pad_row, hipStream_t s) {
    const unsigned wk_magic = Wk > 1 ? (unsigned)((0x100000000ull + (unsigned)Wk - 1) / (unsigned)Wk) : 0u;      // (a grid one token wide does not use it)
    if constexpr (QB == 1) {
        // every key of a small token grid (the windows) resident in LDS: one workgroup of eight waves per (image, head)
        if (T <= 3 * KT + 32 && T > KT && !(Wk == KT && T == Hk * Wk)) {
            if (in_rows || out_rows)
                hipLaunchKernelGGL((sam_attn_kernel<HD, 1, false, 8, true, true>), dim3(1, nb * H), dim3(512), 0, s, (const __bf16*)qkv, ld, (__bf16*)O, ldo,
                                   rel_h, rel_w, T, H, Hk, Wk, c2, wk_magic, out_rows, in_rows, pad_row);
            else
                hipLaunchKernelGGL((sam_attn_kernel<HD, 1, false, 8, true>), dim3(1, nb * H), dim3(512), 0, s, (const __bf16*)qkv, ld, (__bf16*)O, ldo, rel_h,
                                   rel_w, T, H, Hk, Wk, c2, wk_magic, out_rows, in_rows, pad_row);
            MARS_LAUNCH_CHECK();
            return 0;
        }
    }
    MARS_CHECK(!in_rows && !out_rows, "sam_attn_rows: row maps go with the windows' kernel: one 16-query block a wave form, 64 < T <= 224 tokens (T=%d)", T);
    constexpr int QPB = 4 * QB * 16;
    dim3 grid((T + QPB - 1) / QPB, nb * H), block(256);
    // one key row per tile: no division, and whole 16-byte loads of rel_w (rows of 64 floats are 16-byte aligned)
    if (Wk == KT && T == Hk * Wk && ((uintptr_t)rel_w % 16) == 0)
        hipLaunchKernelGGL((sam_attn_kernel<HD, QB, true>), grid, block, 0, s, (const __bf16*)qkv, ld, (__bf16*)O, ldo, rel_h, rel_w, T, H, Hk, Wk, c2, wk_magic, out_rows, in_rows, pad_row);
    else
        hipLaunchKernelGGL((sam_attn_kernel<HD, QB, false>), grid, block, 0, s, (const __bf16*)qkv, ld, (__bf16*)O, ldo, rel_h, rel_w, T, H, Hk, Wk, c2, wk_magic, out_rows, in_rows, pad_row);
    MARS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int marship_sam_attn_rows(const void* qkv, int64_t ld, void* O, int64_t ldo, const float* rel_h, const float* rel_w,
                                     int nb, int T, int H, int hdp, int Hk, int Wk, float scale, const int32_t* in_rows, const void* pad_row,
                                     const int32_t* out_rows, void* stream) {
    MARS_CHECK(qkv && O && rel_h && rel_w, "sam_attn: null pointer");
    MARS_CHECK(!in_rows || (pad_row && ((uintptr_t)pad_row % 16) == 0), "sam_attn: a row map needs the 16-byte aligned row of padding values");
    MARS_CHECK(nb > 0 && T > 0 && H > 0 && Hk > 0 && Wk > 0 && Hk * Wk >= T, "sam_attn: bad shape nb=%d T=%d H=%d Hk=%d Wk=%d", nb, T, H, Hk, Wk);
    MARS_CHECK(hdp == 64 || hdp == 80 || hdp == 128, "sam_attn: head dim as stored %d (64, 80 or 128)", hdp);
    MARS_CHECK(ld >= 3 * (int64_t)H * hdp && ld % 8 == 0 && ldo >= (int64_t)H * hdp && ldo % 4 == 0, "sam_attn: row strides");
    MARS_CHECK(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)O % 8) == 0, "sam_attn: alignment");
    MARS_CHECK((int64_t)nb * H <= 65535, "sam_attn: too many (image, head) pairs for one launch");
    MARS_CHECK(scale > 0.f && scale < 1e6f, "sam_attn: the score scale must be positive and finite");
    MARS_CHECK(T <= 65536 && Wk <= 65535, "sam_attn: token grids up to 65536 tokens (T=%d, Wk=%d)", T, Wk);
    const float c2 = scale * 1.4426950408889634f;
    if (hdp == 64) {
        // (the windows: every key resident, one 16-query block a wave at a time)
        if (T <= 3 * KT + 32 && T > KT && !(Wk == KT && T == Hk * Wk)) return launch<64, 1>(qkv, ld, O, ldo, rel_h, rel_w, nb, T, H, Hk, Wk, c2, out_rows, in_rows, (const __bf16*)pad_row, (hipStream_t)stream);
        return launch<64, 2>(qkv, ld, O, ldo, rel_h, rel_w, nb, T, H, Hk, Wk, c2, out_rows, in_rows, (const __bf16*)pad_row, (hipStream_t)stream);
    }
    if (hdp == 80) {
        // long rows (the global blocks): two 16-query blocks a wave share every K / V tile the workgroup brings in
        // (the one-key-row-per-tile form only: the general bias lookup does not fit beside two blocks' accumulators)
        if (T >= 1024 && Wk == KT && T == Hk * Wk && ((uintptr_t)rel_w % 16) == 0)
            return launch<80, 2>(qkv, ld, O, ldo, rel_h, rel_w, nb, T, H, Hk, Wk, c2, out_rows, in_rows, (const __bf16*)pad_row, (hipStream_t)stream);
        return launch<80, 1>(qkv, ld, O, ldo, rel_h, rel_w, nb, T, H, Hk, Wk, c2, out_rows, in_rows, (const __bf16*)pad_row, (hipStream_t)stream);
    }
    return launch<128, 1>(qkv, ld, O, ldo, rel_h, rel_w, nb, T, H, Hk, Wk, c2, out_rows, in_rows, (const __bf16*)pad_row, (hipStream_t)stream);
}

extern "C" int marship_sam_attn(const void* qkv, int64_t ld, void* O, int64_t ldo, const float* rel_h, const float* rel_w,
                                int nb, int T, int H, int hdp, int Hk, int Wk, float scale, void* stream) {
    return marship_sam_attn_rows(qkv, ld, O, ldo, rel_h, rel_w, nb, T, H, hdp, Hk, Wk, scale, nullptr, nullptr, nullptr, stream);
}
