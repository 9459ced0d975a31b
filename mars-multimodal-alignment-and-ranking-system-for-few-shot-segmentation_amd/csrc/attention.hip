// Fused attention for 64-wide heads (bf16 in, f32 accumulate) and the fused
// head/layer mean of the attention probabilities.
//
// Orientation: each wave owns 16 queries and computes S^T = K * Q^T with
// v_mfma_f32_16x16x32_bf16, so a query is a LANE column (query = lane & 15) and its
// 64 keys of a tile sit in 16 accumulator registers spread over the four 16-lane
// groups.  The online-softmax statistics are then per-lane scalars (two xor
// shuffles finish a row), and the probabilities are already the B operand of
// O^T += V^T * P^T: registers of fragment pair (2s, 2s+1) give the eight k slots of
// k-step s, i.e. keys 32s + 4g + {0..3} and 32s + 16 + 4g + {0..3} for lane group g.
// The A operand V^T[d][those keys] comes straight from the row-major V tile in LDS
// through ds_read_b64_tr_b16 (4 keys x 16 d per 16-lane group, delivered column-wise).
//
// Softmax in the exp2 domain, on an ABSOLUTE scale where the scores allow it.  The query
// fragments are scaled by log2(e)/8 once, when they are loaded, so the matrix pipe delivers
// s' = s log2(e)/8.  The first key tile is worked the classic way (exact row maxima m0).  If
// every |m0| of the wave is at most ABS_RANGE, the wave moves its accumulators to the absolute
// scale (reference maximum 0) and every later full tile costs, per probability, one v_exp_f32
// on the MFMA result and half a conversion - no multiply, no subtraction, and no dependence of
// the exponentials on a maximum of the tile.  Float formats keep their relative precision, and
// output and row sum are normalised by the same sum in the end, so the only thing to rule out is
// overflow: a per-lane max3 chain and one ballot per tile check that no score exceeds
// HEAD_ROOM (probabilities up to 2^64); if one does, the wave returns to the classic online
// softmax (running maximum) for good, starting with that tile.  Underflow cannot empty a row: its first tile holds
// a probability of at least 2^-ABS_RANGE.  Ragged and causal tiles always take the classic path.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/mars_hip.h"

// Diagnostic builds (tools/diag_attn.hip) define MARS_STAMPS and collect per-wave cycle counts of
// the three phases of a tile; in the library build the macros expand to nothing.
#ifdef MARS_STAMPS
__device__ unsigned long long* mars_attn_stamps;
#define MARS_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime();
#define MARS_ACC(dst, a, b) dst += (b) - (a);
#else
#define MARS_T(var)
#define MARS_ACC(dst, a, b)
#endif

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int KT = 64;           // keys per tile
constexpr int ROWB = 128;        // bytes per LDS row (64 bf16)

__device__ __forceinline__ int tile_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ bf16x8 pack_bf16x8(f32x4 a, f32x4 b) {
    bf16x8 r;
    r[0] = (__bf16)a[0]; r[1] = (__bf16)a[1]; r[2] = (__bf16)a[2]; r[3] = (__bf16)a[3];
    r[4] = (__bf16)b[0]; r[5] = (__bf16)b[1]; r[6] = (__bf16)b[2]; r[7] = (__bf16)b[3];
    return r;
}

// V^T fragment for d-block m and k-step s from the row-major [key][d] tile
__device__ __forceinline__ bf16x8 load_vt_frag(const unsigned char* vt, int lane, int m, int s) {
    const int g = lane >> 4, ii = lane & 15;
    const int q = ii >> 2, p = ii & 3;
    const int colb = 32 * m + 8 * p;  // byte offset of d = 16m + 4p inside the row
    const int r0 = 32 * s + 4 * g + q, r1 = r0 + 16;
    const int a0 = tile_off(r0, colb >> 4) + (colb & 15);
    const int a1 = tile_off(r1, colb >> 4) + (colb & 15);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vt + a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vt + a1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

// Plain fmaxf: this file is built with IEEE mode off and NaNs not honoured (see the Makefile), so
// hipcc does not put a NaN-quieting self-max in front of every maximum of an MFMA result.  Inline
// asm is NOT an option for anything that reads MFMA results: the MFMA -> VALU read hazard is
// covered by compiler-inserted wait states, and the compiler does not look inside asm statements.
__device__ __forceinline__ float vmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float vmax2(float a, float b) { return fmaxf(a, b); }

// Reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) on the VALU:
// v_permlane16_swap exchanges odd rows of its first operand with even rows of the second,
// v_permlane32_swap the upper half of the first with the lower half of the second, so feeding
// the same value twice leaves {row, partner row} side by side in the two results.
__device__ __forceinline__ float rows_max(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    x = vmax2(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
    u = __builtin_bit_cast(unsigned, x);
    r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return vmax2(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}

__device__ __forceinline__ float rows_sum(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    x = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    u = __builtin_bit_cast(unsigned, x);
    r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// QB query blocks of 16 per wave, NW waves per workgroup: one K/V tile in LDS serves NW*QB*16
// queries, and a wave re-uses its K and V^T fragments (held in registers) for all its QB blocks,
// which is what takes the LDS read traffic off the critical path.
//
// K/V tiles are double buffered and filled by LDS-DMA (global_load_lds, 1 KiB = 8 key rows per
// wave instruction; the bank swizzle is applied to the per-lane source address).  Per tile: read
// the tile's fragments into registers, then start the DMA of the next tile into the other buffer,
// compute, and only then wait for the DMA and meet at the barrier - the fill has the whole
// compute phase to land and no LDS read is ever issued while a DMA is in flight.
template <int QB, int NW>
__global__ __launch_bounds__(NW * 64, 2) void flash_attn_kernel(const __bf16* __restrict__ qkv, __bf16* __restrict__ O,
                                                             float* __restrict__ lse, int T, int H, int causal, float c2) {
    constexpr int QPB = NW * QB * 16, PPW = 16 / NW;        // queries per workgroup; DMA pieces per wave and tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][KT * ROWB];  // [buf][K|V]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r16 = lane & 15;
    const int D = H * 64;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int q0 = blockIdx.x * QPB + wave * (QB * 16);
    const int64_t row_b = (int64_t)3 * D * 2;  // bytes per token row of qkv
    const unsigned row_b24 = (unsigned)row_b;
    const unsigned char* base = (const unsigned char*)qkv + (int64_t)b * T * row_b + (int64_t)h * 128;
    const bool wave_active = q0 < T;            // trailing waves of the last workgroup only help with the fills

    // Q^T operands: lane = query r16 of block qb, d = 8g + j (+32 for the second k-step), scaled by
    // c2 (log2(e) / 8 for raw queries) so that the MFMAs deliver the scores in the exp2 domain; queries that
    // come scaled (c2 == 1) are used as stored
    bf16x8 qf[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = min(q0 + qb * 16 + r16, T - 1);
        const unsigned char* qp = base + (int64_t)qrow * row_b;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 raw = __builtin_bit_cast(bf16x8, *(const uint4*)(qp + 64 * ks + 16 * g));
            if (c2 == 1.0f) {
                qf[qb][ks] = raw;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[qb][ks][e] = (__bf16)((float)raw[e] * c2);
            }
        }
    }

    // DMA piece i of this wave: piece index p = wave * PPW + i, operand p >> 3 (K, V), key rows 8 (p & 7) .. +7;
    // lane -> key row (lane >> 3) of the piece, 16-byte slot lane & 7 holding chunk slot ^ ((row >> 1) & 7)
    const unsigned char* psrc[PPW];
    int prow[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave * PPW + i;
        prow[i] = 8 * (p & 7) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((prow[i] >> 1) & 7);
        psrc[i] = base + (int64_t)D * 2 * (1 + (p >> 3)) + chunk * 16;
    }
    auto issue_tile = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave * PPW + i;
            const int key = min(t * KT + prow[i], T - 1);
            // one sequence of one image is far below 4 GiB (T < 2^24 rows of < 2^24 bytes: checked by the launcher)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(psrc[i] + (((unsigned)key & 0xffffffu) * (row_b24 & 0xffffffu))),
                                             (lds_ptr_t)(&lds[buf][p >> 3][(p & 7) * 1024]), 16, 0, 0);
        }
    };

    // o[qb][0..3]: un-normalised output, d blocks of 16; o[qb][4]: the same product against a block
    // of ones, i.e. the running sum of the (bf16-rounded) probabilities, in all four registers
    f32x4 o[QB][5];
    float mrun[QB];             // reference max of the scaled scores s' = s * log2(e)/8
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        mrun[qb] = -INFINITY;
#pragma unroll
        for (int m = 0; m < 5; ++m) o[qb][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    const int nt = (T + KT - 1) / KT;
    // causal: tiles past the last query of this workgroup are never needed
    const int nt_eff = causal ? min(nt, (blockIdx.x * QPB + QPB - 1) / KT + 1) : nt;

    bf16x8 kf[4][2], vf[4][2];
    // One key tile against the wave's query blocks (straight-line code over the blocks, so that one
    // block's MFMAs can be scheduled beside the other's VALU work).  MASKED tiles (ragged last
    // tile, causal) skip the fragments that hold no valid key at all.
    auto process = [&](int t, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const int nfrag = MASKED ? min(4, (T - t * KT + 15) >> 4) : 4;     // fragments with a key < T
        f32x4 s[QB][4];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (!MASKED || f < nfrag) {
                    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[f][0], qf[qb][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[f][1], qf[qb][1], acc, 0, 0, 0);
                    s[qb][f] = acc;
                } else {
                    s[qb][f] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
            }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            // lane holds keys t*64 + 16f + 4g + j of query q0 + 16qb + r16
            if constexpr (MASKED) {
                const int myq = q0 + qb * 16 + r16;
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int key = t * KT + 16 * f + 4 * g + j;
                        if (key >= T || (causal && key > myq)) s[qb][f][j] = -INFINITY;
                    }
            }
            float mx = vmax3(s[qb][0][0], s[qb][0][1], s[qb][0][2]);
            mx = vmax3(mx, s[qb][0][3], s[qb][1][0]);
            mx = vmax3(mx, s[qb][1][1], s[qb][1][2]);
            mx = vmax3(mx, s[qb][1][3], s[qb][2][0]);
            mx = vmax3(mx, s[qb][2][1], s[qb][2][2]);
            mx = vmax3(mx, s[qb][2][3], s[qb][3][0]);
            mx = vmax3(mx, s[qb][3][1], s[qb][3][2]);
            mx = vmax2(mx, s[qb][3][3]);
            mx = rows_max(mx);
            const float mnew = vmax2(mrun[qb], mx);
            const bool dead = MASKED && mnew == -INFINITY;                 // every key so far is masked
            const float alpha = dead ? 1.f : __builtin_amdgcn_exp2f(mrun[qb] - mnew);
            mrun[qb] = mnew;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (!MASKED || f < nfrag) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        s[qb][f][j] = dead ? 0.f : __builtin_amdgcn_exp2f(s[qb][f][j] - mnew);
                } else {
                    s[qb][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            if (!__all(alpha == 1.f)) {   // the running max moved for some query of this block
#pragma unroll
                for (int m = 0; m < 5; ++m) { o[qb][m][0] *= alpha; o[qb][m][1] *= alpha; o[qb][m][2] *= alpha; o[qb][m][3] *= alpha; }
            }
            const bf16x8 p0 = pack_bf16x8(s[qb][0], s[qb][1]);
#pragma unroll
            for (int m = 0; m < 4; ++m) o[qb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[m][0], p0, o[qb][m], 0, 0, 0);
            o[qb][4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p0, o[qb][4], 0, 0, 0);
            if (!MASKED || nfrag > 2) {
                const bf16x8 p1 = pack_bf16x8(s[qb][2], s[qb][3]);
#pragma unroll
                for (int m = 0; m < 4; ++m) o[qb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[m][1], p1, o[qb][m], 0, 0, 0);
                o[qb][4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p1, o[qb][4], 0, 0, 0);
            }
        }
    };

    // A full, unmasked key tile after the first one on the absolute scale (see the head of the file).
    constexpr float ABS_RANGE = 32.f, HEAD_ROOM = 64.f;
    bool lazy = false;          // wave-uniform: the accumulators are on the absolute scale (mrun == 0)
    auto process_fast = [&]() -> bool {
        static_assert(QB == 2, "the interleave below is written for two query blocks per wave");
        f32x4 s[QB][4];
        bf16x8 p[QB][2];
        float top = -INFINITY;
        // the two halves of the head dimension as separate sweeps over the eight fragments: an MFMA never
        // follows the one whose result it accumulates onto
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                s[qb][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[f][0], qf[qb][0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 8, 1);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                s[qb][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[f][1], qf[qb][1], s[qb][f], 0, 0, 0);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float mx = vmax3(s[qb][0][0], s[qb][0][1], s[qb][0][2]);
            mx = vmax3(mx, s[qb][0][3], s[qb][1][0]);
            mx = vmax3(mx, s[qb][1][1], s[qb][1][2]);
            mx = vmax3(mx, s[qb][1][3], s[qb][2][0]);
            mx = vmax3(mx, s[qb][2][1], s[qb][2][2]);
            mx = vmax3(mx, s[qb][2][3], s[qb][3][0]);
            mx = vmax3(mx, s[qb][3][1], s[qb][3][2]);
            top = vmax3(top, mx, s[qb][3][3]);
        }
        // a score beyond the head room: nothing has been touched yet; the caller leaves the absolute scale
        // and works this tile the classic way
        if (__builtin_expect(__any(top > HEAD_ROOM), 0)) return false;
        // probabilities of block 0, then the accumulation of block 0 on the matrix pipe beside the exponentials
        // of block 1 on the vector pipe, then the accumulation of block 1
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) s[qb][f][j] = __builtin_amdgcn_exp2f(s[qb][f][j]);
            p[qb][0] = pack_bf16x8(s[qb][0], s[qb][1]);
            p[qb][1] = pack_bf16x8(s[qb][2], s[qb][3]);
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
            for (int m = 0; m < 4; ++m) o[qb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[m][0], p[qb][0], o[qb][m], 0, 0, 0);
            o[qb][4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p[qb][0], o[qb][4], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m) o[qb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[m][1], p[qb][1], o[qb][m], 0, 0, 0);
            o[qb][4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p[qb][1], o[qb][4], 0, 0, 0);
        }
        // scheduling groups (masks: 0x2 vector ALU, 0x8 MFMA, 0x400 transcendental)
        __builtin_amdgcn_sched_group_barrier(0x400, 16, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, 8, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x8, 12, 0);
        return true;
    };
    // after the first tile: move to the absolute scale if every row maximum of the wave is moderate
    auto to_absolute_scale = [&]() {
        bool ok = true;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) ok = ok && fabsf(mrun[qb]) <= ABS_RANGE;
        if (__all(ok)) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const float up = __builtin_amdgcn_exp2f(mrun[qb]);
#pragma unroll
                for (int m = 0; m < 5; ++m) { o[qb][m][0] *= up; o[qb][m][1] *= up; o[qb][m][2] *= up; o[qb][m][3] *= up; }
                mrun[qb] = 0.f;
            }
            lazy = true;
        }
    };

#ifdef MARS_STAMPS
    unsigned long long c_read = 0, c_comp = 0, c_wait = 0;
#endif
    MARS_T(ts0)
    issue_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    MARS_T(ts1)
    // per tile: fragments into registers, the next tile's fill into the other buffer, compute, then wait for
    // the fill and meet at the barrier; every wave of the workgroup runs one head and one tail per tile
    // Fragment addresses inside a tile as a per-lane base plus a compile-time offset (the ds_read offset field).
    // K: row 16 f + r16, chunk g or 4 + g; the swizzle term ((row >> 1) & 7) = (r16 >> 1) & 7 does not depend on f.
    // V^T (load_vt_frag): rows 32 s + 4 g + q (+ 16), byte column 32 m + 8 p; the swizzle term is (2 g + (q >> 1)) & 7
    // for every s and both rows, so only the 16-byte chunk (2 m + (p >> 1)) ^ term depends on m.
    const int ksw = (r16 >> 1) & 7;
    const int kb0 = r16 * ROWB + ((g ^ ksw) << 4), kb1 = r16 * ROWB + (((4 + g) ^ ksw) << 4);
    int vb[4];
    {
        const int q = r16 >> 2, pp = r16 & 3, vsw = (2 * g + (q >> 1)) & 7;
#pragma unroll
        for (int m = 0; m < 4; ++m) vb[m] = (4 * g + q) * ROWB + (((2 * m + (pp >> 1)) ^ vsw) << 4) + 8 * (pp & 1);
    }
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
    auto tile_head = [&](int t) {
        const int buf = t & 1;
        MARS_T(ta)
        if (wave_active) {
            const unsigned char* kt0 = &lds[buf][0][0] + kb0;
            const unsigned char* kt1 = &lds[buf][0][0] + kb1;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                kf[f][0] = __builtin_bit_cast(bf16x8, *(const uint4*)(kt0 + f * 16 * ROWB));
                kf[f][1] = __builtin_bit_cast(bf16x8, *(const uint4*)(kt1 + f * 16 * ROWB));
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const unsigned char* vt_ = &lds[buf][1][0] + vb[m];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vt_ + ks * 32 * ROWB));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vt_ + ks * 32 * ROWB + 16 * ROWB));
                    const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    vf[m][ks] = __builtin_bit_cast(bf16x8, r);
                }
            }
        }
        // the fragments are in registers before the next fill starts: hipcc drains every in-flight
        // LDS-DMA (vmcnt(0)) in front of an LDS read that follows one
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        MARS_T(tb)
        MARS_ACC(c_read, ta, tb)
        if (t + 1 < nt_eff) issue_tile(t + 1, buf ^ 1);
    };
    auto tile_tail = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        MARS_T(tc)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        MARS_T(td)
        MARS_ACC(c_wait, tc, td)
    };
    for (int t = 0; t < nt_eff; ++t) {
        tile_head(t);
        if (wave_active) {
            if ((t * KT + KT > T) || causal) {
                process(t, std::true_type{});
            } else {
                bool done = false;
                if (lazy) {
                    // a run of full tiles on the absolute scale; left for good when a score exceeds the head room
                    while (true) {
                        if (!process_fast()) { lazy = false; break; }
                        if ((t + 2) * KT > T || t + 1 >= nt_eff) { done = true; break; }
                        tile_tail();
                        ++t;
                        tile_head(t);
                    }
                }
                if (!done) {
                    process(t, std::false_type{});
                    if (t == 0) to_absolute_scale();
                }
            }
        }
        tile_tail();
    }
#ifdef MARS_STAMPS
    if (lane == 0) {
        unsigned long long* sb = mars_attn_stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
        sb[4] = __builtin_amdgcn_s_memtime() - ts0;
        sb[0] = ts1 - ts0; sb[1] = c_read; sb[3] = c_wait; sb[2] = sb[4] - sb[0] - c_read - c_wait + c_comp;   // compute = the rest
        sb[5] = wave_active ? 1 : 0;
    }
#endif
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int myq = q0 + qb * 16 + r16;
        if (myq < T) {
            const float lsum = o[qb][4][0];
            const float inv = 1.0f / lsum;
            __bf16* op = O + ((int64_t)b * T + myq) * D + h * 64;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                bf16x4 v;
                v[0] = (__bf16)(o[qb][m][0] * inv); v[1] = (__bf16)(o[qb][m][1] * inv);
                v[2] = (__bf16)(o[qb][m][2] * inv); v[3] = (__bf16)(o[qb][m][3] * inv);
                *(bf16x4*)(op + 16 * m + 4 * g) = v;
            }
            // natural-log LSE of the scaled scores: (max + log2(sum)) * ln 2
            if (lse && g == 0) lse[(int64_t)bh * T + myq] = (mrun[qb] + log2f(lsum)) * 0.69314718055994530942f;
        }
    }
}

// The same attention for the image towers (no causal mask, more than one key tile), with the tile loop
// arranged so that a wave never waits for LDS while it has matrix work:
//
//   part 1 of tile t   scores (K fragments of t are in registers), range check, exponentials -> bf16 probabilities
//   meet               wait for the V fragments of t (requested during tile t-1) and this wave's share of the fill
//                      of tile t+1, workgroup barrier: tile t+1 is in LDS, nobody reads buffer t any more
//   part 2 of tile t   request the K fragments of t+1; accumulate P.V one 16-wide block of d at a time and request
//                      that block's V fragments of t+1 as soon as its MFMAs are issued; then start the fill of
//                      tile t+2 into buffer t
//
// 32 queries per wave (two blocks of 16), four waves, K/V tiles of 64 keys double buffered as in the kernel above.
// Full tiles after the first run on the absolute exp2 scale when the first tile allows it (head of the file);
// the first tile, the ragged last one and everything after a range-check failure use the running maximum.
//
// LEAD1: sequences of 64 k + 1 tokens (a class token in front of a grid of patches: 577 for a 24 x 24 grid).
// Tiled from key 0 they end in a tile with ONE key, which costs a whole trip round the loop - fill, barrier,
// masks, the classic softmax, twenty matrix instructions on zeros.  Instead key 0 becomes the INITIAL STATE of
// the online softmax, worked out on the vector ALU (a 64-long dot product per query: sixteen multiply-adds a
// lane and a sum over the four lane rows; the output starts as p0 v0), and the tiles cover keys 1 .. T-1, all full.
template <bool LEAD1>
__global__ __launch_bounds__(256, 2) void flash_attn_pipe_kernel(const __bf16* __restrict__ qkv, __bf16* __restrict__ O,
                                                                 float* __restrict__ lse, int T, int H, float c2) {
    constexpr int QB = 2, NW = 4, QPB = NW * QB * 16, PPW = 16 / NW;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][KT * ROWB];  // [buf][K|V]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r16 = lane & 15;
    const int D = H * 64;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int q0 = blockIdx.x * QPB + wave * (QB * 16);
    const int64_t row_b = (int64_t)3 * D * 2;
    const unsigned row_b24 = (unsigned)row_b;
    const unsigned char* base = (const unsigned char*)qkv + (int64_t)b * T * row_b + (int64_t)h * 128;
    const bool wave_active = q0 < T;            // trailing waves of the last workgroup only help with the fills

    bf16x8 qf[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = min(q0 + qb * 16 + r16, T - 1);
        const unsigned char* qp = base + (int64_t)qrow * row_b;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 raw = __builtin_bit_cast(bf16x8, *(const uint4*)(qp + 64 * ks + 16 * g));
            if (c2 == 1.0f) {
                qf[qb][ks] = raw;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[qb][ks][e] = (__bf16)((float)raw[e] * c2);
            }
        }
    }
    const unsigned char* psrc[PPW];
    int prow[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave * PPW + i;
        prow[i] = 8 * (p & 7) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((prow[i] >> 1) & 7);
        psrc[i] = base + (int64_t)D * 2 * (1 + (p >> 3)) + chunk * 16;
    }
    auto issue_tile = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave * PPW + i;
            const int key = min((LEAD1 ? 1 : 0) + t * KT + prow[i], T - 1);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(psrc[i] + (((unsigned)key & 0xffffffu) * (row_b24 & 0xffffffu))),
                                             (lds_ptr_t)(&lds[buf][p >> 3][(p & 7) * 1024]), 16, 0, 0);
        }
    };
    f32x4 o[QB][5];
    float mrun[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        mrun[qb] = -INFINITY;
#pragma unroll
        for (int m = 0; m < 5; ++m) o[qb][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    const int TK = LEAD1 ? T - 1 : T;                  // keys the tiles cover (LEAD1: a multiple of KT)
    const int nt = (TK + KT - 1) / KT;

    // fragment addresses: per-lane base plus compile-time offsets (see the kernel above)
    const int ksw = (r16 >> 1) & 7;
    const int kb0 = r16 * ROWB + ((g ^ ksw) << 4), kb1 = r16 * ROWB + (((4 + g) ^ ksw) << 4);
    int vb[4];
    {
        const int q = r16 >> 2, pp = r16 & 3, vsw = (2 * g + (q >> 1)) & 7;
#pragma unroll
        for (int m = 0; m < 4; ++m) vb[m] = (4 * g + q) * ROWB + (((2 * m + (pp >> 1)) ^ vsw) << 4) + 8 * (pp & 1);
    }
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
    bf16x8 kf[4][2], vf[4][2], p[QB][2];
    auto read_k = [&](int buf) {
        const unsigned char* kt0 = &lds[buf][0][0] + kb0;
        const unsigned char* kt1 = &lds[buf][0][0] + kb1;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            kf[f][0] = __builtin_bit_cast(bf16x8, *(const uint4*)(kt0 + f * 16 * ROWB));
            kf[f][1] = __builtin_bit_cast(bf16x8, *(const uint4*)(kt1 + f * 16 * ROWB));
        }
    };
    auto read_v = [&](int buf, int m) {
        const unsigned char* vt_ = &lds[buf][1][0] + vb[m];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vt_ + ks * 32 * ROWB));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vt_ + ks * 32 * ROWB + 16 * ROWB));
            const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            vf[m][ks] = __builtin_bit_cast(bf16x8, r);
        }
    };

    constexpr float ABS_RANGE = 32.f, HEAD_ROOM = 64.f;
    bool lazy = false;          // wave-uniform: the accumulators are on the absolute scale (mrun == 0)

    // part 1 on the absolute scale; false (nothing touched) when a score exceeds the head room
    auto part1_fast = [&]() -> bool {
        f32x4 s[QB][4];
        float top = -INFINITY;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                s[qb][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[f][0], qf[qb][0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 8, 1);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                s[qb][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[f][1], qf[qb][1], s[qb][f], 0, 0, 0);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float mx = vmax3(s[qb][0][0], s[qb][0][1], s[qb][0][2]);
            mx = vmax3(mx, s[qb][0][3], s[qb][1][0]);
            mx = vmax3(mx, s[qb][1][1], s[qb][1][2]);
            mx = vmax3(mx, s[qb][1][3], s[qb][2][0]);
            mx = vmax3(mx, s[qb][2][1], s[qb][2][2]);
            mx = vmax3(mx, s[qb][2][3], s[qb][3][0]);
            mx = vmax3(mx, s[qb][3][1], s[qb][3][2]);
            top = vmax3(top, mx, s[qb][3][3]);
        }
        if (__builtin_expect(__any(top > HEAD_ROOM), 0)) return false;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) s[qb][f][j] = __builtin_amdgcn_exp2f(s[qb][f][j]);
            p[qb][0] = pack_bf16x8(s[qb][0], s[qb][1]);
            p[qb][1] = pack_bf16x8(s[qb][2], s[qb][3]);
        }
        return true;
    };
    // part 1 with the running maximum.  FIRST: the accumulators are empty; if every row maximum of the wave is
    // moderate the reference becomes 0 and the wave turns lazy.  MASKED: the ragged last tile.
    auto part1_classic = [&](int t, auto masked_tag, auto first_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value, FIRST = decltype(first_tag)::value;
        const int nfrag = MASKED ? min(4, (T - t * KT + 15) >> 4) : 4;     // fragments with a key < T
        f32x4 s[QB][4];
        float mx[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (!MASKED || f < nfrag) {
                    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[f][0], qf[qb][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[f][1], qf[qb][1], acc, 0, 0, 0);
                    s[qb][f] = acc;
                } else {
                    s[qb][f] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
            }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if constexpr (MASKED) {
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (t * KT + 16 * f + 4 * g + j >= T) s[qb][f][j] = -INFINITY;
            }
            float m_ = vmax3(s[qb][0][0], s[qb][0][1], s[qb][0][2]);
            m_ = vmax3(m_, s[qb][0][3], s[qb][1][0]);
            m_ = vmax3(m_, s[qb][1][1], s[qb][1][2]);
            m_ = vmax3(m_, s[qb][1][3], s[qb][2][0]);
            m_ = vmax3(m_, s[qb][2][1], s[qb][2][2]);
            m_ = vmax3(m_, s[qb][2][3], s[qb][3][0]);
            m_ = vmax3(m_, s[qb][3][1], s[qb][3][2]);
            mx[qb] = rows_max(vmax2(m_, s[qb][3][3]));
        }
        if constexpr (FIRST && !MASKED) {
            bool ok = true;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) ok = ok && fabsf(mx[qb]) <= ABS_RANGE;
            if (__all(ok)) {
                lazy = true;
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) mx[qb] = 0.f;          // the reference of the absolute scale
            }
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float mnew = FIRST ? mx[qb] : vmax2(mrun[qb], mx[qb]);
            if constexpr (!FIRST) {
                const float alpha = __builtin_amdgcn_exp2f(mrun[qb] - mnew);
                if (!__all(alpha == 1.f)) {   // the running max moved for some query of this block
#pragma unroll
                    for (int m = 0; m < 5; ++m) { o[qb][m][0] *= alpha; o[qb][m][1] *= alpha; o[qb][m][2] *= alpha; o[qb][m][3] *= alpha; }
                }
            }
            mrun[qb] = mnew;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (!MASKED || f < nfrag) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[qb][f][j] = __builtin_amdgcn_exp2f(s[qb][f][j] - mnew);
                } else {
                    s[qb][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            p[qb][0] = pack_bf16x8(s[qb][0], s[qb][1]);
            p[qb][1] = pack_bf16x8(s[qb][2], s[qb][3]);
        }
    };
    // meet + part 2 (see the head of this kernel)
    auto tile_finish = [&](int t) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        const bool more = t + 1 < nt;
        const int nbuf = (t + 1) & 1;
        if (wave_active) {
            if (more) read_k(nbuf);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                o[0][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[m][0], p[0][0], o[0][m], 0, 0, 0);
                o[1][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[m][0], p[1][0], o[1][m], 0, 0, 0);
                o[m & 1][4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p[m & 1][m >> 1], o[m & 1][4], 0, 0, 0);
                o[0][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[m][1], p[0][1], o[0][m], 0, 0, 0);
                o[1][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[m][1], p[1][1], o[1][m], 0, 0, 0);
                if (more) read_v(nbuf, m);
            }
        }
        if (t + 2 < nt) issue_tile(t + 2, t & 1);
    };

    // prologue: tile 0 in LDS, its fragments requested, tile 1 on its way
    issue_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave_active) {
        read_k(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) read_v(0, m);
    }
    if (nt > 1) issue_tile(1, 1);
    if constexpr (LEAD1) {
        if (wave_active) {
            // key 0 as the initial state: s0 = q . k0 in the exp2 domain (the query fragments carry the scale)
            const unsigned char* k0p = base + (int64_t)D * 2 + 16 * g;
            const bf16x8 ka = __builtin_bit_cast(bf16x8, *(const uint4*)k0p);
            const bf16x8 kb = __builtin_bit_cast(bf16x8, *(const uint4*)(k0p + 64));
            float s0[QB];
            bool ok = true;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float part = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    part = fmaf((float)qf[qb][0][e], (float)ka[e], part);
                    part = fmaf((float)qf[qb][1][e], (float)kb[e], part);
                }
                s0[qb] = rows_sum(part);
                ok = ok && fabsf(s0[qb]) <= ABS_RANGE;
            }
            lazy = __all(ok);                           // moderate scores: the reference of the absolute scale is 0
            bf16x4 v0[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) v0[m] = *(const bf16x4*)(base + (int64_t)D * 4 + 32 * m + 8 * g);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                mrun[qb] = lazy ? 0.f : s0[qb];
                const float p0 = lazy ? __builtin_amdgcn_exp2f(s0[qb]) : 1.f;
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[qb][m][j] = p0 * (float)v0[m][j];
                o[qb][4] = (f32x4){p0, p0, p0, p0};
            }
        }
    } else {
        if (wave_active) part1_classic(0, std::false_type{}, std::true_type{});
        tile_finish(0);
    }
    int t = LEAD1 ? 0 : 1;
    while (t < nt) {
        const bool ragged = t * KT + KT > TK;
        if (wave_active) {
            if (lazy && !ragged) {
                // a run of full tiles on the absolute scale
                bool fired = false;
                while (true) {
                    if (!part1_fast()) { fired = true; break; }
                    tile_finish(t);
                    ++t;
                    if (t >= nt || t * KT + KT > TK) break;
                }
                if (!fired) continue;
                lazy = false;
            }
            if (ragged) part1_classic(t, std::true_type{}, std::false_type{});
            else part1_classic(t, std::false_type{}, std::false_type{});
        }
        tile_finish(t);
        ++t;
    }

#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int myq = q0 + qb * 16 + r16;
        if (myq < T) {
            const float lsum = o[qb][4][0];
            const float inv = 1.0f / lsum;
            __bf16* op = O + ((int64_t)b * T + myq) * D + h * 64;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                bf16x4 v;
                v[0] = (__bf16)(o[qb][m][0] * inv); v[1] = (__bf16)(o[qb][m][1] * inv);
                v[2] = (__bf16)(o[qb][m][2] * inv); v[3] = (__bf16)(o[qb][m][3] * inv);
                *(bf16x4*)(op + 16 * m + 4 * g) = v;
            }
            if (lse && g == 0) lse[(int64_t)bh * T + myq] = (mrun[qb] + log2f(lsum)) * 0.69314718055994530942f;
        }
    }
}

// acc[i][j] += scale_out * sum_h exp(q_h[off+i].k_h[off+j]/8 - lse[h][off+i])
// NF = 16-key fragments per wave: 4 (64 keys a workgroup) or 2 (32 keys: twice the waves for the same product)
template <int NF>
__global__ __launch_bounds__(256) void flash_attn_mean_kernel(const __bf16* __restrict__ qkv, const float* __restrict__ lse,
                                                              int T, int H, int off, int Pn, float scale_out,
                                                              float* __restrict__ acc_out, int64_t ldacc, float c2) {
    // the K tile, the query fragments and the log-sum-exp of head h + 1 are requested before head h is worked on
    // (two LDS buffers, one barrier per head): a head is 8 MFMAs and 16 exponentials per lane, far less than a
    // trip to memory
    static_assert(NF == 2 || NF == 4, "32 or 64 keys per workgroup");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][16 * NF * ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    const int D = H * 64;
    const int64_t row_b = (int64_t)3 * D * 2;
    const int i0 = blockIdx.y * 64 + wave * 16, j0 = blockIdx.x * (16 * NF);
    const int myi = i0 + r16;
    const int qrow = off + min(myi, Pn - 1);
    const unsigned char* base = (const unsigned char*)qkv;
    constexpr float LOG2E = 1.44269504088896340736f;
    f32x4 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int kr[2], kc[2];
    int64_t koff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int qd = tid + 256 * i;
        kr[i] = qd >> 3;
        kc[i] = qd & 7;
        koff[i] = (int64_t)D * 2 + (int64_t)(off + min(j0 + kr[i], Pn - 1)) * row_b + kc[i] * 16;
    }
    // (plain variables and a macro: behind a lambda the compiler keeps the fetched registers in scratch memory and
    // waits for the loads at once)
    uint4 kreg0, kreg1, qreg0, qreg1;
    float lreg;
#define MARS_FETCH_HEAD(h_)                                                   \
    {                                                                         \
        const unsigned char* hb_ = base + (int64_t)(h_) * 128;                \
        const unsigned char* qp_ = hb_ + (int64_t)qrow * row_b;               \
        qreg0 = *(const uint4*)(qp_ + 16 * g);                                \
        qreg1 = *(const uint4*)(qp_ + 64 + 16 * g);                           \
        lreg = lse[(int64_t)(h_) * T + qrow];                                 \
        kreg0 = *(const uint4*)(hb_ + koff[0]);                               \
        if constexpr (NF == 4) kreg1 = *(const uint4*)(hb_ + koff[1]);        \
    }
    MARS_FETCH_HEAD(0)
    *(uint4*)(&lds[0][tile_off(kr[0], kc[0])]) = kreg0;
    if constexpr (NF == 4) *(uint4*)(&lds[0][tile_off(kr[1], kc[1])]) = kreg1;
    __syncthreads();
    for (int h = 0; h < H; ++h) {
        // the queries exactly as the attention kernel formed them (scaled by c2 and rounded, or as stored)
        bf16x8 qf0 = __builtin_bit_cast(bf16x8, qreg0), qf1 = __builtin_bit_cast(bf16x8, qreg1);
        if (c2 != 1.0f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { qf0[e] = (__bf16)((float)qf0[e] * c2); qf1[e] = (__bf16)((float)qf1[e] * c2); }
        }
        const float l = lreg * LOG2E;
        if (h + 1 < H) MARS_FETCH_HEAD(h + 1)
        const unsigned char* kt = lds[h & 1];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const bf16x8 k0 = __builtin_bit_cast(bf16x8, *(const uint4*)(kt + tile_off(16 * f + r16, g)));
            const bf16x8 k1 = __builtin_bit_cast(bf16x8, *(const uint4*)(kt + tile_off(16 * f + r16, 4 + g)));
            f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf0, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf1, s, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[f][j] += __builtin_amdgcn_exp2f(s[j] - l);
        }
        if (h + 1 < H) {
            *(uint4*)(&lds[(h + 1) & 1][tile_off(kr[0], kc[0])]) = kreg0;
            if constexpr (NF == 4) *(uint4*)(&lds[(h + 1) & 1][tile_off(kr[1], kc[1])]) = kreg1;
        }
        __syncthreads();   // the next head's tile is in place; everybody is done with this one before it is overwritten
    }
#undef MARS_FETCH_HEAD
    if (myi < Pn) {
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = j0 + 16 * f + 4 * g + j;
                if (col < Pn) acc_out[(int64_t)myi * ldacc + col] += scale_out * acc[f][j];
            }
    }
}

}  // namespace

void mars_flash_attn32_launch(const void* qkv, void* O, float* lse, int B, int T, int H, float score_scale_log2, hipStream_t stream);

// Image towers (no causal mask, more than three key tiles) run the 32x32x16 kernel of attention32.hip.  A/B runs:
// MARS_FLASH_VARIANT=4 keeps them on flash_attn_pipe_kernel (16x16x32, one workgroup per query block), =2 the same
// with class-token sequences tiled from key 0, =1 on the plain tile loop.
static int flash_variant() {
    static const int v = [] { const char* e = getenv("MARS_FLASH_VARIANT"); return e ? atoi(e) : 0; }();
    return v;
}

extern "C" int marship_flash_attn(const void* qkv, void* O, float* lse, int B, int T, int H, int causal,
                                  float score_scale_log2, void* stream) {
    MARS_CHECK(qkv && O, "flash_attn: null pointer");
    MARS_CHECK(score_scale_log2 > 0.f && score_scale_log2 < 1e6f, "flash_attn: score scale must be positive and finite");
    MARS_CHECK(B > 0 && T > 0 && H > 0 && (int64_t)B * H <= 65535, "flash_attn: bad shape B=%d T=%d H=%d", B, T, H);
    // the kernel forms byte offsets inside one image's rows with a 24 x 24 bit multiply into 32 bits
    MARS_CHECK(T < (1 << 24) && (int64_t)H * 384 < (1 << 24) && (int64_t)T * H * 384 < (1LL << 32),
               "flash_attn: one image's qkv rows (T=%d x %d bytes) must stay below 4 GiB", T, H * 384);
    MARS_CHECK(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)O % 8) == 0, "flash_attn: misaligned buffers");
    if (T <= 96) {   // short (text) sequences: 32 queries per wave, 2 waves
        dim3 grid((T + 63) / 64, B * H), block(128);
        hipLaunchKernelGGL((flash_attn_kernel<2, 2>), grid, block, 0, (hipStream_t)stream, (const __bf16*)qkv, (__bf16*)O, lse, T, H, causal, score_scale_log2);
    } else if (causal || flash_variant() == 1) {   // 32 queries per wave, 4 waves: 128 queries per workgroup
        dim3 grid((T + 127) / 128, B * H), block(256);
        hipLaunchKernelGGL((flash_attn_kernel<2, 4>), grid, block, 0, (hipStream_t)stream, (const __bf16*)qkv, (__bf16*)O, lse, T, H, causal, score_scale_log2);
    } else if (flash_variant() == 0 && T > 3 * KT) {   // image towers on the 32x32x16 matrix instruction (attention32.hip)
        mars_flash_attn32_launch(qkv, O, lse, B, T, H, score_scale_log2, (hipStream_t)stream);
    } else {         // image towers: the same tiling with the pipelined tile loop
        dim3 grid((T + 127) / 128, B * H), block(256);
        if (T % KT == 1 && flash_variant() != 2)   // a class token in front of whole tiles of patches (MARS_FLASH_VARIANT=2: tiled from key 0)
            hipLaunchKernelGGL(flash_attn_pipe_kernel<true>, grid, block, 0, (hipStream_t)stream, (const __bf16*)qkv, (__bf16*)O, lse, T, H, score_scale_log2);
        else
            hipLaunchKernelGGL(flash_attn_pipe_kernel<false>, grid, block, 0, (hipStream_t)stream, (const __bf16*)qkv, (__bf16*)O, lse, T, H, score_scale_log2);
    }
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_flash_attn_mean(const void* qkv, const float* lse, int T, int H, int off, int Pn, float scale,
                                       float* acc, int64_t ldacc, float score_scale_log2, void* stream) {
    MARS_CHECK(score_scale_log2 > 0.f && score_scale_log2 < 1e6f, "flash_attn_mean: score scale must be positive and finite");
    MARS_CHECK(qkv && lse && acc, "flash_attn_mean: null pointer");
    MARS_CHECK(T > 0 && H > 0 && off >= 0 && Pn > 0 && off + Pn <= T && ldacc >= Pn, "flash_attn_mean: bad shape");
    MARS_CHECK(((uintptr_t)qkv % 16) == 0, "flash_attn_mean: misaligned qkv");
    // 64 keys a workgroup; the 32-key form (twice the waves) measured 25.1 against 22.9 us a launch on the DINOv2-L maps
    dim3 grid((Pn + 63) / 64, (Pn + 63) / 64), block(256);
    hipLaunchKernelGGL(flash_attn_mean_kernel<4>, grid, block, 0, (hipStream_t)stream, (const __bf16*)qkv, lse, T, H, off, Pn, scale, acc, ldacc, score_scale_log2);
    MARS_LAUNCH_CHECK();
    return 0;
}
