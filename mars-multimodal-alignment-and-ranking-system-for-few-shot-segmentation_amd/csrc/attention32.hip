// Fused attention for 64-wide heads on v_mfma_f32_32x32x16_bf16 (image towers: no causal mask, more than one key
// tile).  Same arithmetic as flash_attn_pipe_kernel (attention.hip) - exp2-domain scores, absolute scale where the
// scores allow it, class-token form - on the larger matrix instruction:
//
//   a wave owns 32 queries.  S^T = K Q^T per 32 keys is four instructions (one per 16 of the 64 head dims): the
//   query is the LANE column (lane & 31) and a lane holds 16 of the 32 keys, the other half-wave the other 16, so
//   row statistics are per-lane partials that meet once, at the end (sums) or through one half-wave swap (maxima).
//   The exponentiated scores are, as they stand, the B operand of O^T += V^T P^T: registers 8u .. 8u+7 of a key
//   block are k-step u, whose keys are 16u + 8(e>>2) + 4(lane>>5) + (e&3); the A operand V^T[d][those keys] comes
//   from the row-major V tile through two ds_read_b64_tr_b16.
//
// Against the 16x16x32 form a matrix instruction blocks the SIMD's vector issue for 8 of 32 cycles instead of 8
// of 16, there are 16 instead of 36 of them per wave and tile, and the row sums are plain adds (no product with a
// block of ones); the kernel is built for three waves per SIMD.
//
// K/V tiles of 64 keys, double buffered, filled by LDS-DMA with the bank swizzle on the source side.  K rows are
// read by ds_read_b128 (16-byte chunk c of row r at slot c ^ ((r >> 1) & 7)); V rows by transposed reads whose
// half-wave covers rows k0 .. k0+3 in four chunks: chunk c of row r at slot c ^ (((r >> 1) & 1) << 2), so that
// rows two apart (same bank half) use the other four slots.
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>
#include "common.h"
#include "../../include/mars_hip.h"

// Diagnostic builds (tools/diag_attn32.hip) define MARS_STAMPS and collect per-wave cycle counts of the kernel's
// phases; in the library build the macros expand to nothing.
#ifdef MARS_STAMPS
__device__ unsigned long long* mars_attn32_stamps;
#define MARS_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime();
#define MARS_ACC(dst, a, b) dst += (b) - (a);
#else
#define MARS_T(var)
#define MARS_ACC(dst, a, b)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef int v2i32 __attribute__((ext_vector_type(2)));
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

constexpr int KT = 64;           // keys per tile
constexpr int ROWB = 128;        // bytes per LDS row (64 bf16)

__device__ __forceinline__ float vmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// x of this lane combined with x of the same lane in the other half-wave: v_permlane32_swap on (x, x) leaves the
// lower half's values in every lane of its first result and the upper half's in the second
__device__ __forceinline__ float halves_sum(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float halves_max(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& s, int u) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (__bf16)s[8 * u + e];
    return r;
}

// Workgroup -> (image-head, query block).  The query blocks of one image-head read the same K and V; dispatched in
// launch order (consecutive workgroups go to consecutive XCDs, each with an L2 of its own) they would fetch them into
// nqb different L2s.  Instead the workgroups that share an XCD (equal id mod 8) walk the query blocks of one
// image-head after the other: K and V are fetched once and found in that L2 by the other blocks.  Image-heads
// beyond the last full group of eight keep the launch order.  (Placement is a matter of speed only.)
__device__ __forceinline__ void attn_block(int id, int nbh, int nqb, int xcd_aware, int& bh, int& qb) {
    const int full = (nbh >> 3) * 8 * nqb;
    if (xcd_aware && id < full) {
        const int x = id & 7, j = id >> 3;
        bh = (j / nqb) * 8 + x;
        qb = j % nqb;
    } else {
        bh = id / nqb;
        qb = id % nqb;
    }
}

template <bool LEAD1>
__global__ __launch_bounds__(256, 2) void flash_attn_w32_kernel(const __bf16* __restrict__ qkv, __bf16* __restrict__ O,
                                                                  float* __restrict__ lse, int T, int H, float c2, int nbh, int xcd_aware,
                                                                  unsigned* __restrict__ shared_count) {
    constexpr int NW = 4, QPB = NW * 32, PPW = 16 / NW;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][3][KT * ROWB];     // [K|V][slot]: rings of three tiles
    __shared__ __attribute__((aligned(16))) unsigned char stage_q[NW][32 * ROWB];   // the next item's queries, a wave's 32 rows each
    __shared__ __attribute__((aligned(16))) unsigned char stage_kv[1024];           // its key 0 (bytes 0-127) and value 0 (128-255)
    __shared__ __attribute__((aligned(32))) int xchg[8];   // after an item: [wave] whether the wave wants it worked again, [4] the item wave 0 took from the shared counter
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    const int D = H * 64;
#ifdef MARS_STAMPS
    unsigned long long st_compute = 0, st_wait = 0, st_top = 0, st_fin = 0, st_fill = 0, st_tail = 0, st_vote = 0, st_topwait = 0;
#endif
    MARS_T(t_begin)
    const int64_t row_b = (int64_t)3 * D * 2;
    const unsigned row_b24 = (unsigned)row_b;
    const int nqb = (T + QPB - 1) / QPB, nitems = nbh * nqb;
    const int TK = LEAD1 ? T - 1 : T;           // keys the tiles cover (LEAD1: a multiple of KT)
    const int nt = (TK + KT - 1) / KT;
    const bool ragged_last = nt * KT > TK;
    // Absolute scale: taken when the first scores of every query of the wave are within +-ABS_RANGE (a row then
    // holds a probability of at least 2^-32 and cannot underflow to an empty row); left - the item is worked again
    // with the running maximum - when a row sum reaches SUM_LIMIT = 2^64, i.e. before a probability could exceed
    // what the products with V (|v| < 2^57) and the sums hold in float.
    constexpr float ABS_RANGE = 32.f, SUM_LIMIT = 18446744073709551616.f;
    constexpr int EARLY = 8;                    // exponentials of a tile issued while its V (and the next K) fragments are on their way

    // A workgroup walks the items (image-head, block of 128 queries) id = blockIdx.x, + gridDim.x, ...  While it works
    // on one, the next one's queries (and key 0 / value 0) are brought into LDS by LDS-DMA and the next one's first K
    // and V tiles follow the current item's last ones through the ring, so that an item starts from LDS, not from
    // a round trip to memory (the start of an item was a third of its time when every item began with one).
    auto item_base = [&](int id, int& qblk) -> const unsigned char* {
        int bh;
        attn_block(id, nbh, nqb, xcd_aware, bh, qblk);
        return (const unsigned char*)qkv + (int64_t)(bh / H) * T * row_b + (int64_t)(bh % H) * 128;
    };
    const int lrow = lane >> 3;
    auto stage_item = [&](const unsigned char* ib, int iq) {
        const int q0n = iq * QPB + wave * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + lrow;
            const int qrow = min(q0n + row, T - 1);
            const unsigned off = ((unsigned)qrow & 0xffffffu) * (row_b24 & 0xffffffu) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ib + off), (lds_ptr_t)(&stage_q[wave][i * 1024]), 16, 0, 0);
        }
        if (LEAD1 && wave == 0) {               // lanes 0-7: key 0, lanes 8-15: value 0 (the other lanes' 16 bytes are not read)
            const unsigned off = (unsigned)D * 2u * (1 + ((lane >> 3) & 1)) + ((lane & 7) << 4);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ib + off), (lds_ptr_t)stage_kv, 16, 0, 0);
        }
    };

    // Fills.  A tile's K and its V are eight 1-KiB pieces each (8 rows; a lane brings 16 bytes of row lane / 8 of
    // the piece).  Waves 0 and 1 fill K tiles, waves 2 and 3 V tiles, four pieces each.
    const int is_v = wave >> 1;                 // wave-uniform
    unsigned poff[2];                           // byte offset of this lane's 16 bytes from key 0's row, without the key: even / odd pieces
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * i + lrow;           // (the swizzles repeat every 16 rows)
        const int sw = is_v ? (((row >> 1) & 1) << 2) : ((row >> 1) & 7);
        poff[i] = (unsigned)D * 2u * (1 + is_v) + (((lane & 7) ^ sw) << 4);
    }
    // tile t of the item at `ib` into ring slot `slot` (K for waves 0 and 1, V for waves 2 and 3)
    auto fill_tile = [&](const unsigned char* ib, int t, int slot) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = (wave & 1) * PPW + i;
            const int key = min((LEAD1 ? 1 : 0) + t * KT + 8 * piece + lrow, T - 1);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ib + (((unsigned)key & 0xffffffu) * (row_b24 & 0xffffffu) + poff[i & 1])),
                                             (lds_ptr_t)(&lds[is_v][slot][piece * 1024]), 16, 0, 0);
        }
    };

    f32x16 o[2];
    float mrun = -INFINITY, lsum = 0.f;         // lsum: this lane's part of the row sum (its half of the keys)

    // fragment addresses inside a tile (ka also addresses the staged queries: same rows, same chunks, same swizzle)
    int ka[4], va[2];
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[s] = r32 * ROWB + (((2 * s + hh) ^ ((r32 >> 1) & 7)) << 4);
    {
        const int i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int db = 0; db < 2; ++db)
            va[db] = (4 * hh + q) * ROWB + (((4 * db + 2 * g1 + (pp >> 1)) ^ ((q >> 1) << 2)) << 4) + 8 * (pp & 1);
    }
    // LDS reads are inline asm on purpose: hipcc treats an LDS-DMA in flight as a pending LDS write and puts
    // s_waitcnt vmcnt(0) in front of every LDS read it can see behind one, which would drain the fills that are meant
    // to stay in flight across iterations.  Whoever issues these reads waits for them (s_waitcnt lgkmcnt) and pins
    // their uses behind the wait (sched_barrier).
    bf16x8 qf[4], kf[2][4];                     // qf[s]: B operand of step s, Q[query r32][d = 16 s + 8 hh + j]; kf[kb][s]
    v2i32 vlo[2][2][2], vhi[2][2][2];           // V^T fragment [db][kb][u]: keys 0-3 / 4-7 of its eight k slots
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)&lds[0][0][0];
    auto read_k = [&](int slot) {
        const unsigned kt = lds_base + slot * (KT * ROWB);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                uint4 r;
                if (kb == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(kt + ka[s]));
                else asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(r) : "v"(kt + ka[s]));
                kf[kb][s] = __builtin_bit_cast(bf16x8, r);
            }
    };
    auto read_v = [&](int slot) {
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const unsigned vt = lds_base + (3 + slot) * (KT * ROWB) + va[db];
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vlo[db][0][0]) : "v"(vt));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(vhi[db][0][0]) : "v"(vt));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(vlo[db][0][1]) : "v"(vt));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:3072" : "=v"(vhi[db][0][1]) : "v"(vt));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(vlo[db][1][0]) : "v"(vt));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:5120" : "=v"(vhi[db][1][0]) : "v"(vt));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(vlo[db][1][1]) : "v"(vt));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:7168" : "=v"(vhi[db][1][1]) : "v"(vt));
        }
    };
    auto vfrag = [&](int db, int kb, int u) -> bf16x8 {
        const v4i32 r = {vlo[db][kb][u][0], vlo[db][kb][u][1], vhi[db][kb][u][0], vhi[db][kb][u][1]};
        return __builtin_bit_cast(bf16x8, r);
    };
    // the workgroup's barrier without the drain of vector memory that __syncthreads() puts in front of it
    auto meet = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto lds_wait = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto scores = [&](f32x16 (&s)[2]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 acc = (f32x16)(0.f);
#pragma unroll
            for (int st = 0; st < 4; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][st], qf[st], acc, 0, 0, 0);
            s[kb] = acc;
        }
    };
    // keys of a ragged last tile beyond the sequence get a score of -inf (probability 0)
    auto mask_tail = [&](f32x16 (&s)[2], int t) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t * KT + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hh >= TK) s[kb][r] = -INFINITY;
    };
    auto tile_max = [&](const f32x16 (&s)[2]) -> float {
        float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;       // four chains: the maxima are latency bound
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            m0 = vmax3(m0, s[kb][0], s[kb][1]);   m1 = vmax3(m1, s[kb][2], s[kb][3]);
            m2 = vmax3(m2, s[kb][4], s[kb][5]);   m3 = vmax3(m3, s[kb][6], s[kb][7]);
            m0 = vmax3(m0, s[kb][8], s[kb][9]);   m1 = vmax3(m1, s[kb][10], s[kb][11]);
            m2 = vmax3(m2, s[kb][12], s[kb][13]); m3 = vmax3(m3, s[kb][14], s[kb][15]);
        }
        return fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    };

    // ---- the state of the item in hand (all wave-uniform) ----
    const unsigned char* base = nullptr;        // its image-head's qkv rows
    const unsigned char* nbase = nullptr;       // the next item's, or null
    const unsigned char* fbase = nullptr;       // where fills beyond the item's last tile read: nbase, or base if there is no next item
    int qblk = 0, nqblk = 0;
    int g = 0;                                  // tiles this workgroup has been through: tile t of the item lives in ring slot (g + t) % 3
    bool wave_active = false;

    // The tile loop is software pipelined over three tiles: iteration t turns the scores of tile t (matrix results of
    // the iteration before, in x) into probabilities e on the vector ALU while the matrix pipe first accumulates
    // P(t-1) V(t-1) and then works out the scores of tile t+1 into x again; the probabilities are packed into p for
    // the iteration after.  x is only ever written by the matrix pipe and e by the vector ALU, so no set of registers
    // changes its role from one iteration to the next.  At the top K(t+1) and V(t-1) are in LDS and their fragments
    // are requested; the fills of K(t+3) (into the slot K(t) had: its fragments were read one iteration ago) and of
    // V(t+1) (into the slot of V(t-2)) are started right behind and have two iterations to land - beyond the item's
    // last tile they are the next item's K tiles 0-2 and V tile 0; one barrier per tile, at the bottom.  The last
    // product P(nt-1) V(nt-1) follows the loop.
    //
    // MODE 0: this wave only helps with fills and barriers.  MODE 1: absolute scale (reference 0); returns true if
    // some score exceeded the head room (the accumulators are then worthless and the workgroup runs the pass again
    // with the running maximum).  MODE 2: running maximum.
    auto pass = [&](f32x16 (&x)[2], auto mode_tag) -> bool {
        constexpr int MODE = decltype(mode_tag)::value;
        f32x16 e[2];
        bf16x8 p[2][2];
        auto pv = [&]() {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(0, kb, u), p[kb][u], o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(1, kb, u), p[kb][u], o[1], 0, 0, 0);
                }
        };
        // one iteration; FIRST: tile 0 (no product of a tile before); LAST: tile nt-1 (the only one that can be
        // ragged; no scores of a next tile)
        auto iteration = [&](int t, auto first_tag, auto last_tag) {
            constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
            MARS_T(t_top)
            if constexpr (MODE != 0) {
                if constexpr (!FIRST) read_v((g + t + 2) % 3);          // tile t-1
                if constexpr (!LAST) read_k((g + t + 1) % 3);
            }
            {
                // (without a next item the tiles beyond the last are this item's first ones again: nobody reads
                // them, and the loop stays free of branches)
                const int tf = is_v ? t + 1 : t + 3;
                const bool own = tf < nt;
                fill_tile(own ? base : fbase, own ? tf : tf - nt, (g + tf) % 3);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE != 0) {
                // while the fragments are on their way: what needs the scores of tile t only
                if constexpr (LAST) {
                    if (ragged_last) mask_tail(x, t);
                }
                float ref = 0.f, alpha = 1.f;
                if constexpr (MODE == 2) {
                    const float mnew = fmaxf(mrun, halves_max(tile_max(x)));
                    alpha = __builtin_amdgcn_exp2f(mrun - mnew);
                    mrun = mnew;
                    ref = mnew;
                }
#pragma unroll
                for (int r = 0; r < EARLY; ++r) e[0][r] = __builtin_amdgcn_exp2f(MODE == 1 ? x[0][r] : x[0][r] - ref);
                lds_wait();
                // P(t-1) V(t-1), then the scores of tile t+1, on the matrix pipe ...
                if constexpr (!FIRST) pv();
                if constexpr (MODE == 2) {
                    if (!__all(alpha == 1.f)) {         // the running max moved for some query of this wave
                        o[0] *= alpha;
                        o[1] *= alpha;
                        lsum *= alpha;
                    }
                }
                // ... beside the probabilities of tile t on the vector ALU
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = (kb == 0 ? EARLY : 0); r < 16; ++r) e[kb][r] = __builtin_amdgcn_exp2f(MODE == 1 ? x[kb][r] : x[kb][r] - ref);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!LAST) scores(x);
                float a0 = e[0][0], a1 = e[0][1], a2 = e[0][2], a3 = e[0][3];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                    for (int r = (kb == 0 ? 4 : 0); r < 16; r += 4) { a0 += e[kb][r]; a1 += e[kb][r + 1]; a2 += e[kb][r + 2]; a3 += e[kb][r + 3]; }
                    p[kb][0] = pack8(e[kb], 0);
                    p[kb][1] = pack8(e[kb], 1);
                }
                lsum += (a0 + a1) + (a2 + a3);
            }
            MARS_T(t_work)
            // the fills started one iteration ago have landed (this iteration's four stay in flight); vector memory
            // operations retire in order, so everything older - the staged queries, the last item's stores - has too
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            MARS_T(t_landed)
            meet();
            MARS_T(t_met)
            MARS_ACC(st_compute, t_top, t_work)
            MARS_ACC(st_wait, t_work, t_met)
            MARS_ACC(st_fill, t_work, t_landed)
        };
        iteration(0, std::true_type{}, std::false_type{});
        for (int t = 1; t + 1 < nt; ++t) iteration(t, std::false_type{}, std::false_type{});
        iteration(nt - 1, std::false_type{}, std::true_type{});
        MARS_T(t_tail)
        if constexpr (MODE != 0) {
            read_v((g + nt - 1) % 3);
            lds_wait();
            pv();
        }
        MARS_T(t_tailed)
        MARS_ACC(st_tail, t_tail, t_tailed)
        // absolute scale: a probability beyond 2^HEAD_ROOM shows in its row's sum (no maxima are taken on the way)
        return MODE == 1 && __any(!(halves_sum(lsum) < SUM_LIMIT));
    };

    // key 0 / value 0 of the item (LEAD1): the dims of the lane's query fragments / of its output registers
    uint4 k0r[4];
    bf16x4 v0r[2][4];
    // The state before tile 0.  LEAD1: key 0 is the initial state, worked out on the vector ALU: s0 = q . k0 in the
    // exp2 domain (the query fragments carry the scale; a lane has half of its query's dims, the other half-wave the
    // rest), the output starts as p0 v0.  `absolute`: take the absolute scale (reference 0) if the first scores are
    // moderate - key 0's, or the row maxima of tile 0; the wave-uniform answer is returned.
    auto first_state = [&](const f32x16 (&sc)[2], bool absolute) -> bool {
        o[0] = (f32x16)(0.f);
        o[1] = (f32x16)(0.f);
        lsum = 0.f;
        mrun = -INFINITY;
        bool lazy = false;
        if (!wave_active) return false;
        if constexpr (LEAD1) {
            float part = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const bf16x8 kk = __builtin_bit_cast(bf16x8, k0r[st]);
#pragma unroll
                for (int e = 0; e < 8; ++e) part = fmaf((float)qf[st][e], (float)kk[e], part);
            }
            const float s0 = halves_sum(part);
            lazy = absolute && __all(fabsf(s0) <= ABS_RANGE);
            mrun = lazy ? 0.f : s0;
            const float p0 = lazy ? __builtin_amdgcn_exp2f(s0) : 1.f;
            lsum = hh == 0 ? p0 : 0.f;                  // counted once: the two halves' parts are added in the end
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[db][4 * rg + j] = p0 * (float)v0r[db][rg][j];
                }
        } else {
            // (a first tile that is also the ragged last one holds at least one key: its maximum is finite)
            float m_ = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (!(ragged_last && nt == 1) || 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hh < TK) m_ = fmaxf(m_, sc[kb][r]);
            lazy = absolute && __all(fabsf(halves_max(m_)) <= ABS_RANGE);
            if (lazy) mrun = 0.f;
        }
        return lazy;
    };
    auto scale_queries = [&]() {
        if (c2 != 1.0f) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[s][e] = (__bf16)((float)qf[s][e] * c2);
        }
    };

    // Which items a workgroup works on.  Its first two are blockIdx.x and blockIdx.x + gridDim.x.  With a shared
    // counter (shared_count: one word per XCD, then the count of finished workgroups) the later ones are taken from
    // the counter of the workgroup's XCD as it goes, two to three items ahead of the one it works on; without, they follow
    // at a stride of gridDim.x.  Why not the fixed stride alone: of the two workgroups of a CU the one dispatched
    // first wins the issue arbitration (oldest first) and runs about a quarter faster; with equal shares it ends
    // early and the other one finishes alone, with nobody to fill the matrix pipe while it waits.
    const int grid = (int)gridDim.x, xcd = blockIdx.x & 7;
    int id = blockIdx.x, nid = min(id + grid, nitems), taking = nitems;
    auto take = [&]() -> unsigned { return __hip_atomic_fetch_add(shared_count + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // the workgroup's third, fourth ... item from the counter's answers 0, 1, ... (the static ones of all workgroups come first)
    auto next_item = [&](unsigned j) -> int { return (int)min((2u * (unsigned)(grid >> 3) + j) * 8u + (unsigned)xcd, (unsigned)nitems); };
    unsigned took = 0;                          // (thread 0) what the counter said; looked at when an item is done
    if (shared_count && tid == 0 && nid < nitems) took = take();
    // the first item's queries (and key / value 0) and its first tiles: K(0), K(1), K(2) and V(0)
    base = item_base(id, qblk);
    stage_item(base, qblk);
    fill_tile(base, 0, 0);
    if (!is_v) {
        fill_tile(base, 1, 1);
        fill_tile(base, 2, 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    while (id < nitems) {
        MARS_T(t_item)
        if (!shared_count) taking = min(nid + grid, nitems);
        nbase = nid < nitems ? item_base(nid, nqblk) : nullptr;
        fbase = nbase ? nbase : base;
        const int q0 = qblk * QPB + wave * 32;
        wave_active = q0 < T;                   // trailing waves of an image-head's last block only help with the fills
        // the item's queries, key 0 and value 0 from where the last item (or the lines above) had them brought, and
        // the K fragments of its tile 0
        {
            const unsigned sq = (unsigned)(uintptr_t)(lds_ptr_t)&stage_q[wave][0];
            const unsigned skv = (unsigned)(uintptr_t)(lds_ptr_t)&stage_kv[0] + 16 * hh;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                uint4 r;
                asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(sq + ka[s]));
                qf[s] = __builtin_bit_cast(bf16x8, r);
            }
            if constexpr (LEAD1) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(k0r[0]) : "v"(skv));
                asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(k0r[1]) : "v"(skv));
                asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(k0r[2]) : "v"(skv));
                asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(k0r[3]) : "v"(skv));
                const unsigned sv = (unsigned)(uintptr_t)(lds_ptr_t)&stage_kv[0] + 128 + 8 * hh;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        v2i32 r;
                        asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(sv + 64 * db + 16 * rg));
                        v0r[db][rg] = __builtin_bit_cast(bf16x4, r);
                    }
            }
        }
        if (wave_active) read_k(g % 3);
        lds_wait();
        MARS_T(t_topmeet)
        meet();                                 // everybody has its queries: the staging area is free for the next item
        MARS_T(t_topmet)
        MARS_ACC(st_topwait, t_item, t_topmet)
        __builtin_amdgcn_sched_barrier(0);
        if (nbase) stage_item(nbase, nqblk);
        __builtin_amdgcn_sched_barrier(0);
        scale_queries();
        f32x16 sc[2];
        if (wave_active) scores(sc);
        const bool lazy = first_state(sc, true);
        MARS_T(t_loop)
        MARS_ACC(st_top, t_item, t_loop)
        bool again;
        if (!wave_active) again = pass(sc, std::integral_constant<int, 0>{});
        else if (lazy) again = pass(sc, std::integral_constant<int, 1>{});
        else again = pass(sc, std::integral_constant<int, 2>{});
        MARS_T(t_voting)
        // The waves tell each other whether one of them wants the item worked again, and wave 0 tells the item it
        // took, through LDS and one barrier (asm accesses, and no drain of the fills in flight: see the LDS reads).
        bool redo;
        {
            const unsigned xa = (unsigned)(uintptr_t)(lds_ptr_t)&xchg[0];
            int mine = again ? 1 : 0;
            if (lane == 0) asm volatile("ds_write_b32 %0, %1" :: "v"(xa + 4 * wave), "v"(mine) : "memory");
            int mine_next = nitems;
            if (shared_count && tid == 0) {
                asm volatile("" : "+v"(took));  // (nothing derived from the counter's answer before this point: it has had the item to arrive)
                if (nid < nitems) mine_next = next_item(took);
                asm volatile("ds_write_b32 %0, %1 offset:16" :: "v"(xa), "v"(mine_next) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            meet();
            // the item after that one: asked for now (the fills in flight are old enough for a wait to cost little,
            // should the compiler want one around the request), looked at when the next item is done
            if (shared_count && tid == 0 && mine_next < nitems) took = take();
            v4i32 all;
            int got;
            asm volatile("ds_read_b128 %0, %1" : "=v"(all) : "v"(xa));
            asm volatile("ds_read_b32 %0, %1 offset:16" : "=v"(got) : "v"(xa));
            lds_wait();
            redo = __builtin_amdgcn_readfirstlane(all[0] | all[1] | all[2] | all[3]) != 0;
            if (shared_count) taking = __builtin_amdgcn_readfirstlane(got);
        }
        MARS_T(t_voted)
        MARS_ACC(st_vote, t_voting, t_voted)
        if (redo) {
            // some wave's scores left the absolute scale's range: those waves start over with the running maximum,
            // the others help with the fills again.  The ring is set up afresh (the pass has left the next item's
            // first tiles in it; the second pass brings them again), key 0 and value 0 come straight from memory.
            if constexpr (LEAD1) {
#pragma unroll
                for (int st = 0; st < 4; ++st) k0r[st] = *(const uint4*)(base + (int64_t)D * 2 + 16 * hh + 32 * st);
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) v0r[db][rg] = *(const bf16x4*)(base + (int64_t)D * 4 + 64 * db + 16 * rg + 8 * hh);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the fills the first pass left in flight
            __syncthreads();
            fill_tile(base, 0, g % 3);
            if (!is_v) {
                fill_tile(base, 1, (g + 1) % 3);
                fill_tile(base, 2, (g + 2) % 3);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (wave_active) read_k(g % 3);
            lds_wait();
            __syncthreads();
            if (wave_active) scores(sc);
            if (again) {
                first_state(sc, false);
                pass(sc, std::integral_constant<int, 2>{});
            } else {
                pass(sc, std::integral_constant<int, 0>{});
            }
        }
        MARS_T(t_done)
        const int myq = q0 + r32;
        const float ltot = halves_sum(lsum);
        if (wave_active) {
            int bh, qb_;
            attn_block(id, nbh, nqb, xcd_aware, bh, qb_);
            const float inv = __builtin_amdgcn_rcpf(ltot);      // (1 ulp: the outputs are rounded to bf16)
            __bf16* op = O + ((int64_t)(bh / H) * T + myq) * D + (bh % H) * 64;
            // A lane holds four consecutive dims of each group of eight, its partner in the other half-wave (same
            // query) the other four.  Two groups at a time, the half-waves trade halves (v_permlane32_swap on the
            // packed pairs: the lower lane ends up with all eight dims of the first group, the upper lane with those
            // of the second) and every lane stores 16 bytes.
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    bf16x4 a, b;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        a[j] = (__bf16)(o[db][8 * rp + j] * inv);
                        b[j] = (__bf16)(o[db][8 * rp + 4 + j] * inv);
                    }
                    const v2i32 ai = __builtin_bit_cast(v2i32, a), bi = __builtin_bit_cast(v2i32, b);
                    const auto s0 = __builtin_amdgcn_permlane32_swap((unsigned)ai[0], (unsigned)bi[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap((unsigned)ai[1], (unsigned)bi[1], false, false);
                    const v4i32 out = {(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
                    if (myq < T) *(v4i32*)(op + 32 * db + 8 * (2 * rp + hh)) = out;
                }
            // natural-log LSE of the scaled scores: (max + log2(sum)) * ln 2
            if (lse && hh == 0 && myq < T) lse[(int64_t)bh * T + myq] = (mrun + __builtin_amdgcn_logf(ltot)) * 0.69314718055994530942f;
        }
        MARS_T(t_fin)
        MARS_ACC(st_fin, t_done, t_fin)
        g += nt;
        base = nbase;
        qblk = nqblk;
        id = nid;
        nid = taking;
    }
    // the last workgroup to finish leaves the counters as it found them: zero
    if (shared_count && tid == 0) {
        const unsigned done = __hip_atomic_fetch_add(shared_count + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (unsigned)grid - 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) __hip_atomic_store(shared_count + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may be in flight when the workgroup gives its LDS back
#ifdef MARS_STAMPS
    MARS_T(t_end)
    if (lane == 0) {
        unsigned long long* sb = mars_attn32_stamps + ((size_t)blockIdx.x * NW + wave) * 16;
        sb[0] = st_top; sb[1] = st_compute; sb[2] = st_wait; sb[3] = st_fin; sb[4] = t_end - t_begin; sb[5] = st_tail; sb[6] = st_fill; sb[7] = st_vote; sb[8] = st_topwait;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        sb[9] = hw; sb[10] = xcc; sb[11] = t_begin; sb[12] = t_end;
    }
#endif
}

// Item counters of the launches.  Launches on one stream run one after the other and each leaves its counters at
// zero, so a stream keeps one set for good; launches on different streams may overlap and get different sets.  The
// sets live in a file-scope device array (no allocation); a process that uses more than MAX_STREAM_SETS streams on
// a device gets no counters for the further ones (those launches walk their items at a fixed stride).
constexpr int MAX_STREAM_SETS = 64, MAX_DEVICES = 16;
__device__ __attribute__((aligned(256))) unsigned item_counters[MAX_STREAM_SETS][16];

unsigned* stream_counters(hipStream_t stream) {
    struct PerDevice { unsigned* sets = nullptr; bool failed = false; std::vector<hipStream_t> streams; };
    static std::mutex mu;
    static PerDevice devices[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> hold(mu);
    PerDevice& d = devices[dev];
    if (d.failed) return nullptr;
    if (!d.sets) {
        // each device has a copy of the array at an address of its own (zero when the code object is loaded)
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(item_counters)) != hipSuccess || !p) {
            (void)hipGetLastError();
            d.failed = true;
            return nullptr;
        }
        d.sets = (unsigned*)p;
    }
    for (size_t i = 0; i < d.streams.size(); ++i)
        if (d.streams[i] == stream) return d.sets + 16 * i;
    if ((int)d.streams.size() >= MAX_STREAM_SETS) return nullptr;
    d.streams.push_back(stream);
    return d.sets + 16 * (d.streams.size() - 1);
}

}  // namespace

// T >= 3 key tiles (the ring runs three tiles ahead), no causal mask; arguments as checked by marship_flash_attn
void mars_flash_attn32_launch(const void* qkv, void* O, float* lse, int B, int T, int H, float score_scale_log2, hipStream_t stream) {
    // one workgroup per slot the chip has for them (a CU holds two: 65 KiB of LDS and 256 registers a lane each), each
    // walking its share of the items; a whole number of workgroups per XCD keeps an item's XCD what its number says
    static const int cus = [] { int dev = 0, n = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; return n; }();
    const int items = ((T + 127) / 128) * B * H;
    static const int persist = [] { const char* e = getenv("MARS_FLASH32_PERSIST"); return e ? atoi(e) : 1; }();
    static const int xcd_aware = [] { const char* e = getenv("MARS_FLASH_XCD"); return e ? atoi(e) : 1; }();
    static const int shared = [] { const char* e = getenv("MARS_FLASH32_SHARED"); return e ? atoi(e) : 1; }();
    int slots = cus * 2;
    slots -= slots % 8;
    const bool walking = persist && items > slots && slots > 0;
    dim3 grid(walking ? slots : items), block(256);
    const int nbh = B * H;
    // a workgroup's third and later items come from counters shared by the launch (if it has that many, and if this
    // stream could be given a set of counters)
    unsigned* counters = walking && shared && items > 2 * slots ? stream_counters(stream) : nullptr;
    if (T % KT == 1) hipLaunchKernelGGL(flash_attn_w32_kernel<true>, grid, block, 0, stream, (const __bf16*)qkv, (__bf16*)O, lse, T, H, score_scale_log2, nbh, xcd_aware, counters);
    else hipLaunchKernelGGL(flash_attn_w32_kernel<false>, grid, block, 0, stream, (const __bf16*)qkv, (__bf16*)O, lse, T, H, score_scale_log2, nbh, xcd_aware, counters);
}
