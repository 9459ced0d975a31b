// TN GEMM for the large-M products of the AlphaCLIP tower.  Block tile 256 x (64*NWC), 2 x NWC
// waves, each wave 128 x 64 outputs = 8 x 4 fragments of v_mfma_f32_16x16x32_bf16 (or four
// v_mfma_f32_16x16x4_f32 per fragment pair for f32 inputs).  Two shapes are built:
//   NWC = 4: 256 x 256, 8 waves, 96 KiB LDS, one workgroup per CU;
//   NWC = 2: 256 x 128, 4 waves, 72 KiB LDS, two workgroups per CU, so that one workgroup's
//            epilogue (store-bound) runs beside the other's main loop (MFMA-bound) and a CU's
//            share of the tiles is balanced at half-tile granularity.
//
// K is consumed in steps of 64 bytes per row (32 bf16 / 16 f32).  LDS holds a ring of three
// steps; a step is fetched by LDS-DMA (global_load_lds, 1 KiB per wave instruction = 16 rows x
// 64 B) two steps before its fragments are read and three before they feed the MFMAs, and the
// DMAs stay in flight across the per-step barrier:
//
//     top of step s:  s_waitcnt vmcnt(n)   this wave's DMAs of step s+1 have landed (s+2 still flying)
//                     s_barrier            => every wave's have, and everybody is done reading step s
//                     DMAs of step s+3 into the slot step s used, dealt out between the MFMA rows
//                     12 x ds_read_b128 of step s+1 into the idle fragment registers, likewise,
//                     32 MFMAs on the fragments of step s (fetched one step earlier)
//
// Bank swizzle (rows are 64 B, four 16-B chunks): chunk c of row r sits at chunk slot
// c ^ g((r >> 2) & 3), g = {0,3,2,1}; the DMA image is lane-linear, so the xor is applied to
// the per-lane SOURCE address and again on the fragment reads.
#include <type_traits>
#include "common.h"
#include "../../include/mars_hip.h"

// Diagnostic builds (tools/diag_gemm.hip) define MARS_STAMPS and read per-wave clock stamps from a
// buffer of their own; in the library build the macro expands to nothing.
#ifdef MARS_STAMPS
__device__ unsigned long long* mars_stamp_buf;
#define MARS_STAMP(i)                                                                                   \
    if (lane == 0) mars_stamp_buf[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 16 + (i)] = __builtin_amdgcn_s_memtime();
#define MARS_STAMP_ID()                                                                                 \
    if (lane == 0) {                                                                                    \
        unsigned long long* sb = mars_stamp_buf + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 16; \
        sb[14] = __builtin_amdgcn_s_memrealtime();                                                      \
        sb[15] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); \
    }
#else
#define MARS_STAMP(i)
#define MARS_STAMP_ID()
#endif

// experiment switches of the diagnostic build: drop the loop's DMAs / fragment reads / MFMAs
#ifndef MARS_EXP_DMA
#define MARS_EXP_DMA 1
#endif
#ifndef MARS_EXP_READ
#define MARS_EXP_READ 1
#endif
#ifndef MARS_EXP_MMA
#define MARS_EXP_MMA 1
#endif

namespace {

constexpr int STEP_B = 64;                               // bytes of K per row and step

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const unsigned char* src, unsigned char* lds_dst) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ int gsw(int x) { return (4 - x) & 3; }

template <bool FAST>
__device__ __forceinline__ float act_fn3(float v, int act) {
    if (act == 1) {
        if constexpr (FAST) {
            // erf by Abramowitz & Stegun 7.1.26 (absolute error below 1.5e-7, far inside a bf16 result) on the
            // hardware exp2 and reciprocal: a third of the instructions of erff
            const float z = fabsf(v) * 0.70710678118654752440f;
            const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
            float p = fmaf(1.061405429f, t, -1.453152027f);
            p = fmaf(p, t, 1.421413741f);
            p = fmaf(p, t, -0.284496736f);
            p = fmaf(p, t, 0.254829592f);
            const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
            const float erf_abs = fmaf(-p * t, e, 1.0f);
            return 0.5f * v + 0.5f * fabsf(v) * erf_abs;      // v * erf(v / sqrt 2) = |v| * erf(|v| / sqrt 2)
        }
        return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    }
    if (act == 2) {
        // v / (1 + 2^(-1.702 log2(e) v)) on the hardware exp2 and reciprocal (1 ulp each; __frcp_rn would be a
        // correctly rounded division, ten more instructions per output that a bf16 result cannot show)
        if constexpr (FAST) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * v));
        else return v / (1.0f + expf(-1.702f * v));
    }
    if (act == 3) return fmaxf(v, 0.f);
    return v;
}

template <int OFF>
__device__ __forceinline__ void lds_read128(uint4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}

template <typename TIN>
__device__ __forceinline__ void mma_step(f32x4& acc, const uint4& fa, const uint4& fb) {
    if constexpr (sizeof(TIN) == 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc, 0, 0, 0);
    } else {
        const f32x4 a = __builtin_bit_cast(f32x4, fa);
        const f32x4 b = __builtin_bit_cast(f32x4, fb);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
    }
}

// NWR x NWC waves, each FM x FN fragments (16 FM x 16 FN outputs); WPE = waves per SIMD the register
// budget must allow (workgroups per CU x waves per workgroup / 4).
//
// PAIR: the two 64-byte halves of a 128-byte line are requested back to back (steps 2p and 2p+1 of every row
// piece in consecutive instructions), so the second request finds the line the first one brought into the
// CU's vector cache instead of fetching it from L2 again one step later, when it has long been evicted.  The
// ring then has an even number of slots, filled a pair at a time at the top of every odd step (the pair of slots
// steps 2p, 2p+1 used is free once the fragments of step 2p+1 are in registers), and one barrier per pair.
//
// PERSIST: the launch has one workgroup per CU slot and each walks the tiles b, b + gridDim.x, ... of the
// dispatch order (the tile -> XCD map stays what a launch of one workgroup per tile gives).  When the main loop
// of a tile is done the ring is free, so the first RING steps of the NEXT tile are requested before the
// accumulators are stored: the fill of the ring (a quarter of a tile's operand bytes at K = 1024) lands under
// the store-bound epilogue instead of in front of an idle matrix pipe.
template <typename TIN, typename TOUT, int NWR, int NWC, int FM, int FN, int WPE, int RING, bool PAIR = false, bool PERSIST = false>
__global__ __launch_bounds__(64 * NWR * NWC, WPE) void gemm_tn_ring_kernel(MarsGemm g, int group_m) {
    static_assert(!PAIR || (RING % 2 == 0 && RING >= 4 && RING <= 8), "pair fills need a ring of four, six or eight slots");
    static_assert(!PERSIST || PAIR, "the persistent form is built on the pair-fill ring");
    constexpr int BM = 16 * FM * NWR, WN = 16 * FN, BN = WN * NWC, NWAVES = NWR * NWC;
    constexpr int A_BYTES = BM * STEP_B, B_BYTES = BN * STEP_B, STAGE = A_BYTES + B_BYTES;
    static_assert((BM / 16) % NWAVES == 0 && (BN / 16) % NWAVES == 0, "DMA pieces must divide among the waves");
    static_assert(FM == 2 || FM == 4 || FM == 8, "wave tile is 32, 64 or 128 rows");
    static_assert(FN == 4 || FN == 8, "wave tile is 64 or 128 columns");
    constexpr int A_PIECES = BM / 16 / NWAVES, B_PIECES = BN / 16 / NWAVES;   // 1-KiB DMAs per wave and step
    constexpr int PIECES = A_PIECES + B_PIECES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[RING * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / NWC, wc = wave % NWC;
    const int r16 = lane & 15, kq = lane >> 4;
    MARS_STAMP_ID()

    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    const int nwg = ntm * ntn;
    // tile of dispatch slot `bid` (the blockIdx.x of a launch with one workgroup per tile)
    auto locate = [&](int bid, int& r0, int& c0) {
        {   // blocks that share an L2 (same blockIdx % 8) walk a contiguous range of tiles ...
            const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        // ... in a grouped raster, so the 32 tiles an XCD has in flight form a compact patch
        const int gsize = group_m * ntn;
        const int first_m = (bid / gsize) * group_m;
        const int gm = min(ntm - first_m, group_m);
        const int tm = first_m + (bid % gsize) % gm, tn = (bid % gsize) / gm;
        r0 = tm * BM;
        c0 = tn * BN;
    };
    int vb = blockIdx.x;                                   // dispatch slot of the tile in hand
    int row0, col0;
    locate(vb, row0, col0);
    const int z = blockIdx.y, z0 = z / g.nb1, z1 = z % g.nb1;
    const unsigned char* Ab = (const unsigned char*)g.A + (z0 * g.sA0 + z1 * g.sA1) * (int64_t)sizeof(TIN);
    const unsigned char* Bb = (const unsigned char*)g.B + (z0 * g.sB0 + z1 * g.sB1) * (int64_t)sizeof(TIN);
    const int64_t lda_b = g.lda * (int64_t)sizeof(TIN), ldb_b = g.ldb * (int64_t)sizeof(TIN);

    // DMA sources: piece i of this wave fills rows (i*NWAVES + wave)*16 .. +15 of A (then of B)
    const unsigned char* pa[A_PIECES];
    const unsigned char* pb[B_PIECES];
    auto set_sources = [&](int r0, int c0) {
        const int lr = lane >> 2, ls = lane & 3;
        const int chunk = ls ^ gsw((lr >> 2) & 3);
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
            int ga = r0 + (i * NWAVES + wave) * 16 + lr; ga = ga < g.M ? ga : g.M - 1;
            pa[i] = Ab + ga * lda_b + (chunk << 4);
        }
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i) {
            int gb = c0 + (i * NWAVES + wave) * 16 + lr; gb = gb < g.N ? gb : g.N - 1;
            pb[i] = Bb + gb * ldb_b + (chunk << 4);
        }
    };
    set_sources(row0, col0);
    // piece p of this wave's share of step s (A pieces first), into ring slot `slot`
    auto issue_piece = [&](int s, int slot, auto p_tag) {
        constexpr int p = decltype(p_tag)::value;
        unsigned char* sa = lds + slot * STAGE;
        const int64_t ko = (int64_t)s * STEP_B;
        if constexpr (p < A_PIECES) glds16(pa[p] + ko, sa + (p * NWAVES + wave) * 1024);
        else glds16(pb[p - A_PIECES] + ko, sa + A_BYTES + ((p - A_PIECES) * NWAVES + wave) * 1024);
    };
    auto issue = [&](int s, int slot) {
        [&]<int... P>(std::integer_sequence<int, P...>) { (issue_piece(s, slot, std::integral_constant<int, P>{}), ...); }
        (std::make_integer_sequence<int, PIECES>{});
    };
    // the pieces of a step that are issued in front of fragment row m's MFMAs: a DMA issue blocks the wave
    // while the CU's load path is busy (about 36 cycles per piece already queued), so the pieces are spread
    // over the step instead of queued behind one another while the matrix pipe waits
    auto issue_row = [&](int s, int slot, auto m_tag) {
        constexpr int m = decltype(m_tag)::value;
        [&]<int... P>(std::integer_sequence<int, P...>) {
            ((P * FM / PIECES == m ? issue_piece(s, slot, std::integral_constant<int, P>{}) : (void)0), ...);
        }
        (std::make_integer_sequence<int, PIECES>{});
    };
    // pair fills: both halves of the lines of piece P, one instruction behind the other
    auto issue_pair_piece = [&](int s, int slot0, int slot1, auto p_tag) {
        issue_piece(s, slot0, p_tag);
        issue_piece(s + 1, slot1, p_tag);
    };
    auto issue_pair_row = [&](int s, int slot0, int slot1, auto m_tag) {
        constexpr int m = decltype(m_tag)::value;
        [&]<int... P>(std::integer_sequence<int, P...>) {
            ((P * FM / PIECES == m ? issue_pair_piece(s, slot0, slot1, std::integral_constant<int, P>{}) : (void)0), ...);
        }
        (std::make_integer_sequence<int, PIECES>{});
    };
    auto issue_pair = [&](int s, int slot0, int slot1) {
        [&]<int... P>(std::integer_sequence<int, P...>) { (issue_pair_piece(s, slot0, slot1, std::integral_constant<int, P>{}), ...); }
        (std::make_integer_sequence<int, PIECES>{});
    };

    f32x4 acc[FM][FN];

    const int ns = (int)(((int64_t)g.K * sizeof(TIN)) / STEP_B);
    // The MFMAs compute C^T fragments (B rows are the first operand), so a lane ends up with four
    // consecutive COLUMNS of one row of C: accumulator [m][n][reg] is C[16m + r16][colmap(n, reg)],
    // and the epilogue stores 16 bytes per lane straight from registers.  For 2-byte outputs the B
    // rows are dealt to the fragment pair (2v, 2v+1) so that the pair gives a lane 8 consecutive
    // columns: row index i of fragment n is column 32(n>>1) + 8(i>>2) + 4(n&1) + (i&3).
    // bf16 products with f32 output deal the rows the same way: 32 consecutive bytes per lane and row, whole
    // 128-byte lines per four lanes, for the residual loads and the stores of the stream.
    constexpr bool PERM = sizeof(TOUT) == 2 || sizeof(TIN) == 2;
    auto frag_off = [&](int row) { return row * STEP_B + ((kq ^ gsw((row >> 2) & 3)) << 4); };
    const unsigned a_off = wr * (16 * FM) * STEP_B + frag_off(r16);
    const int brow = PERM ? 8 * (r16 >> 2) + (r16 & 3) : r16;          // fragment 0; fragment 1 is 4 / 16 rows on
    const unsigned b_off0 = A_BYTES + wc * WN * STEP_B + frag_off(brow);
    const unsigned b_off1 = A_BYTES + wc * WN * STEP_B + frag_off(brow + (PERM ? 4 : 16));
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)lds;

    // ---- main loop: the fragments of step s+1 are read while the MFMAs of step s run ----
    uint4 faA[FM], fbA[FN], faB[FM], fbB[FN];
    auto read_frags = [&](uint4 (&FA)[FM], uint4 (&FB)[FN], int slot_) {
        const unsigned sbase = lds_base + (unsigned)(slot_ * STAGE);
        const unsigned aa = sbase + a_off, ba0 = sbase + b_off0, ba1 = sbase + b_off1;
        lds_read128<0>(FB[0], ba0); lds_read128<0>(FB[1], ba1); lds_read128<2048>(FB[2], ba0); lds_read128<2048>(FB[3], ba1);
        if constexpr (FN == 8) {
            lds_read128<4096>(FB[4], ba0); lds_read128<4096>(FB[5], ba1); lds_read128<6144>(FB[6], ba0); lds_read128<6144>(FB[7], ba1);
        }
        lds_read128<0>(FA[0], aa); lds_read128<1024>(FA[1], aa);
        if constexpr (FM >= 4) { lds_read128<2048>(FA[2], aa); lds_read128<3072>(FA[3], aa); }
        if constexpr (FM == 8) {
            lds_read128<4096>(FA[4], aa); lds_read128<5120>(FA[5], aa); lds_read128<6144>(FA[6], aa); lds_read128<7168>(FA[7], aa);
        }
    };
#define MARS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
    // all of this wave's DMAs but those of the youngest `pairs` pairs of steps have landed
    auto wait_pairs = [&](int pairs) {
        static_assert(!PAIR || (RING / 2 - 1) * 2 * PIECES <= 63, "vmcnt is a 6-bit count");
        if (pairs <= 0) MARS_WAIT_VM(0);
        else if (pairs == 1) MARS_WAIT_VM((2 * PIECES) & 63);
        else if (pairs == 2) MARS_WAIT_VM((4 * PIECES) & 63);
        else MARS_WAIT_VM((6 * PIECES) & 63);
    };
    // fragment read j of a step (B fragments first) into the idle register set
    auto read_one = [&](uint4 (&FA)[FM], uint4 (&FB)[FN], int slot_, auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        const unsigned sbase = lds_base + (unsigned)(slot_ * STAGE);
        if constexpr (j < FN) lds_read128<(j >> 1) * 2048>(FB[j], sbase + ((j & 1) ? b_off1 : b_off0));
        else lds_read128<(j - FN) * 1024>(FA[j - FN], sbase + a_off);
    };
    // One step.  At its top the DMAs of steps s+1 and s+2 are this wave's outstanding ones; step
    // s+1 must have landed (for every wave: barrier) before its fragments are fetched, and the same
    // barrier says that everybody has the fragments of step s in registers, so its slot can take
    // step s+3.  The step's DMAs and fragment reads are dealt out between the rows of MFMAs: issued
    // in a bunch behind the barrier, eight waves' worth of them queue up in the load path and in the
    // LDS, and every wave sits in its issue slot while the matrix pipe idles.  Fragment reads are
    // inline asm on purpose: hipcc treats an in-flight LDS-DMA as a pending LDS write and would put
    // s_waitcnt vmcnt(0) in front of any ds_read it can see, draining the ring.  `slot` is the ring
    // slot of step s.  MAIN steps (s + 3 < ns) have no conditions in them; the reads of the last
    // step fetch a stale slot into registers nobody uses.
#define MARS_PHASE(CA, CB, NA, NB, s_, ODD)                                                          \
    {                                                                                                 \
        const int nslot = slot + 1 == RING ? 0 : slot + 1;                                            \
        bool more;                                                                                    \
        if constexpr (PAIR) {                                                                         \
            if ((ODD) && (s_) + 1 < ns) {                                                             \
                /* pairs issued behind the one step s+1 belongs to stay in flight */                  \
                wait_pairs(min(((s_) + RING - 3) >> 1, (ns >> 1) - 1) - (((s_) + 1) >> 1));           \
                __builtin_amdgcn_s_barrier();                                                         \
                asm volatile("" ::: "memory");                                                        \
            }                                                                                         \
            more = MARS_EXP_DMA && (ODD) && (s_) + RING - 1 < ns;                                     \
        } else {                                                                                      \
            if ((s_) + 1 < ns) {                                                                      \
                if ((s_) + RING - 1 < ns) MARS_WAIT_VM((RING - 2) * PIECES);                          \
                else MARS_WAIT_VM(0);                                                                 \
                __builtin_amdgcn_s_barrier();                                                         \
                asm volatile("" ::: "memory");                                                        \
            }                                                                                         \
            more = MARS_EXP_DMA && (s_) + RING < ns;                                                  \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        auto row = [&](auto mi_tag) {                                                                 \
            constexpr int M_ = decltype(mi_tag)::value;                                               \
            if constexpr (PAIR) {                                                                     \
                if (more) issue_pair_row((s_) + RING - 1, (slot + RING - 1) % RING, slot, mi_tag);    \
            } else {                                                                                  \
                if (more) issue_row((s_) + RING, slot, mi_tag);                                       \
            }                                                                                         \
            if (MARS_EXP_READ) {                                                                      \
                [&]<int... J>(std::integer_sequence<int, J...>) {                                     \
                    ((J * FM / (FM + FN) == M_ ? read_one(NA, NB, nslot, std::integral_constant<int, J>{}) : (void)0), ...); \
                }                                                                                     \
                (std::make_integer_sequence<int, FM + FN>{});                                         \
            }                                                                                         \
            if (MARS_EXP_MMA) _Pragma("unroll") for (int n = 0; n < FN; ++n) mma_step<TIN>(acc[M_][n], CB[n], CA[M_]); \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        };                                                                                            \
        [&]<int... Mi>(std::integer_sequence<int, Mi...>) { (row(std::integral_constant<int, Mi>{}), ...); } \
        (std::make_integer_sequence<int, FM>{});                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        slot = nslot;                                                                                 \
    }

    // the first RING steps of the tile whose sources are set, into slots 0 .. RING - 1
    auto fill_ring = [&]() {
        if constexpr (PAIR) {                              // ns is even: K spans a multiple of 128 bytes
#pragma unroll
            for (int i = 0; i < RING; i += 2)
                if (i < ns) issue_pair(i, i, i + 1);
        } else {
#pragma unroll
            for (int i = 0; i < RING; ++i)
                if (i < ns) issue(i, i);
        }
    };
    fill_ring();
    int after_fill = 0;
    bool warm = false;           // PERSIST: this tile's ring fill went out under the epilogue of the one before
    for (;;) {                   // the tiles of this workgroup (one unless PERSIST)
    MARS_STAMP(0)
#pragma unroll
    for (int m = 0; m < FM; ++m)
#pragma unroll
        for (int n = 0; n < FN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (PERSIST && warm) {
        // the fill is older than everything the epilogue issued behind it, and vector memory operations retire
        // in order: at most `after_fill` outstanding means the ring is filled
        if (after_fill == 6) MARS_WAIT_VM(6);
        else if (after_fill == 12) MARS_WAIT_VM(12);
        else if (after_fill == 15) MARS_WAIT_VM(15);
        else if (after_fill == 16) MARS_WAIT_VM(16);
        else if (after_fill == 32) MARS_WAIT_VM(32);
        else MARS_WAIT_VM(0);
    } else if constexpr (PAIR) {
        wait_pairs(min(RING, ns) / 2 - 1);
    } else {
        if (ns >= RING) MARS_WAIT_VM((RING - 1) * PIECES);
        else MARS_WAIT_VM(0);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(faA, fbA, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MARS_STAMP(1)
    int slot = 0;
    for (int s = 0; s < ns; s += 2) {
        MARS_PHASE(faA, fbA, faB, fbB, s, false)
        if (s + 1 < ns) MARS_PHASE(faB, fbB, faA, fbA, s + 1, true)
    }
    MARS_STAMP(2)

    // ---------------- epilogue ----------------
    // Wave tiles inside the matrix with 16-byte aligned rows store straight from the accumulators
    // (short branch-free variants: plain, QuickGELU, erf GELU, ReLU, residual add).  Everything else -
    // N-edge tiles, unaligned rows, an activation together with a residual - goes through LDS
    // in slabs of 32 rows and a per-element loop.
    constexpr bool FAST = sizeof(TIN) == 2;
    constexpr int NV = PERM ? FN / 2 : FN, VW = PERM ? 8 : 4;     // 16-byte stores per fragment row, columns per store
    constexpr int VS = WN / NV;                                    // column distance between a lane's stores
    // the shapes that keep three or four waves on a SIMD have no registers to spare for the folded LayerNorm forms
    constexpr bool LN_OUT = FAST && sizeof(TOUT) == 4 && WPE <= 2, LN_IN = FAST && sizeof(TOUT) == 2 && WPE <= 2;
    TOUT* Cb = (TOUT*)g.C + (z0 * g.sC0 + z1 * g.sC1);
    const float* Rb = g.residual ? g.residual + (z0 * g.sR0 + z1 * g.sR1) : nullptr;
    const int wrow0 = row0 + wr * (16 * FM), wcol0 = col0 + wc * WN;
    // rows beyond M are masked per lane in the fast variants.  A persistent workgroup decides for the whole
    // tile: the staged path writes to the ring, so no wave may be filling it for the next tile meanwhile
    const bool inside = PERSIST ? col0 + BN <= g.N : wcol0 + WN <= g.N;
    const bool aligned = (g.ldc % (16 / (int)sizeof(TOUT)) == 0) && (((uintptr_t)Cb) % 16 == 0) &&
                         (!Rb || ((g.ldr % 4 == 0) && (((uintptr_t)Rb) % 16 == 0)));
    // The per-column parameters of the register epilogues (and the row statistics of a folded LayerNorm) are
    // requested here, in front of the barrier that ends the main loop, into the registers the operand
    // fragments have left: their latency (a first touch of these lines per tile) passes while the workgroup
    // gathers, not in front of the first store.
    const int lcol = wcol0 + VW * kq;                      // store v of a row starts at lcol + v * VS
    float bias_[NV][VW], gam_[NV][VW], csum_[LN_IN ? NV : 1][VW];
    float mean_[LN_IN ? FM : 1], rstd_[LN_IN ? FM : 1];
    {
        const bool vec = (((uintptr_t)g.bias | (uintptr_t)g.gamma | (uintptr_t)g.ln_colsum) & 15) == 0;
        auto cols = [&](const float* p, float def, auto& dst, int nv) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (v >= nv) break;
#pragma unroll
                for (int q = 0; q < VW / 4; ++q) {
                    f32x4 t = {def, def, def, def};
                    if (p && inside) {
                        const float* src = p + lcol + v * VS + 4 * q;
                        if (vec) t = *(const f32x4*)src;
                        else t = (f32x4){src[0], src[1], src[2], src[3]};
                    }
                    dst[v][4 * q] = t[0]; dst[v][4 * q + 1] = t[1]; dst[v][4 * q + 2] = t[2]; dst[v][4 * q + 3] = t[3];
                }
            }
        };
        cols(g.bias, 0.f, bias_, NV);
        cols(g.gamma, 1.f, gam_, NV);
        if constexpr (LN_IN) {
            cols(g.ln_colsum, 0.f, csum_, NV);
            if (g.ln_stat) {
#pragma unroll
                for (int m = 0; m < FM; ++m) {
                    const int row = min(wrow0 + r16 + 16 * m, g.M - 1);
                    const float2 st = *(const float2*)(g.ln_stat + 2 * (int64_t)row);
                    mean_[m] = st.x;
                    rstd_[m] = st.y;
                }
            }
        }
    }
    // every wave is past its last fragment read before the ring is reused for the epilogue
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    MARS_STAMP(3)
    // PERSIST: locate the next tile of this workgroup and request its first RING steps.  Called by the register
    // epilogues behind their last load (a load issued behind the fill would wait for the whole fill to land)
    // and in front of their stores; by the staged epilogue when it is done with the ring.
    const bool has_next = PERSIST && vb + (int)gridDim.x < nwg;
    int nrow0 = row0, ncol0 = col0;
    // vector memory instructions this wave issues behind the fill (they retire behind it, in order), when the
    // epilogue knows the number: the next tile then waits for the fill alone, not for the stores; 0 = unknown
    after_fill = 0;
    auto prefetch_next = [&]() {
        if constexpr (PERSIST) {
            if (has_next) {
                // nothing of the epilogue may move up across the fill (it would be counted behind it)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                locate(vb + (int)gridDim.x, nrow0, ncol0);
                set_sources(nrow0, ncol0);
                fill_ring();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    auto fast = [&](auto mode_tag, auto ln_tag) {
        constexpr int MODE = decltype(mode_tag)::value;    // 0 plain, 1 QuickGELU, 2 residual add, 3 erf GELU, 4 ReLU
        // a LayerNorm folded into the products around it.  The f32 stream is then kept as two bf16 planes, hi =
        // bf16(x) and lo = bf16(x - hi) (16 mantissa bits between them), so that the plane the next product
        // takes as its operand is part of the stream and costs no traffic of its own.
        // 1 = this product adds to the stream: residual from the planes, result back into them, row sums out;
        // 3 = the same, last of a run: result out as f32;  2 = this product reads hi in place of normalised rows
        constexpr int LN = decltype(ln_tag)::value;
        constexpr bool PLANES = LN == 1 || LN == 3;
        static_assert(LN == 0 || (PLANES && MODE == 2 && sizeof(TOUT) == 4 && VW == 8) || (LN == 2 && MODE <= 1 && sizeof(TOUT) == 2));
        // Lanes of a C^T fragment that follow one another hold different ROWS, so a 16-byte access per lane reaches
        // memory as 64 requests of 16 bytes; four consecutive lanes on 64 consecutive bytes of one row go out as one
        // request and move 1.5-1.8 times the bytes per cycle (tools/diag_store.hip).  The residual is therefore
        // loaded, and the result stored, in that other lane order - lane L on row L >> 2, 16-byte piece L & 3 of the
        // fragment row - and moved between the two orders with ds_bpermute, four dwords per access.
        const int r16s = lane >> 2, kqs = lane & 3;
        const int to_store = (((lane & 3) << 4) | (lane >> 2)) << 2;     // lane L of the store order takes fragment lane (L & 3, L >> 2)
        const int to_frag = (((lane & 15) << 2) | (lane >> 4)) << 2;      // fragment lane (kq, r16) takes lane 4 r16 + kq
        auto xlane = [&](uint4 v, int idx) {
            uint4 r;
            r.x = (unsigned)__builtin_amdgcn_ds_bpermute(idx, (int)v.x);
            r.y = (unsigned)__builtin_amdgcn_ds_bpermute(idx, (int)v.y);
            r.z = (unsigned)__builtin_amdgcn_ds_bpermute(idx, (int)v.z);
            r.w = (unsigned)__builtin_amdgcn_ds_bpermute(idx, (int)v.w);
            return r;
        };
        const int lcol_s = wcol0 + VW * kqs;
        TOUT* cp = Cb + (int64_t)(wrow0 + r16s) * g.ldc + lcol_s;
        // residual rows are requested PD - 1 fragment rows ahead of their use, in the registers the
        // operand fragments no longer need; a load issued after a store would retire behind it
        constexpr int PD = 3;
        uint4 res[PLANES ? 1 : PD][NV][VW / 4];
        uint4 rhi[PLANES ? PD : 1][NV], rlo[PLANES ? PD : 1][NV];
        const float* rp = (MODE == 2 && !PLANES) ? Rb + (int64_t)(wrow0 + r16s) * g.ldr + lcol_s : nullptr;
        __bf16* hp = PLANES ? (__bf16*)g.ln_copy + (int64_t)(wrow0 + r16s) * g.ldcopy + lcol_s : nullptr;
        __bf16* lp = PLANES ? (__bf16*)g.ln_lo + (int64_t)(wrow0 + r16s) * g.ldcopy + lcol_s : nullptr;
        const int rows_left = g.M - (wrow0 + r16);         // fragment row m of this lane exists iff 16 m < rows_left
        const int rows_left_s = g.M - (wrow0 + r16s);      // the same for the row this lane loads and stores
        // every fragment row of this wave exists: the stores below are all issued
        const bool whole = wrow0 + 16 * FM <= g.M;
        constexpr int ST_ROW = LN == 1 ? 2 * NV + 1 : NV * (sizeof(TOUT) == 4 ? VW / 4 : 1);   // stores per fragment row
        if constexpr (MODE != 2) {
            prefetch_next();
            if (whole) after_fill = FM * ST_ROW;
        }
        auto load_res = [&](int m) {
            if (16 * m < rows_left_s) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if constexpr (PLANES) {
                        rhi[m % PD][v] = *(const uint4*)(hp + (int64_t)m * 16 * g.ldcopy + v * VS);
                        rlo[m % PD][v] = *(const uint4*)(lp + (int64_t)m * 16 * g.ldcopy + v * VS);
                    } else {
#pragma unroll
                        for (int q = 0; q < VW / 4; ++q)
                            res[m % PD][v][q] = *(const uint4*)(rp + (int64_t)m * 16 * g.ldr + v * VS + 4 * q);
                    }
                }
            }
        };
        if constexpr (MODE == 2) {
#pragma unroll
            for (int m = 0; m < PD - 1; ++m) load_res(m);
        }
#pragma unroll
        for (int m = 0; m < FM; ++m) {
            if constexpr (MODE == 2) {
                if (m + PD - 1 < FM) load_res(m + PD - 1);
                if (m == (FM > PD ? FM - PD : 0)) {                     // the last residual row is on its way
                    prefetch_next();
                    if (whole) after_fill = (FM - m) * ST_ROW;
                }
            }
            float rs = 0.f, rss = 0.f;                     // LN == 1: this lane's part of the row sums
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float x[VW];
                // the residual of this fragment row, brought into the fragment's lane order
                float radd[VW];
                if constexpr (PLANES) {
                    const bf16x8 fh = __builtin_bit_cast(bf16x8, xlane(rhi[m % PD][v], to_frag));
                    const bf16x8 fl = __builtin_bit_cast(bf16x8, xlane(rlo[m % PD][v], to_frag));
#pragma unroll
                    for (int e = 0; e < 8; ++e) radd[e] = (float)fh[e] + (float)fl[e];
                } else if constexpr (MODE == 2) {
#pragma unroll
                    for (int q = 0; q < VW / 4; ++q) {
                        const f32x4 f = __builtin_bit_cast(f32x4, xlane(res[m % PD][v][q], to_frag));
                        radd[4 * q] = f[0]; radd[4 * q + 1] = f[1]; radd[4 * q + 2] = f[2]; radd[4 * q + 3] = f[3];
                    }
                }
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    const float a0 = PERM ? acc[m][2 * v + (e >> 2)][e & 3] : acc[m][v][e & 3];
                    if constexpr (LN == 2) x[e] = fmaf(rstd_[m], fmaf(-mean_[m], csum_[v][e], a0), bias_[v][e]);
                    else x[e] = g.alpha * a0 + bias_[v][e];
                }
                if constexpr (MODE == 1 && FAST) {
                    // QuickGELU two outputs at a time: the multiplies and the add around the two transcendentals
                    // are packed instructions (act_fn3 per element leaves them single)
#pragma unroll
                    for (int e = 0; e < VW; e += 2) {
                        const f32x2 y = {x[e], x[e + 1]};
                        f32x2 t = y * (f32x2){-2.4554669595930157f, -2.4554669595930157f};
                        t = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                        t = t + (f32x2){1.0f, 1.0f};
                        t = (f32x2){__builtin_amdgcn_rcpf(t[0]), __builtin_amdgcn_rcpf(t[1])};
                        t = y * t;
                        x[e] = t[0];
                        x[e + 1] = t[1];
                    }
                }
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    float y = x[e];
                    if constexpr (MODE == 1 && !FAST) y = act_fn3<FAST>(y, 2);
                    if constexpr (MODE == 3) y = act_fn3<FAST>(y, 1);
                    if constexpr (MODE == 4) y = fmaxf(y, 0.f);
                    y *= gam_[v][e];
                    if constexpr (MODE == 2) y += radd[e];
                    x[e] = y;
                    if constexpr (LN == 1) {
                        rs += y;
                        rss = fmaf(y, y, rss);
                    }
                }
                TOUT* c = cp + (int64_t)m * 16 * g.ldc + v * VS;
                if constexpr (LN == 1) {
                    bf16x8 oh, ol;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        oh[e] = (__bf16)x[e];
                        ol[e] = (__bf16)(x[e] - (float)oh[e]);
                    }
                    const uint4 sh = xlane(__builtin_bit_cast(uint4, oh), to_store), sl = xlane(__builtin_bit_cast(uint4, ol), to_store);
                    if (16 * m < rows_left_s) {
                        *(uint4*)(hp + (int64_t)m * 16 * g.ldcopy + v * VS) = sh;
                        *(uint4*)(lp + (int64_t)m * 16 * g.ldcopy + v * VS) = sl;
                    }
                } else if constexpr (sizeof(TOUT) == 4) {
#pragma unroll
                    for (int q = 0; q < VW / 4; ++q) {
                        const uint4 sv = xlane(__builtin_bit_cast(uint4, (f32x4){x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]}), to_store);
                        if (16 * m < rows_left_s) *(uint4*)(c + 4 * q) = sv;
                    }
                } else {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (__bf16)x[e];
                    const uint4 sv = xlane(__builtin_bit_cast(uint4, o), to_store);
                    if (16 * m < rows_left_s) *(uint4*)c = sv;
                }
            }
            if constexpr (LN == 1) {
                // the four lanes that share a row (one per group of 16) add their parts; the first writes
                rs += __shfl_xor(rs, 16);
                rss += __shfl_xor(rss, 16);
                rs += __shfl_xor(rs, 32);
                rss += __shfl_xor(rss, 32);
                if (kq == 0 && 16 * m < rows_left)
                    *(float2*)(g.ln_part + ((int64_t)(wrow0 + r16 + 16 * m) * (g.N / WN) + wcol0 / WN) * 2) = make_float2(rs, rss);
            }
            if (m == FM / 4 - 1) { MARS_STAMP(4) }
            if (m == FM / 2 - 1) { MARS_STAMP(5) }
            if (m == 3 * FM / 4 - 1) { MARS_STAMP(6) }
            if (m == FM - 1) { MARS_STAMP(7) }
        }
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // the launcher admits the folded LayerNorm forms only for products every wave tile of which is inside and
    // aligned (N a multiple of 64), so they never reach the per-element path
    const bool planes = LN_OUT && g.ln_lo != nullptr;
    if (inside && aligned && g.act == 0 && (Rb || planes)) {
        if constexpr (LN_OUT) {
            if (planes && g.ln_part) fast(I2{}, I1{});
            else if (planes) fast(I2{}, std::integral_constant<int, 3>{});
            else fast(I2{}, I0{});
        } else fast(I2{}, I0{});
    } else if (inside && aligned && g.act == 0 && !Rb) {
        if constexpr (LN_IN) {
            if (g.ln_stat) fast(I0{}, I2{});
            else fast(I0{}, I0{});
        } else fast(I0{}, I0{});
    } else if (inside && aligned && g.act == 2 && !Rb) {
        if constexpr (LN_IN) {
            if (g.ln_stat) fast(I1{}, I2{});
            else fast(I1{}, I0{});
        } else fast(I1{}, I0{});
    } else if (inside && aligned && g.act == 1 && !Rb) fast(std::integral_constant<int, 3>{}, I0{});
    else if (inside && aligned && g.act == 3 && !Rb) fast(std::integral_constant<int, 4>{}, I0{});
    else {
        constexpr int EP_LD = FN * 16 + 4;
        // slabs of 32 rows per wave where the ring has room for them, of 16 otherwise
        constexpr int SLAB = NWAVES * 32 * EP_LD * 4 <= RING * STAGE ? 32 : 16;
        constexpr int CPL = 16 / (int)sizeof(TOUT), LPR = WN / CPL, RPP = 64 / LPR, NPASS = SLAB / RPP;
        static_assert(NWAVES * SLAB * EP_LD * 4 <= RING * STAGE && NPASS >= 1, "epilogue staging does not fit");
        float* ep = (float*)lds + wave * SLAB * EP_LD;
        const int ec = (lane % LPR) * CPL, er = lane / LPR;
        const int col = wcol0 + ec;
#pragma unroll
        for (int half = 0; half < FM * 16 / SLAB; ++half) {
#pragma unroll
            for (int mm = 0; mm < SLAB / 16; ++mm)
#pragma unroll
                for (int n = 0; n < FN; ++n) {
                    const int c0 = PERM ? 32 * (n >> 1) + 8 * kq + 4 * (n & 1) : 16 * n + 4 * kq;
                    *(f32x4*)(ep + (mm * 16 + r16) * EP_LD + c0) = acc[half * (SLAB / 16) + mm][n];
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
            for (int it = 0; it < NPASS; ++it) {
                const int lr2 = it * RPP + er;
                const int row = wrow0 + half * SLAB + lr2;
                if (row >= g.M) continue;
#pragma unroll 1
                for (int e = 0; e < CPL; ++e) {
                    const int c = col + e;
                    if (c >= g.N) break;
                    float x = g.alpha * ep[lr2 * EP_LD + ec + e] + (g.bias ? g.bias[c] : 0.f);
                    x = act_fn3<false>(x, g.act) * (g.gamma ? g.gamma[c] : 1.f);
                    if (Rb) x += Rb[(int64_t)row * g.ldr + c];
                    st_from_f32<TOUT>(Cb + (int64_t)row * g.ldc + c, x);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (half < 4) { MARS_STAMP(4 + half) }
        }
        if constexpr (PERSIST) {
            if (has_next) {
                __builtin_amdgcn_s_barrier();              // every wave is done with its staging slab
                asm volatile("" ::: "memory");
                prefetch_next();
            }
        }
    }
#ifdef MARS_STAMPS
    if (!has_next) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MARS_STAMP(8)
        if (lane == 0) mars_stamp_buf[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 16 + 13] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if (!has_next) break;
    vb += (int)gridDim.x;
    row0 = nrow0;
    col0 = ncol0;
    warm = true;
    }                            // tiles of this workgroup
#undef MARS_PHASE
#undef MARS_WAIT_VM
}

}  // namespace

// M-tiles per raster group (tuning knob of the diagnostic tools)
int mars_ring_group_m = 8;

// one workgroup per CU slot (`slots`: CUs x workgroups a CU holds), each walking its share of the tiles; bf16
// operands, unbatched.  A product with no more tiles than slots is the plain launch.
template <int NWR, int NWC, int FM, int FN, int WPE, int RING>
static void ring_launch_persistent(const MarsGemm& g, hipStream_t s, int slots) {
    constexpr int BM = 16 * FM * NWR, BN = 16 * FN * NWC;
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    if (slots > 8) slots -= slots % 8;                     // a whole number of workgroups per XCD (locality only)
    dim3 grid(tiles < slots || slots <= 0 ? tiles : slots, 1), block(64 * NWR * NWC);
    if (g.out_dtype == 1) hipLaunchKernelGGL((gemm_tn_ring_kernel<__bf16, __bf16, NWR, NWC, FM, FN, WPE, RING, true, true>), grid, block, 0, s, g, mars_ring_group_m);
    else hipLaunchKernelGGL((gemm_tn_ring_kernel<__bf16, float, NWR, NWC, FM, FN, WPE, RING, true, true>), grid, block, 0, s, g, mars_ring_group_m);
}

template <int NWR, int NWC, int FM, int FN, int WPE, int RING, bool PAIR = false>
static void ring_launch(const MarsGemm& g, hipStream_t s) {
    constexpr int BM = 16 * FM * NWR, BN = 16 * FN * NWC;
    dim3 grid(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN), g.nb0 * g.nb1), block(64 * NWR * NWC);
    if (g.in_dtype == 1) {
        if (g.out_dtype == 1) hipLaunchKernelGGL((gemm_tn_ring_kernel<__bf16, __bf16, NWR, NWC, FM, FN, WPE, RING, PAIR>), grid, block, 0, s, g, mars_ring_group_m);
        else hipLaunchKernelGGL((gemm_tn_ring_kernel<__bf16, float, NWR, NWC, FM, FN, WPE, RING, PAIR>), grid, block, 0, s, g, mars_ring_group_m);
    } else {
        if (g.out_dtype == 1) hipLaunchKernelGGL((gemm_tn_ring_kernel<float, __bf16, NWR, NWC, FM, FN, WPE, RING, PAIR>), grid, block, 0, s, g, mars_ring_group_m);
        else hipLaunchKernelGGL((gemm_tn_ring_kernel<float, float, NWR, NWC, FM, FN, WPE, RING, PAIR>), grid, block, 0, s, g, mars_ring_group_m);
    }
}

// shape 0: 256 x 256 tiles (8 waves, one workgroup per CU); 1: 256 x 128 (4 waves, two per CU);
// 2: 128 x 128 (4 waves of 64 x 64, three per CU) for products with few tiles; 3: 64 x 128 (4 waves of
// 32 x 64) for products that would not even give every CU a 128 x 128 tile
// 8: the 256 x 256 pair-fill shape as persistent workgroups (bf16 operands, one batch entry; `slots` = CUs)
void mars_gemm_ring_launch(const MarsGemm& g, hipStream_t s, int shape, int slots) {
    if (shape == 8 && g.in_dtype == 1 && g.nb0 * g.nb1 == 1) ring_launch_persistent<2, 4, 8, 4, 2, 4>(g, s, slots);
    else if (shape == 8) ring_launch<2, 4, 8, 4, 2, 4, true>(g, s);
    else if (shape == 0) ring_launch<2, 4, 8, 4, 2, 3>(g, s);
    else if (shape == 1) ring_launch<2, 2, 8, 4, 2, 3>(g, s);
    else if (shape == 2) ring_launch<2, 2, 4, 4, 3, 3>(g, s);
    else if (shape == 3) ring_launch<2, 2, 2, 4, 4, 3>(g, s);
    else if (shape == 4) ring_launch<2, 4, 8, 4, 2, 4, true>(g, s);         // 256 x 256 with pair fills (128 KiB LDS)
    else if (shape == 5) ring_launch<2, 2, 4, 4, 1, 8, true>(g, s);         // 128 x 128 with pair fills, eight slots (128 KiB)
    else if (shape == 6) ring_launch<2, 2, 2, 4, 2, 6, true>(g, s);         // 64 x 128 with pair fills, six slots (72 KiB)
    else ring_launch<2, 2, 4, 4, 2, 4, true>(g, s);                          // 128 x 128 with pair fills, four slots (64 KiB)
}
