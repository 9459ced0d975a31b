// TN GEMM for the large-M products of the AlphaCLIP tower: 256 x 256 block tile, 8 waves
// (2 x 4), each wave 128 x 64 outputs = 8 x 4 fragments of v_mfma_f32_16x16x32_bf16 (or four
// v_mfma_f32_16x16x4_f32 per fragment pair for f32 inputs).
//
// K is consumed in steps of 64 bytes per row (32 bf16 / 16 f32).  LDS holds a ring of four
// steps (4 x (A 16 KiB + B 16 KiB) = 128 KiB, one workgroup per CU); the step three ahead is
// fetched by LDS-DMA (global_load_lds, 1 KiB per wave instruction = 16 rows x 64 B) while the
// current one feeds the MFMAs, and the DMAs stay in flight across the per-step barrier:
//
//     top of step s:  s_waitcnt vmcnt(8)   this wave's DMAs of step s have landed (s+1, s+2 still flying)
//                     s_barrier            => every wave's have, and everybody is done reading step s-1
//                     issue DMAs of step s+3 into the slot step s-1 used
//                     12 x ds_read_b128, 32 MFMAs
//
// Bank swizzle (rows are 64 B, four 16-B chunks): chunk c of row r sits at chunk slot
// c ^ g((r >> 2) & 3), g = {0,3,2,1}; the DMA image is lane-linear, so the xor is applied to
// the per-lane SOURCE address and again on the fragment reads.
#include "common.h"
#include "../../include/mars_hip.h"

namespace {

constexpr int BM = 256, BN = 256, STEP_B = 64;           // bytes of K per row and step
constexpr int NWAVES = 8, THREADS = 512;
constexpr int A_BYTES = BM * STEP_B, B_BYTES = BN * STEP_B, STAGE = A_BYTES + B_BYTES, RING = 4;
constexpr int FM = 8, FN = 4;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const unsigned char* src, unsigned char* lds_dst) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ int gsw(int x) { return (4 - x) & 3; }

template <bool FAST>
__device__ __forceinline__ float act_fn3(float v, int act) {
    if (act == 1) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == 2) {
        if constexpr (FAST) return v * __frcp_rn(1.0f + __expf(-1.702f * v));
        else return v / (1.0f + expf(-1.702f * v));
    }
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

template <typename TIN, typename TOUT>
__global__ __launch_bounds__(THREADS) void gemm_tn_ring_kernel(MarsGemm g) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[RING * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r16 = lane & 15, kq = lane >> 4;

    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {   // blocks that share an L2 (same blockIdx % 8) walk a contiguous range of tiles ...
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // ... in a grouped raster, so the 32 tiles an XCD has in flight form a compact patch
    constexpr int GROUP_M = 8;
    const int gsize = GROUP_M * ntn;
    const int first_m = (bid / gsize) * GROUP_M;
    const int gm = min(ntm - first_m, GROUP_M);
    const int tm = first_m + (bid % gsize) % gm, tn = (bid % gsize) / gm;
    const int row0 = tm * BM, col0 = tn * BN;
    const int z = blockIdx.y, z0 = z / g.nb1, z1 = z % g.nb1;
    const unsigned char* Ab = (const unsigned char*)g.A + (z0 * g.sA0 + z1 * g.sA1) * (int64_t)sizeof(TIN);
    const unsigned char* Bb = (const unsigned char*)g.B + (z0 * g.sB0 + z1 * g.sB1) * (int64_t)sizeof(TIN);
    const int64_t lda_b = g.lda * (int64_t)sizeof(TIN), ldb_b = g.ldb * (int64_t)sizeof(TIN);

    // DMA sources: instruction i (0,1) of this wave fills rows (i*8 + wave)*16 .. +15 of A and of B
    const unsigned char* pa[2];
    const unsigned char* pb[2];
    {
        const int lr = lane >> 2, ls = lane & 3;
        const int chunk = ls ^ gsw((lr >> 2) & 3);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (i * NWAVES + wave) * 16 + lr;
            int ga = row0 + r; ga = ga < g.M ? ga : g.M - 1;
            int gb = col0 + r; gb = gb < g.N ? gb : g.N - 1;
            pa[i] = Ab + ga * lda_b + (chunk << 4);
            pb[i] = Bb + gb * ldb_b + (chunk << 4);
        }
    }
    auto issue = [&](int s) {
        unsigned char* sa = lds + (s & (RING - 1)) * STAGE;
        const int64_t ko = (int64_t)s * STEP_B;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(pa[i] + ko, sa + (i * NWAVES + wave) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(pb[i] + ko, sa + A_BYTES + (i * NWAVES + wave) * 1024);
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int m = 0; m < FM; ++m)
#pragma unroll
        for (int n = 0; n < FN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ns = (int)(((int64_t)g.K * sizeof(TIN)) / STEP_B);
    const int lane_off = r16 * STEP_B + ((kq ^ gsw((r16 >> 2) & 3)) << 4);
    const unsigned a_off = wr * 128 * STEP_B + lane_off;
    const unsigned b_off = A_BYTES + wc * 64 * STEP_B + lane_off;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)lds;

    issue(0);
    if (ns > 1) issue(1);
    if (ns > 2) issue(2);
    for (int s = 0; s < ns; ++s) {
        // the 4 DMAs of step s are the oldest outstanding ones of this wave
        if (s + 2 < ns) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (s + 1 < ns) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 3 < ns) issue(s + 3);
        // Fragment reads are inline asm on purpose: hipcc treats an in-flight LDS-DMA as a pending LDS
        // write and would put s_waitcnt vmcnt(0) in front of any ds_read it can see, draining the ring.
        const unsigned sbase = lds_base + (unsigned)((s & (RING - 1)) * STAGE);
        const unsigned aa = sbase + a_off, ba = sbase + b_off;
        uint4 fa[FM], fb[FN];
        lds_read128<0>(fb[0], ba); lds_read128<1024>(fb[1], ba); lds_read128<2048>(fb[2], ba); lds_read128<3072>(fb[3], ba);
        lds_read128<0>(fa[0], aa); lds_read128<1024>(fa[1], aa); lds_read128<2048>(fa[2], aa); lds_read128<3072>(fa[3], aa);
        lds_read128<4096>(fa[4], aa); lds_read128<5120>(fa[5], aa); lds_read128<6144>(fa[6], aa); lds_read128<7168>(fa[7], aa);
#define MARS_ROW(m, cnt)                                                                           \
        asm volatile("s_waitcnt lgkmcnt(" #cnt ")" ::: "memory");                                   \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        _Pragma("unroll") for (int n = 0; n < FN; ++n) mma_step<TIN>(acc[m][n], fa[m], fb[n]);
        MARS_ROW(0, 7) MARS_ROW(1, 6) MARS_ROW(2, 5) MARS_ROW(3, 4)
        MARS_ROW(4, 3) MARS_ROW(5, 2) MARS_ROW(6, 1) MARS_ROW(7, 0)
#undef MARS_ROW
    }
    // all fragment reads are complete (their data fed the MFMAs); meet before the ring is reused
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---------------- epilogue through LDS: 32 rows of the wave's tile at a time ----------------
    constexpr bool FAST = sizeof(TIN) == 2;
    constexpr int EP_LD = FN * 16 + 4;
    static_assert(NWAVES * 32 * EP_LD * 4 <= RING * STAGE, "epilogue staging does not fit");
    TOUT* Cb = (TOUT*)g.C + (z0 * g.sC0 + z1 * g.sC1);
    const float* Rb = g.residual ? g.residual + (z0 * g.sR0 + z1 * g.sR1) : nullptr;
    float* ep = (float*)lds + wave * 32 * EP_LD;
    const int wrow0 = row0 + wr * 128, wcol0 = col0 + wc * 64;
    const int ec = (lane & 15) * 4, er = lane >> 4;    // 16 four-column chunks per row, 4 rows per pass
    const bool vec_out = (g.ldc % 4 == 0) && (((uintptr_t)Cb) % 16 == 0);
    const bool vec_res = Rb && (g.ldr % 4 == 0) && (((uintptr_t)Rb) % 16 == 0);
    const int col = wcol0 + ec;
    const int nv = col < g.N ? (g.N - col >= 4 ? 4 : g.N - col) : 0;
    float bias4[4], gam4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        bias4[e] = (g.bias && e < nv) ? g.bias[col + e] : 0.f;
        gam4[e] = (g.gamma && e < nv) ? g.gamma[col + e] : 1.f;
    }
#pragma unroll
    for (int half = 0; half < FM / 2; ++half) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int n = 0; n < FN; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) ep[(mm * 16 + kq * 4 + j) * EP_LD + n * 16 + r16] = acc[half * 2 + mm][n][j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int lr2 = it * 4 + er;
            const int row = wrow0 + half * 32 + lr2;
            if (row >= g.M || nv == 0) continue;
            f32x4 v = *(const f32x4*)(ep + lr2 * EP_LD + ec);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fn3<FAST>(g.alpha * v[e] + bias4[e], g.act) * gam4[e];
            if (Rb) {
                const float* rp = Rb + (int64_t)row * g.ldr + col;
                if (vec_res && nv == 4) { const f32x4 r = *(const f32x4*)rp; v += r; }
                else { for (int e = 0; e < nv; ++e) v[e] += rp[e]; }
            }
            TOUT* cp = Cb + (int64_t)row * g.ldc + col;
            if (vec_out && nv == 4) {
                if constexpr (sizeof(TOUT) == 4) *(f32x4*)cp = v;
                else { bf16x4 o; o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3]; *(bf16x4*)cp = o; }
            } else {
                for (int e = 0; e < nv; ++e) st_from_f32<TOUT>(cp + e, v[e]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

}  // namespace

void mars_gemm_ring_launch(const MarsGemm& g, hipStream_t s) {
    dim3 grid(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN), g.nb0 * g.nb1), block(THREADS);
    if (g.in_dtype == 1) {
        if (g.out_dtype == 1) hipLaunchKernelGGL((gemm_tn_ring_kernel<__bf16, __bf16>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm_tn_ring_kernel<__bf16, float>), grid, block, 0, s, g);
    } else {
        if (g.out_dtype == 1) hipLaunchKernelGGL((gemm_tn_ring_kernel<float, __bf16>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm_tn_ring_kernel<float, float>), grid, block, 0, s, g);
    }
}
