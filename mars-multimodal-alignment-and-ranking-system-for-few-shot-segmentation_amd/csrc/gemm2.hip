// TN GEMM, LDS-DMA staged variant: global_load_lds (16 B per lane, 1 KiB per wave
// instruction) writes the k-tiles straight into LDS, so the main loop holds no
// staging registers and issues no ds_write.  The LDS image of an instruction is
// lane-linear (8 rows x 128 B), hence the bank swizzle lives on the SOURCE address:
// lane l of the instruction that fills rows r0..r0+7 fetches chunk (l & 7) ^ swz(row)
// of row r0 + (l >> 3), and the fragment reads apply the same xor.
//
// Loop shape per k-tile: issue the next tile's DMAs, read fragments and run the MFMAs
// of the current one, then wait for the DMAs and meet at one barrier.
//
// Epilogue: accumulators go through LDS (the staging buffers are free by then) so
// that every lane stores 4 consecutive columns -- 16-byte (f32) or 8-byte (bf16) stores,
// whole 256-/128-byte row segments per 16 lanes, with bias / LayerScale / residual
// fetched as float4.
#include "common.h"
#include "../../include/mars_hip.h"

namespace {

constexpr int BKB = 128;  // bytes of K per tile row

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == 1) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == 2) return v / (1.0f + expf(-1.702f * v));
    if (act == 3) return fmaxf(v, 0.f);
    return v;
}

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * BKB + ((chunk ^ ((row >> 1) & 7)) << 4); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const unsigned char* src, unsigned char* lds_dst) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_dst, 16, 0, 0);
}

// WM x WN waves, each wave owns (BM/WM) x (BN/WN) = FM*16 x FN*16 outputs
template <typename TIN, typename TOUT, int WM, int WN, int FM, int FN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_tn_glds_kernel(MarsGemm g) {
    constexpr int BM = WM * FM * 16, BN = WN * FN * 16;
    constexpr int NW = WM * WN;
    constexpr int A_BYTES = BM * BKB, B_BYTES = BN * BKB, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;  // 1-KiB DMA instructions per wave and tile
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split evenly over the waves");
    // epilogue staging: per wave 32 rows x (FN*16 + 4) floats, must fit in the two stages
    constexpr int EP_LD = FN * 16 + 4;
    static_assert(NW * 32 * EP_LD * 4 <= 2 * STAGE, "epilogue tile does not fit the staging buffers");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int r16 = lane & 15, kq = lane >> 4;

    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {   // blocks that share an L2 (same blockIdx % 8) walk a contiguous range of tiles
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // grouped raster inside that range: GROUP_M row-tiles x all column-tiles, walked down the rows
    // first, so the ~64 tiles an XCD has in flight form a square patch whose A and B panels fit its L2
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

    // per-lane DMA sources: instruction i of this wave fills rows (i*NW + wave)*8 .. +7
    const unsigned char* pa[A_INSTR];
    const unsigned char* pb[B_INSTR];
    const int lr = lane >> 3, ls = lane & 7;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int r = (i * NW + wave) * 8 + lr;
        int gr = row0 + r; gr = gr < g.M ? gr : g.M - 1;
        pa[i] = Ab + gr * lda_b + ((ls ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int r = (i * NW + wave) * 8 + lr;
        int gr = col0 + r; gr = gr < g.N ? gr : g.N - 1;
        pb[i] = Bb + gr * ldb_b + ((ls ^ ((r >> 1) & 7)) << 4);
    }
    auto stage = [&](int buf, int kt) {
        unsigned char* sa = lds + buf * STAGE;
        unsigned char* sb = sa + A_BYTES;
        const int64_t ko = (int64_t)kt * BKB;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) glds16(pa[i] + ko, sa + (i * NW + wave) * 1024);
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) glds16(pb[i] + ko, sb + (i * NW + wave) * 1024);
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int m = 0; m < FM; ++m)
#pragma unroll
        for (int n = 0; n < FN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (int)(((int64_t)g.K * sizeof(TIN)) / BKB);
    stage(0, 0);
    __syncthreads();  // hipcc drains the DMAs (vmcnt(0)) before the barrier
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
        const unsigned char* la = lds + buf * STAGE;
        const unsigned char* lb = la + A_BYTES;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            uint4 fa[FM], fb[FN];
#pragma unroll
            for (int m = 0; m < FM; ++m) fa[m] = *(const uint4*)(la + swz_off(wr * FM * 16 + m * 16 + r16, kc * 4 + kq));
#pragma unroll
            for (int n = 0; n < FN; ++n) fb[n] = *(const uint4*)(lb + swz_off(wc * FN * 16 + n * 16 + r16, kc * 4 + kq));
#pragma unroll
            for (int m = 0; m < FM; ++m)
#pragma unroll
                for (int n = 0; n < FN; ++n) {
                    if constexpr (sizeof(TIN) == 2) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, fa[m]), __builtin_bit_cast(bf16x8, fb[n]), acc[m][n], 0, 0, 0);
                    } else {
                        const f32x4 a = __builtin_bit_cast(f32x4, fa[m]);
                        const f32x4 b = __builtin_bit_cast(f32x4, fb[n]);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc[m][n], 0, 0, 0);
                    }
                }
        }
        __syncthreads();  // next tile landed (vmcnt(0)) and every wave is done reading this one
    }

    // ---------------- epilogue through LDS: 32 rows of the wave's tile at a time ----------------
    TOUT* Cb = (TOUT*)g.C + (z0 * g.sC0 + z1 * g.sC1);
    const float* Rb = g.residual ? g.residual + (z0 * g.sR0 + z1 * g.sR1) : nullptr;
    float* ep = (float*)lds + wave * 32 * EP_LD;
    const int wrow0 = row0 + wr * FM * 16, wcol0 = col0 + wc * FN * 16;
    constexpr int CPR = FN * 4;             // 4-column chunks per row of the wave tile
    constexpr int RPI = 64 / CPR;           // rows covered per pass of the 64 lanes
    const int ec = (lane % CPR) * 4, er = lane / CPR;
    const bool vec_out = (g.ldc % 4 == 0) && (((uintptr_t)Cb) % 16 == 0);
    const bool vec_res = Rb && (g.ldr % 4 == 0) && (((uintptr_t)Rb) % 16 == 0);
#pragma unroll
    for (int half = 0; half < FM / 2; ++half) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int n = 0; n < FN; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) ep[(mm * 16 + kq * 4 + j) * EP_LD + n * 16 + r16] = acc[half * 2 + mm][n][j];
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes are visible to its lanes
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int lr2 = it * RPI + er;
            const int row = wrow0 + half * 32 + lr2, col = wcol0 + ec;
            if (row >= g.M || col >= g.N) continue;
            f32x4 v = *(const f32x4*)(ep + lr2 * EP_LD + ec);
            const int nv = g.N - col >= 4 ? 4 : g.N - col;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e < nv) {
                    float x = g.alpha * v[e] + (g.bias ? g.bias[col + e] : 0.f);
                    x = act_fn(x, g.act);
                    if (g.gamma) x *= g.gamma[col + e];
                    v[e] = x;
                }
            }
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
        __builtin_amdgcn_s_waitcnt(0xC07F);  // reads of this half are done before the next half overwrites it
    }
}

template <typename TIN, typename TOUT>
void launch_cfg(const MarsGemm& g, hipStream_t s, int variant) {
    const int batches = g.nb0 * g.nb1;
    if (variant == 2) {  // 256 x 128 tile, 8 waves
        constexpr int BM = 256, BN = 128;
        dim3 grid(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN), batches), block(512);
        hipLaunchKernelGGL((gemm_tn_glds_kernel<TIN, TOUT, 4, 2, 4, 4>), grid, block, 0, s, g);
    } else {             // 128 x 128 tile, 4 waves
        constexpr int BM = 128, BN = 128;
        dim3 grid(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN), batches), block(256);
        hipLaunchKernelGGL((gemm_tn_glds_kernel<TIN, TOUT, 2, 2, 4, 4>), grid, block, 0, s, g);
    }
}

}  // namespace

void mars_gemm_glds_launch(const MarsGemm& g, hipStream_t s, int variant) {
    if (g.in_dtype == 1) {
        if (g.out_dtype == 1) launch_cfg<__bf16, __bf16>(g, s, variant);
        else launch_cfg<__bf16, float>(g, s, variant);
    } else {
        if (g.out_dtype == 1) launch_cfg<float, __bf16>(g, s, variant);
        else launch_cfg<float, float>(g, s, variant);
    }
}
