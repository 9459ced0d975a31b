// TN GEMM on the gfx950 matrix cores: C[M,N] = A[M,K] * B[N,K]^T with a fused epilogue
//   out = residual + gamma[n] * act(alpha * acc + bias[n]).
//
// One kernel body serves both input types because the LDS image is defined in BYTES:
// a k-tile is 128 bytes per row (64 bf16 or 32 f32), cut into eight 16-byte chunks.
// Lane l of a wave owns row (l & 15) of a 16-row fragment and chunk (l >> 4) of each
// 64-byte half.  For bf16 that chunk is exactly the 8-element A/B operand of
// v_mfma_f32_16x16x32_bf16.  For f32 the four floats of the chunk feed four
// v_mfma_f32_16x16x4_f32: MFMA s takes float s of every lane, i.e. k = 4*(l>>4)+s --
// a permutation of k inside the 16-float block that A and B share, so the sum is
// unchanged.  Both MFMAs have the same C/D map (col = l & 15, row = 4*(l>>4)+reg).
//
// Tile 128x128, 4 waves as 2x2, each wave 64x64 = 4x4 fragments.  Global -> register ->
// LDS staging with the next tile's loads in flight during the MFMAs, two LDS buffers,
// one barrier per k-tile.  Chunk c of row r is stored at chunk c ^ ((r >> 1) & 7) so the
// 16 rows a ds_read_b128 lane group touches land on 16 distinct 16-byte bank slots.
// Out-of-range rows are clamped on the source address (results never stored).
#include <atomic>
#include <cstdlib>
#include "common.h"
#include "../../include/mars_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BKB = 128;  // BKB: bytes of K per tile row
constexpr int THREADS = 256;

struct Epi {
    const float* bias; const float* gamma; const float* residual;
    int64_t ldr; float alpha; int act;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == 2) return v / (1.0f + expf(-1.702f * v));
    if (act == 3) return fmaxf(v, 0.f);
    return v;
}

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * BKB + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <typename TIN, typename TOUT>
__global__ __launch_bounds__(THREADS) void gemm_tn_kernel(MarsGemm g) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][BM * BKB];  // [buf][A|B]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r16 = lane & 15, kq = lane >> 4;

    // XCD-aware tile order: blocks that share an L2 (same blockIdx % 8) walk a contiguous tile range
    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / ntn, tn = bid % ntn;
    const int row0 = tm * BM, col0 = tn * BN;
    const int z = blockIdx.y, z0 = z / g.nb1, z1 = z % g.nb1;

    const unsigned char* Ab = (const unsigned char*)g.A + (z0 * g.sA0 + z1 * g.sA1) * (int64_t)sizeof(TIN);
    const unsigned char* Bb = (const unsigned char*)g.B + (z0 * g.sB0 + z1 * g.sB1) * (int64_t)sizeof(TIN);
    const int64_t lda_b = g.lda * (int64_t)sizeof(TIN), ldb_b = g.ldb * (int64_t)sizeof(TIN);

    // staging assignment: 1024 chunks per operand tile, 4 per thread
    const unsigned char* pa[4];
    const unsigned char* pb[4];
    int dst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + THREADS * i, r = q >> 3, c = q & 7;
        int ra = row0 + r; ra = ra < g.M ? ra : g.M - 1;
        int rb = col0 + r; rb = rb < g.N ? rb : g.N - 1;
        pa[i] = Ab + ra * lda_b + c * 16;
        pb[i] = Bb + rb * ldb_b + c * 16;
        dst[i] = lds_off(r, c);
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (int)(((int64_t)g.K * sizeof(TIN)) / BKB);
    uint4 ra4[4], rb4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ra4[i] = *(const uint4*)(pa[i]); rb4[i] = *(const uint4*)(pb[i]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { *(uint4*)(&lds[0][0][dst[i]]) = ra4[i]; *(uint4*)(&lds[0][1][dst[i]]) = rb4[i]; }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            const int64_t ko = (int64_t)(kt + 1) * BKB;
#pragma unroll
            for (int i = 0; i < 4; ++i) { ra4[i] = *(const uint4*)(pa[i] + ko); rb4[i] = *(const uint4*)(pb[i] + ko); }
        }
        const unsigned char* la = &lds[buf][0][0];
        const unsigned char* lb = &lds[buf][1][0];
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            uint4 fa[4], fb[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) fa[m] = *(const uint4*)(la + lds_off(wr * 64 + m * 16 + r16, kc * 4 + kq));
#pragma unroll
            for (int n = 0; n < 4; ++n) fb[n] = *(const uint4*)(lb + lds_off(wc * 64 + n * 16 + r16, kc * 4 + kq));
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
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
        if (kt + 1 < nk) {
            const int nb = buf ^ 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) { *(uint4*)(&lds[nb][0][dst[i]]) = ra4[i]; *(uint4*)(&lds[nb][1][dst[i]]) = rb4[i]; }
        }
        __syncthreads();
    }

    // epilogue: lane holds rows 4*kq + j (j = 0..3) and column r16 of each 16x16 fragment
    TOUT* Cb = (TOUT*)g.C + (z0 * g.sC0 + z1 * g.sC1);
    const float* Rb = g.residual ? g.residual + (z0 * g.sR0 + z1 * g.sR1) : nullptr;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int col = col0 + wc * 64 + n * 16 + r16;
        if (col >= g.N) continue;
        const float bias = g.bias ? g.bias[col] : 0.f;
        const float gam = g.gamma ? g.gamma[col] : 1.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = row0 + wr * 64 + m * 16 + kq * 4 + j;
                if (row >= g.M) continue;
                float v = apply_act(g.alpha * acc[m][n][j] + bias, g.act) * gam;
                if (Rb) v += Rb[(int64_t)row * g.ldr + col];
                st_from_f32<TOUT>(Cb + (int64_t)row * g.ldc + col, v);
            }
        }
    }
}

}  // namespace

void mars_gemm_glds_launch(const MarsGemm& g, hipStream_t s, int variant);
void mars_gemm_ring_launch(const MarsGemm& g, hipStream_t s, int shape, int slots);
// The two tuning hooks below set process-wide state that every later marship_gemm call of every thread reads.
// They are relaxed atomics: reading them while another thread sets them is not a data race, but a setter is not
// ordered with products other threads are launching - set them at start-up, or while nobody else launches.
static std::atomic<int> g_gemm_variant{4};

// Tuning hook: 0 = register-staged 128x128, 1 = LDS-DMA 128x128, 2 = LDS-DMA 256x128,
// 3 = LDS-DMA ring 256x256, 4 = choose by shape (default), 5 = LDS-DMA ring 256x128,
// 6 = LDS-DMA ring 128x128, 7 = LDS-DMA ring 64x128, 8 = LDS-DMA ring 256x256 with pair fills,
// 9 = 128x128 with pair fills and eight slots, 10 = 64x128 with pair fills and six slots,
// 11 = 128x128 with pair fills and four slots, 12 = 8 as persistent workgroups (one per CU, each filling the
// ring for its next tile under the epilogue of the one in hand).
extern "C" int marship_set_gemm_variant(int v) {
    if (v >= 0 && v <= 12) return g_gemm_variant.exchange(v, std::memory_order_relaxed);
    return g_gemm_variant.load(std::memory_order_relaxed);
}

// Tuning hook: 1 (default) lets "choose by shape" cut a tall product into two launches when the tiles of one
// launch would leave most of the chip idle in their last round; 0 keeps every product in one launch.
static std::atomic<int> g_gemm_split{1};
extern "C" int marship_set_gemm_split(int on) { return g_gemm_split.exchange(on != 0, std::memory_order_relaxed); }

// the kernel variant the launch policy picks for a product (the variant set by the tuning hook, or, for
// "choose by shape", the ring shape: 5, 6, 7, 8, 10, 11 or 12)
// MARS_GEMM_PERSIST=1 runs the 256 x 256 shape as persistent workgroups (variant 12; read once, for A/B runs).
// Off by default: per product it is within 2 % of one workgroup per tile either way, and a whole step of the
// benchmark is 4 % slower with it (44.2 against 42.5 ms, three interleaved runs).
static bool persist_default() {
    static const bool on = [] { const char* e = getenv("MARS_GEMM_PERSIST"); return e && e[0] == '1'; }();
    return on;
}

static int pick_variant(const MarsGemm* g) {
    int v = g_gemm_variant.load(std::memory_order_relaxed);
    if (v == 4) {
        // the 256 x 128 ring shape needs about a full wave of workgroups to pay off; below that
        // the 128 x 128 shape spreads the product over more CUs (bf16: with pair fills)
        const int64_t big = (int64_t)((g->M + 255) / 256) * ((g->N + 255) / 256) * g->nb0 * g->nb1;
        v = big >= 192 ? 5 : (g->in_dtype == 1 ? 11 : 6);
        // long products without an activation run faster on the 256 x 256 shape with pair fills (one barrier
        // per 128 bytes of K); an activation epilogue, or a short product whose time is mostly its residual
        // traffic, is better hidden behind the second 256 x 128 workgroup of the CU
        // (QuickGELU, an exp2 and a reciprocal per output, is cheap enough to go with them)
        // (as persistent workgroups when the product is one batch entry)
        if (v == 5 && g->in_dtype == 1 && (g->act == 0 || g->act == 2) && (g->N >= 2048 || g->K >= 2048)) v = g->nb0 * g->nb1 == 1 && persist_default() ? 12 : 8;
        // a short, narrow bf16 product (the attention projection) is mostly its f32 residual traffic: the
        // 128 x 128 pair-fill shape balances 2.3 rounds of 256-row tiles better (4.5 rounds of its own)
        else if (v == 5 && g->in_dtype == 1 && g->act == 0 && g->out_dtype == 0) v = 11;
        // few 128 x 128 tiles: halve the tile height to spread the work (bf16: up to about one tile per
        // CU, on the pair-fill shape; f32: up to about half the CUs)
        const int64_t t128 = (int64_t)((g->M + 127) / 128) * ((g->N + 127) / 128) * g->nb0 * g->nb1;
        // (long products pay the smaller tile's extra operand bytes 128 steps long: 264 tiles at K = 4096, the rows
        // of the AlphaCLIP fc2 beyond its full rounds, take 68 us on 128 x 128 against 79; 176 tiles 57 against 51)
        if (v == 11 && t128 < (g->K >= 2048 ? 220 : 300)) v = 10;
        if (v == 6 && t128 < 150) v = 7;
    }
    return v;
}

// Tuning hook: workgroups a persistent launch (variant 12) uses; 0 (default) = one per CU.  Fewer make every
// workgroup walk more tiles (tests use that to reach the multi-tile path on small products).
static std::atomic<int> g_gemm_slots{0};
extern "C" int marship_set_gemm_slots(int slots) { return g_gemm_slots.exchange(slots > 0 ? slots : 0, std::memory_order_relaxed); }

static int cu_count() {
    static std::atomic<int> cached{0};
    int n = cached.load(std::memory_order_relaxed);
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            n = cus;
        else
            n = 256;
        cached.store(n, std::memory_order_relaxed);
    }
    return n;
}

// Rows the first of two launches covers, or M when the product runs as one launch.  The 256 x 256 shape keeps
// one workgroup on a CU, so a product runs in rounds of that many tiles; when the last round is less than 0.6
// full (the AlphaCLIP FC2: 2.27 rounds) the rows of the full rounds go out as one launch and the rest as a launch
// of their own, which the policy then gives a smaller tile.  (The same cut on the two-per-CU shapes was slower:
// 256 x 128 137 -> 149 us, the projection on 128 x 128 with its 4.5 rounds 136 -> 145 us.)
static int split_rows(const MarsGemm* g, int v) {
    if (!g_gemm_split.load(std::memory_order_relaxed) || g_gemm_variant.load(std::memory_order_relaxed) != 4 || g->nb0 * g->nb1 != 1 || (v != 8 && v != 12)) return g->M;
    const int slots = cu_count();
    const int64_t ntm = (g->M + 255) / 256, ntn = (g->N + 255) / 256, tiles = ntm * ntn;
    const int64_t full = tiles / slots, rem = tiles % slots;
    if (full < 1 || full > 4 || rem == 0 || rem * 10 > (int64_t)slots * 6) return g->M;
    const int64_t row_tiles = full * slots / ntn;
    if (row_tiles < 1 || row_tiles >= ntm) return g->M;
    return (int)(row_tiles * 256);
}

// Columns the first of two launches covers, or N when the product is not cut by columns.  A bf16 product of between one and
// two rounds of 256 x 256 tiles whose row tiles fit a round (the fc1 of a ViT-H layer on 4096 tokens: 16 x 20 tiles
// = 1.25 rounds) goes out as whole columns of tiles that fill one round on the 256 x 256 pair-fill shape (16 x 16) and
// the remaining columns as a product of their own (80 -> 68 us); whatever the policy would pick for the whole
// product leaves a quarter of a second round to a few CUs.
static int split_cols(const MarsGemm* g) {
    if (!g_gemm_split.load(std::memory_order_relaxed) || g_gemm_variant.load(std::memory_order_relaxed) != 4) return g->N;
    if (g->in_dtype != 1 || g->nb0 * g->nb1 != 1 || g->ln_copy || g->ln_lo || g->ln_part || g->ln_stat || g->ln_colsum) return g->N;
    const int slots = cu_count();
    const int64_t ntm = (g->M + 255) / 256, ntn = (g->N + 255) / 256, tiles = ntm * ntn;
    if (g->K < 1024 || tiles <= slots || tiles >= 2 * (int64_t)slots || ntm > slots) return g->N;
    const int64_t cols = slots / ntm;                    // whole columns of tiles in one round
    if (cols < 1 || cols >= ntn || cols * ntm * 4 < (int64_t)slots * 3) return g->N;
    return (int)(cols * 256);
}

extern "C" int marship_gemm_variant_for(const MarsGemm* g) {
    MARS_CHECK(g && g->M > 0 && g->N > 0 && g->nb0 > 0 && g->nb1 > 0, "gemm_variant_for: bad shape");
    return pick_variant(g);
}

extern "C" int marship_gemm_split_rows(const MarsGemm* g) {
    MARS_CHECK(g && g->M > 0 && g->N > 0 && g->nb0 > 0 && g->nb1 > 0, "gemm_split_rows: bad shape");
    return split_rows(g, pick_variant(g));
}

extern "C" int marship_gemm(const MarsGemm* g, void* stream) {
    MARS_CHECK(g && g->A && g->B && g->C, "gemm: null operand");
    MARS_CHECK(g->M > 0 && g->N > 0 && g->K > 0 && g->nb0 > 0 && g->nb1 > 0, "gemm: bad shape M=%d N=%d K=%d", g->M, g->N, g->K);
    const int es = g->in_dtype == 1 ? 2 : 4;
    MARS_CHECK(g->in_dtype == 0 || g->in_dtype == 1, "gemm: in_dtype %d", g->in_dtype);
    MARS_CHECK(g->out_dtype == 0 || g->out_dtype == 1, "gemm: out_dtype %d", g->out_dtype);
    MARS_CHECK(((int64_t)g->K * es) % BKB == 0, "gemm: K=%d must span a multiple of 128 bytes (pad with zeros)", g->K);
    MARS_CHECK((g->lda * es) % 16 == 0 && (g->ldb * es) % 16 == 0, "gemm: lda/ldb rows must be 16-byte multiples");
    MARS_CHECK(((uintptr_t)g->A % 16) == 0 && ((uintptr_t)g->B % 16) == 0, "gemm: A/B must be 16-byte aligned");
    MARS_CHECK((g->sA0 * es) % 16 == 0 && (g->sA1 * es) % 16 == 0 && (g->sB0 * es) % 16 == 0 && (g->sB1 * es) % 16 == 0,
               "gemm: batch strides must keep 16-byte alignment");
    MARS_CHECK(g->lda >= g->K && g->ldb >= g->K && g->ldc >= 1, "gemm: leading dimension smaller than K");
    // the tile-relative operand offsets (up to 256 rows) are 32-bit
    MARS_CHECK(g->lda * es < (1 << 22) && g->ldb * es < (1 << 22), "gemm: lda/ldb rows longer than 4 MiB are not supported");
    const int ntm = (g->M + BM - 1) / BM, ntn = (g->N + BN - 1) / BN;
    dim3 grid(ntm * ntn, g->nb0 * g->nb1), block(THREADS);
    hipStream_t s = (hipStream_t)stream;
    const bool ln_out = g->ln_part || g->ln_lo, ln_in = g->ln_stat || g->ln_colsum;
    if (ln_out || ln_in) {
        // the folded LayerNorm forms exist in the register epilogues of the ring kernels only: every wave
        // tile must take them, so the conditions of those epilogues are requirements here
        MARS_CHECK(!(ln_out && ln_in), "gemm: a product either adds to the stream or reads it");
        MARS_CHECK(g->in_dtype == 1 && g->nb0 * g->nb1 == 1 && g->N % 64 == 0, "gemm: folded LayerNorm needs one bf16 product with N a multiple of 64");
        const int v0 = g_gemm_variant.load(std::memory_order_relaxed);
        MARS_CHECK(v0 == 3 || v0 == 4 || v0 == 5 || v0 >= 8, "gemm: folded LayerNorm needs one of the ring kernels with at most two waves per SIMD (variant %d set)", v0);
        MARS_CHECK(((uintptr_t)g->C % 16) == 0 && g->ldc % (g->out_dtype == 1 ? 8 : 4) == 0, "gemm: folded LayerNorm needs 16-byte aligned output rows");
        if (ln_out) {
            MARS_CHECK(g->ln_copy && g->ln_lo && g->out_dtype == 0 && g->act == 0 && !g->residual,
                       "gemm: a product that adds to the stream is f32 = planes + gamma*(A.B^T + bias): ln_copy and ln_lo set, no residual pointer");
            MARS_CHECK(((uintptr_t)g->ln_copy % 16) == 0 && ((uintptr_t)g->ln_lo % 16) == 0 && g->ldcopy % 8 == 0 && g->ldcopy >= g->N &&
                       ((uintptr_t)g->ln_part % 8) == 0, "gemm: ln_copy / ln_lo / ln_part alignment");
        } else {
            MARS_CHECK(g->ln_stat && g->ln_colsum && g->out_dtype == 1 && (g->act == 0 || g->act == 2) && !g->residual && g->alpha == 1.0f,
                       "gemm: the product that reads the stream is bf16 = gamma*act(...), act none or QuickGELU, alpha 1, no residual, ln_stat and ln_colsum both set");
            MARS_CHECK(((uintptr_t)g->ln_stat % 8) == 0, "gemm: ln_stat alignment");
        }
    }
    if (g_gemm_variant.load(std::memory_order_relaxed) != 0) {
        const int forced = g_gemm_slots.load(std::memory_order_relaxed);
        const int slots = forced > 0 ? forced : cu_count();
        const int cols = split_cols(g);
        if (cols < g->N) {
            // columns [0, cols) are one round of 256 x 256 tiles; the rest is a product of its own
            MarsGemm head = *g, tail = *g;
            head.N = cols;
            tail.N = g->N - cols;
            tail.B = (const char*)g->B + (int64_t)cols * g->ldb * es;
            tail.C = (char*)g->C + (int64_t)cols * (g->out_dtype == 1 ? 2 : 4);
            if (g->bias) tail.bias = g->bias + cols;
            if (g->gamma) tail.gamma = g->gamma + cols;
            if (g->residual) tail.residual = g->residual + cols;
            mars_gemm_ring_launch(head, s, 8 - 4, slots);
            MARS_LAUNCH_CHECK();
            const int vt = pick_variant(&tail);
            if (vt == 3 || vt >= 5) mars_gemm_ring_launch(tail, s, vt == 3 ? 0 : vt - 4, slots);
            else mars_gemm_glds_launch(tail, s, vt);
            MARS_LAUNCH_CHECK();
            return 0;
        }
        const int v = pick_variant(g);
        const int rows = split_rows(g, v);
        if (rows < g->M) {
            // rows [0, rows) fill whole rounds of tiles; the rest is a product of its own
            MarsGemm head = *g, tail = *g;
            head.M = rows;
            tail.M = g->M - rows;
            tail.A = (const char*)g->A + (int64_t)rows * g->lda * es;
            tail.C = (char*)g->C + (int64_t)rows * g->ldc * (g->out_dtype == 1 ? 2 : 4);
            if (g->residual) tail.residual = g->residual + (int64_t)rows * g->ldr;
            if (g->ln_copy) tail.ln_copy = (char*)g->ln_copy + (int64_t)rows * g->ldcopy * 2;
            if (g->ln_lo) tail.ln_lo = (char*)g->ln_lo + (int64_t)rows * g->ldcopy * 2;
            if (g->ln_part) tail.ln_part = g->ln_part + (int64_t)rows * (g->N / 64) * 2;
            if (g->ln_stat) tail.ln_stat = g->ln_stat + (int64_t)rows * 2;
            mars_gemm_ring_launch(head, s, v - 4, slots);
            MARS_LAUNCH_CHECK();
            const int vt = pick_variant(&tail);
            mars_gemm_ring_launch(tail, s, vt - 4, slots);
            MARS_LAUNCH_CHECK();
            return 0;
        }
        if (v == 3 || v >= 5) mars_gemm_ring_launch(*g, s, v == 3 ? 0 : v - 4, slots);
        else mars_gemm_glds_launch(*g, s, v);
        MARS_LAUNCH_CHECK();
        return 0;
    }
    if (g->in_dtype == 1) {
        if (g->out_dtype == 1) hipLaunchKernelGGL((gemm_tn_kernel<__bf16, __bf16>), grid, block, 0, s, *g);
        else hipLaunchKernelGGL((gemm_tn_kernel<__bf16, float>), grid, block, 0, s, *g);
    } else {
        if (g->out_dtype == 1) hipLaunchKernelGGL((gemm_tn_kernel<float, __bf16>), grid, block, 0, s, *g);
        else hipLaunchKernelGGL((gemm_tn_kernel<float, float>), grid, block, 0, s, *g);
    }
    MARS_LAUNCH_CHECK();
    return 0;
}
