// Kernels of the Segment-Anything proposal stage that the transformer kernels do not already cover:
// row gathers (window partition / un-partition, 3x3 neighbourhoods, pixel shuffles), the decomposed
// relative-position bias, a row softmax with that bias, the random-Fourier position code, and the
// per-mask statistics of the automatic mask generator.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/mars_hip.h"

namespace {

// dst[r, 0:ncols] = idx[r] >= 0 ? src[idx[r], src_col : src_col + ncols] : 0   (elements of ES bytes)
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, int64_t ld_src, int src_col,
                                                          const int32_t* __restrict__ idx, T* __restrict__ dst, int64_t ld_dst,
                                                          int dst_col, int64_t rows, int ncols) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const int s = idx[r];
    T* d = dst + r * ld_dst + dst_col;
    if (s < 0) {
        for (int c = lane; c < ncols; c += 64) d[c] = (T)0.f;
    } else {
        const T* p = src + (int64_t)s * ld_src + src_col;
        for (int c = lane; c < ncols; c += 64) d[c] = p[c];
    }
}

// rel_h[bh, t, kh] = q[bh, t, :] . Rh[qh(t), kh, :],  rel_w[bh, t, kw] = q[bh, t, :] . Rw[qw(t), kw, :]
// q is read from the packed qkv buffer: row (b*T + t), column head*hdp + d.
// A workgroup owns one table slice: blockIdx.z = 0 fixes the query row qh (the gw queries of that row
// against Rh[qh]), blockIdx.z = 1 fixes the query column qw (the gh queries of that column against
// Rw[qw]).  Queries and table slice are staged in LDS with rows padded by one float, then every
// thread forms dot products of length hd in increasing d.
template <typename T>
__global__ __launch_bounds__(256) void relpos_bias_kernel(const T* __restrict__ qkv, int64_t ld, int Tn, int H, int hd, int hdp,
                                                          int gw, int Hk, int Wk, const float* __restrict__ Rh,
                                                          const float* __restrict__ Rw, float* __restrict__ rel_h,
                                                          float* __restrict__ rel_w) {
    extern __shared__ float shm[];
    const int gh = Tn / gw;
    const bool along_w = blockIdx.z == 1;
    const int fixed = blockIdx.x;
    if (fixed >= (along_w ? gw : gh)) return;
    const int nr = along_w ? gh : gw, nk = along_w ? Wk : Hk, pitch = hd + 1;
    const int64_t bh = blockIdx.y;
    const int head = (int)(bh % H);
    const int64_t b = bh / H;
    float* qs = shm;
    float* ts = shm + nr * pitch;
    const float* table = along_w ? Rw + (int64_t)fixed * Wk * hd : Rh + (int64_t)fixed * Hk * hd;
    const int nt = blockDim.x;
    if (sizeof(T) == 2 && (hd & 7) == 0 && (hdp & 7) == 0 && (ld & 7) == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)table & 15) == 0) {
        // 16-byte pieces: eight values of a query row, four of a table row
        const int qp = hd >> 3, tp = hd >> 2;
        for (int i = threadIdx.x; i < nr * qp; i += nt) {
            const int r = i / qp, d = (i - r * qp) * 8;
            const int t = along_w ? r * gw + fixed : fixed * gw + r;
            const bf16x8 v = *(const bf16x8*)((const __bf16*)qkv + (b * Tn + t) * ld + (int64_t)head * hdp + d);
#pragma unroll
            for (int e = 0; e < 8; ++e) qs[r * pitch + d + e] = (float)v[e];
        }
        for (int i = threadIdx.x; i < nk * tp; i += nt) {
            const int r = i / tp, d = (i - r * tp) * 4;
            const f32x4 v = ((const f32x4*)table)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) ts[r * pitch + d + e] = v[e];
        }
    } else {
        for (int i = threadIdx.x; i < nr * hd; i += nt) {
            const int r = i / hd, d = i % hd;
            const int t = along_w ? r * gw + fixed : fixed * gw + r;
            qs[r * pitch + d] = ld_as_f32<T>(qkv + (b * Tn + t) * ld + (int64_t)head * hdp + d);
        }
        for (int i = threadIdx.x; i < nk * hd; i += nt) ts[(i / hd) * pitch + i % hd] = table[i];
    }
    __syncthreads();
    float* out = along_w ? rel_w : rel_h;
    // TS x TS outputs a thread: 2 TS LDS reads per TS^2 multiply-adds (every sum still runs over d in increasing order)
    auto tiled = [&](auto ts_tag) {
        constexpr int TS = decltype(ts_tag)::value;
        const int tk = nk / TS;
        for (int tile = threadIdx.x; tile < (nr / TS) * tk; tile += nt) {
            const int r4 = (tile / tk) * TS, k4 = (tile % tk) * TS;
            float acc[TS][TS] = {};
#pragma unroll 4
            for (int d = 0; d < hd; ++d) {
                float qv[TS], tv[TS];
#pragma unroll
                for (int i = 0; i < TS; ++i) { qv[i] = qs[(r4 + i) * pitch + d]; tv[i] = ts[(k4 + i) * pitch + d]; }
#pragma unroll
                for (int i = 0; i < TS; ++i)
#pragma unroll
                    for (int j = 0; j < TS; ++j) acc[i][j] = fmaf(qv[i], tv[j], acc[i][j]);
            }
#pragma unroll
            for (int i = 0; i < TS; ++i) {
                const int t = along_w ? (r4 + i) * gw + fixed : fixed * gw + r4 + i;
#pragma unroll
                for (int j = 0; j < TS; ++j) out[(bh * Tn + t) * nk + k4 + j] = acc[i][j];
            }
        }
    };
    if ((nr & 3) == 0 && (nk & 3) == 0) { tiled(std::integral_constant<int, 4>{}); return; }
    if ((nr & 1) == 0 && (nk & 1) == 0) { tiled(std::integral_constant<int, 2>{}); return; }
    for (int o = threadIdx.x; o < nr * nk; o += nt) {
        const int r = o / nk, k = o % nk;
        const float* q = qs + r * pitch;
        const float* tb = ts + k * pitch;
        float acc = 0.f;
        for (int d = 0; d < hd; ++d) acc = fmaf(q[d], tb[d], acc);
        const int t = along_w ? r * gw + fixed : fixed * gw + r;
        out[(bh * Tn + t) * nk + k] = acc;
    }
}

// The same terms for small token grids (the 14 x 14 windows; sides and tables up to 16, bf16 queries) on the matrix
// pipe: a wave owns one table slice - the queries of grid row qh against Rh[qh], or of grid column qw against Rw[qw] -
// which is one 16 x 16 tile of v_mfma_f32_16x16x32_bf16: table rows on M, queries on N, head dims on K.  The queries
// are the B operand as they are stored; the f32 table is the A operand as three bf16 terms that add up to its value
// (8 + 8 + 8 mantissa bits), accumulated in f32.  With one workgroup per slice on the vector ALU (above) a window
// layer launched 11200 workgroups of which a fifth of the threads had work.
__global__ __launch_bounds__(256) void relpos_bias_mma_kernel(const __bf16* __restrict__ qkv, int64_t ld, int Tn, int H, int hd, int hdp, int gw,
                                                              int Hk, int Wk, const float* __restrict__ Rh, const float* __restrict__ Rw,
                                                              float* __restrict__ rel_h, float* __restrict__ rel_w, int64_t nslices,
                                                              const int32_t* __restrict__ in_rows, const __bf16* __restrict__ pad_row) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
    const int gh = Tn / gw, nsl = gh + gw;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= nslices) return;                           // (whole waves: no barrier in this kernel)
    const int64_t bh = wid / nsl;
    const int sl = (int)(wid - bh * nsl);
    const bool along_w = sl >= gh;
    const int fixed = along_w ? sl - gh : sl;
    const int nq = along_w ? gh : gw, nk = along_w ? Wk : Hk;
    const int head = (int)(bh % H);
    const int64_t b = bh / H;
    const int n = min(r16, nq - 1);
    const int t = along_w ? n * gw + fixed : fixed * gw + n;
    // (with a row map: the row of qkv that the map names for token row b Tn + t, the row of padding values for a negative entry)
    const int64_t qr = in_rows ? in_rows[b * Tn + t] : b * Tn + t;
    const __bf16* qp = (qr >= 0 ? qkv + qr * ld : pad_row) + (int64_t)head * hdp;
    const float* table = (along_w ? Rw + (int64_t)fixed * Wk * hd : Rh + (int64_t)fixed * Hk * hd) + (int64_t)min(r16, nk - 1) * hd;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // every load of the slice is requested before the first product (head dims up to 128: four K steps; the same trips
    // for every lane: a matrix instruction takes its operands from all of them)
    bf16x8 qb[4];
    f32x4 t0[4], t1[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int d0 = 32 * ks + 8 * g;
        qb[ks] = bf16x8{};
        t0[ks] = t1[ks] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (d0 < hd) {                                    // lane groups past the head's last dims add zeros
            qb[ks] = *(const bf16x8*)(qp + d0);
            t0[ks] = *(const f32x4*)(table + d0);
            t1[ks] = *(const f32x4*)(table + d0 + 4);
        }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (32 * ks >= hd) break;
        bf16x8 a0, a1, a2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = r16 < nk ? (e < 4 ? t0[ks][e & 3] : t1[ks][e & 3]) : 0.f;
            const __bf16 hi = (__bf16)v;
            const float r1 = v - (float)hi;
            const __bf16 mid = (__bf16)r1;
            a0[e] = hi;
            a1[e] = mid;
            a2[e] = (__bf16)(r1 - (float)mid);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qb[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qb[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, qb[ks], acc, 0, 0, 0);
    }
    // this lane: table rows 4 g + j against query r16
    if (r16 < nq) {
        float* out = (along_w ? rel_w : rel_h) + (bh * Tn + t) * nk;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * g + j < nk) out[4 * g + j] = acc[j];
    }
}

// P[row, k] = softmax_k(S[row, k] + rel_h[row, k / Wk] + rel_w[row, k % Wk]) for k < T, 0 for T <= k < ldp.
// One workgroup per row; the biased row is kept in LDS between the three passes.
template <typename TOUT>
__global__ __launch_bounds__(256) void softmax_bias_rows_kernel(const float* __restrict__ S, int64_t lds_, TOUT* __restrict__ P,
                                                                int64_t ldp, int T, const float* __restrict__ rel_h,
                                                                const float* __restrict__ rel_w, int Hk, int Wk) {
    extern __shared__ float rowbuf[];
    __shared__ float sm[8];
    const int64_t row = blockIdx.x;
    const float* sr = S + row * lds_;
    const float* rh = rel_h ? rel_h + row * Hk : nullptr;
    const float* rw = rel_w ? rel_w + row * Wk : nullptr;
    float mx = -INFINITY;
    for (int k = threadIdx.x; k < T; k += 256) {
        float v = sr[k];
        if (rh) v = (v + rh[k / Wk]) + rw[k % Wk];      // the order of the reference's two additions
        rowbuf[k] = v;
        mx = fmaxf(mx, v);
    }
    mx = block_reduce_max(mx, sm);
    float sum = 0.f;
    for (int k = threadIdx.x; k < T; k += 256) {
        const float e = expf(rowbuf[k] - mx);
        rowbuf[k] = e;
        sum += e;
    }
    sum = block_reduce_sum(sum, sm);
    TOUT* pr = P + row * ldp;
    for (int k = threadIdx.x; k < ldp; k += 256) st_from_f32<TOUT>(pr + k, k < T ? rowbuf[k] / sum : 0.f);
}

// The same for rows of at most 256 entries (the 14 x 14 windows): one wave per row, the row in registers.
template <typename TOUT>
__global__ __launch_bounds__(256) void softmax_bias_rows_small_kernel(const float* __restrict__ S, int64_t lds_, TOUT* __restrict__ P,
                                                                      int64_t ldp, int T, const float* __restrict__ rel_h,
                                                                      const float* __restrict__ rel_w, int Hk, int Wk, int64_t nrows) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int lane = threadIdx.x & 63;
    const float* sr = S + row * lds_;
    const float* rh = rel_h ? rel_h + row * Hk : nullptr;
    const float* rw = rel_w ? rel_w + row * Wk : nullptr;
    float v[4];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        v[j] = -INFINITY;
        if (k < T) {
            float x = sr[k];
            if (rh) x = (x + rh[k / Wk]) + rw[k % Wk];
            v[j] = x;
        }
        mx = fmaxf(mx, v[j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = lane + 64 * j < T ? expf(v[j] - mx) : 0.f;
        sum += v[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    TOUT* pr = P + row * ldp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        if (k < ldp) st_from_f32<TOUT>(pr + k, k < T ? v[j] / sum : 0.f);
    }
}

// out[i, :] = [sin(2 pi ((2c - 1) . G)), cos(...)] for coordinate pairs c in [0,1]^2.  With labels:
// label -1 replaces the code by table row 0, labels 0 / 1 add table rows 1 / 2 (prompt_encoder.py:86-90),
// labels 2 / 3 (the two corners of a box, :93-101) add table rows 3 / 4.
__global__ __launch_bounds__(256) void fourier_pe_kernel(const float* __restrict__ coords, const float* __restrict__ G, int F,
                                                         const int32_t* __restrict__ labels, const float* __restrict__ table,
                                                         float* __restrict__ out, int64_t ldo, int64_t n) {
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    const float x = 2.f * coords[2 * i] - 1.f, y = 2.f * coords[2 * i + 1] - 1.f;
    const int lab = labels ? labels[i] : -2;          // no labels: the bare position code
    for (int f = threadIdx.x; f < F; f += 256) {
        float v = x * G[f] + y * G[F + f];          // row vector times the [2, F] matrix
        v = 2.f * 3.14159265358979323846f * v;
        float sv = sinf(v), cv = cosf(v);
        if (lab == -1) { sv = 0.f + table[f]; cv = 0.f + table[F + f]; }
        else if (lab >= 0 && lab <= 3) { sv += table[(int64_t)(lab + 1) * 2 * F + f]; cv += table[(int64_t)(lab + 1) * 2 * F + F + f]; }
        out[i * ldo + f] = sv;
        out[i * ldo + F + f] = cv;
    }
}

// Dense embedding of a mask prompt (prompt_encoder.py:51-58, 102-105): conv 2x2 s2 (1 -> m4) + LayerNorm2d + GELU,
// conv 2x2 s2 (m4 -> m) + LayerNorm2d + GELU, conv 1x1 (m -> C), fused per output pixel: one workgroup forms the
// 4 x 4 input patch's four stage-1 pixels, the stage-2 pixel and the C output channels, token-major out[b, pixel, :].
__global__ __launch_bounds__(256) void mask_embed_kernel(const float* __restrict__ masks, int G, int m4, int m, int C,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ g1, const float* __restrict__ h1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         const float* __restrict__ g2, const float* __restrict__ h2,
                                                         const float* __restrict__ w3, const float* __restrict__ b3,
                                                         float eps, float* __restrict__ out) {
    __shared__ float r1[256], a1[256], r2[256], a2[256];
    const int pix = blockIdx.x, gy = pix / G, gx = pix % G, t = threadIdx.x;
    const int64_t W = 4 * (int64_t)G;
    const float* in = masks + (int64_t)blockIdx.y * W * W;
    if (t < 4 * m4) {
        const int pos = t / m4, ch = t % m4, py = pos >> 1, px = pos & 1;
        const int64_t y0 = 2 * (2 * gy + py), x0 = 2 * (2 * gx + px);
        float acc = b1[ch];
        for (int ky = 0; ky < 2; ++ky)
            for (int kx = 0; kx < 2; ++kx) acc = fmaf(in[(y0 + ky) * W + x0 + kx], w1[ch * 4 + ky * 2 + kx], acc);
        r1[t] = acc;
    }
    __syncthreads();
    if (t < 4 * m4) {
        const int pos = t / m4, ch = t % m4;
        float u = 0.f, v = 0.f;
        for (int c = 0; c < m4; ++c) u += r1[pos * m4 + c];
        u /= (float)m4;
        for (int c = 0; c < m4; ++c) { const float d = r1[pos * m4 + c] - u; v += d * d; }
        v /= (float)m4;
        const float y = g1[ch] * ((r1[t] - u) / sqrtf(v + eps)) + h1[ch];
        a1[t] = 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
    }
    __syncthreads();
    if (t < m) {
        float acc = b2[t];
        for (int c = 0; c < m4; ++c)
            for (int pos = 0; pos < 4; ++pos) acc = fmaf(a1[pos * m4 + c], w2[(t * m4 + c) * 4 + pos], acc);   // w2[co, c, py, px]
        r2[t] = acc;
    }
    __syncthreads();
    if (t < m) {
        float u = 0.f, v = 0.f;
        for (int c = 0; c < m; ++c) u += r2[c];
        u /= (float)m;
        for (int c = 0; c < m; ++c) { const float d = r2[c] - u; v += d * d; }
        v /= (float)m;
        const float y = g2[t] * ((r2[t] - u) / sqrtf(v + eps)) + h2[t];
        a2[t] = 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
    }
    __syncthreads();
    float* o = out + ((int64_t)blockIdx.y * G * G + pix) * C;
    for (int co = t; co < C; co += 256) {
        float acc = b3[co];
        for (int c = 0; c < m; ++c) acc = fmaf(a2[c], w3[(int64_t)co * m + c], acc);
        o[co] = acc;
    }
}

// out[r, :] = a[r, :] + b[r % b_rows, :]  (b may be null), written in the output type
template <typename TOUT>
__global__ __launch_bounds__(256) void add_cast_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t b_rows,
                                                       TOUT* __restrict__ out, int64_t rows, int D) {
    const int64_t total = rows * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float v = a[i];
        if (b) v += b[((i / D) % b_rows) * D + i % D];
        st_from_f32<TOUT>(out + i, v);
    }
}

// One pass over the rows of the image-side stream of the mask decoder (D = 64 * NCH * 4 at most, a wave per row):
//   y = LayerNorm(x) (w, b given; written back as f32 when `y` is set) or y = x,
//   c0 = y + add[r % add_rows]  and  c1 = y  in the compute type (either may be null),
// i.e. the LayerNorm that ends a layer together with the two casts the next layer's projections read.
template <typename TOUT, int NCH>
__global__ __launch_bounds__(256) void stream_copies_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                            float eps, float* __restrict__ y, const float* __restrict__ add, int64_t add_rows,
                                                            TOUT* __restrict__ c0, TOUT* __restrict__ c1, int64_t rows, int D) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, nc = D >> 2;
    const f32x4* xr = (const f32x4*)(x + row * D);
    f32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = i * 64 + lane;
        if (c < nc) { v[i] = xr[c]; s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
    }
    if (w) {
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * 64 + lane;
            if (c < nc) {
                v[i] -= mean;
                q += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * 64 + lane;
            if (c < nc) v[i] = v[i] * rstd * ((const f32x4*)w)[c] + ((const f32x4*)b)[c];
        }
    }
    const f32x4* ar = add ? (const f32x4*)(add + (row % add_rows) * D) : nullptr;
    auto put = [&](TOUT* dst, const f32x4& o, int c) {
        if constexpr (sizeof(TOUT) == 4) ((f32x4*)dst)[c] = o;
        else { bf16x4 ob; ob[0] = (__bf16)o[0]; ob[1] = (__bf16)o[1]; ob[2] = (__bf16)o[2]; ob[3] = (__bf16)o[3]; ((bf16x4*)dst)[c] = ob; }
    };
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = i * 64 + lane;
        if (c < nc) {
            if (y) ((f32x4*)(y + row * D))[c] = v[i];
            if (c1) put(c1 + row * D, v[i], c);
            if (c0) put(c0 + row * D, ar ? v[i] + ar[c] : v[i], c);
        }
    }
}

// LayerNorm over rows of D (a multiple of 4, at most 256) followed by an activation (0 none, 1 erf GELU), f32 in,
// compute type out: LayerNorm2d + GELU between the two transposed convolutions (mask_decoder.py:53-59).  Sixteen
// lanes a row (D = 64: one float4 a lane), so a wave covers four rows.
template <typename TOUT>
__global__ __launch_bounds__(256) void layernorm_act_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                                 float eps, TOUT* __restrict__ out, int64_t ldo, int64_t rows, int D, int act) {
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int l16 = threadIdx.x & 15, nc = D >> 2;
    const bool live = row < rows;
    const f32x4* xr = (const f32x4*)(x + (live ? row : 0) * D);
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 16 + l16;
        if (c < nc) { v[i] = xr[c]; s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 16 + l16;
        if (c < nc) {
            v[i] -= mean;
            q += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)D + eps);
    if (!live) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 16 + l16;
        if (c < nc) {
            f32x4 o = v[i] * rstd * ((const f32x4*)w)[c] + ((const f32x4*)b)[c];
            if (act == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = 0.5f * o[e] * (1.0f + erff(o[e] * 0.70710678118654752440f));
            }
            TOUT* dst = out + row * ldo;
            if constexpr (sizeof(TOUT) == 4) ((f32x4*)dst)[c] = o;
            else { bf16x4 ob; ob[0] = (__bf16)o[0]; ob[1] = (__bf16)o[1]; ob[2] = (__bf16)o[2]; ob[3] = (__bf16)o[3]; ((bf16x4*)dst)[c] = ob; }
        }
    }
}

// masks[b, m, p] = sum_c hyper[b, m, c] * up[b*P + p, c]  for NM = 4 masks and C <= 64 channels (a multiple of 4): the
// product of the hypernetwork outputs with the upscaled embedding (mask_decoder.py:141-143).  A thread owns one
// pixel: it reads its row of `up` once (16-byte loads) and the prompt's four vectors from LDS; a workgroup's pixels are
// consecutive, so the four result rows are written coalesced.
__global__ __launch_bounds__(256) void hyper_product_kernel(const float* __restrict__ hyper, int64_t ldh, const float* __restrict__ up, int64_t ldu,
                                                            float* __restrict__ masks, int64_t P, int C) {
    __shared__ float hs[4 * 64];
    const int b = blockIdx.y;
    if (threadIdx.x < 4 * C) hs[(threadIdx.x / C) * 64 + threadIdx.x % C] = hyper[((int64_t)b * 4 + threadIdx.x / C) * ldh + threadIdx.x % C];
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const f32x4* ur = (const f32x4*)(up + ((int64_t)b * P + p) * ldu);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int c = 0; c < C; c += 4) {
        const f32x4 u = ur[c >> 2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a0 = fmaf(hs[c + e], u[e], a0);
            a1 = fmaf(hs[64 + c + e], u[e], a1);
            a2 = fmaf(hs[128 + c + e], u[e], a2);
            a3 = fmaf(hs[192 + c + e], u[e], a3);
        }
    }
    float* mo = masks + (int64_t)b * 4 * P + p;
    mo[0] = a0; mo[P] = a1; mo[2 * P] = a2; mo[3 * P] = a3;
}

// elementwise activation (1 = erf GELU, 3 = ReLU) with a type change
template <typename TOUT>
__global__ __launch_bounds__(256) void act_cast_kernel(const float* __restrict__ in, TOUT* __restrict__ out, int64_t n, int act) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = in[i];
        if (act == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        else if (act == 3) v = fmaxf(v, 0.f);
        st_from_f32<TOUT>(out + i, v);
    }
}

// per mask: #(logit > thr + off), #(logit > thr - off), and the bounding box of (logit > thr)
__global__ __launch_bounds__(256) void mask_stats_kernel(const float* __restrict__ masks, int H, int W, float thr, float off,
                                                         int32_t* __restrict__ out) {
    __shared__ int sh[6][4];
    const float* m = masks + (int64_t)blockIdx.x * H * W;
    int hi = 0, lo = 0, x0 = W, y0 = H, x1 = -1, y1 = -1;
    for (int i = threadIdx.x; i < H * W; i += 256) {
        const float v = m[i];
        hi += v > thr + off;
        lo += v > thr - off;
        if (v > thr) {
            const int y = i / W, x = i % W;
            x0 = min(x0, x); y0 = min(y0, y); x1 = max(x1, x); y1 = max(y1, y);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        hi += __shfl_xor(hi, o, 64); lo += __shfl_xor(lo, o, 64);
        x0 = min(x0, __shfl_xor(x0, o, 64)); y0 = min(y0, __shfl_xor(y0, o, 64));
        x1 = max(x1, __shfl_xor(x1, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][wave] = hi; sh[1][wave] = lo; sh[2][wave] = x0; sh[3][wave] = y0; sh[4][wave] = x1; sh[5][wave] = y1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t* o = out + (int64_t)blockIdx.x * 6;
        o[0] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        o[1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        const int bx0 = min(min(sh[2][0], sh[2][1]), min(sh[2][2], sh[2][3])), by0 = min(min(sh[3][0], sh[3][1]), min(sh[3][2], sh[3][3]));
        const int bx1 = max(max(sh[4][0], sh[4][1]), max(sh[4][2], sh[4][3])), by1 = max(max(sh[5][0], sh[5][1]), max(sh[5][2], sh[5][3]));
        const bool empty = bx1 < bx0 || by1 < by0;      // an empty mask has the box [0, 0, 0, 0]
        o[2] = empty ? 0 : bx0; o[3] = empty ? 0 : by0; o[4] = empty ? 0 : bx1; o[5] = empty ? 0 : by1;
    }
}

// bits[k, i] = masks[k, i] > thr  (as f32 0/1, the proposal format of the ranking path)
__global__ __launch_bounds__(256) void threshold_masks_kernel(const float* __restrict__ in, float thr, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = in[i] > thr ? 1.f : 0.f;
}

// Sam.postprocess_masks (two bilinear passes and a crop, composed into one banded table an axis) evaluated on the fly
// from the low-resolution logits, a band of output rows at a time: the source rows the band needs are copied to LDS,
// resampled along x into a second LDS array, then along y out of that - the two sums of resample_axis_kernel in the same order, so the values are the
// ones the two-pass form writes - and consumed in place: by the mask statistics (STATS: a workgroup walks all bands of
// one plane) or by the threshold (a workgroup owns one band of one selected plane and writes 0 / 1).  The image-size
// logits never exist in memory.
struct AxisTable { const int32_t* lo; const int32_t* size; const float* w; int kmax; };

// KT: taps unrolled per sum (tables with kmax <= KT; weights past an output's size are zero in the table and the index is
// clamped, so the extra terms add an exact zero); KT = 0: loops of the table's own length.
template <int KT>
__device__ __forceinline__ float banded_sum(const float* __restrict__ v, int stride, int m0, int ms, int last, const float* __restrict__ w, int kmax) {
    float acc = 0.f;
    if constexpr (KT > 0) {
        float x[KT], wk[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) {                   // all reads requested before the first is used
            x[k] = v[max(min(m0 + k, last), 0) * stride];
            wk[k] = k < kmax ? w[k] : 0.f;                // (a row of the table holds kmax weights)
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) acc = k == 0 ? x[0] * wk[0] : __builtin_fmaf(x[k], wk[k], acc);
    } else {
        for (int k = 0; k < ms; ++k) acc = k == 0 ? v[m0 * stride] * w[0] : __builtin_fmaf(v[(m0 + k) * stride], w[k], acc);
    }
    return acc;
}

template <bool STATS, int KT>
__global__ __launch_bounds__(256) void resample_bands_kernel(const float* __restrict__ low, int64_t stride_b, int64_t stride_c, int nm,
                                                             const int32_t* __restrict__ sel, int L, int H, int W, AxisTable tx, AxisTable ty,
                                                             int band, int max_rows, float thr, float off, int32_t* __restrict__ stats,
                                                             float* __restrict__ out) {
    extern __shared__ float bands_lds[];
    float* srow = bands_lds;                             // [max_rows][L]: the source rows of the band
    float* tmp = bands_lds + (size_t)max_rows * L;       // [max_rows][W]: those rows resampled along x
    __shared__ int sr[2];
    __shared__ int sh[6][4];
    const int slot = STATS ? blockIdx.x : blockIdx.y;
    const int plane = (!STATS && sel) ? sel[slot] : slot;
    const float* src = low + (int64_t)(plane / nm) * stride_b + (int64_t)(plane % nm) * stride_c;
    int hi = 0, lo = 0, x0 = W, y0 = H, x1 = -1, y1 = -1;
    const int nb = (H + band - 1) / band;
    for (int b = STATS ? 0 : blockIdx.x; b < (STATS ? nb : blockIdx.x + 1); ++b) {
        const int ya = b * band, yb = min(H, ya + band);
        __syncthreads();                                  // (the previous band is done with the LDS rows and sr)
        if (threadIdx.x < 64) {                           // source rows [r0, r1) of this band
            int r0 = L, r1 = 0;
            for (int y = ya + (int)threadIdx.x; y < yb; y += 64) {
                r0 = min(r0, ty.lo[y]);
                r1 = max(r1, ty.lo[y] + ty.size[y]);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                r0 = min(r0, __shfl_xor(r0, o, 64));
                r1 = max(r1, __shfl_xor(r1, o, 64));
            }
            if (threadIdx.x == 0) { sr[0] = r0; sr[1] = min(min(r1, L), r0 + max_rows); }
        }
        __syncthreads();
        const int r0 = min(sr[0], L - 1), nr = max(sr[1] - r0, 0);        // (a band whose outputs read nothing: nr = 0, the sums are empty)
        // the rows are consecutive in memory: one coalesced copy
        const float* sp = src + (int64_t)r0 * L;
        if ((L & 3) == 0 && ((uintptr_t)sp & 15) == 0) {
            for (int i = threadIdx.x; i < nr * L / 4; i += 256) ((f32x4*)srow)[i] = ((const f32x4*)sp)[i];
        } else {
            for (int i = threadIdx.x; i < nr * L; i += 256) srow[i] = sp[i];
        }
        __syncthreads();
        for (int x = threadIdx.x; x < W; x += 256) {
            const int m0 = tx.lo[x], ms = tx.size[x];
            const float* w = tx.w + (int64_t)x * tx.kmax;
            for (int r = 0; r < nr; ++r) tmp[r * W + x] = banded_sum<KT>(srow + r * L, 1, m0, ms, L - 1, w, tx.kmax);
        }
        __syncthreads();
        for (int y = ya; y < yb; ++y) {
            const int m0 = ty.lo[y] - r0, ms = min(ty.size[y], nr - m0);
            const float* w = ty.w + (int64_t)y * ty.kmax;
            for (int x = threadIdx.x; x < W; x += 256) {
                const float acc = banded_sum<KT>(tmp + x, W, m0, ms, nr - 1, w, ty.kmax);
                if (STATS) {
                    hi += acc > thr + off;
                    lo += acc > thr - off;
                    if (acc > thr) { x0 = min(x0, x); y0 = min(y0, y); x1 = max(x1, x); y1 = max(y1, y); }
                    if (out) out[((int64_t)slot * H + y) * W + x] = acc > thr ? 1.f : 0.f;       // (callers that keep every mask)
                } else {
                    out[((int64_t)slot * H + y) * W + x] = acc > thr ? 1.f : 0.f;
                }
            }
        }
    }
    if (!STATS) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        hi += __shfl_xor(hi, o, 64); lo += __shfl_xor(lo, o, 64);
        x0 = min(x0, __shfl_xor(x0, o, 64)); y0 = min(y0, __shfl_xor(y0, o, 64));
        x1 = max(x1, __shfl_xor(x1, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][wave] = hi; sh[1][wave] = lo; sh[2][wave] = x0; sh[3][wave] = y0; sh[4][wave] = x1; sh[5][wave] = y1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t* o = stats + (int64_t)plane * 6;
        o[0] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        o[1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        const int bx0 = min(min(sh[2][0], sh[2][1]), min(sh[2][2], sh[2][3])), by0 = min(min(sh[3][0], sh[3][1]), min(sh[3][2], sh[3][3]));
        const int bx1 = max(max(sh[4][0], sh[4][1]), max(sh[4][2], sh[4][3])), by1 = max(max(sh[5][0], sh[5][1]), max(sh[5][2], sh[5][3]));
        const bool empty = bx1 < bx0 || by1 < by0;      // an empty mask has the box [0, 0, 0, 0]
        o[2] = empty ? 0 : bx0; o[3] = empty ? 0 : by0; o[4] = empty ? 0 : bx1; o[5] = empty ? 0 : by1;
    }
}

// The statistics again, for tables of at most KT taps and images at most blockDim.x CPT wide: a thread owns CPT columns of the
// image and walks down the plane once.  Per source row it forms the row's x-resampled value for its columns (the x table
// of a column sits in registers) and keeps the last KT of them in registers; an output row y whose first source row is
// ylo[y] is due when the window holds rows ylo[y] .. ylo[y] + KT - 1 (rows past the last repeat it, as the clamped
// index of banded_sum does; weights past a row's size are zero in the table).  Against the band form no x-resampled
// value goes through LDS, no source row is brought twice, and the tables are read once a plane, not once a band: LDS
// holds the y table and a chunk of R source rows (each followed by KT copies of its last value, so column reads need no
// clamp), the next chunk on its way in registers while the current one is worked on.  Same sums in the same order as
// banded_sum: the same values.  A y table whose first rows do not grow with y is still served (the window starts over
// at such a row), at the price of rows brought again.
template <int KT, int CPT>
__global__ __launch_bounds__(256) void resample_stats_columns_kernel(const float* __restrict__ low, int64_t stride_b, int64_t stride_c, int nm, int L,
                                                                     int H, int W, AxisTable tx, AxisTable ty, int R, float thr, float off,
                                                                     int32_t* __restrict__ stats, float* __restrict__ out) {
    extern __shared__ float cols_lds[];
    const int Ls = L + KT, L4 = L >> 2;
    float* yw = cols_lds;                                // [H][KT]
    float* srow = yw + (size_t)H * KT;                   // [R][Ls]
    int* ylo = (int*)(srow + (size_t)R * Ls);            // [H]
    __shared__ int sh[6][4];
    const int plane = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;        // NT: whole waves, at most 256
    const float* src = low + (int64_t)(plane / nm) * stride_b + (int64_t)(plane % nm) * stride_c;
    for (int i = tid; i < H; i += NT) {
        ylo[i] = ty.lo[i];
#pragma unroll
        for (int k = 0; k < KT; ++k) yw[i * KT + k] = k < ty.kmax ? ty.w[(int64_t)i * ty.kmax + k] : 0.f;
    }
    int m0x[CPT];
    float wx[CPT][KT], win[CPT][KT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int xx = min(tid + NT * j, W - 1);
        m0x[j] = min(max(tx.lo[xx], 0), L - 1);
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            wx[j][k] = k < tx.kmax ? tx.w[(int64_t)xx * tx.kmax + k] : 0.f;
            win[j][k] = 0.f;
        }
    }
    // source rows [cs, ce) are in LDS; rows [ps, ps + R) are on their way in pf (a thread's 16-byte pieces tid, tid + NT, ...)
    constexpr int PF = 4;                                // R L <= NT PF 4 values
    int cs = 0, ce = 0, ps = 0;
    f32x4 pf[PF];
    auto fetch = [&](int start) {
        ps = start;
        const int n4 = min(R, L - start) * L4;
        const f32x4* sp = (const f32x4*)(src + (int64_t)start * L);
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int i = tid + NT * p;
            pf[p] = i < n4 ? sp[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto commit = [&]() {
        __syncthreads();                                  // (everybody is done with the rows in LDS)
        const int n4 = min(R, L - ps) * L4;
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int i = tid + NT * p;
            if (i < n4) {
                const int row = i / L4, c4 = i - row * L4;
                ((f32x4*)(srow + row * Ls))[c4] = pf[p];
                if (c4 == L4 - 1) {
#pragma unroll
                    for (int k = 0; k < KT; ++k) srow[row * Ls + L + k] = pf[p][3];
                }
            }
        }
        cs = ps;
        ce = min(ps + R, L);
        __syncthreads();
        if (ce < L) fetch(ce);
    };
    fetch(0);
    __syncthreads();                                      // the y table
    const float t_hi = thr + off, t_lo = thr - off;
    int hi = 0, lo = 0, y0 = H, y1 = -1;                  // (wave-uniform)
    unsigned long long hit[CPT] = {}, okm[CPT];           // okm: the lanes whose column j is inside the image
#pragma unroll
    for (int j = 0; j < CPT; ++j) okm[j] = __builtin_amdgcn_ballot_w64(tid + NT * j < W);
    int r = -1;                                           // the window holds rows r - KT + 1 .. r
    for (int y = 0; y < H; ++y) {
        const int l0 = min(max(ylo[y], 0), L - 1), top = l0 + KT - 1;
        if (top < r || l0 > r + 1) r = l0 - 1;           // rows skipped, or a table that goes back: the window starts over
        while (r < top) {
            ++r;
            const int rr = min(r, L - 1);
            if (rr < cs || rr >= ce) {                   // (uniform: r depends on y alone)
                if (ps != rr) fetch(rr);
                commit();
            }
            const float* sp = srow + (rr - cs) * Ls;
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                float v[KT];
#pragma unroll
                for (int k = 0; k < KT; ++k) v[k] = sp[m0x[j] + k];
                float acc = v[0] * wx[j][0];
#pragma unroll
                for (int k = 1; k < KT; ++k) acc = __builtin_fmaf(v[k], wx[j][k], acc);
#pragma unroll
                for (int k = 0; k + 1 < KT; ++k) win[j][k] = win[j][k + 1];
                win[j][KT - 1] = acc;
            }
        }
        float wy[KT];
#pragma unroll
        for (int k4 = 0; k4 < KT / 4; ++k4) {
            const f32x4 w4 = ((const f32x4*)yw)[y * (KT / 4) + k4];
#pragma unroll
            for (int k = 0; k < 4; ++k) wy[4 * k4 + k] = w4[k];
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int x = tid + NT * j;
            const bool ok = x < W;
            float acc = win[j][0] * wy[0];
#pragma unroll
            for (int k = 1; k < KT; ++k) acc = __builtin_fmaf(win[j][k], wy[k], acc);
            // counts and the box on the wave's lane masks: scalar work beside the sums
            const unsigned long long mh = __builtin_amdgcn_ballot_w64(acc > t_hi) & okm[j], ml = __builtin_amdgcn_ballot_w64(acc > t_lo) & okm[j],
                                     mi = __builtin_amdgcn_ballot_w64(acc > thr) & okm[j];
            hi += __popcll(mh);
            lo += __popcll(ml);
            hit[j] |= mi;
            if (mi) { y0 = min(y0, y); y1 = y; }
            if (out && ok) out[((int64_t)plane * H + y) * W + x] = acc > thr ? 1.f : 0.f;
        }
    }
    // hi, lo, y0 and y1 are the wave's; the columns a lane's outputs were set in give x0 and x1
    int x0 = W, x1 = -1;
#pragma unroll
    for (int j = 0; j < CPT; ++j)
        if ((hit[j] >> (tid & 63)) & 1) { x0 = min(x0, tid + NT * j); x1 = max(x1, tid + NT * j); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        x0 = min(x0, __shfl_xor(x0, o, 64));
        x1 = max(x1, __shfl_xor(x1, o, 64));
    }
    const int wave = tid >> 6;
    if ((tid & 63) == 0) { sh[0][wave] = hi; sh[1][wave] = lo; sh[2][wave] = x0; sh[3][wave] = y0; sh[4][wave] = x1; sh[5][wave] = y1; }
    __syncthreads();
    if (tid == 0) {
        int32_t* o = stats + (int64_t)plane * 6;
        int a = 0, b = 0, bx0 = W, by0 = H, bx1 = -1, by1 = -1;
        for (int w = 0; w < NT / 64; ++w) {
            a += sh[0][w]; b += sh[1][w];
            bx0 = min(bx0, sh[2][w]); by0 = min(by0, sh[3][w]); bx1 = max(bx1, sh[4][w]); by1 = max(by1, sh[5][w]);
        }
        o[0] = a;
        o[1] = b;
        const bool empty = bx1 < bx0 || by1 < by0;      // an empty mask has the box [0, 0, 0, 0]
        o[2] = empty ? 0 : bx0; o[3] = empty ? 0 : by0; o[4] = empty ? 0 : bx1; o[5] = empty ? 0 : by1;
    }
}

int check_bands(const char* what, const float* low, int nm, int L, int H, int W, const MarsAxisTable* tx, const MarsAxisTable* ty, int band,
                int max_rows) {
    MARS_CHECK(low && tx && ty && tx->lo && tx->size && tx->w && ty->lo && ty->size && ty->w, "%s: null pointer", what);
    MARS_CHECK(nm > 0 && L > 0 && H > 0 && W > 0 && tx->kmax > 0 && ty->kmax > 0, "%s: bad shape", what);
    MARS_CHECK(band > 0 && max_rows > 0 && max_rows <= L, "%s: band %d / rows %d", what, band, max_rows);
    MARS_CHECK((size_t)max_rows * (W + L) * sizeof(float) <= 64 * 1024, "%s: %d source rows of %d + %d samples do not fit in LDS; use a smaller band", what,
               max_rows, L, W);
    return 0;
}

}  // namespace

extern "C" int marship_resample_mask_stats(const float* low, int64_t stride_b, int64_t stride_c, int nm, int n, int L, int H, int W,
                                           const MarsAxisTable* tx, const MarsAxisTable* ty, int band, int max_rows, float thr, float off,
                                           int32_t* stats, float* bits, void* stream) {
    if (check_bands("resample_mask_stats", low, nm, L, H, W, tx, ty, band, max_rows)) return -1;
    MARS_CHECK(stats && n > 0, "resample_mask_stats: bad arguments");
    const AxisTable ax = {tx->lo, tx->size, tx->w, tx->kmax}, ay = {ty->lo, ty->size, ty->w, ty->kmax};
    const size_t shm = (size_t)max_rows * (W + L) * sizeof(float);
    const int kt = ax.kmax <= 4 && ay.kmax <= 4 ? 4 : ax.kmax <= 8 && ay.kmax <= 8 ? 8 : 0;
    // short tables, images up to 1024 wide, planes whose rows are whole 16-byte pieces: a thread walks its columns down
    // the plane (MARS_RESAMPLE_COLUMNS=0, read at every call: the band form, for comparisons)
    const char* env = getenv("MARS_RESAMPLE_COLUMNS");
    if (kt != 0 && W <= 1024 && L % 4 == 0 && L <= 4096 && ((uintptr_t)low % 16) == 0 && stride_b % 4 == 0 && stride_c % 4 == 0 && !(env && env[0] == '0')) {
        // columns a thread at 256 threads a workgroup, then only as many whole waves as those columns need.  (Six such
        // workgroups fit a CU; wider ones with fewer columns a thread left fewer planes in flight and were slower: 640
        // columns as 2 x 320 threads 1.2 ms for 3072 planes, 3 x 256 0.9, 4 x 192 1.0.)
        const int cpt = (W + 255) / 256, nt = ((W + cpt - 1) / cpt + 63) / 64 * 64;
        const int R = 16 * nt / L < 16 ? (16 * nt / L < 1 ? 1 : 16 * nt / L) : 16;       // R L / 4 <= 4 nt pieces
        const size_t lds = ((size_t)R * (L + kt) + (size_t)H * (kt + 1)) * sizeof(float);
        if ((int64_t)R * L <= 16 * nt && lds <= 64 * 1024) {
#define MARS_COLS(KT, CPT)                                                                                                                    \
    hipLaunchKernelGGL((resample_stats_columns_kernel<KT, CPT>), dim3((unsigned)n), dim3(nt), lds, (hipStream_t)stream, low, stride_b, stride_c, \
                       nm, L, H, W, ax, ay, R, thr, off, stats, bits)
            if (kt == 4) { if (cpt == 1) MARS_COLS(4, 1); else if (cpt == 2) MARS_COLS(4, 2); else if (cpt == 3) MARS_COLS(4, 3); else MARS_COLS(4, 4); }
            else { if (cpt == 1) MARS_COLS(8, 1); else if (cpt == 2) MARS_COLS(8, 2); else if (cpt == 3) MARS_COLS(8, 3); else MARS_COLS(8, 4); }
#undef MARS_COLS
            MARS_LAUNCH_CHECK();
            return 0;
        }
    }
#define MARS_BANDS(KT)                                                                                                                       \
    hipLaunchKernelGGL((resample_bands_kernel<true, KT>), dim3((unsigned)n), dim3(256), shm, (hipStream_t)stream, low, stride_b, stride_c, nm,    \
                       (const int32_t*)nullptr, L, H, W, ax, ay, band, max_rows, thr, off, stats, bits)
    if (kt == 4) MARS_BANDS(4); else if (kt == 8) MARS_BANDS(8); else MARS_BANDS(0);
#undef MARS_BANDS
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_resample_threshold(const float* low, int64_t stride_b, int64_t stride_c, int nm, const int32_t* sel, int nsel, int L,
                                          int H, int W, const MarsAxisTable* tx, const MarsAxisTable* ty, int band, int max_rows, float thr,
                                          float* out, void* stream) {
    if (check_bands("resample_threshold", low, nm, L, H, W, tx, ty, band, max_rows)) return -1;
    MARS_CHECK(out && nsel > 0 && nsel <= 65535, "resample_threshold: %d planes (1 .. 65535 a launch)", nsel);
    const AxisTable ax = {tx->lo, tx->size, tx->w, tx->kmax}, ay = {ty->lo, ty->size, ty->w, ty->kmax};
    const size_t shm = (size_t)max_rows * (W + L) * sizeof(float);
    const dim3 grid((unsigned)((H + band - 1) / band), (unsigned)nsel);
    const int kt = ax.kmax <= 4 && ay.kmax <= 4 ? 4 : ax.kmax <= 8 && ay.kmax <= 8 ? 8 : 0;
#define MARS_BANDS(KT)                                                                                                                       \
    hipLaunchKernelGGL((resample_bands_kernel<false, KT>), grid, dim3(256), shm, (hipStream_t)stream, low, stride_b, stride_c, nm, sel, L, H, W, ax, ay, \
                       band, max_rows, thr, 0.f, (int32_t*)nullptr, out)
    if (kt == 4) MARS_BANDS(4); else if (kt == 8) MARS_BANDS(8); else MARS_BANDS(0);
#undef MARS_BANDS
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_gather_rows(const void* src, int64_t ld_src, int src_col, const int32_t* idx, void* dst, int64_t ld_dst,
                                   int dst_col, int64_t rows, int ncols, int dtype, void* stream) {
    MARS_CHECK(src && idx && dst && rows > 0 && ncols > 0 && src_col >= 0 && dst_col >= 0, "gather_rows: bad arguments");
    MARS_CHECK(ld_src >= src_col + ncols && ld_dst >= dst_col + ncols, "gather_rows: columns exceed the row");
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == 1) hipLaunchKernelGGL(gather_rows_kernel<__bf16>, grid, block, 0, (hipStream_t)stream, (const __bf16*)src, ld_src, src_col, idx, (__bf16*)dst, ld_dst, dst_col, rows, ncols);
    else hipLaunchKernelGGL(gather_rows_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)src, ld_src, src_col, idx, (float*)dst, ld_dst, dst_col, rows, ncols);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_relpos_bias_rows(const void* qkv, int64_t ld, int B, int T, int H, int hd, int hdp, int gw, int Hk, int Wk,
                                        const float* Rh, const float* Rw, float* rel_h, float* rel_w, const int32_t* in_rows, const void* pad_row,
                                        int dtype, void* stream) {
    MARS_CHECK(qkv && Rh && Rw && rel_h && rel_w, "relpos_bias: null pointer");
    MARS_CHECK(B > 0 && T > 0 && H > 0 && hd > 0 && hdp >= hd && gw > 0 && T % gw == 0 && Hk > 0 && Wk > 0 && ld >= (int64_t)H * hdp,
               "relpos_bias: bad shape");
    const int gh = T / gw;
    const int nmax = gh > gw ? gh : gw, kmax = Hk > Wk ? Hk : Wk;
    // small grids of bf16 queries: a wave per table slice on the matrix pipe (MARS_RELPOS_MMA=0, read once: the vector form)
    static const bool mma_on = [] { const char* e = getenv("MARS_RELPOS_MMA"); return !(e && e[0] == '0'); }();
    const bool mma = mma_on || in_rows;
    MARS_CHECK(!in_rows || (pad_row && ((uintptr_t)pad_row % 16) == 0), "relpos_bias_rows: a row map needs the 16-byte aligned row of padding values");
    if (mma && dtype == 1 && nmax <= 16 && kmax <= 16 && hd % 8 == 0 && hd <= 128 && hdp % 8 == 0 && ld % 8 == 0 && ((uintptr_t)qkv % 16) == 0 &&
        ((uintptr_t)Rh % 16) == 0 && ((uintptr_t)Rw % 16) == 0) {
        const int64_t nslices = (int64_t)B * H * (gh + gw);
        hipLaunchKernelGGL(relpos_bias_mma_kernel, dim3((unsigned)((nslices + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)qkv, ld, T, H,
                           hd, hdp, gw, Hk, Wk, Rh, Rw, rel_h, rel_w, nslices, in_rows, (const __bf16*)pad_row);
        MARS_LAUNCH_CHECK();
        return 0;
    }
    MARS_CHECK(!in_rows, "relpos_bias_rows: a row map needs bf16 queries, grids and tables up to 16 a side, head dims a multiple of 8 up to 128");
    const size_t shm = (size_t)(nmax + kmax) * (hd + 1) * sizeof(float);
    MARS_CHECK(shm <= 64 * 1024 && (int64_t)B * H <= 65535, "relpos_bias: grid side %d, table side %d or head dim %d too large", nmax, kmax, hd);
    dim3 grid((unsigned)nmax, (unsigned)(B * H), 2), block(256);
    if (dtype == 1) hipLaunchKernelGGL(relpos_bias_kernel<__bf16>, grid, block, shm, (hipStream_t)stream, (const __bf16*)qkv, ld, T, H, hd, hdp, gw, Hk, Wk, Rh, Rw, rel_h, rel_w);
    else hipLaunchKernelGGL(relpos_bias_kernel<float>, grid, block, shm, (hipStream_t)stream, (const float*)qkv, ld, T, H, hd, hdp, gw, Hk, Wk, Rh, Rw, rel_h, rel_w);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_relpos_bias(const void* qkv, int64_t ld, int B, int T, int H, int hd, int hdp, int gw, int Hk, int Wk,
                                   const float* Rh, const float* Rw, float* rel_h, float* rel_w, int dtype, void* stream) {
    return marship_relpos_bias_rows(qkv, ld, B, T, H, hd, hdp, gw, Hk, Wk, Rh, Rw, rel_h, rel_w, nullptr, nullptr, dtype, stream);
}

extern "C" int marship_softmax_bias_rows(const float* S, int64_t lds_, void* P, int64_t ldp, int64_t nrows, int T, const float* rel_h,
                                         const float* rel_w, int Hk, int Wk, int out_dtype, void* stream) {
    MARS_CHECK(S && P && nrows > 0 && T > 0 && lds_ >= T && ldp >= T, "softmax_bias_rows: bad shape");
    MARS_CHECK(T <= 16384, "softmax_bias_rows: rows longer than 16384 are not supported");
    MARS_CHECK((rel_h == nullptr) == (rel_w == nullptr), "softmax_bias_rows: both bias terms or none");
    MARS_CHECK(!rel_h || (Hk > 0 && Wk > 0 && (int64_t)Hk * Wk == T), "softmax_bias_rows: Hk*Wk must equal T");
    MARS_CHECK(nrows < (1ll << 31), "softmax_bias_rows: too many rows");
    if (T <= 256 && ldp <= 256) {
        dim3 grid((unsigned)((nrows + 3) / 4)), block(256);
        if (out_dtype == 1) hipLaunchKernelGGL(softmax_bias_rows_small_kernel<__bf16>, grid, block, 0, (hipStream_t)stream, S, lds_, (__bf16*)P, ldp, T, rel_h, rel_w, Hk, Wk, nrows);
        else hipLaunchKernelGGL(softmax_bias_rows_small_kernel<float>, grid, block, 0, (hipStream_t)stream, S, lds_, (float*)P, ldp, T, rel_h, rel_w, Hk, Wk, nrows);
        MARS_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid((unsigned)nrows), block(256);
    const size_t shm = (size_t)T * sizeof(float);
    if (out_dtype == 1) hipLaunchKernelGGL(softmax_bias_rows_kernel<__bf16>, grid, block, shm, (hipStream_t)stream, S, lds_, (__bf16*)P, ldp, T, rel_h, rel_w, Hk, Wk);
    else hipLaunchKernelGGL(softmax_bias_rows_kernel<float>, grid, block, shm, (hipStream_t)stream, S, lds_, (float*)P, ldp, T, rel_h, rel_w, Hk, Wk);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_fourier_pe(const float* coords, const float* G, int F, const int32_t* labels, const float* table, float* out,
                                  int64_t ldo, int64_t n, void* stream) {
    MARS_CHECK(coords && G && out && F > 0 && n > 0 && ldo >= 2 * F && n < (1ll << 31), "fourier_pe: bad arguments");
    MARS_CHECK(!labels || table, "fourier_pe: labels need the embedding table");
    hipLaunchKernelGGL(fourier_pe_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, coords, G, F, labels, table, out, ldo, n);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_mask_embed(const float* masks, int B, int G, int m4, int m, int C, const float* w1, const float* b1,
                                  const float* g1, const float* h1, const float* w2, const float* b2, const float* g2,
                                  const float* h2, const float* w3, const float* b3, float eps, float* out, void* stream) {
    MARS_CHECK(masks && out && w1 && b1 && g1 && h1 && w2 && b2 && g2 && h2 && w3 && b3, "mask_embed: null pointer");
    MARS_CHECK(B > 0 && B <= 65535 && G > 0 && G <= 1024 && m4 > 0 && 4 * m4 <= 256 && m > 0 && m <= 256 && C > 0,
               "mask_embed: bad shape (B=%d G=%d m4=%d m=%d C=%d)", B, G, m4, m, C);
    hipLaunchKernelGGL(mask_embed_kernel, dim3((unsigned)(G * G), (unsigned)B), dim3(256), 0, (hipStream_t)stream, masks, G, m4, m, C,
                       w1, b1, g1, h1, w2, b2, g2, h2, w3, b3, eps, out);
    MARS_LAUNCH_CHECK();
    return 0;
}

static unsigned blocks_for(int64_t n) { const int64_t b = (n + 255) / 256; return (unsigned)(b < 65536 ? b : 65536); }

extern "C" int marship_add_cast(const float* a, const float* b, int64_t b_rows, void* out, int64_t rows, int D, int out_dtype, void* stream) {
    MARS_CHECK(a && out && rows > 0 && D > 0 && (!b || b_rows > 0), "add_cast: bad arguments");
    if (out_dtype == 1) hipLaunchKernelGGL(add_cast_kernel<__bf16>, dim3(blocks_for(rows * D)), dim3(256), 0, (hipStream_t)stream, a, b, b_rows, (__bf16*)out, rows, D);
    else hipLaunchKernelGGL(add_cast_kernel<float>, dim3(blocks_for(rows * D)), dim3(256), 0, (hipStream_t)stream, a, b, b_rows, (float*)out, rows, D);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_stream_copies(const float* x, const float* w, const float* b, float eps, float* y, const float* add, int64_t add_rows,
                                     void* c0, void* c1, int64_t rows, int D, int out_dtype, void* stream) {
    MARS_CHECK(x && rows > 0 && D > 0 && D % 4 == 0 && D <= 2048, "stream_copies: bad shape rows=%lld D=%d", (long long)rows, D);
    MARS_CHECK((w == nullptr) == (b == nullptr) && (!y || w), "stream_copies: LayerNorm needs weight and bias; y is its f32 result");
    MARS_CHECK(!add || add_rows > 0, "stream_copies: add_rows");
    MARS_CHECK((((uintptr_t)x | (uintptr_t)w | (uintptr_t)b | (uintptr_t)y | (uintptr_t)add) % 16) == 0 && (((uintptr_t)c0 | (uintptr_t)c1) % 8) == 0,
               "stream_copies: alignment");
    MARS_CHECK(out_dtype == 0 || out_dtype == 1, "stream_copies: out_dtype %d", out_dtype);
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int nch = (D / 4 + 63) / 64;
#define MARS_SC(N)                                                                                                                     \
    if (out_dtype == 1) hipLaunchKernelGGL((stream_copies_kernel<__bf16, N>), grid, block, 0, (hipStream_t)stream, x, w, b, eps, y, add, add_rows, \
                                           (__bf16*)c0, (__bf16*)c1, rows, D);                                                          \
    else hipLaunchKernelGGL((stream_copies_kernel<float, N>), grid, block, 0, (hipStream_t)stream, x, w, b, eps, y, add, add_rows, (float*)c0, (float*)c1, rows, D);
    if (nch <= 1) { MARS_SC(1) } else if (nch <= 2) { MARS_SC(2) } else if (nch <= 4) { MARS_SC(4) } else { MARS_SC(8) }
#undef MARS_SC
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_layernorm_act_rows(const float* x, const float* w, const float* b, float eps, void* out, int64_t ldo, int64_t rows, int D,
                                          int act, int out_dtype, void* stream) {
    MARS_CHECK(x && w && b && out && rows > 0, "layernorm_act_rows: bad arguments");
    MARS_CHECK(D > 0 && D % 4 == 0 && D <= 256 && ldo >= D && (act == 0 || act == 1), "layernorm_act_rows: D=%d act=%d", D, act);
    MARS_CHECK((((uintptr_t)x | (uintptr_t)w | (uintptr_t)b) % 16) == 0 && ((uintptr_t)out % 16) == 0 && (ldo * (out_dtype == 1 ? 2 : 4)) % 8 == 0,
               "layernorm_act_rows: alignment");
    const dim3 grid((unsigned)((rows + 15) / 16)), block(256);
    if (out_dtype == 1) hipLaunchKernelGGL(layernorm_act_rows_kernel<__bf16>, grid, block, 0, (hipStream_t)stream, x, w, b, eps, (__bf16*)out, ldo, rows, D, act);
    else hipLaunchKernelGGL(layernorm_act_rows_kernel<float>, grid, block, 0, (hipStream_t)stream, x, w, b, eps, (float*)out, ldo, rows, D, act);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_hyper_product(const float* hyper, int64_t ldh, const float* up, int64_t ldu, float* masks, int B, int64_t P, int C,
                                     void* stream) {
    MARS_CHECK(hyper && up && masks && B > 0 && B <= 65535 && P > 0, "hyper_product: bad arguments");
    MARS_CHECK(C > 0 && C <= 64 && C % 4 == 0 && ldh >= C && ldu >= C && ldu % 4 == 0 && ((uintptr_t)up % 16) == 0, "hyper_product: C=%d / row layout", C);
    hipLaunchKernelGGL(hyper_product_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream, hyper, ldh, up, ldu, masks, P, C);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_act_cast(const float* in, void* out, int64_t n, int act, int out_dtype, void* stream) {
    MARS_CHECK(in && out && n > 0 && (act == 0 || act == 1 || act == 3), "act_cast: bad arguments");
    if (out_dtype == 1) hipLaunchKernelGGL(act_cast_kernel<__bf16>, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, in, (__bf16*)out, n, act);
    else hipLaunchKernelGGL(act_cast_kernel<float>, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, in, (float*)out, n, act);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_mask_stats(const float* masks, int n, int H, int W, float thr, float off, int32_t* out, void* stream) {
    MARS_CHECK(masks && out && n > 0 && H > 0 && W > 0, "mask_stats: bad arguments");
    hipLaunchKernelGGL(mask_stats_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, masks, H, W, thr, off, out);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_threshold_masks(const float* in, float thr, float* out, int64_t n, void* stream) {
    MARS_CHECK(in && out && n > 0, "threshold_masks: bad arguments");
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(threshold_masks_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, in, thr, out, n);
    MARS_LAUNCH_CHECK();
    return 0;
}
