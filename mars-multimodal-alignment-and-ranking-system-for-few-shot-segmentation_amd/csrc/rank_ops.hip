// HBM-bound alignment / refinement / pooling / ranking kernels of the MARS path.
// Everything here streams its input once with coalesced access; reductions are
// wave shuffles plus one LDS stage, and results that must be order-independent
// (IoU counts) are integer-valued float atomics.
#include "common.h"
#include "../../include/mars_hip.h"
#include <limits.h>
#include <mutex>

namespace {

// ------------------------------------------------------------ VVA prior
// partial[chunk][4][P]: sum_fg, max_fg, sum_bg, max_bg over this chunk's rows
__global__ __launch_bounds__(256) void vva_partial_kernel(const float* __restrict__ S, int64_t ldS,
                                                          const uint8_t* __restrict__ fg, int R, int P, int chunks,
                                                          float* __restrict__ ws) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    const int rpc = (R + chunks - 1) / chunks;
    const int r0 = c * rpc, r1 = min(R, r0 + rpc);
    float sf = 0.f, mf = -INFINITY, sb = 0.f, mb = -INFINITY;
    if (j < P) {
        for (int r = r0; r < r1; ++r) {
            const float v = S[(int64_t)r * ldS + j];
            if (fg[r]) { sf += v; mf = fmaxf(mf, v); } else { sb += v; mb = fmaxf(mb, v); }
        }
        float* w = ws + (int64_t)c * 4 * P;
        w[j] = sf; w[P + j] = mf; w[2 * P + j] = sb; w[3 * P + j] = mb;
    }
}

__global__ __launch_bounds__(1024) void vva_final_kernel(const float* __restrict__ ws, const uint8_t* __restrict__ fg, int R,
                                                         int P, int chunks, float* __restrict__ out) {
    __shared__ float sm[16];
    float cf = 0.f;
    for (int r = threadIdx.x; r < R; r += blockDim.x) cf += fg[r] ? 1.f : 0.f;
    const float nfg = block_reduce_sum(cf, sm);
    const float nbg = (float)R - nfg;
    float lmin = INFINITY, lmax = -INFINITY;
    for (int j = threadIdx.x; j < P; j += blockDim.x) {
        float sf = 0.f, mf = -INFINITY, sb = 0.f, mb = -INFINITY;
        for (int c = 0; c < chunks; ++c) {
            const float* w = ws + (int64_t)c * 4 * P;
            sf += w[j]; mf = fmaxf(mf, w[P + j]); sb += w[2 * P + j]; mb = fmaxf(mb, w[3 * P + j]);
        }
        float v = (sf / nfg) * mf;
        if (nbg > 0.f) v -= (sb / nbg) * mb;
        out[j] = v;
        lmin = fminf(lmin, v); lmax = fmaxf(lmax, v);
    }
    const float gmin = block_reduce_min(lmin, sm);
    const float gmax = block_reduce_max(lmax, sm);
    const float den = 1e-7f + gmax - gmin;
    for (int j = threadIdx.x; j < P; j += blockDim.x) out[j] = (out[j] - gmin) / den;
}

__global__ __launch_bounds__(1024) void minmax_kernel(const float* __restrict__ x, float* __restrict__ y, int n) {
    __shared__ float sm[16];
    float lmin = INFINITY, lmax = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { lmin = fminf(lmin, x[i]); lmax = fmaxf(lmax, x[i]); }
    const float gmin = block_reduce_min(lmin, sm);
    const float gmax = block_reduce_max(lmax, sm);
    const float den = 1e-7f + gmax - gmin;
    for (int i = threadIdx.x; i < n; i += blockDim.x) y[i] = (x[i] - gmin) / den;
}

// ------------------------------------------------------------ box mask of the thresholded prior
constexpr int BOX_MAXPIX = 48 * 48;
__global__ __launch_bounds__(1024) void box_mask_kernel(const float* __restrict__ prior, int G, double thr,
                                                        float* __restrict__ Bm) {
    __shared__ int label[BOX_MAXPIX];
    __shared__ int bx0[BOX_MAXPIX], bx1[BOX_MAXPIX], by0[BOX_MAXPIX], by1[BOX_MAXPIX];
    __shared__ int roots[BOX_MAXPIX];
    __shared__ int nroots;
    __shared__ float sm[16];
    const int n = G * G;
    float lmax = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int u = (int)(unsigned char)(prior[i] * 255.0f);
        lmax = fmaxf(lmax, (float)u);
    }
    const int maxv = (int)block_reduce_max(lmax, sm);
    const int thr_i = (int)(thr * (double)maxv);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int u = (int)(unsigned char)(prior[i] * 255.0f);
        label[i] = u > thr_i ? i : -1;
        bx0[i] = INT_MAX; by0[i] = INT_MAX; bx1[i] = -1; by1[i] = -1;
    }
    if (threadIdx.x == 0) nroots = 0;
    __syncthreads();
    // min-label propagation over the 8-neighbourhood until a full sweep changes nothing
    for (int it = 0; it < n + 1; ++it) {
        int changed = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            int l = label[i];
            if (l < 0) continue;
            const int y = i / G, x = i % G;
            int m = l;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= G || xx < 0 || xx >= G) continue;
                    const int ln = ((volatile int*)label)[yy * G + xx];
                    if (ln >= 0 && ln < m) m = ln;
                }
            // pointer jumping: a label is the index of a pixel of the same component, whose own label is at
            // least as small; following it a few times shortens the sweeps from the component's diameter to
            // about its logarithm and leaves the fixed point (the smallest index of the component) as it is
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int lj = ((volatile int*)label)[m];
                if (lj >= 0 && lj < m) m = lj;
            }
            if (m < l) { ((volatile int*)label)[i] = m; changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int l = label[i];
        if (l < 0) continue;
        const int y = i / G, x = i % G;
        atomicMin(&bx0[l], x); atomicMax(&bx1[l], x); atomicMin(&by0[l], y); atomicMax(&by1[l], y);
        if (l == i) roots[atomicAdd(&nroots, 1)] = i;
    }
    __syncthreads();
    const int nr = nroots;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / G, x = i % G;
        float b = 0.f;
        for (int k = 0; k < nr; ++k) {
            const int r = roots[k];
            const int x0 = bx0[r], y0 = by0[r];
            const int x1 = min(bx1[r] + 1, G - 1), y1 = min(by1[r] + 1, G - 1);
            if (x >= x0 && x < x1 && y >= y0 && y < y1) { b = 1.f; break; }
        }
        Bm[i] = b;
    }
}

// ------------------------------------------------------------ one Sinkhorn step
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ A, int64_t ldA, int P, int chunks,
                                                             float* __restrict__ part) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    const int rpc = (P + chunks - 1) / chunks;
    const int r0 = c * rpc, r1 = min(P, r0 + rpc);
    if (j >= P) return;
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += A[(int64_t)r * ldA + j];
    part[(int64_t)c * P + j] = s;
}
__global__ void colsum_final_kernel(const float* __restrict__ part, int P, int chunks, float* __restrict__ colsum) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P) return;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += part[(int64_t)c * P + j];
    colsum[j] = s;
}
__global__ __launch_bounds__(256) void sinkhorn_rows_kernel(const float* __restrict__ A, int64_t ldA,
                                                            const float* __restrict__ colsum, float* __restrict__ D,
                                                            int64_t ldD, int P) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= P) return;
    const int lane = threadIdx.x & 63;
    const float* ar = A + (int64_t)row * ldA;
    float v[32];
    float s = 0.f;
    const int n = (P + 63) / 64;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (i < n) {
            const int c = i * 64 + lane;
            v[i] = c < P ? ar[c] / colsum[c] : 0.f;
            s += v[i];
        }
    }
    s = wave_sum(s);
    float* dr = D + (int64_t)row * ldD;
    const int np = (int)((ldD + 63) / 64);
#pragma unroll
    for (int i = 0; i < 33; ++i) {
        if (i < np) {
            const int c = i * 64 + lane;
            if (c < ldD) dr[c] = (i < 32 && c < P) ? v[i < 32 ? i : 0] / s : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void max_transpose_kernel(const float* __restrict__ D, const float* __restrict__ DDT,
                                                            int64_t ld, int P, float* __restrict__ R, float* __restrict__ RT) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        float v = 0.f;
        if (r < P && c < ld) {
            if (c < P) v = fmaxf(D[(int64_t)r * ld + c], DDT[(int64_t)r * ld + c]);
            R[(int64_t)r * ld + c] = v;
        }
        tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < P && r < P) RT[(int64_t)c * ld + r] = tile[tx][ty + 8 * k];
    }
}
__global__ void zero_pad_cols_kernel(float* __restrict__ X, int64_t ld, int P) {
    // zero columns P..ld-1 of rows 0..P-1
    const int pad = (int)(ld - P);
    const int64_t total = (int64_t)P * pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        X[(i / pad) * ld + P + (i % pad)] = 0.f;
}

__global__ __launch_bounds__(256) void masked_matvec_kernel(const float* __restrict__ R2, int64_t ld, const float* __restrict__ Bm,
                                                            const float* __restrict__ prior, int P, float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= P) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int c = lane; c < P; c += 64) s += (R2[(int64_t)row * ld + c] * Bm[c]) * prior[c];
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
}

// ------------------------------------------------------------ separable resampling
template <typename TOUT>
__global__ void resample_axis_kernel(const float* __restrict__ in, int planes, int H, int W, int axis, int nout,
                                     const int32_t* __restrict__ xmin, const int32_t* __restrict__ xsize,
                                     const float* __restrict__ wt, int kmax, int quant_u8, float shift, float scale,
                                     TOUT* __restrict__ out) {
    const int Ho = axis == 0 ? nout : H, Wo = axis == 1 ? nout : W;
    const int64_t total = (int64_t)planes * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
        const int64_t pl = i / ((int64_t)Wo * Ho);
        const int o = axis == 1 ? x : y;
        const int m0 = xmin[o], ms = xsize[o];
        const float* src = in + pl * (int64_t)H * W + (axis == 1 ? (int64_t)y * W + m0 : (int64_t)m0 * W + x);
        const int64_t step = axis == 1 ? 1 : W;
        const float* w = wt + (int64_t)o * kmax;
        // the first tap initialises the sum, as the vectorised CPU kernel of ATen does
        float acc = 0.f;
        if (ms <= 8) {
            // short filters (every resize of the path): all taps requested before the first is used
            float v[8], wk[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[k] = k < ms ? src[k * step] : 0.f;
                wk[k] = k < ms ? w[k] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k < ms) {
                    float x = v[k];
                    if (quant_u8) x = (float)(unsigned char)(x * 255.0f) / 255.0f;
                    acc = k == 0 ? x * wk[0] : acc + x * wk[k];
                }
            }
        } else {
            for (int k = 0; k < ms; ++k) {
                float v = src[k * step];
                if (quant_u8) v = (float)(unsigned char)(v * 255.0f) / 255.0f;
                acc = k == 0 ? v * w[0] : acc + v * w[k];
            }
        }
        st_from_f32<TOUT>(out + i, (acc - shift) / scale);
    }
}

__global__ void resample_axis_u8_kernel(const uint8_t* __restrict__ in, int planes, int H, int W, int axis, int nout,
                                        const int32_t* __restrict__ xmin, const int32_t* __restrict__ xsize,
                                        const int32_t* __restrict__ kk, int kmax, uint8_t* __restrict__ out) {
    const int Ho = axis == 0 ? nout : H, Wo = axis == 1 ? nout : W;
    const int64_t total = (int64_t)planes * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
        const int64_t pl = i / ((int64_t)Wo * Ho);
        const int o = axis == 1 ? x : y;
        const int m0 = xmin[o], ms = xsize[o];
        const uint8_t* src = in + pl * (int64_t)H * W + (axis == 1 ? (int64_t)y * W + m0 : (int64_t)m0 * W + x);
        const int64_t step = axis == 1 ? 1 : W;
        const int32_t* w = kk + (int64_t)o * kmax;
        int ss = 1 << 21;  // rounding term of Pillow's 22-bit fixed point
        for (int k = 0; k < ms; ++k) ss += (int)src[k * step] * w[k];
        ss >>= 22;
        out[i] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
    }
}

__global__ void f32_to_u8_kernel(const float* __restrict__ in, uint8_t* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (uint8_t)(in[i] * 255.0f);
}

// ------------------------------------------------------------ proposal pooling
// one block per (mask n, patch row gy): column maxima over the row window, then per-patch max
__global__ __launch_bounds__(256) void pool_masks_kernel(const float* __restrict__ masks, int S, int G, uint8_t* __restrict__ bits) {
    extern __shared__ float colmax[];
    const int gy = blockIdx.x % G;
    const int64_t n = blockIdx.x / G;
    const int ys = (gy * S) / G, ye = ((gy + 1) * S + G - 1) / G;
    const float* mp = masks + n * (int64_t)S * S;
    if (S % 2 == 0 && ((uintptr_t)masks) % 8 == 0) {
        // pairs of columns, the rows of the window sixteen at a time: every load of a pass is requested before
        // the first maximum is taken
        for (int x2 = threadIdx.x; x2 < S / 2; x2 += blockDim.x) {
            float mx = -INFINITY, my = -INFINITY;
            for (int y0 = ys; y0 < ye; y0 += 16) {
                float2 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    v[u] = y0 + u < ye ? *(const float2*)(mp + (int64_t)(y0 + u) * S + 2 * x2) : make_float2(-INFINITY, -INFINITY);
#pragma unroll
                for (int u = 0; u < 16; ++u) { mx = fmaxf(mx, v[u].x); my = fmaxf(my, v[u].y); }
            }
            colmax[2 * x2] = mx;
            colmax[2 * x2 + 1] = my;
        }
    } else {
        for (int x = threadIdx.x; x < S; x += blockDim.x) {
            float m = -INFINITY;
            for (int y = ys; y < ye; ++y) m = fmaxf(m, mp[(int64_t)y * S + x]);
            colmax[x] = m;
        }
    }
    __syncthreads();
    for (int gx = threadIdx.x; gx < G; gx += blockDim.x) {
        const int xs = (gx * S) / G, xe = ((gx + 1) * S + G - 1) / G;
        float m = -INFINITY;
        for (int x = xs; x < xe; ++x) m = fmaxf(m, colmax[x]);
        bits[n * G * G + gy * G + gx] = m > 0.f ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void proposal_stats_kernel(const uint8_t* __restrict__ bits, int N, int P,
                                                             const float* __restrict__ vva, const float* __restrict__ vta,
                                                             int32_t* __restrict__ cnt, float* __restrict__ svv,
                                                             float* __restrict__ svt, int32_t* __restrict__ ucnt) {
    __shared__ float sm[16];
    const int n = blockIdx.x;
    if (n < N) {
        float c = 0.f, a = 0.f, b = 0.f;
        for (int j = threadIdx.x; j < P; j += blockDim.x) {
            if (bits[(int64_t)n * P + j]) { c += 1.f; a += vva[j]; b += vta[j]; }
        }
        c = block_reduce_sum(c, sm); a = block_reduce_sum(a, sm); b = block_reduce_sum(b, sm);
        if (threadIdx.x == 0) { cnt[n] = (int)c; svv[n] = a; svt[n] = b; }
    } else {
        // union of the proposals: one patch per thread, blocks N.. share the patches (integer atomics: exact;
        // the caller clears *ucnt first)
        const int j = (blockIdx.x - N) * blockDim.x + threadIdx.x;
        int any = 0;
        if (j < P) {
            int k = 0;
            for (; k + 8 <= N; k += 8) {
                uint8_t b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) b[u] = bits[(int64_t)(k + u) * P + j];
#pragma unroll
                for (int u = 0; u < 8; ++u) any |= b[u];
            }
            for (; k < N; ++k) any |= bits[(int64_t)k * P + j];
        }
        const float c = block_reduce_sum(any ? 1.f : 0.f, sm);
        if (threadIdx.x == 0 && c > 0.f) atomicAdd(ucnt, (int)c);
    }
}

__global__ __launch_bounds__(256) void cosine_scores_kernel(const float* __restrict__ feats, int64_t ldf, int N, int E,
                                                            const float* __restrict__ text, float* __restrict__ scores) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    const float* f = feats + (int64_t)n * ldf;
    float ff = 0.f, tt = 0.f;
    for (int c = lane; c < E; c += 64) { ff += f[c] * f[c]; tt += text[c] * text[c]; }
    const float nf = sqrtf(wave_sum(ff)), nt = sqrtf(wave_sum(tt));
    float d = 0.f;
    for (int c = lane; c < E; c += 64) d += (f[c] / nf) * (text[c] / nt);
    d = wave_sum(d);
    if (lane == 0) scores[n] = d;
}

// ------------------------------------------------------------ ranking (single block)
// Number types follow the numpy expressions of FilteringMergingModule._score_proposals (:108-136): the pooled
// sums are float32 sums divided in float64, the EMD min-max runs in float64 (ot.emd2 returns float64), the
// AlphaCLIP min-max stays in float32 (float32 cosines with a Python-float epsilon), the mean of four is float64.
// A non-finite score (the host solver returns NaN for a proposal without patches) ranks last and is never kept.
__device__ __forceinline__ double rank_key(double s) { return isfinite(s) ? s : -INFINITY; }

__global__ __launch_bounds__(1024) void rank_kernel(const double* __restrict__ emd, const float* __restrict__ aclip,
                                                    const int32_t* __restrict__ cnt, const float* __restrict__ svv,
                                                    const float* __restrict__ svt, const int32_t* __restrict__ ucnt, int N,
                                                    double alpha, double sthr, double dthr, double* __restrict__ score,
                                                    int32_t* __restrict__ order, uint8_t* __restrict__ keep) {
    __shared__ double sc[1024];
    __shared__ double red[2];  // emd min/max, then the keep threshold
    __shared__ float redf[2];  // aclip min/max
    const int i = threadIdx.x;
    if (i == 0) {
        double emn = INFINITY, emx = -INFINITY;
        float amn = INFINITY, amx = -INFINITY;
        for (int k = 0; k < N; ++k) {
            emn = fmin(emn, emd[k]); emx = fmax(emx, emd[k]);
            amn = fminf(amn, aclip[k]); amx = fmaxf(amx, aclip[k]);
        }
        red[0] = emn; red[1] = emx; redf[0] = amn; redf[1] = amx;
    }
    __syncthreads();
    if (i < N) {
        const double c = (double)cnt[i];
        const double cov = c / (1e-7 + (double)(*ucnt));
        const double pvv = alpha * ((double)svv[i] / (1e-7 + c)) + (1.0 - alpha) * cov;
        const double pvt = alpha * ((double)svt[i] / (1e-7 + c)) + (1.0 - alpha) * cov;
        const double es = (emd[i] - red[0]) / (1e-7 + red[1] - red[0]);
        const float as = __fdiv_rn(__fsub_rn(aclip[i], redf[0]), __fsub_rn(__fadd_rn(1e-7f, redf[1]), redf[0]));
        sc[i] = (es + (double)as + pvv + pvt) / 4.0;
        score[i] = sc[i];
    }
    __syncthreads();
    if (i < N) {
        int rank = 0;
        const double s = rank_key(sc[i]);
        for (int k = 0; k < N; ++k) {
            const double t = rank_key(sc[k]);
            rank += (t > s || (t == s && k < i)) ? 1 : 0;
        }
        order[rank] = i;
    }
    __syncthreads();
    if (i == 0) {
        double top = -INFINITY;
        for (int k = 0; k < N; ++k) top = fmax(top, rank_key(sc[k]));
        red[0] = top < sthr ? dthr * top : sthr;
    }
    __syncthreads();
    if (i < N) keep[i] = (isfinite(sc[i]) && sc[i] >= red[0]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void merge_masks_kernel(const float* __restrict__ masks, const uint8_t* __restrict__ keep,
                                                          int N, int64_t HW, float* __restrict__ merged) {
    __shared__ int kept[1024];
    __shared__ int nk;
    if (threadIdx.x == 0) {
        int c = 0;
        for (int n = 0; n < N; ++n) if (keep[n]) kept[c++] = n;
        nk = c;
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nk; ++k) s += masks[(int64_t)kept[k] * HW + i];
        merged[i] = s > 0.f ? 1.f : 0.f;
    }
}

__device__ __forceinline__ int hist_bin(float v) { return (v >= 0.f && v <= 1.f) ? (v < 0.5f ? 0 : 1) : -1; }

__global__ __launch_bounds__(256) void iou_counts_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                         const float* __restrict__ ign, int64_t HW, float* __restrict__ counts) {
    __shared__ float sm[16];
    float i0 = 0.f, i1 = 0.f, u0 = 0.f, u1 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (int64_t)gridDim.x * blockDim.x) {
        float p = pred[i], g = gt[i];
        if (ign && ign[i] != 0.f) { g = g + ign[i] * 255.f; }
        if (ign && g == 255.f) p = 255.f;
        const int bp = hist_bin(p), bg = hist_bin(g);
        const int bi = (p == g) ? bp : -1;
        // union_k = pred_k + gt_k - inter_k is linear, so it can be summed per pixel
        if (bp == 0) u0 += 1.f; else if (bp == 1) u1 += 1.f;
        if (bg == 0) u0 += 1.f; else if (bg == 1) u1 += 1.f;
        if (bi == 0) { i0 += 1.f; u0 -= 1.f; } else if (bi == 1) { i1 += 1.f; u1 -= 1.f; }
    }
    i0 = block_reduce_sum(i0, sm); i1 = block_reduce_sum(i1, sm);
    u0 = block_reduce_sum(u0, sm); u1 = block_reduce_sum(u1, sm);
    if (threadIdx.x == 0) {
        atomicAdd(counts + 0, i0); atomicAdd(counts + 1, i1); atomicAdd(counts + 2, u0); atomicAdd(counts + 3, u1);
    }
}

inline int grid_for(int64_t n, int block = 256, int cap = 4096) {
    int64_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int marship_vva_prior(const float* S, int64_t ldS, const uint8_t* fg, int R, int P, float* ws, int chunks,
                                 float* out, void* stream) {
    MARS_CHECK(S && fg && ws && out, "vva_prior: null pointer");
    MARS_CHECK(R > 0 && P > 0 && chunks > 0 && chunks <= 1024 && ldS >= P, "vva_prior: bad shape");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(vva_partial_kernel, dim3((P + 255) / 256, chunks), dim3(256), 0, s, S, ldS, fg, R, P, chunks, ws);
    hipLaunchKernelGGL(vva_final_kernel, dim3(1), dim3(1024), 0, s, ws, fg, R, P, chunks, out);
    MARS_LAUNCH_CHECK();
    return 0;
}

// ---- the same prior for callers that cannot allocate: the partial sums live in a file-scope device array
constexpr int VVA_STATIC_CHUNKS = 32;
constexpr int VVA_STATIC_MAX_P = 4096;       // query patches (a 64 x 64 grid; the path uses 37 x 37 = 1369)
constexpr int VVA_STATIC_MAX_DEVICES = 64;
// 4 x 32 x 4096 floats = 2 MiB on every GPU the library runs on, held until the process ends.  In host code this
// name is only a handle for hipGetSymbolAddress: the kernels get the per-device address resolved from it.
static __device__ __align__(256) float scratch[4 * VVA_STATIC_CHUNKS * VVA_STATIC_MAX_P];
static float* g_scratch_ptr[VVA_STATIC_MAX_DEVICES];
static std::mutex g_scratch_mutex;

extern "C" int marship_vva_prior_static(const float* S, int64_t ldS, const uint8_t* fg, int R, int P, float* out,
                                        void* stream) {
    MARS_CHECK(S && fg && out, "vva_prior_static: null pointer");
    MARS_CHECK(R > 0 && P > 0 && ldS >= P, "vva_prior_static: bad shape");
    MARS_CHECK(P <= VVA_STATIC_MAX_P, "vva_prior_static: %d query patches exceed the static scratch (at most %d)", P,
               VVA_STATIC_MAX_P);
    int dev = -1;
    MARS_CHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < VVA_STATIC_MAX_DEVICES,
               "vva_prior_static: no current device (or its index is beyond %d)", VVA_STATIC_MAX_DEVICES - 1);
    float* ws = nullptr;
    {
        // every GPU holds its own copy of the array at its own address: resolved once per device, with that
        // device current, on the first call there
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[dev]) {
            void* p = nullptr;
            const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            MARS_CHECK(e == hipSuccess && p, "vva_prior_static: hipGetSymbolAddress failed on device %d: %s", dev,
                       hipGetErrorString(e));
            g_scratch_ptr[dev] = (float*)p;
        }
        ws = g_scratch_ptr[dev];
    }
    // the first kernel writes every partial the second one reads (all chunks, all P columns), so nothing has to
    // start at zero; what a previous, larger call left behind lies beyond 4 * chunks * P and is never read
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(vva_partial_kernel, dim3((P + 255) / 256, VVA_STATIC_CHUNKS), dim3(256), 0, s, S, ldS, fg, R, P,
                       VVA_STATIC_CHUNKS, ws);
    hipLaunchKernelGGL(vva_final_kernel, dim3(1), dim3(1024), 0, s, ws, fg, R, P, VVA_STATIC_CHUNKS, out);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_minmax(const float* x, float* y, int n, void* stream) {
    MARS_CHECK(x && y && n > 0, "minmax: bad arguments");
    hipLaunchKernelGGL(minmax_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, y, n);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_box_mask(const float* prior, int G, double threshold, float* Bm, void* stream) {
    MARS_CHECK(prior && Bm, "box_mask: null pointer");
    MARS_CHECK(G > 0 && G * G <= BOX_MAXPIX, "box_mask: grid %d exceeds %d pixels", G, BOX_MAXPIX);
    hipLaunchKernelGGL(box_mask_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, prior, G, threshold, Bm);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_sinkhorn1(const float* A, int64_t ldA, float* D, int64_t ldD, int P, float* colsum, void* stream) {
    MARS_CHECK(A && D && colsum, "sinkhorn1: null pointer");
    MARS_CHECK(P > 0 && P <= 2048 && ldA >= P && ldD >= P && ldD <= 2112, "sinkhorn1: P=%d out of range", P);
    hipStream_t s = (hipStream_t)stream;
    const int chunks = 32;  // colsum scratch: (chunks + 1) * P floats
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((P + 255) / 256, chunks), dim3(256), 0, s, A, ldA, P, chunks, colsum + P);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((P + 255) / 256), dim3(256), 0, s, colsum + P, P, chunks, colsum);
    hipLaunchKernelGGL(sinkhorn_rows_kernel, dim3((P + 3) / 4), dim3(256), 0, s, A, ldA, colsum, D, ldD, P);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_max_and_transpose(const float* D, const float* DDT, int64_t ld, int P, float* R, float* RT,
                                         void* stream) {
    MARS_CHECK(D && DDT && R && RT && P > 0 && ld >= P, "max_and_transpose: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (ld > P) hipLaunchKernelGGL(zero_pad_cols_kernel, dim3(grid_for((int64_t)P * (ld - P))), dim3(256), 0, s, RT, ld, P);
    hipLaunchKernelGGL(max_transpose_kernel, dim3((unsigned)((ld + 31) / 32), (P + 31) / 32), dim3(256), 0, s, D, DDT, ld, P, R, RT);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_masked_matvec(const float* R2, int64_t ld, const float* Bm, const float* prior, int P, float* out,
                                     void* stream) {
    MARS_CHECK(R2 && Bm && prior && out && P > 0 && ld >= P, "masked_matvec: bad arguments");
    hipLaunchKernelGGL(masked_matvec_kernel, dim3((P + 3) / 4), dim3(256), 0, (hipStream_t)stream, R2, ld, Bm, prior, P, out);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_resample_axis(const float* in, int planes, int H, int W, int axis, int nout, const int32_t* xmin,
                                     const int32_t* xsize, const float* weights, int kmax, int quant_u8, float shift,
                                     float scale, void* out, int out_dtype, void* stream) {
    MARS_CHECK(in && out && xmin && xsize && weights, "resample_axis: null pointer");
    MARS_CHECK(planes > 0 && H > 0 && W > 0 && nout > 0 && kmax > 0 && (axis == 0 || axis == 1) && scale != 0.f, "resample_axis: bad shape");
    const int64_t total = (int64_t)planes * (axis == 0 ? nout : H) * (axis == 1 ? nout : W);
    dim3 grid(grid_for(total, 256, 16384)), block(256);
    if (out_dtype == 1) hipLaunchKernelGGL(resample_axis_kernel<__bf16>, grid, block, 0, (hipStream_t)stream, in, planes, H, W, axis, nout, xmin, xsize, weights, kmax, quant_u8, shift, scale, (__bf16*)out);
    else hipLaunchKernelGGL(resample_axis_kernel<float>, grid, block, 0, (hipStream_t)stream, in, planes, H, W, axis, nout, xmin, xsize, weights, kmax, quant_u8, shift, scale, (float*)out);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_resample_axis_u8(const uint8_t* in, int planes, int H, int W, int axis, int nout,
                                        const int32_t* xmin, const int32_t* xsize, const int32_t* kk, int kmax,
                                        uint8_t* out, void* stream) {
    MARS_CHECK(in && out && xmin && xsize && kk, "resample_axis_u8: null pointer");
    MARS_CHECK(planes > 0 && H > 0 && W > 0 && nout > 0 && kmax > 0 && (axis == 0 || axis == 1), "resample_axis_u8: bad shape");
    const int64_t total = (int64_t)planes * (axis == 0 ? nout : H) * (axis == 1 ? nout : W);
    hipLaunchKernelGGL(resample_axis_u8_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, planes, H, W, axis, nout, xmin, xsize, kk, kmax, out);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_f32_to_u8(const float* in, uint8_t* out, int64_t n, void* stream) {
    MARS_CHECK(in && out && n > 0, "f32_to_u8: bad arguments");
    hipLaunchKernelGGL(f32_to_u8_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, out, n);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_pool_masks(const float* masks, int N, int S, int G, uint8_t* bits, void* stream) {
    MARS_CHECK(masks && bits, "pool_masks: null pointer");
    MARS_CHECK(N > 0 && S > 0 && G > 0 && G <= S && S <= 8192, "pool_masks: bad shape N=%d S=%d G=%d", N, S, G);
    hipLaunchKernelGGL(pool_masks_kernel, dim3((unsigned)((int64_t)N * G)), dim3(256), S * sizeof(float), (hipStream_t)stream, masks, S, G, bits);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_proposal_stats(const uint8_t* bits, int N, int P, const float* vva, const float* vta, int32_t* cnt,
                                      float* sum_vva, float* sum_vta, int32_t* union_cnt, void* stream) {
    MARS_CHECK(bits && vva && vta && cnt && sum_vva && sum_vta && union_cnt, "proposal_stats: null pointer");
    MARS_CHECK(N > 0 && P > 0, "proposal_stats: bad shape");
    MARS_CHECK(hipMemsetAsync(union_cnt, 0, sizeof(int32_t), (hipStream_t)stream) == hipSuccess, "proposal_stats: memset failed");
    hipLaunchKernelGGL(proposal_stats_kernel, dim3(N + (P + 255) / 256), dim3(256), 0, (hipStream_t)stream, bits, N, P, vva, vta, cnt, sum_vva, sum_vta, union_cnt);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_cosine_scores(const float* feats, int64_t ldf, int N, int E, const float* text, float* scores,
                                     void* stream) {
    MARS_CHECK(feats && text && scores && N > 0 && E > 0 && ldf >= E, "cosine_scores: bad arguments");
    hipLaunchKernelGGL(cosine_scores_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, feats, ldf, N, E, text, scores);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_rank(const double* emd, const float* aclip, const int32_t* cnt, const float* sum_vva,
                            const float* sum_vta, const int32_t* union_cnt, int N, double alpha, double static_thr,
                            double dynamic_thr, double* score, int32_t* order, uint8_t* keep, void* stream) {
    MARS_CHECK(emd && aclip && cnt && sum_vva && sum_vta && union_cnt && score && order && keep, "rank: null pointer");
    MARS_CHECK(N > 0 && N <= 1024, "rank: N=%d exceeds 1024 proposals", N);
    hipLaunchKernelGGL(rank_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, emd, aclip, cnt, sum_vva, sum_vta, union_cnt, N, alpha, static_thr, dynamic_thr, score, order, keep);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_merge_masks(const float* masks, const uint8_t* keep, int N, int64_t HW, float* merged, void* stream) {
    MARS_CHECK(masks && keep && merged && N > 0 && N <= 1024 && HW > 0, "merge_masks: bad arguments");
    hipLaunchKernelGGL(merge_masks_kernel, dim3(grid_for(HW)), dim3(256), 0, (hipStream_t)stream, masks, keep, N, HW, merged);
    MARS_LAUNCH_CHECK();
    return 0;
}

// one block per mask; lanes stride over the points
__global__ __launch_bounds__(256) void points_in_masks_kernel(const float* __restrict__ masks, int H, int W,
                                                              const float* __restrict__ points, int K, int32_t* __restrict__ counts) {
    const float* m = masks + (int64_t)blockIdx.x * H * W;
    int c = 0;
    for (int i = threadIdx.x; i < K; i += 256) {
        // astype(int64) truncates toward zero; then clip to the image
        int x = (int)points[2 * i], y = (int)points[2 * i + 1];
        x = min(max(x, 0), W - 1);
        y = min(max(y, 0), H - 1);
        c += m[(int64_t)y * W + x] != 0.f;
    }
    __shared__ float sm[8];
    const float tot = block_reduce_sum((float)c, sm);      // exact: a count below 2^24
    if (threadIdx.x == 0) counts[blockIdx.x] = (int32_t)tot;
}

// one wave per (row, prompt) pair: three dot products, f32 accumulation in lane-strided order
__global__ __launch_bounds__(256) void cosine_table_kernel(const float* __restrict__ feats, int64_t ldf, int N, int E,
                                                           const float* __restrict__ text, int P, float scale, float* __restrict__ out) {
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pair >= N * P) return;
    const int i = pair / P, p = pair % P;
    const float* f = feats + (int64_t)i * ldf;
    const float* t = text + (int64_t)p * E;
    double ff = 0.0, tt = 0.0, ft = 0.0;
    for (int e = lane; e < E; e += 64) {
        const double a = f[e], b = t[e];
        ff += a * a; tt += b * b; ft += a * b;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ff += __shfl_xor(ff, o, 64); tt += __shfl_xor(tt, o, 64); ft += __shfl_xor(ft, o, 64); }
    if (lane == 0) out[(int64_t)i * P + p] = (float)(scale * ft / (sqrt(ff) * sqrt(tt)));
}

extern "C" int marship_points_in_masks(const float* masks, int n, int H, int W, const float* points, int K, int32_t* counts,
                                       void* stream) {
    MARS_CHECK(masks && points && counts && n > 0 && H > 0 && W > 0 && K > 0, "points_in_masks: bad arguments");
    hipLaunchKernelGGL(points_in_masks_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, masks, H, W, points, K, counts);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_cosine_table(const float* feats, int64_t ldf, int N, int E, const float* text, int P, float scale,
                                    float* out, void* stream) {
    MARS_CHECK(feats && text && out && N > 0 && E > 0 && P > 0 && ldf >= E, "cosine_table: bad arguments");
    const int64_t pairs = (int64_t)N * P;
    MARS_CHECK(pairs < (1ll << 30), "cosine_table: table too large");
    hipLaunchKernelGGL(cosine_table_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, feats, ldf, N, E,
                       text, P, scale, out);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_iou_counts(const float* pred, const float* gt, const float* ignore, int64_t HW, float* counts,
                                  void* stream) {
    MARS_CHECK(pred && gt && counts && HW > 0, "iou_counts: bad arguments");
    hipLaunchKernelGGL(iou_counts_kernel, dim3(grid_for(HW, 256, 64)), dim3(256), 0, (hipStream_t)stream, pred, gt, ignore, HW, counts);
    MARS_LAUNCH_CHECK();
    return 0;
}


// ---- bit-packed proposal cache ---------------------------------------------------------------
// words[mask, j] bit i = masks[mask, 32 j + i] > 0.  A wave covers 64 consecutive pixels; its ballot is two words.
__global__ __launch_bounds__(256) void pack_masks_kernel(const float* __restrict__ masks, int64_t HW, int64_t nwords,
                                                         uint32_t* __restrict__ words) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float* m = masks + (int64_t)blockIdx.y * HW;
    const unsigned long long b = __ballot(e < HW && m[e < HW ? e : 0] > 0.f);
    if ((threadIdx.x & 63) == 0) {
        const int64_t w = e >> 5;
        uint32_t* o = words + (int64_t)blockIdx.y * nwords;
        if (w < nwords) o[w] = (uint32_t)b;
        if (w + 1 < nwords) o[w + 1] = (uint32_t)(b >> 32);
    }
}

__global__ __launch_bounds__(256) void unpack_masks_kernel(const uint32_t* __restrict__ words, int64_t HW, int64_t nwords,
                                                           float* __restrict__ masks) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= HW) return;
    const uint32_t w = words[(int64_t)blockIdx.y * nwords + (e >> 5)];
    masks[(int64_t)blockIdx.y * HW + e] = (w >> (e & 31)) & 1u ? 1.f : 0.f;
}

extern "C" int marship_pack_masks(const float* masks, int n, int64_t HW, uint32_t* words, void* stream) {
    MARS_CHECK(masks && words && n > 0 && n <= 65535 && HW > 0, "pack_masks: bad arguments");
    hipLaunchKernelGGL(pack_masks_kernel, dim3((unsigned)((HW + 255) / 256), n), dim3(256), 0, (hipStream_t)stream, masks, HW,
                       (HW + 31) / 32, words);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_unpack_masks(const uint32_t* words, int n, int64_t HW, float* masks, void* stream) {
    MARS_CHECK(masks && words && n > 0 && n <= 65535 && HW > 0, "unpack_masks: bad arguments");
    hipLaunchKernelGGL(unpack_masks_kernel, dim3((unsigned)((HW + 255) / 256), n), dim3(256), 0, (hipStream_t)stream, words, HW,
                       (HW + 31) / 32, masks);
    MARS_LAUNCH_CHECK();
    return 0;
}
