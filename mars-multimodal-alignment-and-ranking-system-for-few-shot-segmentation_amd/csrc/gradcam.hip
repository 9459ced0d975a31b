// Closed-form backward pieces of the Softmax-GradCAM through the last CLIP block.
// The reference obtains d softmax_logit[0] / d ln_1(x) with autograd
// (pytorch_grad_cam/base_cam.py:92-99); here every factor of that chain rule is an
// explicit kernel and the matrix products reuse the TN GEMM.
#include "common.h"
#include "../../include/mars_hip.h"

namespace {

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ dy, int64_t dys, float dy_scale,
                                                            int skip_rows, const float* __restrict__ add,
                                                            float* __restrict__ dx, int rows, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (int64_t)row * D;
    const float* dyr = dy + (int64_t)row * dys;
    const bool live = row >= skip_rows;
    float v[32], g[32];
    float s = 0.f;
    const int n = (D + 63) / 64;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (i < n) {
            const int c = i * 64 + lane;
            v[i] = c < D ? xr[c] : 0.f;
            s += v[i];
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (i < n) {
            const int c = i * 64 + lane;
            v[i] = c < D ? v[i] - mean : 0.f;
            q += v[i] * v[i];
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (i < n) {
            const int c = i * 64 + lane;
            v[i] *= rstd;  // x_hat
            g[i] = (c < D && live) ? dyr[c] * dy_scale * w[c] : 0.f;
            sg += g[i];
            sgx += g[i] * v[i];
        }
    }
    const float mg = wave_sum(sg) / (float)D, mgx = wave_sum(sgx) / (float)D;
    float* dr = dx + (int64_t)row * D;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (i < n) {
            const int c = i * 64 + lane;
            if (c < D) {
                float o = rstd * (g[i] - mg - v[i] * mgx);
                if (add) o += add[(int64_t)row * D + c];
                dr[c] = o;
            }
        }
    }
}

__global__ void quickgelu_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ pre, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float z = pre[i];
        const float sg = 1.0f / (1.0f + expf(-1.702f * z));
        out[i] = dh[i] * (sg + 1.702f * z * sg * (1.0f - sg));
    }
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P, const float* __restrict__ dP,
                                                          float* __restrict__ dS, int64_t ld, int64_t nrows, int T, float scale) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int lane = threadIdx.x & 63;
    const float* pr = P + row * ld;
    const float* dr = dP + row * ld;
    float s = 0.f;
    for (int c = lane; c < T; c += 64) s += pr[c] * dr[c];
    s = wave_sum(s);
    float* o = dS + row * ld;
    for (int c = lane; c < ld; c += 64) o[c] = c < T ? pr[c] * (dr[c] - s) * scale : 0.f;
}

// 64 columns x 16 row groups per block
__global__ __launch_bounds__(1024) void col_mean_kernel(const float* __restrict__ x, int64_t ld, int r0, int r1, int D,
                                                        float* __restrict__ out) {
    __shared__ float sm[16][64];
    const int lane = threadIdx.x & 63, gq = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < D)
        for (int r = r0 + gq; r < r1; r += 16) s += x[(int64_t)r * ld + c];
    sm[gq][lane] = s;
    __syncthreads();
    if (gq == 0 && c < D) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][lane];
        out[c] = t / (float)(r1 - r0);
    }
}

__global__ __launch_bounds__(1024) void cam_head_kernel(const float* __restrict__ f, const float* __restrict__ proj,
                                                        const float* __restrict__ text, int D, int E, float ls,
                                                        float* __restrict__ logits, float* __restrict__ df) {
    __shared__ __attribute__((aligned(16))) float g[1024], dg[1024];
    __shared__ float sm[16];
    const int t = threadIdx.x;
    // g = f . proj.  One workgroup reads the whole matrix, so what counts is bytes in flight: with E a multiple of
    // four (and at most 1024) a thread takes four columns of every (1024 / (E/4))-th row as 16-byte loads, eight rows
    // at a time; otherwise column t % E, the rows dealt to the 1024 / E thread groups.
    const bool vec = (E % 4 == 0) && E <= 1024 && E >= 4 && (1024 % (E / 4) == 0) && (((uintptr_t)proj) % 16 == 0);
    float ge = 0.f;
    if (vec) {
        const int e4n = E / 4, ngrp = 1024 / e4n;
        const int e4 = t % e4n, grp = t / e4n;
        float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
        int d = grp;
        for (; d + 7 * ngrp < D; d += 8 * ngrp) {
            float4 v[8];
            float fd[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = *(const float4*)(proj + (int64_t)(d + u * ngrp) * E + 4 * e4);
                fd[u] = f[d + u * ngrp];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                part.x += fd[u] * v[u].x; part.y += fd[u] * v[u].y; part.z += fd[u] * v[u].z; part.w += fd[u] * v[u].w;
            }
        }
        for (; d < D; d += ngrp) {
            const float4 v = *(const float4*)(proj + (int64_t)d * E + 4 * e4);
            const float fd = f[d];
            part.x += fd * v.x; part.y += fd * v.y; part.z += fd * v.z; part.w += fd * v.w;
        }
        // column sums over the row groups, a group at a time through dg (E floats)
        for (int k = 0; k < ngrp; ++k) {
            if (grp == k) {
                float* q = dg + 4 * e4;
                if (k == 0) { q[0] = part.x; q[1] = part.y; q[2] = part.z; q[3] = part.w; }
                else { q[0] += part.x; q[1] += part.y; q[2] += part.z; q[3] += part.w; }
            }
            __syncthreads();
        }
        if (t < E) { ge = dg[t]; g[t] = ge; }
        __syncthreads();
    } else {
        const int ngrp = 1024 / E > 0 ? 1024 / E : 1;
        float part = 0.f;
        const int e = t % E, grp = t / E;
        if (grp < ngrp)
            for (int d = grp; d < D; d += ngrp) part += f[d] * proj[(int64_t)d * E + e];
        dg[t] = part;
        __syncthreads();
        if (t < E) {
            for (int k = 0; k < ngrp; ++k) ge += dg[k * E + t];
            g[t] = ge;
        }
        __syncthreads();
    }
    const float t0 = t < E ? text[t] : 0.f, t1 = t < E ? text[E + t] : 0.f;
    const float ng = sqrtf(block_reduce_sum(ge * ge, sm));
    const float n0 = sqrtf(block_reduce_sum(t0 * t0, sm));
    const float n1 = sqrtf(block_reduce_sum(t1 * t1, sm));
    const float gh = ge / ng, th0 = t0 / n0, th1 = t1 / n1;
    const float l0 = block_reduce_sum(ls * gh * th0, sm);
    const float l1 = block_reduce_sum(ls * gh * th1, sm);
    const float mx = fmaxf(l0, l1);
    const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
    const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
    if (t == 0) { logits[0] = p0; logits[1] = p1; }
    const float dl0 = p0 * (1.f - p0), dl1 = -p0 * p1;
    const float dgh = ls * (dl0 * th0 + dl1 * th1);
    const float dot = block_reduce_sum(gh * dgh, sm);
    if (t < E) dg[t] = (dgh - gh * dot) / ng;
    __syncthreads();
    // df = proj . dg: one wave per row, lanes along the row
    const int lane = t & 63, wv = t >> 6;
    if (vec && E <= 1024) {
        // four rows per wave and pass, 16-byte loads: E/4 <= 256 chunks of a row over the 64 lanes
        const int e4n = E / 4;
        for (int d0 = wv * 4; d0 < D; d0 += 64) {
            float s[4] = {0.f, 0.f, 0.f, 0.f};
            for (int c = lane; c < e4n; c += 64) {
                const float4 w4 = *(const float4*)(dg + 4 * c);
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    v[u] = d0 + u < D ? *(const float4*)(proj + (int64_t)(d0 + u) * E + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 4; ++u) s[u] += v[u].x * w4.x + v[u].y * w4.y + v[u].z * w4.z + v[u].w * w4.w;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float r = wave_sum(s[u]);
                if (lane == 0 && d0 + u < D) df[d0 + u] = r;
            }
        }
        return;
    }
    for (int d = wv; d < D; d += 16) {
        float s = 0.f;
        for (int e = lane; e < E; e += 64) s += proj[(int64_t)d * E + e] * dg[e];
        s = wave_sum(s);
        if (lane == 0) df[d] = s;
    }
}

__global__ __launch_bounds__(256) void cam_rows_kernel(const float* __restrict__ act, int64_t ld, const float* __restrict__ w,
                                                       int r0, int rows, int D, float* __restrict__ cam) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* a = act + (int64_t)(r0 + r) * ld;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += w[c] * a[c];
    s = wave_sum(s);
    if (lane == 0) cam[r] = fmaxf(s, 0.f);
}

inline int grid_for(int64_t n, int block = 256, int cap = 4096) {
    int64_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int marship_layernorm_bwd(const float* x, const float* w, const float* dy, int64_t dy_stride, float dy_scale,
                                     int skip_rows, const float* add, float* dx, int rows, int D, float eps, void* stream) {
    MARS_CHECK(x && w && dy && dx, "layernorm_bwd: null pointer");
    MARS_CHECK(rows > 0 && D > 0 && D <= 2048, "layernorm_bwd: D=%d out of range", D);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, dy, dy_stride, dy_scale, skip_rows, add, dx, rows, D, eps);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_quickgelu_bwd(const float* dh, const float* pre, float* out, int64_t n, void* stream) {
    MARS_CHECK(dh && pre && out && n > 0, "quickgelu_bwd: bad arguments");
    hipLaunchKernelGGL(quickgelu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dh, pre, out, n);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_softmax_bwd(const float* P, const float* dP, float* dS, int64_t ld, int nb, int T, float scale,
                                   void* stream) {
    MARS_CHECK(P && dP && dS && nb > 0 && T > 0 && ld >= T, "softmax_bwd: bad arguments");
    const int64_t nrows = (int64_t)nb * T;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P, dP, dS, ld, nrows, T, scale);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_col_mean(const float* x, int64_t ld, int r0, int r1, int D, float* out, void* stream) {
    MARS_CHECK(x && out && r1 > r0 && D > 0 && ld >= D, "col_mean: bad arguments");
    hipLaunchKernelGGL(col_mean_kernel, dim3((D + 63) / 64), dim3(1024), 0, (hipStream_t)stream, x, ld, r0, r1, D, out);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_cam_head(const float* f, const float* proj, const float* text, int D, int E, float logit_scale_exp,
                                float* logits, float* df, void* stream) {
    MARS_CHECK(f && proj && text && logits && df, "cam_head: null pointer");
    MARS_CHECK(D > 0 && E > 0 && E <= 1024, "cam_head: E=%d exceeds 1024", E);
    hipLaunchKernelGGL(cam_head_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, f, proj, text, D, E, logit_scale_exp, logits, df);
    MARS_LAUNCH_CHECK();
    return 0;
}

extern "C" int marship_cam_rows(const float* act, int64_t ld, const float* w, int r0, int rows, int D, float* cam,
                                void* stream) {
    MARS_CHECK(act && w && cam && rows > 0 && D > 0 && ld >= D, "cam_rows: bad arguments");
    hipLaunchKernelGGL(cam_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, act, ld, w, r0, rows, D, cam);
    MARS_LAUNCH_CHECK();
    return 0;
}
