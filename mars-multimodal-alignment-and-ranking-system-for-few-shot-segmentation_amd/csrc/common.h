// Shared device helpers and the error plumbing of the C-ABI (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MARS_WAVE 64

// thread-local last error text, returned by marship_last_error()
void mars_set_error(const char* fmt, ...);

#define MARS_CHECK(cond, ...)                    \
    do {                                         \
        if (!(cond)) {                           \
            mars_set_error(__VA_ARGS__);         \
            return -1;                           \
        }                                        \
    } while (0)

#define MARS_LAUNCH_CHECK()                                                  \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) {                                             \
            mars_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,    \
                           hipGetErrorString(e__));                          \
            return -2;                                                       \
        }                                                                    \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(__bf16 v) { return (float)v; }
__device__ __forceinline__ __bf16 f32_to_bf16(float v) { return (__bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// typed element load/store used by kernels templated on the activation type
template <typename T> __device__ __forceinline__ float ld_as_f32(const T* p);
template <> __device__ __forceinline__ float ld_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_as_f32<__bf16>(const __bf16* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void st_from_f32(T* p, float v);
template <> __device__ __forceinline__ void st_from_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_from_f32<__bf16>(__bf16* p, float v) { *p = (__bf16)v; }

// block-wide reductions through one LDS stage; every thread of the block must call them
__device__ __forceinline__ float block_reduce_sum(float v, float* sm) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sm[i];
    return r;
}
__device__ __forceinline__ float block_reduce_max(float v, float* sm) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < nw; ++i) r = fmaxf(r, sm[i]);
    return r;
}
__device__ __forceinline__ float block_reduce_min(float v, float* sm) {
    v = wave_min(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    float r = INFINITY;
    for (int i = 0; i < nw; ++i) r = fminf(r, sm[i]);
    return r;
}

