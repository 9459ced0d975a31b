// What the matrix pipes sustain with nothing else in the way: every SIMD issues independent
// v_mfma_f32_32x32x16_bf16 (or 16x16x32) back to back, for `waves` waves per SIMD.  Prints TFLOP/s by the wall clock.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/diag_mfma_peak.hip -o tools/diag_bin/diag_mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int BIG>
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
    bf16x8 a, b, a1, a2, a3, a4, a5, a6, a7;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1));
        a1[i] = (__bf16)(0.003f * (threadIdx.x + i)); a2[i] = (__bf16)(0.005f * (threadIdx.x + i)); a3[i] = (__bf16)(0.007f * (threadIdx.x + i));
        a4[i] = (__bf16)(0.009f * (threadIdx.x + i)); a5[i] = (__bf16)(0.011f * (threadIdx.x + i)); a6[i] = (__bf16)(0.013f * (threadIdx.x + i));
        a7[i] = (__bf16)(0.015f * (threadIdx.x + i));
    }
    float s = 0.f;
    if (BIG) {
        f32x16 c0 = (f32x16)(0.f), c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b, c3, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    } else {
        f32x4 c0 = (f32x4)(0.f), c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a4, b, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a5, b, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a6, b, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a7, b, c7, 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i] + c4[i] + c5[i] + c6[i] + c7[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out; CK(hipMalloc(&out, 4096 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int big = 1; big >= 0; --big)
        for (int wps = 1; wps <= 2; ++wps) {
            const int blocks = 256 * wps, iters = 40000;        // one workgroup of 4 waves per CU and wave-per-SIMD step
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (big) hipLaunchKernelGGL(mfma_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL(mfma_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
                CK(hipEventRecord(e1, 0));
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double flops = (double)blocks * 4 * iters * (big ? 4 : 8) * 32768.0;
                if (rep == 2) printf("%s, %d wave(s) a SIMD: %.2f ms, %.0f TFLOP/s (%.2f GHz if the pipe never idles)\n", big ? "32x32x16" : "16x16x32", wps, ms,
                                     flops / ms / 1e9, flops / ms / 1e9 / 2500.0 * 2.4);
            }
        }
    return 0;
}
