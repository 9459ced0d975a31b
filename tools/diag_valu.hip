// Issue rates of vector ALU instructions on one SIMD, alone and with a second wave beside it.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/diag_valu.hip -o tools/diag_bin/diag_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int WHICH>
__global__ __launch_bounds__(512) void rate_kernel(float* out, unsigned long long* cyc, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = -0.001f * (threadIdx.x + i);
    float b0 = 1.0001f, b1 = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[4] = {{a[0], a[1]}, {a[2], a[3]}, {a[4], a[5]}, {a[6], a[7]}};
    f2 q = {b0, b1};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (WHICH == 0) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));) }
        if (WHICH == 1) { REP8(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b0));) }
        if (WHICH == 2) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(q));) }
        if (WHICH == 3) { REP8(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));) }
        if (WHICH == 4) { REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %0" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));) }
        if (WHICH == 5) { REP8(asm volatile("v_exp_f32 %0, %0\n v_pk_mul_f32 %4, %4, %6\n v_exp_f32 %1, %1\n v_pk_mul_f32 %5, %5, %6\n v_exp_f32 %2, %2\n v_pk_mul_f32 %4, %4, %6\n v_exp_f32 %3, %3\n v_pk_mul_f32 %5, %5, %6" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(p[0]), "+v"(p[1]) : "v"(q));) }
        if (WHICH == 6) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(q));) }
        if (WHICH == 7) { REP8(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b1));) }
        if (WHICH == 8) { REP8(asm volatile("v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %4\n v_ldexp_f32 %2, %2, %4\n v_ldexp_f32 %3, %3, %4" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(1));) }
        if (WHICH == 10) { REP32(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b0));) }
        if (WHICH == 11) { REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b0));) }
        if (WHICH == 12) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));) }
        if (WHICH == 13) { REP8(asm volatile("v_exp_f32 %0, %0\n v_mul_f32 %4, %4, %8\n v_exp_f32 %1, %1\n v_mul_f32 %5, %5, %8\n v_exp_f32 %2, %2\n v_mul_f32 %6, %6, %8\n v_exp_f32 %3, %3\n v_mul_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b0));) }
        if (WHICH == 9) { REP8(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int WHICH>
int run(const char* name, int per_iter, float* out, unsigned long long* cyc) {
    for (int waves = 4; waves <= 16; waves *= 2) {       // waves per workgroup: 4 = one a SIMD, 8 = two a SIMD
        const int iters = 2000;
        hipLaunchKernelGGL(rate_kernel<WHICH>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
        CK(hipDeviceSynchronize());
        unsigned long long h[16];
        CK(hipMemcpy(h, cyc, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
        printf("  %-22s %d wave(s) a SIMD: %.2f clocks an instruction (per wave)\n", name, waves / 4, (double)h[0] / ((double)iters * per_iter));
    }
    return 0;
}

int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&cyc, 64 * 8));
    // (s_memtime counts a constant 100 MHz clock on this part: clocks below are of that counter unless they look like core clocks)
    run<10>("v_mul_f32 1 chain", 32, out, cyc);
    run<1>("v_mul_f32 4 chains", 32, out, cyc);
    run<11>("v_mul_f32 8 chains", 64, out, cyc);
    run<12>("v_exp_f32 8 chains", 64, out, cyc);
    run<13>("exp+mul 8 chains", 64, out, cyc);
    run<7>("v_add_f32", 32, out, cyc);
    run<0>("v_exp_f32", 32, out, cyc);
    run<3>("v_exp_f16", 32, out, cyc);
    run<2>("v_pk_mul_f32", 32, out, cyc);
    run<6>("v_pk_fma_f32", 32, out, cyc);
    run<4>("v_cvt_pk_bf16_f32", 32, out, cyc);
    run<8>("v_ldexp_f32", 32, out, cyc);
    run<9>("v_max3_f32", 32, out, cyc);
    run<5>("exp + pk_mul pairs", 64, out, cyc);
    return 0;
}
