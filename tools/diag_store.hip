// Store-shape micro-benchmark: the epilogue of a 64-column bf16 wave tile writes 128 bytes per row.  Shape A is
// what a C^T fragment gives (one instruction = 16 rows x 64 bytes, the other half of each line in the next
// instruction), shape B whole lines (one instruction = 8 rows x 128 bytes).  Same bytes, same rows.
//   hipcc -O3 --offload-arch=gfx950 tools/diag_store.hip -o tools/diag_bin/diag_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// grid: (M / 256) x (row_bytes / 512) tiles; 8 waves: wave = (wr, wc), wave tile 128 rows x 128 bytes
template <int SHAPE>
__global__ __launch_bounds__(512) void store_kernel(unsigned char* out, int M, int64_t row_bytes, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
    const int ntn = (int)(row_bytes / 512);
    const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
    unsigned char* base = out + (int64_t)(tm * 256 + wr * 128) * row_bytes + tn * 512 + wc * 128;
    f32x4 v = {(float)lane, 1.f, 2.f, (float)wave};
    for (int r = 0; r < reps; ++r) {
        if (SHAPE == 0) {
            const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    *(f32x4*)(base + (int64_t)(16 * m + r16) * row_bytes + 64 * h + 16 * kq) = v;
        } else if (SHAPE == 2) {
            // half lines as shape A, but consecutive lanes next to each other: lane = 4 * row + chunk
            const int r16 = lane >> 2, kq = lane & 3;
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    *(f32x4*)(base + (int64_t)(16 * m + r16) * row_bytes + 64 * h + 16 * kq) = v;
        } else if (SHAPE == 3) {
            // today's f32 fragment: a lane owns 32 consecutive bytes and writes them in two instructions, so one
            // instruction covers every other 16 bytes of 128 (the tile is 128 rows x 128 bytes here as well)
            const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    *(f32x4*)(base + (int64_t)(16 * m + r16) * row_bytes + 32 * kq + 16 * q) = v;
        } else {
            const int r8 = lane >> 3, c = lane & 7;
#pragma unroll
            for (int m = 0; m < 16; ++m)
                *(f32x4*)(base + (int64_t)(8 * m + r8) * row_bytes + 16 * c) = v;
        }
        v[1] += 1.f;
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 36864;
    const int64_t row_bytes = argc > 2 ? atoll(argv[2]) : 6144;
    const int reps = argc > 3 ? atoi(argv[3]) : 1;
    unsigned char* out;
    CK(hipMalloc(&out, (size_t)M * row_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 grid((M / 256) * (int)(row_bytes / 512)), block(512);
    for (int shape = 0; shape < 4; ++shape)
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0, 0));
            if (shape == 0) hipLaunchKernelGGL(store_kernel<0>, grid, block, 0, 0, out, M, row_bytes, reps);
            else if (shape == 1) hipLaunchKernelGGL(store_kernel<1>, grid, block, 0, 0, out, M, row_bytes, reps);
            else if (shape == 2) hipLaunchKernelGGL(store_kernel<2>, grid, block, 0, 0, out, M, row_bytes, reps);
            else hipLaunchKernelGGL(store_kernel<3>, grid, block, 0, 0, out, M, row_bytes, reps);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it) printf("shape %c: %8.1f us  %6.2f TB/s\n", "ABCD"[shape], ms * 1e3, (double)M * row_bytes * reps / ms / 1e9);
        }
    return 0;
}
