// L2 -> LDS delivery rate of the ring GEMM's LDS-DMA fill, by request shape.  One 512-thread workgroup per
// CU streams the K extent of a 512-row panel (256 rows of A, 256 of B; all workgroups of an XCD share a few
// panels, so the rows are L2-resident) into an LDS ring, nothing else in the loop.
//   mode 0: one wave instruction = 16 rows x 64 B (half a 128-B line per row), K walked in 64-B steps
//   mode 1: one wave instruction = 8 rows x 128 B (whole lines), K walked in 128-B steps
//   mode 2: as mode 0, but the two halves of a line are requested back to back (steps s and s+1 paired)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/diag_dma.hip -o tools/_build/diag_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__device__ __forceinline__ void glds16(const unsigned char* src, unsigned char* lds_dst) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_dst, 16, 0, 0);
}

constexpr int ROWS = 512;            // rows per workgroup panel pair
constexpr int SLOT = ROWS * 64;      // bytes of one 64-B step
constexpr int NSLOT = 4;

template <int MODE>
__global__ __launch_bounds__(512, 2) void dma_kernel(const unsigned char* A, int npanel, int64_t row_bytes, int reps, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[NSLOT * SLOT];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int panel = (blockIdx.x >> 3) % npanel;     // blocks of one XCD (same blockIdx % 8) walk the same few panels
    const unsigned char* base = A + (int64_t)panel * ROWS * row_bytes;
    const int nstep = (int)(row_bytes / 64);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if constexpr (MODE == 0 || MODE == 2) {
            const int lr = lane >> 2, ch = lane & 3;
            // 32 pieces of 16 rows per step, 4 per wave
            for (int s = 0; s < nstep; s += 2) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = (p * 8 + wave) * 16 + lr;
                    const unsigned char* src = base + row * row_bytes + s * 64 + ch * 16;
                    unsigned char* dst = lds + (s & 3) * SLOT + (p * 8 + wave) * 1024;
                    if constexpr (MODE == 2) {
                        glds16(src, dst);
                        glds16(src + 64, dst + SLOT);
                    } else {
                        glds16(src, dst);
                    }
                }
                if constexpr (MODE == 0) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int row = (p * 8 + wave) * 16 + lr;
                        const unsigned char* src = base + row * row_bytes + (s + 1) * 64 + ch * 16;
                        unsigned char* dst = lds + ((s + 1) & 3) * SLOT + (p * 8 + wave) * 1024;
                        glds16(src, dst);
                    }
                }
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // one pair of steps stays in flight
            }
        } else {
            const int lr = lane >> 3, ch = lane & 7;
            // 64 pieces of 8 rows per 128-B step, 8 per wave
            for (int s = 0; s < nstep; s += 2) {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int row = (p * 8 + wave) * 8 + lr;
                    const unsigned char* src = base + row * row_bytes + s * 64 + ch * 16;
                    unsigned char* dst = lds + ((s >> 1) & 1) * 2 * SLOT + (p * 8 + wave) * 1024;
                    glds16(src, dst);
                }
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    // keep the ring alive
    if (lds[threadIdx.x] == 0x7f && reps < 0) cyc[0] = 1;
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 1024;            // bf16 elements per row
    const int npanel = argc > 2 ? atoi(argv[2]) : 4;
    const int reps = argc > 3 ? atoi(argv[3]) : 32;
    const int nblk = argc > 4 ? atoi(argv[4]) : 256;
    const int64_t row_bytes = (int64_t)K * 2;
    const size_t bytes = (size_t)npanel * ROWS * row_bytes;
    unsigned char* A; unsigned long long* cyc;
    CK(hipMalloc(&A, bytes)); CK(hipMemset(A, 1, bytes)); CK(hipMalloc(&cyc, nblk * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        for (int it = 0; it < 3; ++it) {
            CK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(dma_kernel<0>, dim3(nblk), dim3(512), 0, 0, A, npanel, row_bytes, reps, cyc);
            else if (mode == 1) hipLaunchKernelGGL(dma_kernel<1>, dim3(nblk), dim3(512), 0, 0, A, npanel, row_bytes, reps, cyc);
            else hipLaunchKernelGGL(dma_kernel<2>, dim3(nblk), dim3(512), 0, 0, A, npanel, row_bytes, reps, cyc);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(nblk);
            CK(hipMemcpy(h.data(), cyc, nblk * 8, hipMemcpyDeviceToHost));
            double sum = 0; for (auto v : h) sum += (double)v;
            const double per_blk = (double)ROWS * row_bytes * reps;
            // s_memtime ticks at 100 MHz: report bytes per microsecond per CU and chip-wide TB/s from the event time
            printf("mode %d K=%d panels=%d blocks=%d: %.1f us, %.2f TB/s chip, %.1f GB/s per block (ticks avg %.0f)\n", mode, K, npanel, nblk,
                   ms * 1e3, per_blk * nblk / (ms * 1e-3) / 1e12, per_blk / (ms * 1e-3) / 1e9, sum / nblk);
        }
    }
    return 0;
}
