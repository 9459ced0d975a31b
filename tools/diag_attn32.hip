// Diagnostic build of the 32x32x16 fused attention kernel with per-wave phase clocks.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-honor-nans -mno-amdgpu-ieee tools/diag_attn32.hip -o tools/diag_bin/diag_attn32
//   diag_attn32 B T H
#define MARS_STAMPS 1
#include "../mars-multimodal-alignment-and-ranking-system-for-few-shot-segmentation_amd/csrc/attention32.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

void mars_set_error(const char* fmt, ...) { fprintf(stderr, "error: %s\n", fmt); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 577, H = argc > 3 ? atoi(argv[3]) : 16;
    const size_t n = (size_t)B * T * 3 * H * 64;
    std::vector<unsigned short> hq(n);
    unsigned x = 99u;
    for (auto& v : hq) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((x >> 9) & 0x3ffu) + ((x >> 3) & 0x8000u)); }
    void *qkv, *O; float* lse;
    CK(hipMalloc(&qkv, n * 2)); CK(hipMalloc(&O, n * 2 / 3)); CK(hipMalloc(&lse, (size_t)B * H * T * 4));
    CK(hipMemcpy(qkv, hq.data(), n * 2, hipMemcpyHostToDevice));
    const int nblk = ((T + 127) / 128) * B * H, NW = 4;   // (stamps: one record per launched workgroup and wave; the buffer is sized for one workgroup per item)
    unsigned long long* st; const size_t nst = (size_t)nblk * NW * 8;
    CK(hipMalloc(&st, nst * 8)); CK(hipMemset(st, 0, nst * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(mars_attn32_stamps), &st, sizeof(st)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) mars_flash_attn32_launch(qkv, O, lse, B, T, H, 0.18033688f, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    mars_flash_attn32_launch(qkv, O, lse, B, T, H, 0.18033688f, nullptr);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nst);
    CK(hipMemcpy(h.data(), st, nst * 8, hipMemcpyDeviceToHost));
    printf("B=%d T=%d H=%d: %.1f us, %d blocks\n", B, T, H, ms * 1e3, nblk);
    const char* names[7] = {"item top", "compute", "wait+bar", "finalize", "total", "", "fill wait"};
    for (int k = 0; k < 7; ++k) {
        if (k == 5) continue;
        std::vector<double> d;
        for (size_t w = 0; w < (size_t)nblk * NW; ++w) if (h[w * 8 + 5]) d.push_back((double)h[w * 8 + k]);
        std::sort(d.begin(), d.end());
        printf("  %-9s median %8.0f  p10 %8.0f  p90 %8.0f cyc (active waves: %zu)\n", names[k], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10], d.size());
    }
    return 0;
}
