// Diagnostic build of the 32x32x16 fused attention kernel with per-wave phase clocks.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-honor-nans -mno-amdgpu-ieee -fno-slp-vectorize tools/diag_attn32.hip -o tools/diag_bin/diag_attn32
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
    unsigned long long* st; const size_t nst = (size_t)nblk * NW * 16;
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
    const char* names[9] = {"item top", "compute", "wait+bar", "finalize", "total", "tail PV", "fill wait", "vote", "top wait"};
    for (int k = 0; k < 9; ++k) {
        std::vector<double> d;
        for (size_t w = 0; w < (size_t)nblk * NW; ++w) if (h[w * 16 + 4]) d.push_back((double)h[w * 16 + k]);
        std::sort(d.begin(), d.end());
        double mean = 0; for (double v : d) mean += v; mean /= (double)d.size();
        printf("  %-9s mean %8.0f median %8.0f  p10 %8.0f  p90 %8.0f cyc (waves: %zu)\n", names[k], mean, d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10], d.size());
    }
    // where the workgroups ran (wave 0's record): workgroups and mean / max life per XCD, and how many CUs got 1, 2, 3+ of them
    {
        double sum[8] = {0}, mx[8] = {0}; int cnt[8] = {0};
        std::vector<int> per_cu(8 * 64, 0);
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int b = 0; b < (int)(nst / 16 / NW); ++b) {
            const unsigned long long* r = &h[(size_t)b * NW * 16];
            if (!r[4]) continue;
            const unsigned hw = (unsigned)r[9], xcc = (unsigned)r[10] & 15;
            const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            sum[xcc & 7] += (double)r[4]; mx[xcc & 7] = std::max(mx[xcc & 7], (double)r[4]); cnt[xcc & 7]++;
            per_cu[(xcc & 7) * 64 + ((se & 3) * 16 + (sh * 8 + (cu & 7)) % 16) % 64]++;
            if (b < 0) printf("    wg %3d xcc %u se %u sh %u cu %2u life %llu start %llu\n", b, xcc, se, sh, cu, r[4], r[11]);
            tmin = std::min(tmin, r[11]); tmax = std::max(tmax, r[12]);
        }
        for (int x = 0; x < 8; ++x) printf("  xcd %d: %d workgroups, mean life %.0f, longest %.0f\n", x, cnt[x], cnt[x] ? sum[x] / cnt[x] : 0.0, mx[x]);
        {
            const int nwg = (int)(nst / 16 / NW);
            for (int lo = 0; lo < nwg && lo < 512; lo += 32) {
                double m = 0; int c = 0; unsigned long long s0 = ~0ull;
                for (int b = lo; b < lo + 32 && b < nwg; ++b) { const unsigned long long* r = &h[(size_t)b * NW * 16]; if (r[4]) { m += (double)r[4]; ++c; if (((unsigned)r[10] & 7) == 0) s0 = std::min(s0, r[11]); } }
                if (c) printf("    wgs %3d-%3d: mean life %.0f (n=%d) earliest xcd0 start %llu\n", lo, lo + 31, m / c, c, s0);
            }
        }
        int hist[8] = {0};
        for (int v : per_cu) hist[std::min(v, 7)]++;
        printf("  CUs by workgroups held: 0:%d 1:%d 2:%d 3:%d 4+:%d; first start to last end %llu\n", hist[0], hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7], tmax - tmin);
    }
    return 0;
}
