// Diagnostic build of the ring GEMM with per-wave clock stamps: where does a tile's time go?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/diag_gemm.hip -o gpurun_out/diag_gemm
//   diag_gemm M N K shape out_f32 residual [group_m]
// Stamps (shader cycles, per wave): 0 entry, 1 first fragments in registers, 2 main loop done,
// 3 ring free, 4..7 epilogue slabs issued, 8 stores retired; 13/14 = 100 MHz wall clock at end/entry.
#define MARS_STAMPS 1
#include "../mars-multimodal-alignment-and-ranking-system-for-few-shot-segmentation_amd/csrc/gemm3.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 36928, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024;
    const int shape = argc > 4 ? atoi(argv[4]) : 1;   // 0: 256x256, 1: 256x128, 2: 128x128, 3: 64x128, 4: 256x256 with pair fills
    const bool out_f32 = argc > 5 ? atoi(argv[5]) != 0 : true;
    const bool resid = argc > 6 ? atoi(argv[6]) != 0 : true;
    
    
    if (argc > 7) mars_ring_group_m = atoi(argv[7]);
    // 5 / 7: 128x128 with pair fills (eight / four slots), 6: 64x128 with pair fills
    const int BMt = (shape == 2 || shape == 5 || shape == 7) ? 128 : (shape == 3 || shape == 6) ? 64 : 256;
    const int BN = (shape == 0 || shape == 4 || shape == 8) ? 256 : 128, NW = (shape == 0 || shape == 4 || shape == 8) ? 8 : 4;
    std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K);
    unsigned x = 12345u;
    for (auto& v : ha) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((x >> 9) & 0x3ffu) + ((x >> 3) & 0x8000u)); }
    for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3a00u + ((x >> 9) & 0x3ffu) + ((x >> 3) & 0x8000u)); }
    void *A, *B, *C; float* bias;
    CK(hipMalloc(&A, ha.size() * 2)); CK(hipMalloc(&B, hb.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&bias, N * 4));
    CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(C, 0, (size_t)M * N * 4)); CK(hipMemset(bias, 0, N * 4));
    int nblk = ((M + BMt - 1) / BMt) * ((N + BN - 1) / BN);
    const int ntiles = nblk;
    if (shape == 8 && nblk > 256) nblk = 256;            // persistent workgroups: stamps of each workgroup's LAST tile
    unsigned long long* stamps; const size_t nst = (size_t)nblk * 8 * 16;
    CK(hipMalloc(&stamps, nst * 8)); CK(hipMemset(stamps, 0, nst * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(mars_stamp_buf), &stamps, sizeof(stamps)));
    MarsGemm g{};
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.nb0 = 1; g.nb1 = 1;
    g.in_dtype = 1; g.out_dtype = out_f32 ? 0 : 1; g.alpha = 1.f; g.bias = bias; g.act = out_f32 ? 0 : 2;
    if (resid) { g.residual = (const float*)C; g.ldr = N; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) mars_gemm_ring_launch(g, 0, shape, 256);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    mars_gemm_ring_launch(g, 0, shape, 256);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nst);
    CK(hipMemcpy(h.data(), stamps, nst * 8, hipMemcpyDeviceToHost));
    printf("M=%d N=%d K=%d shape=%d f32=%d res=%d: %.1f us, %d blocks, %d tiles\n", M, N, K, shape, out_f32, resid, ms * 1e3, nblk, ntiles);
    const char* names[9] = {"entry", "prologue", "mainloop", "ringfree", "slab0", "slab1", "slab2", "slab3", "retired"};
    for (int i = 1; i <= 8; ++i) {
        std::vector<double> d;
        for (int b = 0; b < nblk; ++b) for (int w = 0; w < NW; ++w) {
            const unsigned long long* s = &h[((size_t)b * 8 + w) * 16];
            d.push_back((double)(s[i] - s[i - 1]));
        }
        std::sort(d.begin(), d.end());
        printf("  %-9s median %8.0f  p10 %8.0f  p90 %8.0f cyc\n", names[i], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
    }
    // wall-clock timeline: block start/end (100 MHz ticks) relative to the first start, and blocks per CU
    unsigned long long t0 = ~0ull, t1 = 0;
    std::map<unsigned long long, int> per_cu;
    std::vector<double> life;
    for (int b = 0; b < nblk; ++b) {
        const unsigned long long* s = &h[((size_t)b * 8) * 16];
        t0 = std::min(t0, s[14]); t1 = std::max(t1, s[13]);
        life.push_back((double)(s[13] - s[14]) / 100.0);
        per_cu[(s[15] >> 32 & 0xf) << 16 | (s[15] & 0xff00)]++;
    }
    std::sort(life.begin(), life.end());
    int mn = 1 << 30, mx = 0; for (auto& kv : per_cu) { mn = std::min(mn, kv.second); mx = std::max(mx, kv.second); }
    printf("  span %.1f us; block life median %.1f us (p10 %.1f, p90 %.1f); %zu CUs seen, blocks per CU %d..%d\n",
           (double)(t1 - t0) / 100.0, life[life.size() / 2], life[life.size() / 10], life[life.size() * 9 / 10], per_cu.size(), mn, mx);
    // starts per 5-us bucket
    std::map<int, int> starts, ends;
    for (int b = 0; b < nblk; ++b) { const unsigned long long* s = &h[((size_t)b * 8) * 16]; starts[(int)((s[14] - t0) / 500)]++; ends[(int)((s[13] - t0) / 500)]++; }
    printf("  starts per 5us:"); for (auto& kv : starts) printf(" %d:%d", kv.first * 5, kv.second); printf("\n");
    printf("  ends   per 5us:"); for (auto& kv : ends) printf(" %d:%d", kv.first * 5, kv.second); printf("\n");
    return 0;
}
