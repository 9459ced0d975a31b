/* Exact earth mover's distance between two uniform histograms (oracle, test
 * infrastructure only).
 *
 * Reference call sites: /root/reference/mars/components/FilteringMergingModule.py:160-166
 * and /root/reference/matcher/Matcher.py:1188-1193 call POT 0.9.4 `ot.emd2(a, b, M)` with
 * a = 1/n, b = 1/m.  POT is not installed here; `emd2` returns the optimum of the
 * transportation linear programme
 *     min sum_ij g_ij M_ij   s.t.  sum_j g_ij = 1/n,  sum_i g_ij = 1/m,  g >= 0,
 * and that optimum is unique whatever exact algorithm reaches it.  This file solves
 * the same programme by successive shortest augmenting paths with node potentials on
 * integer masses (each source ships m units, each sink takes n units, result divided
 * by n*m), which is exact up to floating-point summation of the costs.  It is pinned
 * against scipy's HiGHS LP solver in tests/test_oracle_emd.py.
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/libemd_oracle.so oracle/emd_ssp.c
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

double emd_uniform_ssp(const double *C, int n, int m) {
    if (n <= 0 || m <= 0) return 0.0;
    const int N = n + m;
    long long *excess = (long long *)malloc(sizeof(long long) * N); /* >0 source excess, <0 sink deficit */
    double *pot = (double *)calloc(N, sizeof(double));
    double *dist = (double *)malloc(sizeof(double) * N);
    int *prev = (int *)malloc(sizeof(int) * N);
    char *done = (char *)malloc(N);
    long long *flow = (long long *)calloc((size_t)n * m, sizeof(long long));
    for (int i = 0; i < n; ++i) excess[i] = m;
    for (int j = 0; j < m; ++j) excess[n + j] = -(long long)n;
    /* feasible duals: u_i = 0, v_j = min_i C_ij  => reduced cost C_ij - u_i - v_j >= 0 */
    for (int j = 0; j < m; ++j) {
        double mn = C[j];
        for (int i = 1; i < n; ++i) if (C[(size_t)i * m + j] < mn) mn = C[(size_t)i * m + j];
        pot[n + j] = mn;
    }
    long long remaining = (long long)n * m;
    while (remaining > 0) {
        for (int k = 0; k < N; ++k) { dist[k] = INFINITY; prev[k] = -1; done[k] = 0; }
        for (int i = 0; i < n; ++i) if (excess[i] > 0) dist[i] = 0.0;
        int target = -1;
        for (;;) {
            int k = -1; double best = INFINITY;
            for (int q = 0; q < N; ++q) if (!done[q] && dist[q] < best) { best = dist[q]; k = q; }
            if (k < 0) break;
            done[k] = 1;
            if (k >= n && excess[k] < 0) { target = k; break; }
            if (k < n) {
                const double *row = C + (size_t)k * m;
                for (int j = 0; j < m; ++j) {
                    if (done[n + j]) continue;
                    double rc = row[j] - pot[k] - pot[n + j];
                    if (rc < 0) rc = 0;
                    double nd = best + rc;
                    if (nd < dist[n + j]) { dist[n + j] = nd; prev[n + j] = k; }
                }
            } else {
                int j = k - n;
                for (int i = 0; i < n; ++i) {
                    if (done[i] || flow[(size_t)i * m + j] <= 0) continue;
                    double rc = -(C[(size_t)i * m + j] - pot[i] - pot[k]);
                    if (rc < 0) rc = 0;
                    double nd = best + rc;
                    if (nd < dist[i]) { dist[i] = nd; prev[i] = k; }
                }
            }
        }
        if (target < 0) break; /* cannot happen for a balanced problem */
        double dt = dist[target];
        for (int k = 0; k < N; ++k) {
            if (!done[k] || k == target) continue;
            /* keep reduced costs non-negative and zero along the shortest-path tree */
            if (k < n) pot[k] += dt - dist[k]; else pot[k] -= dt - dist[k];
        }
        /* bottleneck */
        long long amt = -excess[target];
        int k = target;
        while (prev[k] >= 0) {
            int p = prev[k];
            if (k < n) { /* reverse arc sink p -> source k */
                long long f = flow[(size_t)k * m + (p - n)];
                if (f < amt) amt = f;
            }
            k = p;
        }
        if (excess[k] < amt) amt = excess[k];
        int root = k;
        k = target;
        while (prev[k] >= 0) {
            int p = prev[k];
            if (k >= n) flow[(size_t)p * m + (k - n)] += amt;
            else flow[(size_t)k * m + (p - n)] -= amt;
            k = p;
        }
        excess[root] -= amt;
        excess[target] += amt;
        remaining -= amt;
    }
    long double total = 0.0L;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j)
            if (flow[(size_t)i * m + j]) total += (long double)flow[(size_t)i * m + j] * C[(size_t)i * m + j];
    free(excess); free(pot); free(dist); free(prev); free(done); free(flow);
    return (double)(total / ((long double)n * (long double)m));
}
