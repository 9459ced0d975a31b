// Host routine of the prompt-sampling stage: the rectangular linear assignment problem that the reference
// solves with scipy.optimize.linear_sum_assignment (/root/reference/matcher/Matcher.py:449-450, 471-472,
// 364-365, 376).  Shortest augmenting paths with dual potentials in double precision; for a matrix with
// more rows than columns the transposed problem is solved.  Like scipy it returns min(n, m) pairs ordered
// by row index.  With distinct optima (continuous similarities) the assignment is the unique optimum, so
// it equals scipy's.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "../../include/mars_hip.h"

void mars_set_error(const char* fmt, ...);

namespace {

// rows <= cols; cost(r, c) is read through `at`; assigned[r] = column of row r
template <typename At>
bool solve_rows_le_cols(int n, int m, At at, std::vector<int>& assigned) {
    const double INF = std::numeric_limits<double>::infinity();
    std::vector<double> u(n + 1, 0.0), v(m + 1, 0.0), minv(m + 1);
    std::vector<int> p(m + 1, 0), way(m + 1, 0);
    std::vector<char> used(m + 1);
    for (int i = 1; i <= n; ++i) {
        p[0] = i;
        int j0 = 0;
        std::fill(minv.begin(), minv.end(), INF);
        std::fill(used.begin(), used.end(), 0);
        do {
            used[j0] = 1;
            const int i0 = p[j0];
            double delta = INF;
            int j1 = -1;
            for (int j = 1; j <= m; ++j) {
                if (used[j]) continue;
                const double cur = at(i0 - 1, j - 1) - u[i0] - v[j];
                if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                if (minv[j] < delta) { delta = minv[j]; j1 = j; }
            }
            if (j1 < 0 || !std::isfinite(delta)) return false;       // non-finite costs
            for (int j = 0; j <= m; ++j) {
                if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
                else minv[j] -= delta;
            }
            j0 = j1;
        } while (p[j0] != 0);
        do {
            const int j1 = way[j0];
            p[j0] = p[j1];
            j0 = j1;
        } while (j0);
    }
    assigned.assign(n, -1);
    for (int j = 1; j <= m; ++j)
        if (p[j]) assigned[p[j] - 1] = j - 1;
    return true;
}

}  // namespace

extern "C" int marshost_lap(const float* w, int64_t ld, int n, int m, int maximize, int64_t* row_ind, int64_t* col_ind) {
    if (!w || !row_ind || !col_ind || n <= 0 || m <= 0 || ld < m) {
        mars_set_error("lap: bad arguments (n=%d m=%d ld=%lld)", n, m, (long long)ld);
        return -1;
    }
    const double sign = maximize ? -1.0 : 1.0;
    std::vector<int> assigned;
    bool ok;
    if (n <= m) {
        ok = solve_rows_le_cols(n, m, [&](int r, int c) { return sign * (double)w[(int64_t)r * ld + c]; }, assigned);
        if (ok)
            for (int r = 0; r < n; ++r) { row_ind[r] = r; col_ind[r] = assigned[r]; }
    } else {
        ok = solve_rows_le_cols(m, n, [&](int r, int c) { return sign * (double)w[(int64_t)c * ld + r]; }, assigned);
        if (ok) {
            // assigned[c] = row matched to column c; order the pairs by row
            std::vector<int> cols(m);
            for (int c = 0; c < m; ++c) cols[c] = c;
            std::sort(cols.begin(), cols.end(), [&](int a, int b) { return assigned[a] < assigned[b]; });
            for (int k = 0; k < m; ++k) { row_ind[k] = assigned[cols[k]]; col_ind[k] = cols[k]; }
        }
    }
    if (!ok) {
        mars_set_error("lap: the cost matrix holds non-finite entries");
        return -2;
    }
    return n <= m ? n : m;
}
