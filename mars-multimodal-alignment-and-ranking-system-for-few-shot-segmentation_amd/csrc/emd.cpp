// Host side of the ranking path: exact earth mover's distance between two uniform
// histograms, the value POT's ot.emd2 returns at
// /root/reference/mars/components/FilteringMergingModule.py:160-166 and
// /root/reference/matcher/Matcher.py:1188-1193.
//
// Algorithm: primal network simplex on the dense transportation graph (n sources,
// m sinks), spanning-tree basis stored as parent pointers with child lists and subtree
// sizes, block pricing by whole rows (a vector minimum per row).  A pivot re-hangs the
// subtree below the leaving arc and moves the potentials of the SMALLER side of the cut by
// the entering arc's reduced cost (sources one way, sinks the other: the same dual solution
// whichever side moves); when no arc prices out, the potentials are rebuilt from the tree
// once and the pricing is repeated, so the answer does not carry the rounding of the shifts.  Degeneracy -- severe for uniform marginals -- is removed with Orden's
// perturbation on INTEGER masses: scale so that each source ships m(n+1)+1 units,
// each sink takes n(n+1) units and the last sink n more.  Every basic solution of
// that instance is strictly positive on its n+m-1 tree arcs, so each pivot moves a
// positive amount and the simplex cannot cycle; an optimal basis of the perturbed
// instance is optimal for the original one.  The optimum is returned through the
// dual, sum_i u_i/n + sum_j v_j/m, which needs no unperturbed flows.
//
// One problem per worker thread, largest problems first; the GPU keeps running the AlphaCLIP tower meanwhile.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <immintrin.h>
#include <limits>
#include <numeric>
#include <thread>
#include <vector>

#include "../../include/mars_hip.h"

void mars_set_error(const char* fmt, ...);

namespace {


// min over j of (double)row[j] - v[j]: the pricing scan.  Written with intrinsics because the compiler does
// not vectorise a conditional minimum without finite-math flags (which this file must not have: the solver
// checks its input for non-finite costs).
static inline double row_min_sse2(const double* row, const double* v, int m) {
    __m128d mn = _mm_set1_pd(std::numeric_limits<double>::infinity());
    int j = 0;
    for (; j + 2 <= m; j += 2) mn = _mm_min_pd(mn, _mm_sub_pd(_mm_loadu_pd(row + j), _mm_loadu_pd(v + j)));
    double a[2]; _mm_storeu_pd(a, mn);
    double r = a[0] < a[1] ? a[0] : a[1];
    for (; j < m; ++j) { const double x = row[j] - v[j]; r = x < r ? x : r; }
    return r;
}
static inline double row_min_sse2(const float* row, const double* v, int m) {
    __m128d mn = _mm_set1_pd(std::numeric_limits<double>::infinity());
    int j = 0;
    for (; j + 4 <= m; j += 4) {
        const __m128 f = _mm_loadu_ps(row + j);
        mn = _mm_min_pd(mn, _mm_sub_pd(_mm_cvtps_pd(f), _mm_loadu_pd(v + j)));
        mn = _mm_min_pd(mn, _mm_sub_pd(_mm_cvtps_pd(_mm_movehl_ps(f, f)), _mm_loadu_pd(v + j + 2)));
    }
    double a[2]; _mm_storeu_pd(a, mn);
    double r = a[0] < a[1] ? a[0] : a[1];
    for (; j < m; ++j) { const double x = (double)row[j] - v[j]; r = x < r ? x : r; }
    return r;
}
__attribute__((target("avx2"))) static double row_min_avx2(const double* row, const double* v, int m) {
    __m256d mn = _mm256_set1_pd(std::numeric_limits<double>::infinity());
    int j = 0;
    for (; j + 4 <= m; j += 4) mn = _mm256_min_pd(mn, _mm256_sub_pd(_mm256_loadu_pd(row + j), _mm256_loadu_pd(v + j)));
    double a[4]; _mm256_storeu_pd(a, mn);
    double r = a[0];
    for (int k = 1; k < 4; ++k) r = a[k] < r ? a[k] : r;
    for (; j < m; ++j) { const double x = row[j] - v[j]; r = x < r ? x : r; }
    return r;
}
__attribute__((target("avx2"))) static double row_min_avx2(const float* row, const double* v, int m) {
    __m256d mn = _mm256_set1_pd(std::numeric_limits<double>::infinity());
    int j = 0;
    for (; j + 4 <= m; j += 4) mn = _mm256_min_pd(mn, _mm256_sub_pd(_mm256_cvtps_pd(_mm_loadu_ps(row + j)), _mm256_loadu_pd(v + j)));
    double a[4]; _mm256_storeu_pd(a, mn);
    double r = a[0];
    for (int k = 1; k < 4; ++k) r = a[k] < r ? a[k] : r;
    for (; j < m; ++j) { const double x = (double)row[j] - v[j]; r = x < r ? x : r; }
    return r;
}
// MARS_EMD_NO_AVX2 in the environment keeps the SSE2 loop (so that both can be tested on one machine)
static const bool g_avx2 = __builtin_cpu_supports("avx2") && !getenv("MARS_EMD_NO_AVX2");
template <typename CT> static inline double row_min(const CT* row, const double* v, int m) {
    return g_avx2 ? row_min_avx2(row, v, m) : row_min_sse2(row, v, m);
}

template <typename CT>
class TransportSimplex {
public:
    TransportSimplex(const CT* C, int n, int m) : C_(C), n_(n), m_(m), N_(n + m) {}

    double solve() {
        const double nan = std::numeric_limits<double>::quiet_NaN();
        if (n_ <= 0 || m_ <= 0) return nan;
        for (int64_t e = 0; e < (int64_t)n_ * m_; ++e)
            if (!std::isfinite((double)C_[e])) return nan;
        init_basis();
        const int64_t narcs = (int64_t)n_ * m_;
        int64_t block = (int64_t)std::sqrt((double)narcs);
        if (block < 16) block = 16;
        const int rows_per_block = (int)std::max<int64_t>(1, (block + m_ - 1) / m_);
        int next_row = 0;
        const double eps = 1e-11;
        const int64_t max_pivots = 64 * (int64_t)N_ * N_ + 100000;
        const double* v = pot_.data() + n_;
        bool verified = false;
        for (int64_t pivots = 0;; ++pivots) {
            if (pivots > max_pivots || broken_) return nan;
            // block pricing by whole rows: the most negative reduced cost of the first block of rows that has one
            int bi = -1;
            double best_rc = -eps;
            int scanned_rows = 0, i = next_row;
            while (scanned_rows < n_) {
                const int cnt = std::min(rows_per_block, n_ - scanned_rows);
                for (int k = 0; k < cnt; ++k) {
                    const CT* row = C_ + (int64_t)i * m_;
                    const double rc = row_min(row, v, m_) - pot_[i];
                    if (rc < best_rc) { best_rc = rc; bi = i; }
                    if (++i == n_) i = 0;
                }
                scanned_rows += cnt;
                if (bi >= 0) break;
            }
            if (bi < 0) {
                if (verified) break;
                // dual feasible on the running potentials: rebuild them from the tree (the shifts of the pivots
                // have rounded thousands of times) and price once more
                refresh_potentials();
                verified = true;
                continue;
            }
            verified = false;
            next_row = i;
            // the column of the minimum in row bi
            const CT* row = C_ + (int64_t)bi * m_;
            int bj = 0;
            double mn = std::numeric_limits<double>::infinity();
            for (int j = 0; j < m_; ++j) {
                const double x = (double)row[j] - v[j];
                if (x < mn) { mn = x; bj = j; }
            }
            pivot(bi, bj, mn - pot_[bi]);
        }
        double su = 0.0, sv = 0.0;
        for (int i = 0; i < n_; ++i) su += pot_[i];
        for (int j = 0; j < m_; ++j) sv += pot_[n_ + j];
        return su / n_ + sv / m_;
    }

private:
    const CT* C_;
    int n_, m_, N_;
    std::vector<int> parent_, first_child_, next_sib_, prev_sib_, size_, stamp_;
    std::vector<int64_t> flow_;
    std::vector<double> pot_;
    int stamp_now_ = 0;

    double arc_cost(int a, int b) const {
        return a < n_ ? (double)C_[(int64_t)a * m_ + (b - n_)] : (double)C_[(int64_t)b * m_ + (a - n_)];
    }
    void add_child(int p, int c) {
        prev_sib_[c] = -1;
        next_sib_[c] = first_child_[p];
        if (first_child_[p] >= 0) prev_sib_[first_child_[p]] = c;
        first_child_[p] = c;
    }
    void remove_child(int p, int c) {
        if (prev_sib_[c] >= 0) next_sib_[prev_sib_[c]] = next_sib_[c];
        else first_child_[p] = next_sib_[c];
        if (next_sib_[c] >= 0) prev_sib_[next_sib_[c]] = prev_sib_[c];
        prev_sib_[c] = next_sib_[c] = -1;
    }
    void refresh_potentials() {
        stack_.clear();
        pot_[0] = 0.0;
        for (int c = first_child_[0]; c >= 0; c = next_sib_[c]) stack_.push_back(c);
        while (!stack_.empty()) {
            const int node = stack_.back(); stack_.pop_back();
            const int par = parent_[node];
            pot_[node] = arc_cost(par, node) - pot_[par];
            for (int c = first_child_[node]; c >= 0; c = next_sib_[c]) stack_.push_back(c);
        }
    }

    void init_basis() {
        parent_.assign(N_, -1); size_.assign(N_, 1); stamp_.assign(N_, 0);
        first_child_.assign(N_, -1); next_sib_.assign(N_, -1); prev_sib_.assign(N_, -1);
        flow_.assign(N_, 0); pot_.assign(N_, 0.0);
        std::vector<int64_t> demand(m_, (int64_t)n_ * (n_ + 1));
        demand[m_ - 1] += n_;
        struct Arc { int i, j; int64_t f; };
        std::vector<Arc> arcs;
        arcs.reserve(N_);
        std::vector<char> open(m_, 1);
        for (int i = 0; i < n_; ++i) {
            int64_t supply = (int64_t)m_ * (n_ + 1) + 1;
            const CT* row = C_ + (int64_t)i * m_;
            while (supply > 0) {
                int bj = -1; double bc = std::numeric_limits<double>::infinity();
                for (int j = 0; j < m_; ++j) if (open[j] && (double)row[j] < bc) { bc = (double)row[j]; bj = j; }
                const int64_t x = std::min(supply, demand[bj]);
                arcs.push_back({i, bj, x});
                supply -= x; demand[bj] -= x;
                if (demand[bj] == 0) open[bj] = 0;
            }
        }
        std::vector<std::vector<std::pair<int, int64_t>>> adj(N_);
        for (const Arc& a : arcs) { adj[a.i].push_back({n_ + a.j, a.f}); adj[n_ + a.j].push_back({a.i, a.f}); }
        std::vector<int> queue; queue.reserve(N_);
        std::vector<char> seen(N_, 0);
        queue.push_back(0); seen[0] = 1;
        for (size_t h = 0; h < queue.size(); ++h) {
            const int x = queue[h];
            for (auto& pr : adj[x]) {
                const int y = pr.first;
                if (seen[y]) continue;
                seen[y] = 1;
                parent_[y] = x; flow_[y] = pr.second;
                pot_[y] = arc_cost(x, y) - pot_[x];
                add_child(x, y);
                queue.push_back(y);
            }
        }
        for (size_t h = queue.size(); h-- > 1;) size_[parent_[queue[h]]] += size_[queue[h]];   // BFS order: children after parents
    }

    void pivot(int i, int j, double rc) {
        const int a = i, b = n_ + j;
        // apex of the cycle: mark the ancestors of a, climb from b to the first marked node
        ++stamp_now_;
        for (int x = a; x >= 0; x = parent_[x]) stamp_[x] = stamp_now_;
        int apex = b;
        while (stamp_[apex] != stamp_now_) { apex = parent_[apex]; if (apex < 0) { broken_ = true; return; } }
        // leaving arc: smallest flow among the arcs whose flow decreases around the cycle (the perturbed masses
        // make it unique); on the a side those are the arcs below sources, on the b side below sinks
        int leave = -1; bool leave_on_a = false;
        int64_t theta = std::numeric_limits<int64_t>::max();
        for (int x = a; x != apex; x = parent_[x]) if (x < n_ && flow_[x] < theta) { theta = flow_[x]; leave = x; leave_on_a = true; }
        for (int y = b; y != apex; y = parent_[y]) if (y >= n_ && flow_[y] < theta) { theta = flow_[y]; leave = y; leave_on_a = false; }
        if (leave < 0) { broken_ = true; return; }
        for (int x = a; x != apex; x = parent_[x]) flow_[x] += (x < n_) ? -theta : theta;
        for (int y = b; y != apex; y = parent_[y]) flow_[y] += (y >= n_) ? -theta : theta;
        const int q = leave_on_a ? a : b, p = leave_on_a ? b : a;
        path_.clear();
        for (int x = q; x != leave; x = parent_[x]) path_.push_back(x);
        path_.push_back(leave);
        const int moved = size_[leave];
        for (int x = parent_[leave]; x >= 0; x = parent_[x]) size_[x] -= moved;
        remove_child(parent_[leave], leave);
        // reverse the parent pointers along q .. leave; subtree sizes along the path follow
        old_size_.resize(path_.size());
        for (size_t t = 0; t < path_.size(); ++t) old_size_[t] = size_[path_[t]];
        int64_t carry = theta;
        int newpar = p;
        for (size_t t = 0; t < path_.size(); ++t) {
            const int node = path_[t];
            const int oldpar = parent_[node];
            const int64_t oldflow = flow_[node];
            if (t + 1 < path_.size()) remove_child(oldpar, node);
            parent_[node] = newpar;
            flow_[node] = carry;
            add_child(newpar, node);
            carry = oldflow;
            newpar = node;
        }
        {
            int below = 0;      // new size of the next node down the new chain
            for (size_t t = path_.size(); t-- > 0;) {
                const int own = old_size_[t] - (t > 0 ? old_size_[t - 1] : 0);
                size_[path_[t]] = own + below;
                below = size_[path_[t]];
            }
        }
        for (int x = p; x >= 0; x = parent_[x]) size_[x] += moved;
        // potentials: the entering arc becomes tight.  Inside the moved subtree every source would move by d
        // and every sink by -d (d = +rc when its top is a source); moving the REST of the tree the other way is
        // the same dual solution, so the smaller side is walked.
        const double d = (q < n_) ? rc : -rc;
        if (2 * moved <= N_) {
            stack_.clear();
            stack_.push_back(q);
            while (!stack_.empty()) {
                const int node = stack_.back(); stack_.pop_back();
                pot_[node] += node < n_ ? d : -d;
                for (int c = first_child_[node]; c >= 0; c = next_sib_[c]) stack_.push_back(c);
            }
        } else {
            stack_.clear();
            stack_.push_back(0);
            while (!stack_.empty()) {
                const int node = stack_.back(); stack_.pop_back();
                pot_[node] += node < n_ ? -d : d;
                for (int c = first_child_[node]; c >= 0; c = next_sib_[c]) if (c != q) stack_.push_back(c);
            }
        }
    }

    std::vector<int> path_, stack_, old_size_;
    bool broken_ = false;
};

}  // namespace

extern "C" double marshost_emd2_uniform(const double* C, int n, int m) {
    TransportSimplex<double> s(C, n, m);
    return s.solve();
}

extern "C" int marshost_emd_scores(const float* S, int64_t ldS, int R, int P, const uint8_t* fg, const uint8_t* bits,
                                   int N, int is_cost, double* out, int threads) {
    if (!S || !fg || !bits || !out || R <= 0 || P <= 0 || N <= 0 || ldS < P) {
        mars_set_error("emd_scores: bad arguments");
        return -1;
    }
    std::vector<int> rows;
    for (int r = 0; r < R; ++r) if (fg[r]) rows.push_back(r);
    const int n = (int)rows.size();
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    threads = std::max(1, std::min(threads, N));
    // largest problems first (n is common, so by the number of sinks): the last ones to finish are small
    std::vector<int> msize(N, 0), order(N);
    for (int k = 0; k < N; ++k)
        for (int j = 0; j < P; ++j) msize[k] += bits[(int64_t)k * P + j] != 0;
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return msize[a] > msize[b]; });
    std::atomic<int> next(0);
    auto work = [&]() {
        std::vector<int> cols;
        std::vector<float> M;          // the costs as the reference stores them (float32); the solver widens on the fly
        for (;;) {
            const int t = next.fetch_add(1);
            if (t >= N) break;
            const int k = order[t];
            cols.clear();
            for (int j = 0; j < P; ++j) if (bits[(int64_t)k * P + j]) cols.push_back(j);
            const int m = (int)cols.size();
            if (n == 0 || m == 0) { out[k] = std::numeric_limits<double>::quiet_NaN(); continue; }
            M.resize((size_t)n * m);
            for (int a = 0; a < n; ++a) {
                const float* sr = S + (int64_t)rows[a] * ldS;
                for (int b = 0; b < m; ++b) {
                    // cost_matrix = (1 - S) / 2 in float32, as the reference stores it
                    const float c = is_cost ? sr[cols[b]] : (1.0f - sr[cols[b]]) / 2.0f;
                    M[(size_t)a * m + b] = c;
                }
            }
            TransportSimplex<float> s(M.data(), n, m);
            out[k] = 1.0 - s.solve();
        }
    };
    if (threads == 1) { work(); return 0; }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(work);
    for (auto& t : pool) t.join();
    return 0;
}
