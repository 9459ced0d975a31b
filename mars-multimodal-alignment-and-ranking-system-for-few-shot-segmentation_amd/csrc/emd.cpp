// Host side of the ranking path: exact earth mover's distance between two uniform
// histograms, the value POT's ot.emd2 returns at
// /root/reference/mars/components/FilteringMergingModule.py:160-166 and
// /root/reference/matcher/Matcher.py:1188-1193.
//
// Algorithm: primal network simplex on the dense transportation graph.  The LARGER of the two
// sides supplies ("rows", nr of them), the smaller one takes ("columns", nc): the distance is
// symmetric, and a pricing scan runs along the short side with the potentials of all columns in a
// contiguous vector.  The basis is a spanning tree stored as parent pointers, subtree sizes and a
// THREAD: the nodes in pre-order as a doubly linked list, with the last node of every subtree, so that
// a subtree is one run of the list (walked without a stack, cut out and put back in with a few links).
// With nr > nc most rows are LEAVES of the tree (nr + nc - 1 arcs, every row has one): a row without
// children is kept out of the thread and carries no potential of its own - it is the cost of its one
// arc minus its column's potential, formed when the row is priced - so a pivot walks only the columns
// and the few rows that branch.  Block pricing by whole rows (a vector minimum per row; the winning
// column is found by a second pass over that one row); for float costs the scan that SELECTS entering
// arcs runs in single precision, the pivots and the final optimality scan in double.  The cycle of an entering arc
// is closed by climbing from both ends, the end with the smaller subtree first (an ancestor's subtree is
// strictly larger), which visits the cycle's nodes and nothing above its apex.  A pivot re-hangs the
// subtree below the leaving arc and moves the potentials of the SMALLER side of the cut by the entering
// arc's reduced cost (rows one way, columns the other: the same dual solution whichever side moves);
// when no arc prices out, the potentials are rebuilt from the tree once and the pricing is repeated,
// so the answer does not carry the rounding of the shifts.
//
// Degeneracy - severe for uniform marginals - is removed with Orden's perturbation on INTEGER
// masses: each row ships nc(nr+1)+1 units, each column takes nr(nr+1) units and the last column nr
// more.  Every basic solution of that instance is strictly positive on its nr+nc-1 tree arcs, so each
// pivot moves a positive amount and the simplex cannot cycle; an optimal basis of the perturbed
// instance is optimal for the original one.  The optimum is returned through the dual,
// sum_i u_i/nr + sum_j v_j/nc, which needs no unperturbed flows.
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
#include <type_traits>
#include <vector>

#include "../../include/mars_hip.h"

void mars_set_error(const char* fmt, ...);

namespace {

// min over j of (double)row[j] - v[j]: the pricing scan.  Written with intrinsics because the compiler does
// not vectorise a conditional minimum without finite-math flags (which this file must not have: the solver
// checks its input for non-finite costs).
static inline double hmin2(__m128d mn) {
    double a[2]; _mm_storeu_pd(a, mn);
    return a[0] < a[1] ? a[0] : a[1];
}
static inline double row_min_sse2(const double* row, const double* v, int m) {
    __m128d mn = _mm_set1_pd(std::numeric_limits<double>::infinity());
    int j = 0;
    for (; j + 2 <= m; j += 2) mn = _mm_min_pd(mn, _mm_sub_pd(_mm_loadu_pd(row + j), _mm_loadu_pd(v + j)));
    double r = hmin2(mn);
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
    double r = hmin2(mn);
    for (; j < m; ++j) { const double x = (double)row[j] - v[j]; r = x < r ? x : r; }
    return r;
}
__attribute__((target("avx2"))) static double row_min_avx2(const double* row, const double* v, int m) {
    __m256d mn = _mm256_set1_pd(std::numeric_limits<double>::infinity());
    int j = 0;
    for (; j + 4 <= m; j += 4) mn = _mm256_min_pd(mn, _mm256_sub_pd(_mm256_loadu_pd(row + j), _mm256_loadu_pd(v + j)));
    double r = hmin2(_mm_min_pd(_mm256_castpd256_pd128(mn), _mm256_extractf128_pd(mn, 1)));
    for (; j < m; ++j) { const double x = row[j] - v[j]; r = x < r ? x : r; }
    return r;
}
__attribute__((target("avx2"))) static double row_min_avx2(const float* row, const double* v, int m) {
    const __m256d inf = _mm256_set1_pd(std::numeric_limits<double>::infinity());
    __m256d m0 = inf, m1 = inf;
    int j = 0;
    for (; j + 8 <= m; j += 8) {
        m0 = _mm256_min_pd(m0, _mm256_sub_pd(_mm256_cvtps_pd(_mm_loadu_ps(row + j)), _mm256_loadu_pd(v + j)));
        m1 = _mm256_min_pd(m1, _mm256_sub_pd(_mm256_cvtps_pd(_mm_loadu_ps(row + j + 4)), _mm256_loadu_pd(v + j + 4)));
    }
    for (; j + 4 <= m; j += 4) m0 = _mm256_min_pd(m0, _mm256_sub_pd(_mm256_cvtps_pd(_mm_loadu_ps(row + j)), _mm256_loadu_pd(v + j)));
    m0 = _mm256_min_pd(m0, m1);
    double r = hmin2(_mm_min_pd(_mm256_castpd256_pd128(m0), _mm256_extractf128_pd(m0, 1)));
    for (; j < m; ++j) { const double x = (double)row[j] - v[j]; r = x < r ? x : r; }
    return r;
}
// The same scan in single precision (costs and a float copy of the column potentials): eight lanes, no widening.
// It only SELECTS entering arcs - an arc whose float reduced cost is below -1e-6 is negative in exact arithmetic
// too (the float error is below 5e-7 for costs and potentials of magnitude <= 2), and the pivot is carried out
// with the exact value - so the answer does not depend on it; optimality is always established by the double scan.
__attribute__((target("avx"))) static float row_min_f32_avx(const float* row, const float* vf, int m) {
    const __m256 inf = _mm256_set1_ps(std::numeric_limits<float>::infinity());
    __m256 m0 = inf, m1 = inf;
    int j = 0;
    for (; j + 16 <= m; j += 16) {
        m0 = _mm256_min_ps(m0, _mm256_sub_ps(_mm256_loadu_ps(row + j), _mm256_loadu_ps(vf + j)));
        m1 = _mm256_min_ps(m1, _mm256_sub_ps(_mm256_loadu_ps(row + j + 8), _mm256_loadu_ps(vf + j + 8)));
    }
    for (; j + 8 <= m; j += 8) m0 = _mm256_min_ps(m0, _mm256_sub_ps(_mm256_loadu_ps(row + j), _mm256_loadu_ps(vf + j)));
    m0 = _mm256_min_ps(m0, m1);
    __m128 q = _mm_min_ps(_mm256_castps256_ps128(m0), _mm256_extractf128_ps(m0, 1));
    q = _mm_min_ps(q, _mm_movehl_ps(q, q));
    q = _mm_min_ss(q, _mm_shuffle_ps(q, q, 1));
    float r = _mm_cvtss_f32(q);
    for (; j < m; ++j) { const float x = row[j] - vf[j]; r = x < r ? x : r; }
    return r;
}
static inline float row_min_f32_sse(const float* row, const float* vf, int m) {
    __m128 q = _mm_set1_ps(std::numeric_limits<float>::infinity());
    int j = 0;
    for (; j + 4 <= m; j += 4) q = _mm_min_ps(q, _mm_sub_ps(_mm_loadu_ps(row + j), _mm_loadu_ps(vf + j)));
    q = _mm_min_ps(q, _mm_movehl_ps(q, q));
    q = _mm_min_ss(q, _mm_shuffle_ps(q, q, 1));
    float r = _mm_cvtss_f32(q);
    for (; j < m; ++j) { const float x = row[j] - vf[j]; r = x < r ? x : r; }
    return r;
}
// MARS_EMD_NO_AVX2 in the environment keeps the SSE2 loop (so that both can be tested on one machine)
static const bool g_avx2 = __builtin_cpu_supports("avx2") && !getenv("MARS_EMD_NO_AVX2");
template <typename CT> static inline double row_min(const CT* row, const double* v, int m) {
    return g_avx2 ? row_min_avx2(row, v, m) : row_min_sse2(row, v, m);
}
static const bool g_avx = __builtin_cpu_supports("avx") && !getenv("MARS_EMD_NO_AVX2");
static inline float row_min_f32(const float* row, const float* vf, int m) {
    return g_avx ? row_min_f32_avx(row, vf, m) : row_min_f32_sse(row, vf, m);
}
// MARS_EMD_EXACT_PRICING=1: every scan in double precision (the float prefilter off)
static const bool g_float_pricing = !getenv("MARS_EMD_EXACT_PRICING");

// Transportation simplex over a row-major cost matrix [nr, nc]; nr >= nc is what it is laid out for, not a requirement.
template <typename CT>
class TransportSimplex {
public:
    TransportSimplex(const CT* C, int nr, int nc) : C_(C), nr_(nr), nc_(nc), N_(nr + nc) {}

    double solve() {
        const double nan = std::numeric_limits<double>::quiet_NaN();
        if (nr_ <= 0 || nc_ <= 0) return nan;
        for (int64_t e = 0; e < (int64_t)nr_ * nc_; ++e)
            if (!std::isfinite((double)C_[e])) return nan;
        init_basis();
        const int64_t narcs = (int64_t)nr_ * nc_;
        int64_t block = (int64_t)std::sqrt((double)narcs);
        if (block < 16) block = 16;
        const int rows_per_block = 2 * (int)std::max<int64_t>(1, (block + nc_ - 1) / nc_);
        int next_row = 0;
        const double eps = 1e-11;
        const int64_t max_pivots = 64 * (int64_t)N_ * N_ + 100000;
        const double* v = pot_.data() + nr_;
        bool verified = false;
        // float costs: entering arcs are selected by the single-precision scan until it finds none below -1e-6,
        // then (and for double costs from the start) by the exact scan, which also establishes optimality
        bool fast = std::is_same<CT, float>::value && g_float_pricing;
        if (fast) vf_.resize(nc_);
        for (int64_t pivots = 0;; ++pivots) {
            if (pivots > max_pivots || broken_) return nan;
            // block pricing by whole rows: the most negative reduced cost of the first block of rows that has one
            int bi = -1, bj = -1;
            double best_u = 0.0;
            int scanned_rows = 0, i = next_row;
            if (fast) {
                for (int j = 0; j < nc_; ++j) vf_[j] = (float)v[j];
                const float* vf = vf_.data();
                float best_rc = -1e-6f, best_mn = 0.0f;
                while (scanned_rows < nr_) {
                    const int cnt = std::min(rows_per_block, nr_ - scanned_rows);
                    for (int k = 0; k < cnt; ++k) {
                        const float* row = (const float*)C_ + (int64_t)i * nc_;
                        const double u = row_potential(i);
                        const float mn = row_min_f32(row, vf, nc_);
                        const float rc = mn - (float)u;
                        if (rc < best_rc) { best_rc = rc; bi = i; best_u = u; best_mn = mn; }
                        if (++i == nr_) i = 0;
                    }
                    scanned_rows += cnt;
                    if (bi >= 0) break;
                }
                if (bi < 0) { fast = false; --pivots; continue; }       // the exact scan takes over (from the same row)
                const float* row = (const float*)C_ + (int64_t)bi * nc_;
                bj = 0;
                while (bj < nc_ && row[bj] - vf[bj] != best_mn) ++bj;
                if (bj == nc_ || ((double)row[bj] - v[bj]) - best_u >= -eps) { fast = false; --pivots; continue; }   // (never met: see row_min_f32)
            } else {
                double best_rc = -eps, best_mn = 0.0;
                while (scanned_rows < nr_) {
                    const int cnt = std::min(rows_per_block, nr_ - scanned_rows);
                    for (int k = 0; k < cnt; ++k) {
                        const CT* row = C_ + (int64_t)i * nc_;
                        const double u = row_potential(i);
                        const double mn = row_min(row, v, nc_);
                        if (mn - u < best_rc) { best_rc = mn - u; bi = i; best_u = u; best_mn = mn; }
                        if (++i == nr_) i = 0;
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
                // the column of the minimum in row bi (same arithmetic as the scan, so the value is met exactly)
                const CT* row = C_ + (int64_t)bi * nc_;
                bj = 0;
                while (bj < nc_ && (double)row[bj] - v[bj] != best_mn) ++bj;
                if (bj == nc_) { broken_ = true; continue; }
            }
            next_row = i;
            const CT* row = C_ + (int64_t)bi * nc_;
            pot_[bi] = best_u;      // a leaf row's potential exists from here on (the pivot may give it children)
            pivot(bi, bj, ((double)row[bj] - v[bj]) - best_u);
        }
        refresh_potentials();
        double su = 0.0, sv = 0.0;
        for (int i = 0; i < nr_; ++i) su += pot_[i];
        for (int j = 0; j < nc_; ++j) sv += pot_[nr_ + j];
        return su / nr_ + sv / nc_;
    }

private:
    // One node of the basis tree.  Rows are nodes 0 .. nr-1 (node 0 is the root), columns nr .. nr+nc-1, node N_ is
    // the end mark of the thread.  thread / rev / last describe the pre-order of the LISTED nodes (the root, every
    // column, the rows that have children): the listed nodes of a subtree are the run thread-linked from its top to
    // last.  An unlisted row has its parent, flow and size only.
    struct Node {
        int parent, size, nchild;       // size: all nodes of the subtree; nchild: all children
        int thread, rev, last;
        int64_t flow;                   // on the arc to the parent
    };
    const CT* C_;
    int nr_, nc_, N_;
    std::vector<Node> t_;
    std::vector<double> pot_;           // valid for the listed nodes
    std::vector<int> path_, old_size_, side_a_, side_b_, scratch_;
    std::vector<float> vf_;
    bool broken_ = false;

    double arc_cost(int a, int b) const {
        return a < nr_ ? (double)C_[(int64_t)a * nc_ + (b - nr_)] : (double)C_[(int64_t)b * nc_ + (a - nr_)];
    }
    bool listed(int x) const { return x >= nr_ || x == 0 || t_[x].nchild > 0; }
    double row_potential(int i) const {
        if (i == 0 || t_[i].nchild > 0) return pot_[i];
        const int c = t_[i].parent;
        return (double)C_[(int64_t)i * nc_ + (c - nr_)] - pot_[c];
    }
    // the run [first .. last] goes in right behind p, as the first part of p's subtree
    void paste(int p, int first, int last) {
        Node* t = t_.data();
        const int after = t[p].thread;
        t[p].thread = first; t[first].rev = p;
        t[last].thread = after; t[after].rev = last;
        for (int x = p; x >= 0 && t[x].last == p; x = t[x].parent) t[x].last = last;     // subtrees that ended with p
    }
    // the run [first .. last], all of it inside the subtree of every ancestor from `from` up, comes out
    void cut(int from, int first, int last) {
        Node* t = t_.data();
        const int before = t[first].rev, after = t[last].thread;
        t[before].thread = after; t[after].rev = before;
        for (int x = from; x >= 0 && t[x].last == last; x = t[x].parent) t[x].last = before;   // subtrees that ended with it
    }

    void refresh_potentials() {
        const Node* t = t_.data();
        pot_[0] = 0.0;
        for (int x = t[0].thread; x != N_; x = t[x].thread)        // pre-order: parents first
            pot_[x] = arc_cost(t[x].parent, x) - pot_[t[x].parent];
        for (int i = 1; i < nr_; ++i)       // the leaf rows hang below columns, which are all set now
            if (t[i].nchild == 0) pot_[i] = arc_cost(t[i].parent, i) - pot_[t[i].parent];
    }

    void init_basis() {
        t_.assign(N_ + 1, Node{-1, 1, 0, -1, -1, -1, 0});
        pot_.assign(N_, 0.0);
        Node* t = t_.data();
        std::vector<int64_t> demand(nc_, (int64_t)nr_ * (nr_ + 1));
        demand[nc_ - 1] += nr_;
        struct Arc { int i, j; int64_t f; };
        std::vector<Arc> arcs;
        arcs.reserve(N_);
        std::vector<char> open(nc_, 1);
        for (int i = 0; i < nr_; ++i) {
            int64_t supply = (int64_t)nc_ * (nr_ + 1) + 1;
            const CT* row = C_ + (int64_t)i * nc_;
            while (supply > 0) {
                int bj = -1; CT bc = std::numeric_limits<CT>::infinity();
                for (int j = 0; j < nc_; ++j) if (open[j] && row[j] < bc) { bc = row[j]; bj = j; }
                const int64_t x = std::min(supply, demand[bj]);
                arcs.push_back({i, bj, x});
                supply -= x; demand[bj] -= x;
                if (demand[bj] == 0) open[bj] = 0;
            }
        }
        // adjacency in compressed form, then a depth-first orientation from node 0 that visits nodes in pre-order
        std::vector<int> deg(N_ + 1, 0);
        for (const Arc& a : arcs) { ++deg[a.i + 1]; ++deg[nr_ + a.j + 1]; }
        for (int x = 0; x < N_; ++x) deg[x + 1] += deg[x];
        std::vector<int> fill(deg.begin(), deg.end() - 1), nb(2 * arcs.size());
        std::vector<int64_t> nbf(2 * arcs.size());
        for (const Arc& a : arcs) {
            nb[fill[a.i]] = nr_ + a.j; nbf[fill[a.i]++] = a.f;
            nb[fill[nr_ + a.j]] = a.i; nbf[fill[nr_ + a.j]++] = a.f;
        }
        std::vector<int> order; order.reserve(N_);
        std::vector<int> stack; stack.push_back(0);
        std::vector<char> seen(N_, 0);
        seen[0] = 1;
        while (!stack.empty()) {
            const int x = stack.back(); stack.pop_back();
            order.push_back(x);
            for (int e = deg[x]; e < deg[x + 1]; ++e) {
                const int y = nb[e];
                if (seen[y]) continue;
                seen[y] = 1;
                t[y].parent = x; t[y].flow = nbf[e];
                pot_[y] = arc_cost(x, y) - pot_[x];         // x was popped, so its potential is set
                ++t[x].nchild;
                stack.push_back(y);
            }
        }
        if ((int)order.size() != N_) { broken_ = true; return; }
        for (size_t h = order.size(); h-- > 1;) t[t[order[h]].parent].size += t[order[h]].size;    // children come after their parents
        // thread of the listed nodes in that pre-order; last: the listed node seen last inside each subtree
        int prev = -1;
        for (int x : order) {
            if (!listed(x)) continue;
            if (prev >= 0) { t[prev].thread = x; t[x].rev = prev; }
            prev = x;
        }
        t[prev].thread = N_; t[N_].rev = prev;
        for (size_t h = order.size(); h-- > 0;) {
            const int x = order[h];
            if (!listed(x)) continue;
            if (t[x].last < 0) t[x].last = x;
            const int par = t[x].parent;
            if (par >= 0 && t[par].last < 0) t[par].last = t[x].last;    // reverse pre-order: the first one met is the last of the parent
        }
    }

    // Arc (row a, column nr_ + j) enters with reduced cost rc < 0; pot_[a] is valid.
    void pivot(int a, int j, double rc) {
        Node* t = t_.data();
        const int b = nr_ + j;
        // the cycle: climb from both ends, the smaller subtree first, until they meet.  The arcs whose flow falls are
        // the ones below rows on a's side and below columns on b's side; the smallest of them leaves (unique: the
        // perturbed masses).
        side_a_.clear(); side_b_.clear();
        int leave = -1; bool leave_on_a = false;
        int64_t theta = std::numeric_limits<int64_t>::max();
        int x = a, y = b;
        while (x != y) {
            if (t[x].size < t[y].size) {
                if (x < nr_ && t[x].flow < theta) { theta = t[x].flow; leave = x; leave_on_a = true; }
                side_a_.push_back(x);
                x = t[x].parent;
            } else {
                if (y >= nr_ && t[y].flow < theta) { theta = t[y].flow; leave = y; leave_on_a = false; }
                side_b_.push_back(y);
                y = t[y].parent;
            }
            if (x < 0 || y < 0) { broken_ = true; return; }
        }
        const int apex = x;
        if (leave < 0) { broken_ = true; return; }
        for (int n : side_a_) t[n].flow += (n < nr_) ? -theta : theta;
        for (int n : side_b_) t[n].flow += (n >= nr_) ? -theta : theta;
        const int q = leave_on_a ? a : b, p = leave_on_a ? b : a;
        const int moved = t[leave].size;
        const int oldpar = t[leave].parent;
        for (int n = oldpar; n != apex; n = t[n].parent) t[n].size -= moved;
        if (q == leave && !listed(q)) {
            // a leaf row changes its column: nothing of it is in the thread and nobody's potential moves
            --t[oldpar].nchild;         // (a column: stays listed)
            t[q].parent = p; t[q].flow = theta;
            ++t[p].nchild;
            for (int n = p; n != apex; n = t[n].parent) t[n].size += moved;
            return;
        }
        if (!listed(a)) {               // a leaf row at either end of the entering arc gets a child: into the thread, behind its column
            t[a].last = a;
            paste(t[a].parent, a, a);
        }
        // the listed nodes below the leaving arc come out of the thread as one run ...
        const int run_first = leave;
        int run_last = t[leave].last;
        cut(oldpar, run_first, run_last);
        --t[oldpar].nchild;
        if (q != leave) {
            // ... which is put into the pre-order of the same nodes hanging from q.  Along the path x0 = q, x1, ..,
            // xk = leave (old parents), the subtree of x_t without that of x_(t-1) is x_t, the children's subtrees in
            // front of x_(t-1) [A_t] and those behind it [B_t]; the new order is x0 S0 | x1 A1 B1 | .. | xk Ak Bk, every
            // x_t being the last child of x_(t-1).
            path_.clear();
            for (int n = q; n != leave; n = t[n].parent) path_.push_back(n);
            path_.push_back(leave);
            const int len = (int)path_.size();
            // the places where the old order is cut, read before anything is relinked
            scratch_.resize(3 * (size_t)len);
            int* o_last = scratch_.data(), *o_rev = o_last + len, *o_after = o_rev + len;
            for (int k = 0; k < len; ++k) {
                const int n = path_[k];
                o_last[k] = t[n].last; o_rev[k] = t[n].rev; o_after[k] = t[o_last[k]].thread;
            }
            int tail = o_last[0];                   // the end of what has been strung together so far
            for (int k = 1; k < len; ++k) {
                const int n = path_[k];
                t[tail].thread = n; t[n].rev = tail;                    // x_k behind the end of R_(k-1)
                const int end_a = o_rev[k - 1];                         // the end of x_k A_k (x_k itself when A_k is empty)
                if (o_last[k] != o_last[k - 1]) {                       // B_k is not empty: it follows A_k now
                    const int b0 = o_after[k - 1];
                    t[end_a].thread = b0; t[b0].rev = end_a;
                    tail = o_last[k];
                } else {
                    tail = end_a;
                }
            }
            run_last = tail;
            // (subtrees of other nodes inside the run are whole runs of the old order and end where they ended)
            // parents and flows turn round along the path; sizes follow
            old_size_.resize(len);
            for (int k = 0; k < len; ++k) old_size_[k] = t[path_[k]].size;
            int64_t carry = theta;
            int newpar = p;
            for (int k = 0; k < len; ++k) {
                const int n = path_[k];
                const int64_t oldflow = t[n].flow;
                t[n].parent = newpar;
                t[n].flow = carry;
                carry = oldflow;
                newpar = n;
            }
            int below = 0;      // new size of the next node down the new chain
            for (int k = len; k-- > 0;) {
                const int own = old_size_[k] - (k > 0 ? old_size_[k - 1] : 0);
                t[path_[k]].size = own + below;
                below = t[path_[k]].size;
                t[path_[k]].last = run_last;        // every path node's subtree now ends where the run ends
            }
            ++t[q].nchild;          // x1 hangs from q; the nodes in between lose one child and get one
            --t[leave].nchild;      // x_(k-1) no longer hangs from leave
            paste(p, q, run_last);
        } else {
            t[q].parent = p; t[q].flow = theta;
            paste(p, q, run_last);
        }
        ++t[p].nchild;
        for (int n = p; n != apex; n = t[n].parent) t[n].size += moved;
        // potentials: the entering arc becomes tight.  Inside the moved subtree every row would move by d and every
        // column by -d (d = +rc when its top is a row); moving the REST of the tree the other way is the same dual
        // solution, so the smaller side is walked - along the thread, where the moved subtree is the run q .. run_last.
        // Leaf rows are not in it: theirs follow their column's.
        const double d = (q < nr_) ? rc : -rc;
        const int run_end = t[run_last].thread;
        if (2 * moved <= N_) {
            for (int n = q; n != run_end; n = t[n].thread) pot_[n] += n < nr_ ? d : -d;
        } else {
            for (int n = 0; n != N_;) {
                if (n == q) { n = run_end; continue; }
                pot_[n] += n < nr_ ? -d : d;
                n = t[n].thread;
            }
        }
        // rows that were left without children leave the thread
        if (leave < nr_ && leave != 0 && t[leave].nchild == 0) cut(t[leave].parent, leave, leave);
        if (oldpar < nr_ && oldpar != 0 && t[oldpar].nchild == 0) cut(t[oldpar].parent, oldpar, oldpar);
    }
};

// costs [n, m] in the solver's orientation: rows = the larger side (transposed into buf when m > n)
template <typename T>
static double solve_oriented(const T* M, int n, int m, std::vector<T>& buf) {
    if (n <= 0 || m <= 0) return std::numeric_limits<double>::quiet_NaN();
    if (n >= m) {
        return TransportSimplex<T>(M, n, m).solve();
    }
    buf.resize((size_t)n * m);
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < m; ++b) buf[(size_t)b * n + a] = M[(size_t)a * m + b];
    return TransportSimplex<T>(buf.data(), m, n).solve();
}

}  // namespace

extern "C" double marshost_emd2_uniform(const double* C, int n, int m) {
    std::vector<double> buf;
    return solve_oriented(C, n, m, buf);
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
            // gathered in the solver's orientation: the larger side as rows
            M.resize((size_t)n * m);
            const bool tr = m > n;
            for (int a = 0; a < n; ++a) {
                const float* sr = S + (int64_t)rows[a] * ldS;
                for (int b = 0; b < m; ++b) {
                    // cost_matrix = (1 - S) / 2 in float32, as the reference stores it
                    const float c = is_cost ? sr[cols[b]] : (1.0f - sr[cols[b]]) / 2.0f;
                    M[tr ? (size_t)b * n + a : (size_t)a * m + b] = c;
                }
            }
            TransportSimplex<float> s(M.data(), tr ? m : n, tr ? n : m);
            out[k] = 1.0 - s.solve();
        }
    };
    if (threads == 1) { work(); return 0; }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(work);
    for (auto& t : pool) t.join();
    return 0;
}
