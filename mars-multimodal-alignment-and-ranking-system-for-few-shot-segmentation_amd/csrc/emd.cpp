// Host side of the ranking path: exact earth mover's distance between two uniform
// histograms, the value POT's ot.emd2 returns at
// /root/reference/mars/components/FilteringMergingModule.py:160-166 and
// /root/reference/matcher/Matcher.py:1188-1193.
//
// Algorithm: primal network simplex on the dense transportation graph (n sources,
// m sinks), spanning-tree basis stored as parent pointers with child lists, block
// pricing.  Degeneracy -- severe for uniform marginals -- is removed with Orden's
// perturbation on INTEGER masses: scale so that each source ships m(n+1)+1 units,
// each sink takes n(n+1) units and the last sink n more.  Every basic solution of
// that instance is strictly positive on its n+m-1 tree arcs, so each pivot moves a
// positive amount and the simplex cannot cycle; an optimal basis of the perturbed
// instance is optimal for the original one.  The optimum is returned through the
// dual, sum_i u_i/n + sum_j v_j/m, which needs no unperturbed flows.
//
// One problem per worker thread; the GPU keeps running the AlphaCLIP tower meanwhile.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <limits>
#include <thread>
#include <vector>

#include "../../include/mars_hip.h"

void mars_set_error(const char* fmt, ...);

namespace {

class TransportSimplex {
public:
    TransportSimplex(const double* C, int n, int m) : C_(C), n_(n), m_(m), N_(n + m) {}

    double solve() {
        const double nan = std::numeric_limits<double>::quiet_NaN();
        if (n_ <= 0 || m_ <= 0) return nan;
        // the pivoting below relies on ordered comparisons; a NaN or infinite cost would let it pick
        // arcs that break the tree, so such a problem has no answer instead
        for (int64_t e = 0; e < (int64_t)n_ * m_; ++e)
            if (!std::isfinite(C_[e])) return nan;
        init_basis();
        const int64_t narcs = (int64_t)n_ * m_;
        int64_t block = (int64_t)std::sqrt((double)narcs);
        if (block < 16) block = 16;
        int64_t next = 0;  // arc where the next pricing block starts
        const double eps = 1e-11;
        // Orden's perturbation rules out degenerate pivots, so the count stays far below this cap;
        // it only bounds the damage of an input that violates the solver's assumptions
        const int64_t max_pivots = 64 * (int64_t)N_ * N_ + 100000;
        for (int64_t pivots = 0;; ++pivots) {
            if (pivots > max_pivots || broken_) return nan;
            // block pricing: most negative reduced cost inside the first block that has one
            int64_t best = -1;
            double best_rc = -eps;
            int64_t scanned = 0;
            int64_t e = next;
            int i = (int)(e / m_), j = (int)(e % m_);
            while (scanned < narcs) {
                int64_t cnt = std::min(block, narcs - scanned);
                for (int64_t k = 0; k < cnt; ++k) {
                    const double rc = C_[e] - pot_[i] - pot_[n_ + j];
                    if (rc < best_rc) { best_rc = rc; best = e; }
                    ++e; ++j;
                    if (j == m_) { j = 0; ++i; if (i == n_) { i = 0; e = 0; } }
                }
                scanned += cnt;
                if (best >= 0) break;
            }
            if (best < 0) break;  // dual feasible: optimal
            next = e;
            pivot((int)(best / m_), (int)(best % m_));
        }
        double val = 0.0, su = 0.0, sv = 0.0;
        for (int i = 0; i < n_; ++i) su += pot_[i];
        for (int j = 0; j < m_; ++j) sv += pot_[n_ + j];
        val = su / n_ + sv / m_;
        return val;
    }

private:
    const double* C_;
    int n_, m_, N_;
    std::vector<int> parent_, depth_, first_child_, next_sib_, prev_sib_;
    std::vector<int64_t> flow_;  // flow on the arc joining a node to its parent
    std::vector<double> pot_;    // u_i (sources), v_j (sinks); tree arcs satisfy c_ij = u_i + v_j

    double arc_cost(int a, int b) const {  // a, b adjacent tree nodes of opposite kinds
        return a < n_ ? C_[(int64_t)a * m_ + (b - n_)] : C_[(int64_t)b * m_ + (a - n_)];
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

    void init_basis() {
        parent_.assign(N_, -1); depth_.assign(N_, 0);
        first_child_.assign(N_, -1); next_sib_.assign(N_, -1); prev_sib_.assign(N_, -1);
        flow_.assign(N_, 0); pot_.assign(N_, 0.0);
        // greedy row-minimum start: every source ships to its cheapest open sinks
        std::vector<int64_t> demand(m_, (int64_t)n_ * (n_ + 1));
        demand[m_ - 1] += n_;
        struct Arc { int i, j; int64_t f; };
        std::vector<Arc> arcs;
        arcs.reserve(N_);
        std::vector<char> open(m_, 1);
        for (int i = 0; i < n_; ++i) {
            int64_t supply = (int64_t)m_ * (n_ + 1) + 1;
            const double* row = C_ + (int64_t)i * m_;
            while (supply > 0) {
                int bj = -1; double bc = std::numeric_limits<double>::infinity();
                for (int j = 0; j < m_; ++j) if (open[j] && row[j] < bc) { bc = row[j]; bj = j; }
                const int64_t x = std::min(supply, demand[bj]);
                arcs.push_back({i, bj, x});
                supply -= x; demand[bj] -= x;
                if (demand[bj] == 0) open[bj] = 0;
            }
        }
        // adjacency of the n+m-1 basic arcs, then a BFS from node 0 orients the tree
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
                parent_[y] = x; depth_[y] = depth_[x] + 1; flow_[y] = pr.second;
                pot_[y] = arc_cost(x, y) - pot_[x];
                add_child(x, y);
                queue.push_back(y);
            }
        }
    }

    void pivot(int i, int j) {
        const int a = i, b = n_ + j;
        // leaving arc: smallest flow among the arcs whose flow decreases around the cycle
        int x = a, y = b, leave = -1; bool leave_on_a = false;
        int64_t theta = std::numeric_limits<int64_t>::max();
        while (x != y) {
            if (depth_[x] >= depth_[y]) {
                if (x < n_ && flow_[x] < theta) { theta = flow_[x]; leave = x; leave_on_a = true; }
                x = parent_[x];
            } else {
                if (y >= n_ && flow_[y] < theta) { theta = flow_[y]; leave = y; leave_on_a = false; }
                y = parent_[y];
            }
        }
        if (leave < 0) { broken_ = true; return; }   // entering arc closed no cycle: not a spanning tree any more
        const int apex = x;
        for (x = a; x != apex; x = parent_[x]) flow_[x] += (x < n_) ? -theta : theta;
        for (y = b; y != apex; y = parent_[y]) flow_[y] += (y >= n_) ? -theta : theta;
        // the subtree hanging below `leave` is cut off and re-hung from the entering arc
        const int q = leave_on_a ? a : b, p = leave_on_a ? b : a;
        path_.clear();
        for (x = q; x != leave; x = parent_[x]) path_.push_back(x);
        path_.push_back(leave);
        remove_child(parent_[leave], leave);
        // reverse the parent pointers along q .. leave
        int64_t carry = theta;  // flow of the arc that becomes (node -> new parent)
        int newpar = p;
        for (size_t t = 0; t < path_.size(); ++t) {
            const int node = path_[t];
            const int oldpar = parent_[node];
            const int64_t oldflow = flow_[node];
            if (t + 1 < path_.size()) remove_child(oldpar, node);  // oldpar == path_[t+1]
            parent_[node] = newpar;
            flow_[node] = carry;
            add_child(newpar, node);
            carry = oldflow;
            newpar = node;
        }
        // refresh depth and potentials of the moved subtree
        stack_.clear();
        stack_.push_back(q);
        while (!stack_.empty()) {
            const int node = stack_.back(); stack_.pop_back();
            const int par = parent_[node];
            depth_[node] = depth_[par] + 1;
            pot_[node] = arc_cost(par, node) - pot_[par];
            for (int c = first_child_[node]; c >= 0; c = next_sib_[c]) stack_.push_back(c);
        }
    }

    std::vector<int> path_, stack_;
    bool broken_ = false;
};

}  // namespace

extern "C" double marshost_emd2_uniform(const double* C, int n, int m) {
    TransportSimplex s(C, n, m);
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
    std::atomic<int> next(0);
    auto work = [&]() {
        std::vector<int> cols;
        std::vector<double> M;
        for (;;) {
            const int k = next.fetch_add(1);
            if (k >= N) break;
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
                    M[(size_t)a * m + b] = (double)c;
                }
            }
            TransportSimplex s(M.data(), n, m);
            out[k] = 1.0 - s.solve();
        }
    };
    if (threads == 1) { work(); return 0; }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(work);
    for (auto& t : pool) t.join();
    return 0;
}
