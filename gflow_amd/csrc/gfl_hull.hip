// Concave hull of a 2-D point set -- HOST code (no kernel in this file): the ring behind GFlow's moving-region mask
// (gflow/utils/concave_hull.py:73-92, called at trainer.py:604-609; the reference uses the `concave_hull` package, a binding
// of mapbox's concaveman, C++ as well).  The algorithm is gflow_amd/hull.py's `concave_hull_py` statement by statement
// (convex hull by Andrew's monotone chain; every hull edge, in queue order, is dug in towards the nearest free point that is
// closer to it than to its two neighbour edges, within edge length / concavity, and whose two new edges cross no hull edge);
// tests/test_host_logic.py holds the two against each other.  Why native: the numpy version scans every free point with a
// dozen array operations per edge -- 125 ms for the ~6 000 moving splats of a 480p frame, once per clip when the trajectories'
// grid seeds are chosen (fit_video.py:163-211) -- this one takes ~2 ms.
#include "../../include/gflow_hip.h"

#include <algorithm>
#include <vector>

namespace {

struct Pt { double x, y; };

inline double cross3(const Pt& o, const Pt& a, const Pt& b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }

inline double sq_seg_dist(const Pt& p, const Pt& a, const Pt& b) {
    const double abx = b.x - a.x, aby = b.y - a.y;
    const double den = abx * abx + aby * aby;
    if (den == 0.0) return (p.x - a.x) * (p.x - a.x) + (p.y - a.y) * (p.y - a.y);
    double t = ((p.x - a.x) * abx + (p.y - a.y) * aby) / den;
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    const double qx = a.x + t * abx, qy = a.y + t * aby;
    return (p.x - qx) * (p.x - qx) + (p.y - qy) * (p.y - qy);
}

inline double orient(const Pt& p, const Pt& q, const Pt& r) { return (q.y - p.y) * (r.x - q.x) - (q.x - p.x) * (r.y - q.y); }

// the segment P[a] - P[b] crosses none of the live hull edges; edges that share an end point with it do not count
// (concaveman's `intersects`: p1 !== q2 && q1 !== p2 && the two orientation tests)
bool no_intersections(const Pt* P, int a, int b, const std::vector<int>& ea, const std::vector<int>& eb) {
    const Pt &p2 = P[a], &q2 = P[b];
    for (size_t i = 0; i < ea.size(); ++i) {
        if (ea[i] == b || eb[i] == a) continue;
        const Pt &p1 = P[ea[i]], &q1 = P[eb[i]];
        const bool o1 = orient(p1, q1, p2) > 0, o2 = orient(p1, q1, q2) > 0;
        const bool o3 = orient(p2, q2, p1) > 0, o4 = orient(p2, q2, q1) > 0;
        if (o1 != o2 && o3 != o4) return false;
    }
    return true;
}

}  // namespace

extern "C" int gfl_concave_hull(const double* points_xy, int n, double concavity, double length_threshold, double* ring_xy,
                                int cap_vertices) {
    if (!points_xy || !ring_xy || n < 0 || cap_vertices < 0 || !(concavity > 0.0)) return GFL_ERR_INVALID;
    const Pt* P = reinterpret_cast<const Pt*>(points_xy);
    if (n <= 3) {
        if (cap_vertices < n) return GFL_ERR_WORKSPACE;
        for (int i = 0; i < n; ++i) { ring_xy[2 * i] = P[i].x; ring_xy[2 * i + 1] = P[i].y; }
        return n;
    }
    // ---- convex hull (the points arrive sorted by x, then y, without duplicates: numpy.unique(axis=0))
    std::vector<int> lower, upper;
    for (int i = 0; i < n; ++i) {
        while (lower.size() >= 2 && cross3(P[lower[lower.size() - 2]], P[lower.back()], P[i]) <= 0) lower.pop_back();
        lower.push_back(i);
    }
    for (int i = n - 1; i >= 0; --i) {
        while (upper.size() >= 2 && cross3(P[upper[upper.size() - 2]], P[upper.back()], P[i]) <= 0) upper.pop_back();
        upper.push_back(i);
    }
    std::vector<int> hull(lower.begin(), lower.end() - 1);
    hull.insert(hull.end(), upper.begin(), upper.end() - 1);
    const int m0 = (int)hull.size();
    const int cap = 2 * n + 8;
    std::vector<int> node_p(cap, 0), nxt(cap, 0), prv(cap, 0);
    std::vector<char> alive(cap, 0), is_free(n, 1);
    for (int i = 0; i < m0; ++i) {
        node_p[i] = hull[i];
        nxt[i] = (i + 1) % m0;
        prv[i] = (i + m0 - 1) % m0;
        alive[i] = 1;
        is_free[hull[i]] = 0;
    }
    int n_nodes = m0, n_free = n - m0;
    std::vector<int> queue(m0);
    for (int i = 0; i < m0; ++i) queue[i] = i;
    const double sq_conc = concavity * concavity, sq_len_thr = length_threshold * length_threshold;
    std::vector<std::pair<double, int>> cand;
    std::vector<int> ea, eb;
    for (size_t head = 0; head < queue.size(); ++head) {
        const int node = queue[head];
        if (!alive[node]) continue;
        const int a_i = node_p[node], b_i = node_p[nxt[node]];
        const Pt a = P[a_i], b = P[b_i];
        const double sq_len = (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y);
        if (sq_len < sq_len_thr || n_free == 0) continue;
        const double max_sq = sq_len / sq_conc;
        const Pt prev_p = P[node_p[prv[node]]], next_p = P[node_p[nxt[nxt[node]]]];
        cand.clear();
        for (int c = 0; c < n; ++c) {
            if (!is_free[c]) continue;
            const double d = sq_seg_dist(P[c], a, b);
            if (!(d <= max_sq)) continue;
            if (d < sq_seg_dist(P[c], prev_p, a) && d < sq_seg_dist(P[c], b, next_p)) cand.emplace_back(d, c);
        }
        if (cand.empty()) continue;
        std::stable_sort(cand.begin(), cand.end(), [](const std::pair<double, int>& x, const std::pair<double, int>& y) { return x.first < y.first; });
        ea.clear(); eb.clear();
        for (int i = 0; i < n_nodes; ++i)
            if (alive[i]) { ea.push_back(node_p[i]); eb.push_back(node_p[nxt[i]]); }
        int chosen = -1;
        for (const auto& dc : cand)
            if (no_intersections(P, b_i, dc.second, ea, eb) && no_intersections(P, a_i, dc.second, ea, eb)) { chosen = dc.second; break; }
        if (chosen < 0) continue;
        const Pt pc = P[chosen];
        const double da = (pc.x - a.x) * (pc.x - a.x) + (pc.y - a.y) * (pc.y - a.y);
        const double db = (pc.x - b.x) * (pc.x - b.x) + (pc.y - b.y) * (pc.y - b.y);
        if (std::min(da, db) > max_sq) continue;
        const int fresh = n_nodes++;                       // a -> chosen -> b: the new node sits behind `node`
        node_p[fresh] = chosen;
        nxt[fresh] = nxt[node]; prv[fresh] = node;
        prv[nxt[node]] = fresh;
        nxt[node] = fresh;
        alive[fresh] = 1;
        is_free[chosen] = 0;
        --n_free;
        queue.push_back(node);
        queue.push_back(fresh);
    }
    int start = 0;
    while (!alive[start]) ++start;
    int m = 0, i = start;
    do {
        if (m >= cap_vertices) return GFL_ERR_WORKSPACE;
        ring_xy[2 * m] = P[node_p[i]].x; ring_xy[2 * m + 1] = P[node_p[i]].y;
        ++m;
        i = nxt[i];
    } while (i != start);
    return m;
}
