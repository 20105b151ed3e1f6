// Host-side pieces of the segment tracker: the two-frame cost matrix of
// segment_tracking.py:46-102,179-254 and the assignment solve behind apply_hungarian_algorithm (:257-263).
// Plain C++ (no GPU): the reference runs this part on the host too; it is per-frame sequential logic on
// matrices of a few dozen rows.  The Python bookkeeping (statuses, shared history lists, events) lives in
// swiftwatcher_amd/segment_tracking.py.
#include <math.h>
#include <stdint.h>
#include <float.h>
#include <vector>
#include <algorithm>
#include "swk.h"

#pragma GCC visibility push(default)
extern "C" {

// Cost matrix of formulate_cost_matrix (:82-102).  Square, size n_prev + n_curr, row-major.
//   every cell            1 + DBL_EPSILON                        (:186, "impossible" cells)
//   [i, n_prev + j]       0.5 * 2^(dist - 25) + 0.5 * angle_cost  (:93-97) for prev i, curr j
//   [i, i]                1                                       (:99-100, D / A cells)
// angle_cost = 2^(diff - 90) with diff the folded angle between the motion path of prev i (from its first
// history centroid hist0 to prev i) and the step prev i -> curr j; 1 when prev i has no history (:215-245).
// Centroids are (row, col) float64 pairs.  The libm calls are the ones CPython's math module makes
// (atan2, pow), so costs are bit-identical to the reference's on the same machine.
int32_t swk_track_costs(const double *prev_c, const double *prev_hist0, const uint8_t *prev_has_hist,
                        const double *curr_c, int32_t n_prev, int32_t n_curr, double *cost)
{
    if (n_prev < 0 || n_curr < 0 || !cost) return SWK_ERR_ARG;
    const int n = n_prev + n_curr;
    const double filler = 1.0 + DBL_EPSILON;
    for (int i = 0; i < n * n; ++i) cost[i] = filler;
    const double rad2deg = 180.0 / M_PI;                         // math.degrees(x) = x * (180 / pi)
    for (int i = 0; i < n_prev; ++i) {
        const double pr = prev_c[2 * i], pc = prev_c[2 * i + 1];
        double old_angle = 0.0;
        const bool hist = prev_has_hist[i] != 0;
        if (hist) {
            const double del_y = prev_hist0[2 * i] - pr, del_x = prev_hist0[2 * i + 1] - pc;      // :225-227
            old_angle = atan2(del_y, -1 * del_x) * rad2deg;
        }
        for (int j = 0; j < n_curr; ++j) {
            const double cr = curr_c[2 * j], cc = curr_c[2 * j + 1];
            const double dy = pr - cr, dx = pc - cc;
            const double dist = sqrt(dy * dy + dx * dx);                                          // :193
            const double d_cost = pow(2.0, dist - 25);                                            // :195
            double a_cost = 1.0;
            if (hist) {
                const double new_angle = atan2(dy, -1 * dx) * rad2deg;                            // :230-232
                double diff = fabs(new_angle - old_angle);
                diff = diff < 360 - diff ? diff : 360 - diff;                                     // :238
                a_cost = pow(2.0, diff - 90);                                                     // :241
            }
            cost[i * n + n_prev + j] = 0.5 * d_cost + 0.5 * a_cost;                               // :97
        }
    }
    for (int i = 0; i < n; ++i) cost[i * n + i] = 1.0;                                            // :99-100
    return SWK_OK;
}

// Rectangular linear sum assignment, shortest augmenting path (Crouse 2016), the algorithm behind
// scipy.optimize.linear_sum_assignment since SciPy 1.4 -- restated with the same scan order and the same
// tie rule (among equal shortest-path costs prefer an unassigned column; columns are scanned from the
// `remaining` list that starts as n_cols-1 .. 0), so equal-cost ties resolve as in SciPy.
// cost is row-major n_rows x n_cols with n_rows <= n_cols; col4row[i] receives the column of row i.
//
// Which SciPy: the reference pins scipy==1.3.1 (requirements.txt:15), whose linear_sum_assignment is the older Munkres
// implementation; parity here is defined against SciPy >= 1.4 (fixtures recorded under 1.7.1, cross-checked with 1.15.3).
// The two can differ only in which of several equal-cost permutations they return, and
// tests/test_tracking_counts.py::test_ties_among_structural_cells_cannot_change_statuses shows by brute force that every
// optimal assignment of this cost structure yields the same matches and D / A statuses.
//
// This function follows scipy/optimize/rectangular_lsap/rectangular_lsap.cpp closely (variable names included, so the
// two can be compared line by line).  That file is Copyright (c) 2019 PM Larsen and the SciPy developers, distributed
// under the 3-clause BSD licence; the notice is reproduced in THIRD_PARTY_NOTICES.md at the repository root.
int32_t swk_lsap(const double *cost, int32_t n_rows, int32_t n_cols, int32_t *col4row)
{
    if (!cost || !col4row || n_rows < 0 || n_cols < n_rows) return SWK_ERR_ARG;
    const int nr = n_rows, nc = n_cols;
    std::vector<double> u(nr, 0.0), v(nc, 0.0), spc(nc);
    std::vector<int> path(nc, -1), row4col(nc, -1), remaining(nc);
    std::vector<char> SR(nr), SC(nc);
    for (int i = 0; i < nr; ++i) col4row[i] = -1;
    for (int cur = 0; cur < nr; ++cur) {
        double min_val = 0.0;
        int num_remaining = nc;
        for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
        std::fill(SR.begin(), SR.end(), 0);
        std::fill(SC.begin(), SC.end(), 0);
        std::fill(spc.begin(), spc.end(), INFINITY);
        int sink = -1, i = cur;
        while (sink == -1) {
            int index = -1;
            double lowest = INFINITY;
            SR[i] = 1;
            for (int it = 0; it < num_remaining; ++it) {
                const int j = remaining[it];
                const double r = min_val + cost[(size_t)i * nc + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            min_val = lowest;
            if (min_val == INFINITY) return SWK_ERR_ARG;            // infeasible cost matrix
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        u[cur] += min_val;
        for (int r = 0; r < nr; ++r)
            if (SR[r] && r != cur) u[r] += min_val - spc[col4row[r]];
        for (int j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= min_val - spc[j];
        int j = sink;
        for (;;) {
            const int r = path[j];
            row4col[j] = r;
            std::swap(col4row[r], j);
            if (r == cur) break;
        }
    }
    return SWK_OK;
}

}  // extern "C"
#pragma GCC visibility pop
