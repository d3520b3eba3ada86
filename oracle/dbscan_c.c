/* CPU oracle, TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * All-pairs C restatement of what sklearn.cluster.DBSCAN computes for one chunk
 * (reference call site: utils/tower_extraction.py:107-112):
 *   - rdist(x,y) = sum_j (x_j - y_j)^2, float32 inputs promoted to float64, squares
 *     accumulated in axis order (sklearn/metrics/_dist_metrics.pxd.tp euclidean_rdist),
 *     neighbour iff rdist <= eps*eps (sklearn/neighbors/_binary_tree.pxi.tp:1953-1958);
 *   - core iff neighbour count (self included) >= min_samples (_dbscan.py:423-434);
 *   - labels by the sweep of sklearn/cluster/_dbscan_inner.pyx:10-41.  The sweep there
 *     labels a point when it is popped; every pop of one sweep happens before label_num
 *     is incremented, so labelling at first reach (done here, which bounds the stack at n
 *     entries) yields the same labels.  oracle/dbscan.py:dbscan_inner_literal keeps the
 *     literal form and the tests compare the two.
 * Built with -ffp-contract=off: no FMA contraction, as in sklearn's baseline x86-64 wheels.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline double rdist3(const float* a, const float* b) {
    double d = 0.0;
    for (int j = 0; j < 3; ++j) {
        double t = (double)a[j] - (double)b[j];
        d += t * t;
    }
    return d;
}

int oracle_dbscan_f32(const float* X, int64_t n, double eps, int32_t min_samples,
                      int32_t* labels, uint8_t* core) {
    const double r2 = eps * eps;
    if (n < 0) return -1;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n; ++i) {
        int64_t cnt = 0;
        const float* xi = X + 3 * i;
        for (int64_t j = 0; j < n; ++j)
            cnt += (rdist3(xi, X + 3 * j) <= r2);
        core[i] = (uint8_t)(cnt >= (int64_t)min_samples);
    }
    for (int64_t i = 0; i < n; ++i) labels[i] = -1;
    int64_t* stack = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    if (!stack) return -2;
    int32_t label_num = 0;
    for (int64_t s = 0; s < n; ++s) {
        if (labels[s] != -1 || !core[s]) continue;
        int64_t top = 0;
        labels[s] = label_num;
        stack[top++] = s;
        while (top > 0) {
            int64_t i = stack[--top];
            if (!core[i]) continue;                 /* border point: labelled, not expanded */
            const float* xi = X + 3 * i;
            for (int64_t v = 0; v < n; ++v) {
                if (labels[v] == -1 && rdist3(xi, X + 3 * v) <= r2) {
                    labels[v] = label_num;
                    stack[top++] = v;
                }
            }
        }
        ++label_num;
    }
    free(stack);
    return 0;
}

/* Sequential float32 column sums: what np.mean(a, axis=0) accumulates for a C-order
 * (n,3) float32 array (utils/tower_extraction.py:63); see oracle/ground_filter.py. */
int oracle_seqsum3_f32(const float* X, int64_t n, float* out3) {
    volatile float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int64_t i = 0; i < n; ++i) {
        s0 = s0 + X[3 * i + 0];
        s1 = s1 + X[3 * i + 1];
        s2 = s2 + X[3 * i + 2];
    }
    out3[0] = s0; out3[1] = s1; out3[2] = s2;
    return 0;
}
