// Prioritised replay on the device (reference super_sac/replay.py:140-190 ReplayBuffer.sample / update_priorities,
// :207-353 SegmentTree / SumSegmentTree / MinSegmentTree).
//
// The trees are the reference's: implicit binary heaps of float64 over the next power of two >= capacity, node 1 the
// root, leaves at [cap, 2 cap).  They live in HBM, so that the AFBC / PER update path never leaves the device: the
// host only draws the B uniforms from numpy's GLOBAL generator (replay.py:166: the stream the reference consumes) and
// uploads them; total mass, prefix-sum descent, importance weights and the priority refresh are kernels.  The float64
// host trees of replay.PrioritySampler stay as the checker the tests compare against.
//
// Everything here is ONE workgroup per call: a batch touches B leaves and their <= B ancestors per level, a level
// needs the level below finished, and a block barrier is the cheapest barrier there is.  (Bulk loads -- more rows than
// a workgroup walks comfortably -- set the leaves with a grid and rebuild the inner nodes level by level.)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ssac_internal.h"

namespace {

constexpr int PER_THREADS = 1024;

// SegmentTree.reduce(0, end_exclusive) for the sum tree, in the reference's own association order
// (replay.py:229-258: _reduce_helper returns op(left part, right part) recursively; with start = 0 every left part
// is a whole node, so the value is v[n1] + (v[n2] + (... + v[nk])) along one root-to-leaf descent).
__device__ double range_sum0(const double *tree, int64_t cap, int64_t end_incl) {
    int64_t nodes[48];
    int k = 0;
    int64_t node = 1, ns = 0, ne = cap - 1;
    while (true) {
        if (end_incl == ne) { nodes[k++] = node; break; }
        const int64_t mid = (ns + ne) / 2;
        if (end_incl <= mid) { node = 2 * node; ne = mid; continue; }
        nodes[k++] = 2 * node;          // reduce(ns, mid, 2 node, ns, mid) == v[2 node]
        node = 2 * node + 1; ns = mid + 1;
    }
    double acc = tree[nodes[k - 1]];
    for (int i = k - 2; i >= 0; --i) acc = tree[nodes[i]] + acc;
    return acc;
}

struct AssignArgs {
    double *sum_tree, *min_tree; int64_t cap;
    const int64_t *rows; int n;
    const void *prio; int prio_f64;       // null: the current max priority
    double alpha; double *max_priority; int update_max; int64_t n_filled;
    int *err;                             // pinned host word: 1 = priority <= 0, 2 = row out of range
    int32_t *win;                         // cap entries, all -1 between calls: which batch entry writes a leaf (the LAST
                                          // one that names it, as numpy's fancy assignment `value[idxs] = val` does)
};

// leaves <- prio^alpha, then every ancestor level by level (each thread re-derives the ancestor of ITS rows: rows
// sharing an ancestor write the same value)
__global__ __launch_bounds__(PER_THREADS) void per_assign_kernel(AssignArgs a) {
    const int tid = threadIdx.x;
    const double maxp = *a.max_priority;
    double seen = 0.0;
    for (int i = tid; i < a.n; i += PER_THREADS) {
        const int64_t r = a.rows[i];
        if (r >= 0 && r < a.cap) atomicMax(a.win + r, i);
    }
    __syncthreads();
    for (int i = tid; i < a.n; i += PER_THREADS) {
        const int64_t r = a.rows[i];
        double p = maxp;
        if (a.prio) p = a.prio_f64 ? reinterpret_cast<const double *>(a.prio)[i] : (double)reinterpret_cast<const float *>(a.prio)[i];
        if (a.prio && !(p > 0.0)) { *a.err = 1; continue; }
        if (r < 0 || r >= a.cap || (a.update_max && r >= a.n_filled)) { *a.err = 2; continue; }
        seen = fmax(seen, p);
        if (a.win[r] != i) continue;   // a later entry of the batch names the same row
        const double v = pow(p, a.alpha);
        a.sum_tree[a.cap + r] = v;
        a.min_tree[a.cap + r] = v;
    }
    __syncthreads();
    for (int i = tid; i < a.n; i += PER_THREADS) {
        const int64_t r = a.rows[i];
        if (r >= 0 && r < a.cap) a.win[r] = -1;
    }
    if (a.update_max) {   // _max_priority = max(_max_priority, max(priorities)): positive doubles order like their bits
        for (int o = 32; o > 0; o >>= 1) seen = fmax(seen, __shfl_xor(seen, o, 64));
        if ((tid & 63) == 0 && seen > 0.0)
            atomicMax(reinterpret_cast<unsigned long long *>(a.max_priority), (unsigned long long)__double_as_longlong(seen));
    }
    for (int64_t span = 2; span <= a.cap; span <<= 1) {   // span = leaves under a node of this level
        __syncthreads();
        for (int i = tid; i < a.n; i += PER_THREADS) {
            const int64_t r = a.rows[i];
            if (r < 0 || r >= a.cap) continue;
            const int64_t node = (a.cap + r) / span;
            a.sum_tree[node] = a.sum_tree[2 * node] + a.sum_tree[2 * node + 1];
            a.min_tree[node] = fmin(a.min_tree[2 * node], a.min_tree[2 * node + 1]);
        }
    }
}

__global__ void per_set_leaves_kernel(AssignArgs a) {
    const double maxp = *a.max_priority;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = a.rows[i];
        double p = maxp;
        if (a.prio) p = a.prio_f64 ? reinterpret_cast<const double *>(a.prio)[i] : (double)reinterpret_cast<const float *>(a.prio)[i];
        if (a.prio && !(p > 0.0)) { *a.err = 1; continue; }
        if (r < 0 || r >= a.cap) { *a.err = 2; continue; }
        const double v = pow(p, a.alpha);
        a.sum_tree[a.cap + r] = v;
        a.min_tree[a.cap + r] = v;
    }
}

__global__ void per_level_kernel(double *sum_tree, double *min_tree, int64_t first, int64_t count) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t node = first + i;
        sum_tree[node] = sum_tree[2 * node] + sum_tree[2 * node + 1];
        min_tree[node] = fmin(min_tree[2 * node], min_tree[2 * node + 1]);
    }
}

// mass_b = u_b * sum(0, n_filled - 1); descent "largest i with prefix_sum(i) <= mass" (replay.py:297-336);
// w_b = (p_b n)^-beta / max_weight with p = leaf / total (replay.py:171-177)
__global__ __launch_bounds__(PER_THREADS) void per_sample_kernel(const double *sum_tree, const double *min_tree, int64_t cap,
                                                                 int64_t n_filled, const double *u, int B, double beta,
                                                                 int64_t *idx_out, double *w_out) {
    __shared__ double s_total;
    if (threadIdx.x == 0) s_total = range_sum0(sum_tree, cap, n_filled - 2);   // (the reference's `end` is exclusive after -1)
    __syncthreads();
    const double total = s_total, root = sum_tree[1];
    const double p_min = min_tree[1] / root;
    const double max_weight = pow(p_min * (double)n_filled, -beta);
    for (int b = threadIdx.x; b < B; b += PER_THREADS) {
        double mass = u[b] * total;
        int64_t node = 1;
        while (node < cap) {
            const int64_t left = 2 * node;
            const double ls = sum_tree[left];
            if (ls <= mass) { mass -= ls; node = left + 1; } else { node = left; }
        }
        const int64_t idx = node - cap;
        idx_out[b] = idx;
        const double p_sample = sum_tree[cap + idx] / root;
        w_out[b] = pow(p_sample * (double)n_filled, -beta) / max_weight;
    }
}

}  // namespace

extern "C" int ssac_per_assign(double *sum_tree, double *min_tree, int64_t cap, const int64_t *rows, int n,
                               const void *prio, int prio_is_f64, double alpha, double *max_priority, int update_max,
                               int64_t n_filled, int *err_host_word, int32_t *winner_scratch, void *stream) {
    if (!sum_tree || !min_tree || !rows || !max_priority || !err_host_word || !winner_scratch || cap <= 0 ||
        (cap & (cap - 1)) || n < 0)
        return ssac_fail("ssac_per_assign: bad arguments");
    if (n == 0) return 0;
    AssignArgs a{sum_tree, min_tree, cap, rows, n, prio, prio_is_f64, alpha, max_priority, update_max, n_filled,
                 err_host_word, winner_scratch};
    hipStream_t st = (hipStream_t)stream;
    if (n <= 8 * PER_THREADS) {
        SSAC_LAUNCH(per_assign_kernel, dim3(1), dim3(PER_THREADS), 0, st, a);
        return ssac_check_launch("per_assign");
    }
    // bulk (load_experience): leaves by a grid, then the inner levels bottom-up, one launch per level
    if (update_max) return ssac_fail("ssac_per_assign: a priority refresh of more than 8192 rows");
    SSAC_LAUNCH(per_set_leaves_kernel, dim3(512), dim3(256), 0, st, a);
    for (int64_t first = cap / 2; first >= 1; first /= 2) {
        const int64_t count = first;
        const int grid = (int)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
        SSAC_LAUNCH(per_level_kernel, dim3(grid), dim3(256), 0, st, sum_tree, min_tree, first, count);
    }
    return ssac_check_launch("per_assign (bulk)");
}

extern "C" int ssac_per_sample(const double *sum_tree, const double *min_tree, int64_t cap, int64_t n_filled,
                               const double *u, int n_draws, double beta, int64_t *idx_out, double *weights_out,
                               void *stream) {
    if (!sum_tree || !min_tree || !u || !idx_out || !weights_out || cap <= 0 || (cap & (cap - 1)) || n_draws <= 0 ||
        n_filled < 2 || n_filled > cap)
        return ssac_fail("ssac_per_sample: bad arguments");
    SSAC_LAUNCH(per_sample_kernel, dim3(1), dim3(PER_THREADS), 0, (hipStream_t)stream, sum_tree, min_tree, cap, n_filled, u,
                n_draws, beta, idx_out, weights_out);
    return ssac_check_launch("per_sample");
}
