// Prioritised replay on the device (reference super_sac/replay.py:140-190 ReplayBuffer.sample / update_priorities,
// :207-353 SegmentTree / SumSegmentTree / MinSegmentTree).
//
// The trees are the reference's: implicit binary heaps of float64 over the next power of two >= capacity, node 1 the
// root, leaves at [cap, 2 cap).  They live in HBM, so that the AFBC / PER update path never leaves the device: the
// host only draws the B uniforms from numpy's GLOBAL generator (replay.py:166: the stream the reference consumes) and
// uploads them; total mass, prefix-sum descent, importance weights and the priority refresh are kernels.  The float64
// host trees of replay.PrioritySampler stay as the checker the tests compare against.
//
// Leaves are priority^alpha CORRECTLY ROUNDED (cr_pow below) in the precision the reference computes them in.
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

// ---------------------------------------------------------------------------------------------------------------
// priority^alpha, CORRECTLY ROUNDED.  The reference computes the leaves with numpy's power (replay.py:183-190): float64
// for host float64 / Python-float priorities, float32 -- with the exponent cast to float32 -- for the float32
// priorities adjust_priorities hands over (learning_utils.py:294).  numpy's power is not one function: its SIMD loops
// (SVML) and the libm behind its scalar path differ from each other in the last bit for ~20 % of float32 arguments on
// this very host, and a device libm differs from both.  What all of them approximate is the correctly rounded value,
// and that is what the trees hold here, by construction: x^y = exp(y log x) evaluated in double-double arithmetic
// (~100 bits; error-free transformations with fma) and rounded once -- to float64, or to float32 and widened.  A leaf
// then never depends on which pow a library ships, and the index draw of a given mass is reproducible bit for bit
// (tests/test_hip_kernels.py checks the leaves against an exact decimal evaluation with np.array_equal).
// ---------------------------------------------------------------------------------------------------------------
struct dd { double hi, lo; };
__device__ __forceinline__ dd dd_quick(double a, double b) { const double s = a + b; return {s, b - (s - a)}; }
__device__ __forceinline__ dd dd_two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd dd_two_prod(double a, double b) { const double p = a * b; return {p, __builtin_fma(a, b, -p)}; }
__device__ __forceinline__ dd dd_add(dd a, dd b) {
    dd s = dd_two_sum(a.hi, b.hi);
    const dd t = dd_two_sum(a.lo, b.lo);
    s = dd_quick(s.hi, s.lo + t.hi);
    return dd_quick(s.hi, s.lo + t.lo);
}
__device__ __forceinline__ dd dd_add_d(dd a, double b) {
    const dd s = dd_two_sum(a.hi, b);
    return dd_quick(s.hi, s.lo + a.lo);
}
__device__ __forceinline__ dd dd_mul(dd a, dd b) {
    const dd p = dd_two_prod(a.hi, b.hi);
    return dd_quick(p.hi, p.lo + (a.hi * b.lo + a.lo * b.hi));
}
__device__ __forceinline__ dd dd_mul_d(dd a, double b) {
    const dd p = dd_two_prod(a.hi, b);
    return dd_quick(p.hi, p.lo + a.lo * b);
}
// exp of a double-double: x = k ln2 + r, r / 512 through a degree-9 Taylor polynomial of expm1 (|r| / 512 < 7e-4: the
// first neglected term is below 1e-34), then nine times s <- 2 s + s^2, then 2^k
__device__ dd dd_exp(dd x) {
    const dd LN2 = {0.693147180559945286, 2.319046813846299558e-17};
    const double k = rint(x.hi * 1.4426950408889634);
    dd r = dd_add(x, dd_mul_d(LN2, -k));
    r.hi *= (1.0 / 512.0); r.lo *= (1.0 / 512.0);
    const double inv_fact[10] = {1.0, 1.0, 0.5, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880};
    // Horner on expm1(r) / r = sum_{n >= 1} r^(n-1) / n!   (coefficients below 1/6 as doubles: their relative error
    // 1e-16 meets a term of relative size < 1e-7, far below the 2^-104 the sum needs)
    dd acc = {inv_fact[9], 0.0};
#pragma unroll
    for (int n = 8; n >= 3; --n) acc = dd_add_d(dd_mul(acc, r), inv_fact[n]);
    acc = dd_add(dd_mul(acc, r), dd{0.5, 0.0});
    acc = dd_add(dd_mul(acc, r), dd{1.0, 0.0});
    dd s = dd_mul(acc, r);   // expm1(r)
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const dd s2 = dd_mul(s, s);
        s = dd_add(dd{2.0 * s.hi, 2.0 * s.lo}, s2);
    }
    const dd e = dd_add(dd{1.0, 0.0}, s);
    const int ki = (int)k;
    return {ldexp(e.hi, ki), ldexp(e.lo, ki)};
}
// log of a double as a double-double: one Newton step on exp, y = y0 + (x exp(-y0) - 1)
__device__ dd dd_log(double x) {
    const double y0 = log(x);
    const dd e = dd_exp(dd{-y0, 0.0});
    const dd t = dd_add_d(dd_mul_d(e, x), -1.0);
    return dd_add_d(t, y0);
}
// x^y for x > 0, correctly rounded to float64 (f32 = false) or to float32 and widened (f32 = true; y is then the
// float32-rounded exponent, as numpy's float32 power sees it)
__device__ double cr_pow(double x, double y, bool f32) {
    if (f32) y = (double)(float)y;
    if (y == 0.0 || x == 1.0) return 1.0;
    const dd r = dd_exp(dd_mul_d(dd_log(x), y));
    const dd n = dd_quick(r.hi, r.lo);
    if (!f32) return n.hi;
    // one rounding to float32: hi is already a rounding of the true value, so a hi that sits exactly on a float32 tie
    // takes the direction of lo
    unsigned long long b = (unsigned long long)__double_as_longlong(n.hi);
    if ((b & 0x1FFFFFFFull) == 0x10000000ull && n.lo != 0.0) b = n.lo > 0.0 ? b + 1 : b - 1;
    return (double)(float)__longlong_as_double((long long)b);
}

struct AssignArgs {
    double *sum_tree, *min_tree; int64_t cap;
    const int64_t *rows; int n;
    const void *prio; int prio_f64;       // null: the current max priority
    double alpha; double *max_priority;   // [0] the largest priority seen (replay.py:190), [1] != 0: it came from a FLOAT32
                                          // array -- the reference's _max_priority is then an np.float32, and the power
                                          // of a row pushed at max priority (replay.py:156-161) is a float32 power
    int update_max; int64_t n_filled;
    int *err;                             // pinned host word: 1 = priority <= 0, 2 = row out of range
    int32_t *win;                         // cap entries, all -1 between calls: which batch entry writes a leaf (the LAST
                                          // one that names it, as numpy's fancy assignment `value[idxs] = val` does)
};

// leaves <- prio^alpha, then every ancestor level by level (each thread re-derives the ancestor of ITS rows: rows
// sharing an ancestor write the same value)
__global__ __launch_bounds__(PER_THREADS) void per_assign_kernel(AssignArgs a) {
    const int tid = threadIdx.x;
    const double maxp = a.max_priority[0];
    const bool max_f32 = a.max_priority[1] != 0.0;
    double seen = 0.0;
    // The reference asserts BEFORE it touches the trees (replay.py:183-187): validate the whole batch first; one bad entry
    // and nothing is written -- not a leaf, not max_priority -- only the error word (raised by the next host call).
    int bad_p = 0, bad_r = 0;
    for (int i = tid; i < a.n; i += PER_THREADS) {
        const int64_t r = a.rows[i];
        if (a.prio) {
            const double p = a.prio_f64 ? reinterpret_cast<const double *>(a.prio)[i] : (double)reinterpret_cast<const float *>(a.prio)[i];
            bad_p |= !(p > 0.0);
        }
        bad_r |= r < 0 || r >= a.cap || (a.update_max && r >= a.n_filled);
    }
    bad_p = __syncthreads_or(bad_p);
    bad_r = __syncthreads_or(bad_r);
    if (bad_p || bad_r) {
        if (tid == 0) *a.err = bad_p ? 1 : 2;
        return;
    }
    for (int i = tid; i < a.n; i += PER_THREADS) atomicMax(a.win + a.rows[i], i);
    __syncthreads();
    for (int i = tid; i < a.n; i += PER_THREADS) {
        const int64_t r = a.rows[i];
        double p = maxp;
        if (a.prio) p = a.prio_f64 ? reinterpret_cast<const double *>(a.prio)[i] : (double)reinterpret_cast<const float *>(a.prio)[i];
        seen = fmax(seen, p);
        if (a.win[r] != i) continue;   // a later entry of the batch names the same row
        const double v = cr_pow(p, a.alpha, a.prio ? !a.prio_f64 : max_f32);
        a.sum_tree[a.cap + r] = v;
        a.min_tree[a.cap + r] = v;
    }
    __syncthreads();
    for (int i = tid; i < a.n; i += PER_THREADS) a.win[a.rows[i]] = -1;
    if (a.update_max) {   // _max_priority = max(_max_priority, max(priorities)): positive doubles order like their bits
        for (int o = 32; o > 0; o >>= 1) seen = fmax(seen, __shfl_xor(seen, o, 64));
        if ((tid & 63) == 0 && seen > 0.0)
            atomicMax(reinterpret_cast<unsigned long long *>(a.max_priority), (unsigned long long)__double_as_longlong(seen));
        __syncthreads();
        // max(self._max_priority, np.max(priorities)) keeps the OLD object unless the new maximum is larger: the dtype of
        // the maximum changes with it
        if (tid == 0 && a.max_priority[0] > maxp) a.max_priority[1] = a.prio_f64 ? 0.0 : 1.0;
    }
    for (int64_t span = 2; span <= a.cap; span <<= 1) {   // span = leaves under a node of this level
        __syncthreads();
        for (int i = tid; i < a.n; i += PER_THREADS) {
            const int64_t node = (a.cap + a.rows[i]) / span;
            a.sum_tree[node] = a.sum_tree[2 * node] + a.sum_tree[2 * node + 1];
            a.min_tree[node] = fmin(a.min_tree[2 * node], a.min_tree[2 * node + 1]);
        }
    }
}

__global__ void per_set_leaves_kernel(AssignArgs a) {
    const double maxp = a.max_priority[0];
    const bool max_f32 = a.max_priority[1] != 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = a.rows[i];
        double p = maxp;
        if (a.prio) p = a.prio_f64 ? reinterpret_cast<const double *>(a.prio)[i] : (double)reinterpret_cast<const float *>(a.prio)[i];
        if (a.prio && !(p > 0.0)) { *a.err = 1; continue; }
        if (r < 0 || r >= a.cap) { *a.err = 2; continue; }
        const double v = cr_pow(p, a.alpha, a.prio ? !a.prio_f64 : max_f32);
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
    // A priority refresh (update_max) of ANY size goes through the one-workgroup kernel: it validates before it writes and
    // resolves duplicated rows, and its loops stride over n (the reference accepts any batch size; 65 536 rows = 64 rows per
    // thread and level).  Pushes of up to 8 192 rows too; larger ones are bulk loads.
    if (update_max || n <= 8 * PER_THREADS) {
        SSAC_LAUNCH(per_assign_kernel, dim3(1), dim3(PER_THREADS), 0, st, a);
        return ssac_check_launch("per_assign");
    }
    // bulk (load_experience): leaves by a grid, then the inner levels bottom-up, one launch per level
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
