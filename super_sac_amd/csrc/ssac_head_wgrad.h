// Head-layer weight gradient + Adam/Polyak (out_dim <= 16, VALU): shared by head_wgrad_kernel (ssac_fused.hip)
// and by the merged weight-gradient launch (ssac_gemm.hip), where it runs as extra workgroups beside the fc2/fc1 GEMM tiles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssac_hip.h"

struct HeadWgradArgs {
    float *params; int64_t net_stride; int hidden, out_dim; int64_t off_w, off_b;
    const int32_t *ids; const float *H2, *DQ; int n_rows;
    float *am, *av; const ssac_adam_ctl *ctl; float *grads, *sumsq; int64_t sumsq_stride;
    float *target; float tau;
    // with_b2 (single-output heads over a ReLU layer): this workgroup also takes the fc2 BIAS gradient of its 64 columns,
    // db2[k] = sum_m dz2[m][k] = W3[k] * sum_m dq[m] [h2[m][k] > 0] -- it reads those h2 values anyway and has ~15 us of
    // slack in the merged launch, while the GEMM tile that used to carry the bias was the slowest of its net.  W3[k] is
    // the value this very thread reads before it updates it (nobody else writes W3[k]).
    int with_b2; int64_t off_b2;   // offset of b2 in a net's parameter block
};

// dW3[o][k] = sum_m dq[m][o] h2[m][k];  db3[o] = sum_m dq[m][o].  Workgroup (bx, e): columns [COLS bx, COLS bx + COLS) of
// net e; COLS * GROUPS threads = COLS columns x GROUPS row groups.  lds: >= GROUPS*COLS + GROUPS floats
// (2*GROUPS*COLS + GROUPS with b2_w3s).  COLS = 64 everywhere but in the latency variant of the merged weight-gradient
// launch (wgrad_small_pair_kernel, ssac_gemm.hip), whose 16-column head workgroups walk 16 rows per thread instead of 32.
// Gradient-norm slots: a net's head layer owns hidden / 16 slots (ssac_head_wgrad_tiles); a 64-column workgroup writes
// the first of its four and zeroes the rest, so the slots sum to the same value whichever variant ran last.
// pol / tau: whether and how the Polyak target is updated (the caller resolves a late-bound request, ssac_late_polyak)
constexpr int SSAC_HEAD_SLOT_COLS = 16;
// lds_floats: how many floats of `lds` this call may use.  With room for out_dim x (GROUPS*COLS + GROUPS) of them a head of
// SEVERAL outputs (an actor's 2A) is done in one pass (below) instead of one pass per output.
template <int GROUPS, int COLS = 64>
__device__ __forceinline__ void head_wgrad_body(const HeadWgradArgs &a, float *lds, int bx, int e,
                                                const float *dq_override, bool pol, float tau, int lds_floats = 0);
template <int GROUPS, int COLS = 64>
__device__ __forceinline__ void head_wgrad_body(const HeadWgradArgs &a, float *lds, int bx, int e,
                                                const float *dq_override = nullptr) {
    head_wgrad_body<GROUPS, COLS>(a, lds, bx, e, dq_override, a.target != nullptr, a.tau);
}
template <int GROUPS, int COLS>
__device__ __forceinline__ void head_wgrad_body(const HeadWgradArgs &a, float *lds, int bx, int e,
                                                const float *dq_override, bool pol, float tau, int lds_floats) {
    static_assert(COLS == 64 || COLS == 32 || COLS == 16, "64-column (wave-wide), 32- or 16-column head workgroups");
    float *red = lds;                   // [GROUPS][COLS]
    float *redb = lds + GROUPS * COLS;  // [GROUPS]
    const int tid = threadIdx.x, kk = tid % COLS, mg = tid / COLS;
    const int k = bx * COLS + kk;
    const int hidden = a.hidden, out_dim = a.out_dim, n_rows = a.n_rows;
    const int net = a.ids ? a.ids[e] : e;
    const int64_t base = (int64_t)net * a.net_stride;
    const bool kok = k < hidden;
    const float *h2 = a.H2 + (int64_t)e * n_rows * hidden + (kok ? k : 0);
    // dq_override: this net's dL/dq in LDS (out_dim 1; the merged launch with the loss gradient folded in)
    const float *dq = dq_override ? dq_override : a.DQ + (int64_t)e * n_rows * out_dim;
    float ss = 0.0f;
    const bool with_b2 = a.with_b2 != 0 && out_dim == 1;
    constexpr int NT = GROUPS * COLS, OMAX = 16;
    if (out_dim > 1 && out_dim <= OMAX && out_dim * (NT + GROUPS) <= lds_floats) {
        // SEVERAL outputs (the actor's 2A = 12 at the metric shape) in ONE pass over the rows.  The per-output loop below
        // re-reads the h2 column and pays two barriers and a dependent round trip for the optimizer state PER OUTPUT --
        // 12 x ~2 us for the actor, the longest workgroup of its weight-gradient launch by a factor of two.  Here a thread
        // reads its rows of h2 once, keeps one accumulator per output, all partials meet in LDS behind one barrier, and
        // out_dim x COLS threads finish one weight each: one round trip for all the optimizer state.  Per element the
        // same products are added in the same order as below (rows m = mg, mg + GROUPS, ...; then the row groups in
        // index order): the gradients are bit-identical, only the gradient-norm partial is summed in another order.
        float acc[OMAX], accb[OMAX];
#pragma unroll
        for (int o = 0; o < OMAX; ++o) { acc[o] = 0.0f; accb[o] = 0.0f; }
        // dq (n_rows x out_dim, 24 KB at the metric shape) is the same for every column: it goes through LDS when it fits
        // (one coalesced pass instead of out_dim dependent loads per row and thread); the h2 column values of 8 rows are
        // requested together, the first 8 before the staging barrier
        const bool dq_lds = n_rows * out_dim <= lds_floats;
        constexpr int RU = 8;
        float hv[RU];
        auto load_h = [&](int m0_) {
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int m = m0_ + u * GROUPS;
                hv[u] = m < n_rows ? h2[(int64_t)m * hidden] : 0.0f;
            }
        };
        load_h(mg);
        if (dq_lds) {
            for (int i = tid; i < n_rows * out_dim; i += NT) lds[i] = dq[i];
            __syncthreads();
        }
        // (two instantiations, one per address space: a pointer that may be LDS or global turns every read into a flat
        // load with its own wait)
        auto run = [&](auto dsrc) {
            for (int m0_ = mg; m0_ < n_rows; m0_ += RU * GROUPS) {
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int m = m0_ + u * GROUPS;
                    if (m < n_rows) {
#pragma unroll
                        for (int o = 0; o < OMAX; ++o)
                            if (o < out_dim) {
                                const float d = dsrc[m * out_dim + o];
                                acc[o] += d * hv[u];
                                accb[o] += d;
                            }
                    }
                }
                if (m0_ + RU * GROUPS < n_rows) load_h(m0_ + RU * GROUPS);
            }
        };
        if (dq_lds) run((__attribute__((address_space(3))) const float *)lds);
        else run(dq);
        float *redb2 = lds + out_dim * NT;   // [out_dim][GROUPS]
        __syncthreads();
#pragma unroll
        for (int o = 0; o < OMAX; ++o)
            if (o < out_dim) {
                lds[(o * GROUPS + mg) * COLS + kk] = acc[o];
                if (kk == 0) redb2[o * GROUPS + mg] = accb[o];
            }
        __syncthreads();
        const ssac_adam_ctl c = a.grads ? ssac_adam_ctl{} : *a.ctl;
        auto apply = [&](int64_t i, float g_) {
            if (a.grads) {
                a.grads[i] = g_;
            } else {
                float g2 = g_;
                const float p = a.params[i];
                if (c.weight_decay != 0.0f) g2 = g2 + c.weight_decay * p;
                float mm = a.am[i], vv = a.av[i];
                mm = mm + (1.0f - c.beta1) * (g2 - mm);
                vv = vv * c.beta2 + (1.0f - c.beta2) * g2 * g2;
                const float pn = p - c.step_size * (mm / (sqrtf(vv) / c.bc2_sqrt + c.eps));
                a.am[i] = mm; a.av[i] = vv; a.params[i] = pn;
                if (pol) a.target[i] = a.target[i] * (1.0f - tau) + pn * tau;
            }
        };
        for (int t = tid; t < out_dim * COLS; t += NT) {
            const int o = t / COLS, k2 = bx * COLS + (t - o * COLS);
            float gr = 0.0f;
#pragma unroll
            for (int q = 0; q < GROUPS; ++q) gr += lds[(o * GROUPS + q) * COLS + (t - o * COLS)];
            if (k2 < hidden) {
                ss += gr * gr;
                apply(base + a.off_w + (int64_t)o * hidden + k2, gr);
            }
        }
        if (bx == 0 && tid < out_dim) {   // bias gradients db3[o] = sum_m dq[m][o]
            float gb = 0.0f;
#pragma unroll
            for (int q = 0; q < GROUPS; ++q) gb += redb2[tid * GROUPS + q];
            ss += gb * gb;
            apply(base + a.off_b + tid, gb);
        }
        if (a.sumsq) {   // gradient-norm partial: wave sums, then the waves in index order
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            __syncthreads();   // (the partials have been consumed)
            if ((tid & 63) == 0) lds[tid >> 6] = ss;
            __syncthreads();
            if (tid == 0) {
                float tot = 0.0f;
                for (int w = 0; w < NT / 64; ++w) tot += lds[w];
                constexpr int SPAN = COLS / SSAC_HEAD_SLOT_COLS;
                const int n_slots = (hidden + SSAC_HEAD_SLOT_COLS - 1) / SSAC_HEAD_SLOT_COLS;
                float *slot = a.sumsq + (int64_t)e * a.sumsq_stride + bx * SPAN;
                __hip_atomic_store(slot, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int i = 1; i < SPAN && bx * SPAN + i < n_slots; ++i)
                    __hip_atomic_store(slot + i, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    // one output row o at a time: out_dim is small (1 for continuous critics), rows are the long axis
    float *reds = redb + GROUPS;      // [GROUPS][COLS] (with_b2)
    for (int o = 0; o < out_dim; ++o) {
        float acc = 0.0f, accb = 0.0f, accs = 0.0f;
        int m = mg;
        for (; m + 7 * GROUPS < n_rows; m += 8 * GROUPS) {
            float hv[8], dv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                hv[u] = h2[(int64_t)(m + u * GROUPS) * hidden];
                dv[u] = dq[(int64_t)(m + u * GROUPS) * out_dim + o];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc += dv[u] * hv[u]; accb += dv[u]; accs += hv[u] > 0.0f ? dv[u] : 0.0f; }
        }
        for (; m < n_rows; m += GROUPS) {
            const float d = dq[(int64_t)m * out_dim + o];
            const float h = h2[(int64_t)m * hidden];
            acc += d * h;
            accb += d;
            accs += h > 0.0f ? d : 0.0f;
        }
        __syncthreads();
        red[mg * COLS + kk] = acc;
        if (with_b2) reds[mg * COLS + kk] = accs;
        if (kk == 0) redb[mg] = accb;
        __syncthreads();
        if (mg == 0) {
            float gr = 0.0f;
#pragma unroll
            for (int q = 0; q < GROUPS; ++q) gr += red[q * COLS + kk];
            const ssac_adam_ctl c = a.grads ? ssac_adam_ctl{} : *a.ctl;
            auto apply = [&](int64_t i, float g_) {
                if (a.grads) {
                    a.grads[i] = g_;
                } else {
                    float g2 = g_;
                    const float p = a.params[i];
                    if (c.weight_decay != 0.0f) g2 = g2 + c.weight_decay * p;
                    float mm = a.am[i], vv = a.av[i];
                    mm = mm + (1.0f - c.beta1) * (g2 - mm);
                    vv = vv * c.beta2 + (1.0f - c.beta2) * g2 * g2;
                    const float pn = p - c.step_size * (mm / (sqrtf(vv) / c.bc2_sqrt + c.eps));
                    a.am[i] = mm; a.av[i] = vv; a.params[i] = pn;
                    if (pol) a.target[i] = a.target[i] * (1.0f - tau) + pn * tau;
                }
            };
            if (kok) {
                ss += gr * gr;
                const float w3_old = with_b2 ? a.params[base + a.off_w + k] : 0.0f;   // (before this thread's own update)
                apply(base + a.off_w + (int64_t)o * hidden + k, gr);
                if (with_b2) {
                    float gs = 0.0f;
#pragma unroll
                    for (int q = 0; q < GROUPS; ++q) gs += reds[q * COLS + kk];
                    const float gb2 = w3_old * gs;
                    ss += gb2 * gb2;
                    apply(base + a.off_b2 + k, gb2);
                }
            }
            if (bx == 0 && kk == 0) {  // bias gradient db3[o] = sum_m dq[m][o]
                float gb = 0.0f;
#pragma unroll
                for (int q = 0; q < GROUPS; ++q) gb += redb[q];
                ss += gb * gb;
                apply(base + a.off_b + o, gb);
            }
        }
    }
    if (a.sumsq && tid < 64) {  // the threads of row group 0 hold every contribution, and they all sit in wave 0
        if (mg != 0) ss = 0.0f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        if (tid == 0) {
            constexpr int SPAN = COLS / SSAC_HEAD_SLOT_COLS;   // slots this workgroup's columns cover
            const int n_slots = (hidden + SSAC_HEAD_SLOT_COLS - 1) / SSAC_HEAD_SLOT_COLS;
            float *slot = a.sumsq + (int64_t)e * a.sumsq_stride + bx * SPAN;
            __hip_atomic_store(slot, ss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = 1; i < SPAN && bx * SPAN + i < n_slots; ++i)
                __hip_atomic_store(slot + i, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
