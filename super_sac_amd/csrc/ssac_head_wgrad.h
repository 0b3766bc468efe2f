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
};

// dW3[o][k] = sum_m dq[m][o] h2[m][k];  db3[o] = sum_m dq[m][o].  Workgroup (bx, e): columns [64 bx, 64 bx + 64) of
// net e; 64 * GROUPS threads = 64 columns x GROUPS row groups.  lds: >= GROUPS*64 + GROUPS floats.
// pol / tau: whether and how the Polyak target is updated (the caller resolves a late-bound request, ssac_late_polyak)
template <int GROUPS>
__device__ __forceinline__ void head_wgrad_body(const HeadWgradArgs &a, float *lds, int bx, int e,
                                                const float *dq_override, bool pol, float tau);
template <int GROUPS>
__device__ __forceinline__ void head_wgrad_body(const HeadWgradArgs &a, float *lds, int bx, int e,
                                                const float *dq_override = nullptr) {
    head_wgrad_body<GROUPS>(a, lds, bx, e, dq_override, a.target != nullptr, a.tau);
}
template <int GROUPS>
__device__ __forceinline__ void head_wgrad_body(const HeadWgradArgs &a, float *lds, int bx, int e,
                                                const float *dq_override, bool pol, float tau) {
    float *red = lds;                 // [GROUPS][64]
    float *redb = lds + GROUPS * 64;  // [GROUPS]
    const int tid = threadIdx.x, kk = tid & 63, mg = tid >> 6;
    const int k = bx * 64 + kk;
    const int hidden = a.hidden, out_dim = a.out_dim, n_rows = a.n_rows;
    const int net = a.ids ? a.ids[e] : e;
    const int64_t base = (int64_t)net * a.net_stride;
    const bool kok = k < hidden;
    const float *h2 = a.H2 + (int64_t)e * n_rows * hidden + (kok ? k : 0);
    // dq_override: this net's dL/dq in LDS (out_dim 1; the merged launch with the loss gradient folded in)
    const float *dq = dq_override ? dq_override : a.DQ + (int64_t)e * n_rows * out_dim;
    float ss = 0.0f;
    // one output row o at a time: out_dim is small (1 for continuous critics), rows are the long axis
    for (int o = 0; o < out_dim; ++o) {
        float acc = 0.0f, accb = 0.0f;
        int m = mg;
        for (; m + 7 * GROUPS < n_rows; m += 8 * GROUPS) {
            float hv[8], dv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                hv[u] = h2[(int64_t)(m + u * GROUPS) * hidden];
                dv[u] = dq[(int64_t)(m + u * GROUPS) * out_dim + o];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc += dv[u] * hv[u]; accb += dv[u]; }
        }
        for (; m < n_rows; m += GROUPS) {
            const float d = dq[(int64_t)m * out_dim + o];
            acc += d * h2[(int64_t)m * hidden];
            accb += d;
        }
        __syncthreads();
        red[mg * 64 + kk] = acc;
        if (kk == 0) redb[mg] = accb;
        __syncthreads();
        if (mg == 0) {
            float gr = 0.0f;
#pragma unroll
            for (int q = 0; q < GROUPS; ++q) gr += red[q * 64 + kk];
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
                apply(base + a.off_w + (int64_t)o * hidden + k, gr);
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
    if (a.sumsq && mg == 0) {  // wave 0 holds every contribution
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        if (kk == 0) __hip_atomic_store(a.sumsq + (int64_t)e * a.sumsq_stride + bx, ss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
