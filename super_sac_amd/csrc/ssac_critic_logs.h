// Log finalisation of a critic update (1 workgroup's worth of work): shared by critic_logs_kernel and by the merged
// weight-gradient launch, whose LAST workgroup to finish runs it (ssac_gemm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssac_hip.h"

struct CriticLogsArgs {
    const float *partials; int n_nets, tiles, n_rows; float denom;
    const float *sumsq; int n_ss; const ssac_adam_ctl *scale; float *logs;
    ssac_td_spec tds; float *td_logs; ssac_feed *feed;
};

// Every thread of the workgroup must call this (it contains barriers); the first 256 threads do the work with a
// fixed 256-stride / 4-wave summation order, so the result does not depend on the workgroup size.
// red: 12 floats of LDS.
__device__ __forceinline__ void critic_logs_body(const CriticLogsArgs &a, float *red) {
    const int tid = threadIdx.x;
    const bool act = tid < 256;
    const int n_rows = a.n_rows;
    // every global load of the workgroup is issued up front (one round trip instead of one per phase): the loss
    // partials, the gradient-norm partials, the ring slot, and -- below -- the TD targets, kept in registers for the
    // variance pass (4 per thread cover 1024 rows; longer batches re-read the rest)
    float sl = 0.f, se = 0.f, ss = 0.f;
    const int tot = a.n_nets * a.tiles;
    int slot = 0, w = 0;
    if (a.feed) { slot = (int)a.feed->dst[a.feed->log_slot_word]; w = a.feed->log_width; }
    if (act) {
        for (int i = tid; i < tot; i += 256) {
            sl += a.partials[2 * i];
            if (i / a.tiles == a.n_nets - 1) se += a.partials[2 * i + 1];
        }
        for (int i = tid; i < a.n_ss; i += 256) ss += a.sumsq[i];
    }
    if (a.tds.q_t && a.td_logs) {  // statistics of the targets the critic launch computed (td_target_kernel's logs)
        float s_td = 0.f, s_b = 0.f;
        const float alpha = a.tds.use_entropy ? expf(a.tds.log_alpha[0]) : 0.0f;
        float tv[4] = {0.f, 0.f, 0.f, 0.f};
        if (act) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = tid + 256 * u;
                if (b < n_rows) tv[u] = a.tds.td_out[b];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // (same per-thread order as a 256-stride loop)
                const int b = tid + 256 * u;
                if (b < n_rows) { s_td += tv[u]; s_b += a.tds.use_entropy ? alpha * a.tds.logp[b] : 0.0f; }
            }
            for (int b = tid + 1024; b < n_rows; b += 256) {
                s_td += a.tds.td_out[b];
                s_b += a.tds.use_entropy ? alpha * a.tds.logp[b] : 0.0f;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s_td += __shfl_xor(s_td, o, 64); s_b += __shfl_xor(s_b, o, 64); }
        if (act && (tid & 63) == 0) { red[tid >> 6] = s_td; red[4 + (tid >> 6)] = s_b; }
        __syncthreads();
        const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)n_rows;
        const float mb = (red[4] + red[5] + red[6] + red[7]) / (float)n_rows;
        __syncthreads();
        float sv = 0.f;
        if (act) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = tid + 256 * u;
                if (b < n_rows) { const float dlt = tv[u] - mean; sv += dlt * dlt; }
            }
            for (int b = tid + 1024; b < n_rows; b += 256) {
                const float dlt = a.tds.td_out[b] - mean;
                sv += dlt * dlt;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sv += __shfl_xor(sv, o, 64);
        if (act && (tid & 63) == 0) red[8 + (tid >> 6)] = sv;
        __syncthreads();
        if (tid == 0) {
            const float var = (red[8] + red[9] + red[10] + red[11]) / (float)(n_rows > 1 ? n_rows - 1 : 1);
            a.td_logs[0] = mean;
            a.td_logs[1] = sqrtf(var);
            a.td_logs[2] = mb;
        }
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sl += __shfl_xor(sl, o, 64); se += __shfl_xor(se, o, 64); ss += __shfl_xor(ss, o, 64);
    }
    if (act && (tid & 63) == 0) { red[tid >> 6] = sl; red[4 + (tid >> 6)] = se; red[8 + (tid >> 6)] = ss; }
    __syncthreads();
    if (tid == 0) {
        sl = red[0] + red[1] + red[2] + red[3];
        se = red[4] + red[5] + red[6] + red[7];
        ss = red[8] + red[9] + red[10] + red[11];
        if (a.n_nets > 0) {  // (n_nets == 0: the loss logs were written by ssac_critic_loss_bwd already)
            a.logs[0] += sl / (a.denom * (float)n_rows);   // losses/critic_overall_loss (accumulates over members)
            a.logs[1] = se / (float)n_rows;                // losses/last_member_critic_td_error
        }
        if (a.sumsq) a.logs[2] = sqrtf(ss) * (a.scale ? a.scale->clip_coef : 1.0f);
    }
    if (a.feed) {  // last launch of a captured update: publish the log block, advance the input ring
        __syncthreads();
        if (tid < w) a.feed->log_ring[(int64_t)slot * w + tid] = a.logs[tid];
        __syncthreads();
        if (tid == 0) a.feed->tick += 1;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Loss gradient folded into the weight-gradient launch (continuous critics, one head output).  What the TD target
// decides about the backward pass is ONE scalar per (net, row): dL/dq = -2 pw w (td - (pw q + pb)) / (denom B)
// (learning.py:90-98, 112).  Every workgroup of the merged weight-gradient launch evaluates the n_rows scalars of
// ITS net into LDS (a few KB of L2-resident reads) instead of a separate single-workgroup launch writing them to
// memory first; one workgroup per net also reduces the loss terms, and net slot 0's writes the TD targets out.
// ------------------------------------------------------------------------------------------------------------
struct LossFoldArgs {
    const float *q;           // (n_nets x n_rows) head outputs of the online critics; null = fold off
    const float *td;          // TD targets (n_rows), or null with `tds` set (evaluated here, td_target_kernel order)
    ssac_td_spec tds;
    const float *weight;      // per-row loss weights, or null
    const ssac_popart *popart; int pop; float denom;
    float *partials;          // [n_nets][2]: sum_b w err^2, sum_b err
    int n_rows;
};

__device__ __forceinline__ float ssac_lazy_td(const ssac_td_spec &t, int b, int n_rows, float alpha) {
    float mq = t.q_t[b];
    for (int j = 1; j < t.n_sel; ++j) mq = fminf(mq, t.q_t[(int64_t)j * n_rows + b]);
    const float bonus = t.use_entropy ? alpha * t.logp[b] : 0.0f;
    const float val = mq - bonus;
    return t.rew[b] + t.gamma * (1.0f - t.done[b]) * val;
}

// Every thread of the workgroup calls this (barriers inside when `stats`; `stats` must be workgroup-uniform).
// tab: n_rows floats of LDS; red: 2 * (blockDim.x / 64) floats of LDS.  The caller synchronises before reading tab.
__device__ __forceinline__ void loss_fold_table(const LossFoldArgs &a, int e, float *tab, bool stats, float *red) {
    const int tid = threadIdx.x, n_rows = a.n_rows;
    const float pw = (a.popart && a.pop) ? a.popart->w : 1.0f;
    const float pb = (a.popart && a.pop) ? a.popart->b : 0.0f;
    const float gscale = -2.0f * pw / (a.denom * (float)n_rows);
    const float alpha = (a.tds.q_t && a.tds.use_entropy) ? expf(a.tds.log_alpha[0]) : 0.0f;
    const float *q = a.q + (int64_t)e * n_rows;
    float sl = 0.0f, se = 0.0f;
    for (int b = tid; b < n_rows; b += blockDim.x) {
        const float t = a.tds.q_t ? ssac_lazy_td(a.tds, b, n_rows, alpha) : a.td[b];
        if (stats && e == 0 && a.tds.q_t) a.tds.td_out[b] = t;
        const float w = a.weight ? a.weight[b] : 1.0f;
        const float err = t - (pw * q[b] + pb);
        tab[b] = gscale * w * err;
        sl += w * err * err;
        se += err;
    }
    if (stats) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sl += __shfl_xor(sl, o, 64); se += __shfl_xor(se, o, 64); }
        const int nw = blockDim.x >> 6;
        if ((tid & 63) == 0) { red[tid >> 6] = sl; red[nw + (tid >> 6)] = se; }
        __syncthreads();
        if (tid == 0) {
            float tl = 0.0f, te = 0.0f;
            for (int w = 0; w < nw; ++w) { tl += red[w]; te += red[nw + w]; }
            a.partials[2 * e] = tl;
            a.partials[2 * e + 1] = te;
        }
    }
}
