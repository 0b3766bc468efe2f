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
    if (a.tds.q_t && a.td_logs) {  // statistics of the targets the critic launch computed (td_target_kernel's logs)
        float s_td = 0.f, s_b = 0.f;
        const float alpha = a.tds.use_entropy ? expf(a.tds.log_alpha[0]) : 0.0f;
        if (act)
            for (int b = tid; b < n_rows; b += 256) {
                s_td += a.tds.td_out[b];
                s_b += a.tds.use_entropy ? alpha * a.tds.logp[b] : 0.0f;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s_td += __shfl_xor(s_td, o, 64); s_b += __shfl_xor(s_b, o, 64); }
        if (act && (tid & 63) == 0) { red[tid >> 6] = s_td; red[4 + (tid >> 6)] = s_b; }
        __syncthreads();
        const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)n_rows;
        const float mb = (red[4] + red[5] + red[6] + red[7]) / (float)n_rows;
        __syncthreads();
        float sv = 0.f;
        if (act)
            for (int b = tid; b < n_rows; b += 256) {
                const float dlt = a.tds.td_out[b] - mean;
                sv += dlt * dlt;
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
    float sl = 0.f, se = 0.f, ss = 0.f;
    const int tot = a.n_nets * a.tiles;
    if (act) {
        for (int i = tid; i < tot; i += 256) {
            sl += a.partials[2 * i];
            if (i / a.tiles == a.n_nets - 1) se += a.partials[2 * i + 1];
        }
        for (int i = tid; i < a.n_ss; i += 256) ss += a.sumsq[i];
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
        const int slot = (int)a.feed->dst[a.feed->log_slot_word];
        const int w = a.feed->log_width;
        if (tid < w) a.feed->log_ring[(int64_t)slot * w + tid] = a.logs[tid];
        __syncthreads();
        if (tid == 0) a.feed->tick += 1;
    }
}
