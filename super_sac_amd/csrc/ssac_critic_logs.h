// Log finalisation of a critic update (1 workgroup's worth of work): shared by critic_logs_kernel and by the merged
// weight-gradient launch, whose LAST workgroup to finish runs it (ssac_gemm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssac_hip.h"
#include "ssac_internal.h"

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
    const float mq = ssac_td_min_q(t, b, n_rows);
    const float bonus = t.use_entropy ? alpha * t.logp[b] : 0.0f;
    const float val = mq - bonus;
    return t.rew[b] + t.gamma * (1.0f - t.done[b]) * val;
}

// Every thread of the workgroup calls this (barriers inside when `stats`; `stats` must be workgroup-uniform).
// tab: n_rows floats of LDS; red: 2 * (blockDim.x / 64) floats of LDS.  The caller synchronises before reading tab.
// write_td: this workgroup also stores the (lazily evaluated) TD targets to tds.td_out -- the caller's td_target tensor
// and the input of log_fold_td_stats; exactly one workgroup of a launch does that.
__device__ __forceinline__ void loss_fold_table(const LossFoldArgs &a, int e, float *tab, bool stats, float *red,
                                                bool write_td) {
    const int tid = threadIdx.x, n_rows = a.n_rows;
    const float pw = (a.popart && a.pop) ? a.popart->w : 1.0f;
    const float pb = (a.popart && a.pop) ? a.popart->b : 0.0f;
    const float gscale = -2.0f * pw / (a.denom * (float)n_rows);
    const float alpha = (a.tds.q_t && a.tds.use_entropy) ? expf(a.tds.log_alpha[0]) : 0.0f;
    const float *q = a.q + (int64_t)e * n_rows;
    float sl = 0.0f, se = 0.0f;
    for (int b = tid; b < n_rows; b += blockDim.x) {
        const float t = a.tds.q_t ? ssac_lazy_td(a.tds, b, n_rows, alpha) : a.td[b];
        if (write_td && a.tds.q_t) a.tds.td_out[b] = t;
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
            // (agent-scope stores: with the logs folded into this launch the reader is a workgroup on another XCD)
            __hip_atomic_store(a.partials + 2 * e, tl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.partials + 2 * e + 1, te, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The same table in two steps for the GEMM tiles (no statistics, no TD store): the first pass's operands (row b =
// threadIdx.x) are REQUESTED before the tile's first operand chunk and the table is finished while that chunk is in
// flight -- vector-memory returns are in order, so loads issued behind the operand chunk would wait for it.
// Same arithmetic, operation by operation, as loss_fold_table.
struct LossFoldRegs { float q, t0, t1, t2, t3, lp, rew, done, w, la; };

__device__ __forceinline__ void loss_fold_issue(const LossFoldArgs &a, int e, LossFoldRegs &r) {
    const int b = threadIdx.x, n_rows = a.n_rows;
    // (named scalars, one unconditional assignment of the struct at the end: with conditional stores into the fields the
    // compiler merged two of them into ONE store at a run-time offset -- i.e. the struct went to scratch memory, 12 bytes
    // per lane in every kernel that inlines this)
    float q = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, lp = 0.f, rew = 0.f, done = 0.f, w = 1.f, la = 0.f;
#if defined(SSAC_LAB) && defined(SSAC_EXP_DQ_READY)
    // (measurement build only, WRONG values: the bound of "dL/dq computed once, elsewhere" -- a GEMM workgroup reads ONE
    //  float per row instead of evaluating the TD target: round-5 review, item 1(a))
    if (b < n_rows) q = a.q[(int64_t)e * n_rows + b];
    r = LossFoldRegs{q, t0, t1, t2, t3, lp, rew, done, w, la};
    return;
#endif
    const bool lazy = a.tds.q_t != nullptr;
    if (lazy && a.tds.use_entropy) la = a.tds.log_alpha[0];
    if (b < n_rows) {
        q = a.q[(int64_t)e * n_rows + b];
        if (a.weight) w = a.weight[b];
        // (the first four entries of the [n_sel x n_parts] list: two slots, or two slots of two partials each)
        const int ne = lazy ? a.tds.n_sel * (a.tds.n_parts > 1 ? a.tds.n_parts : 1) : 1;
        const float *tp = lazy ? a.tds.q_t : a.td;
        t0 = tp[b];
        if (ne > 1) t1 = tp[(int64_t)n_rows + b];
        if (ne > 2) t2 = tp[(int64_t)2 * n_rows + b];
        if (ne > 3) t3 = tp[(int64_t)3 * n_rows + b];
        if (lazy) {
            if (a.tds.use_entropy) lp = a.tds.logp[b];
            rew = a.tds.rew[b];
            done = a.tds.done[b];
        }
    }
    r = LossFoldRegs{q, t0, t1, t2, t3, lp, rew, done, w, la};
}

// the caller synchronises (LDS hand-off) before reading tab
__device__ __forceinline__ void loss_fold_finish(const LossFoldArgs &a, int e, const LossFoldRegs &r, float *tab) {
    const int tid = threadIdx.x, n_rows = a.n_rows;
    const float pw = (a.popart && a.pop) ? a.popart->w : 1.0f;
    const float pb = (a.popart && a.pop) ? a.popart->b : 0.0f;
    const float gscale = -2.0f * pw / (a.denom * (float)n_rows);
#if defined(SSAC_LAB) && defined(SSAC_EXP_DQ_READY)
    if (tid < n_rows) tab[tid] = gscale * r.q;
    return;
#endif
    const float alpha = (a.tds.q_t && a.tds.use_entropy) ? expf(r.la) : 0.0f;
    if (tid < n_rows) {
        float t = r.t0;
        if (a.tds.q_t) {
            // min over the slots of the slot's value = the sum of its partials (entry e = j n_parts + s of the list; the
            // first four entries are in registers) -- ssac_td_min_q, operation by operation
            const int np = a.tds.n_parts > 1 ? a.tds.n_parts : 1;
            auto entry = [&](int e_) {
                return e_ == 0 ? r.t0 : e_ == 1 ? r.t1 : e_ == 2 ? r.t2 : e_ == 3 ? r.t3 : a.tds.q_t[(int64_t)e_ * n_rows + tid];
            };
            float mq = 0.0f;
            for (int j = 0; j < a.tds.n_sel; ++j) {
                float v = entry(j * np);
                for (int s_ = 1; s_ < np; ++s_) v += entry(j * np + s_);
                mq = j == 0 ? v : fminf(mq, v);
            }
            const float bonus = a.tds.use_entropy ? alpha * r.lp : 0.0f;
            const float val = mq - bonus;
            t = r.rew + a.tds.gamma * (1.0f - r.done) * val;
        }
        const float err = t - (pw * r.q + pb);
        tab[tid] = gscale * r.w * err;
    }
    const float *q = a.q + (int64_t)e * n_rows;
    for (int b = tid + blockDim.x; b < n_rows; b += blockDim.x) {   // (batches beyond one pass of the workgroup)
        const float t = a.tds.q_t ? ssac_lazy_td(a.tds, b, n_rows, alpha) : a.td[b];
        const float w = a.weight ? a.weight[b] : 1.0f;
        const float err = t - (pw * q[b] + pb);
        tab[b] = gscale * w * err;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Log finalisation folded into the weight-gradient launch (ssac_logfold in include/ssac_hip.h): no logs launch.
//   * the statistics of the TD targets (mean / unbiased std / entropy bonus) are computed by ONE workgroup -- the one
//     that evaluated and wrote the targets -- right after its loss-fold table (log_fold_td_stats);
//   * every workgroup's cross-workgroup outputs (loss partials, gradient-norm partial, the three TD statistics) leave as
//     agent-scope (write-through) stores; the workgroup then drains ITS stores (s_waitcnt vmcnt(0) in the storing lane)
//     and bumps a device-scope arrival counter (log_fold_arrive);
//   * the workgroup that draws the last ticket reads the partials back with agent-scope loads, sums them in index order
//     (the result does not depend on WHICH workgroup is last), writes the log block, publishes it to its ring slot and
//     advances the input ring (log_fold_finish): ~270 floats and one shuffle tree on one wave -- no fence, no second
//     launch, no L2 write-back of the 17 MB of optimizer state the launch has just dirtied.
// ------------------------------------------------------------------------------------------------------------
struct LogFoldArgs {
    unsigned *done;                 // null = fold off
    float *logs, *td_logs; ssac_feed *feed;
    float *deferred_stats;          // != null: DEFERRED mode (below): the TD statistics go here, no ticket is drawn, and
                                    // the launch only advances the input ring; the next update's first launch finishes
    const float *partials; int n_nets;     // [n_nets][2] from loss_fold_table
    const float *sumsq; int n_ss;          // every gradient-norm partial of the launch
    int n_rows; float denom;
    // ACTOR mode (round 6: the online actor update's two logs, ssac_actor_logs' work, in its weight-gradient launch): partials
    // = the tiles' loss terms (stride 1, n_nets = their count), logs[0] += a_scale * their sum, *a_gn = sqrt(sum of sumsq);
    // a_pub != null: the finished block (a_width floats at a_block) also goes to its slot of the log ring
    int actor; float a_scale; float *a_gn; const float *a_block; int a_width; float *a_pub;
};

// all threads of the (>= 256-thread) workgroup call this; red: 12 floats of LDS.  td_out was written by this workgroup.
__device__ __forceinline__ void log_fold_td_stats(const LogFoldArgs &f, const ssac_td_spec &tds, float *red) {
    const int tid = threadIdx.x, n_rows = f.n_rows;
    const bool act = tid < 256;
    __syncthreads();   // (td_out of every row is written; red is free)
    float s_td = 0.f, s_b = 0.f;
    const float alpha = tds.use_entropy ? expf(tds.log_alpha[0]) : 0.0f;
    if (act)
        for (int b = tid; b < n_rows; b += 256) {
            s_td += tds.td_out[b];
            s_b += tds.use_entropy ? alpha * tds.logp[b] : 0.0f;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s_td += __shfl_xor(s_td, o, 64); s_b += __shfl_xor(s_b, o, 64); }
    if (act && (tid & 63) == 0) { red[tid >> 6] = s_td; red[4 + (tid >> 6)] = s_b; }
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)n_rows;
    const float mb = (red[4] + red[5] + red[6] + red[7]) / (float)n_rows;
    float sv = 0.f;
    if (act)
        for (int b = tid; b < n_rows; b += 256) { const float d = tds.td_out[b] - mean; sv += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sv += __shfl_xor(sv, o, 64);
    if (act && (tid & 63) == 0) red[8 + (tid >> 6)] = sv;
    __syncthreads();
    if (tid == 0 && f.deferred_stats) {   // read by the NEXT launch: plain stores
        const float var = (red[8] + red[9] + red[10] + red[11]) / (float)(n_rows > 1 ? n_rows - 1 : 1);
        f.deferred_stats[0] = mean; f.deferred_stats[1] = sqrtf(var); f.deferred_stats[2] = mb;
    } else if (tid == 0 && f.td_logs) {
        const float var = (red[8] + red[9] + red[10] + red[11]) / (float)(n_rows > 1 ? n_rows - 1 : 1);
        __hip_atomic_store(f.td_logs + 0, mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(f.td_logs + 1, sqrtf(var), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(f.td_logs + 2, mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
}

// thread 0 of every workgroup, AFTER it has issued its last cross-workgroup store: true for the last arriver
__device__ __forceinline__ bool log_fold_arrive(const LogFoldArgs &f, unsigned total) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this lane's stores have left the CU (write-through) ...
    const unsigned ticket = __hip_atomic_fetch_add(f.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ticket == total - 1;                        // ... before the ticket can be seen
}

// wave 0 (all 64 lanes) of the last workgroup
__device__ __forceinline__ void log_fold_finish(const LogFoldArgs &f) {
    const int lane = threadIdx.x & 63;
    if (f.actor) {
        float s = 0.f, q = 0.f;
        for (int i = lane; i < f.n_nets; i += 64) s += f.partials[i];   // (written by an earlier launch)
        for (int i = lane; i < f.n_ss; i += 64) q += __hip_atomic_load(f.sumsq + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        const float loss = f.logs[0] + f.a_scale * s, gn = sqrtf(q);
        if (lane == 0) { f.logs[0] = loss; if (f.a_gn) *f.a_gn = gn; }
        if (f.a_pub)
            for (int i = lane; i < f.a_width; i += 64) {
                float v = f.a_block[i];
                if (f.a_block + i == f.logs) v = loss;
                if (f.a_gn && f.a_block + i == f.a_gn) v = gn;
                f.a_pub[i] = v;
            }
        if (lane == 0) __hip_atomic_store(f.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        return;
    }
    float sl = 0.f, se = 0.f, ss = 0.f;
    for (int i = lane; i < f.n_nets; i += 64) {
        sl += __hip_atomic_load(f.partials + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (i == f.n_nets - 1) se = __hip_atomic_load(f.partials + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = lane; i < f.n_ss; i += 64) ss += __hip_atomic_load(f.sumsq + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sl += __shfl_xor(sl, o, 64); se += __shfl_xor(se, o, 64); ss += __shfl_xor(ss, o, 64);
    }
    float td0 = 0.f, td1 = 0.f, td2 = 0.f;
    if (f.td_logs) {
        td0 = __hip_atomic_load(f.td_logs + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        td1 = __hip_atomic_load(f.td_logs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        td2 = __hip_atomic_load(f.td_logs + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the log block was cleared at the start of this update (another launch): [0] accumulates over ensemble members
    const float l0 = f.logs[0] + sl / (f.denom * (float)f.n_rows), l1 = se / (float)f.n_rows, l2 = sqrtf(ss);
    if (lane == 0) { f.logs[0] = l0; f.logs[1] = l1; f.logs[2] = l2; }
    if (f.feed) {  // publish the block to its ring slot, advance the input ring
        const int slot = (int)f.feed->dst[f.feed->log_slot_word], w = f.feed->log_width;
        const int tdo = f.td_logs ? (int)(f.td_logs - f.logs) : -1;
        for (int i = lane; i < w; i += 64) {
            float v = i == 0 ? l0 : i == 1 ? l1 : i == 2 ? l2 : f.logs[i];
            if (tdo >= 0 && i >= tdo && i < tdo + 3) v = i == tdo ? td0 : (i == tdo + 1 ? td1 : td2);
            f.feed->log_ring[(int64_t)slot * w + i] = v;
        }
        if (lane == 0) f.feed->tick += 1;
    }
    if (lane == 0) __hip_atomic_store(f.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
}


// ------------------------------------------------------------------------------------------------------------
// Deferred log finalisation of a RECORDED update (ssac_deferred_logs in include/ssac_hip.h): nothing of update k's log
// block depends on update k+1, but finishing it inside update k costs a dependent chain of memory round trips behind
// the LAST weight-gradient workgroup (5 us as a launch of its own, the same again as a last-arriver epilogue).  So the
// weight-gradient launch only leaves the inputs behind -- per-net loss partials, gradient-norm partials, the three TD
// statistics -- and advances the input ring; ONE extra workgroup of the NEXT update's first launch (beside ~220 busy
// ones, off every critical path) sums them in index order and writes update k's slot of the log ring.  Reading a log
// value of the newest update before another update has been issued launches ssac_deferred_logs_flush (the same body).
// ------------------------------------------------------------------------------------------------------------
struct DeferredLogsArgs {
    const float *partials; int n_nets;      // [n_nets][2]: sum_b w err^2, sum_b err  (loss_fold_table)
    const float *sumsq; int n_ss;           // gradient-norm partials of the weight-gradient launch
    const float *td_stats;                  // [3] mean / std / entropy bonus of the TD targets, or null
    int td_off, n_rows; float denom;        // td_off: index of the 3 TD statistics inside a log block
    const ssac_feed *feed;                  // the log ring + the input ring (which ring slot update k's block goes to)
};

// first wave of the calling workgroup.  ring_slot < 0: the update BEFORE the one this launch belongs to (its log-ring
// slot is read from its slot of the input ring); >= 0: that slot (flush).
__device__ __forceinline__ void deferred_logs_body(const DeferredLogsArgs &d, int ring_slot) {
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    const ssac_feed f = *d.feed;
    int slot = ring_slot;
    if (slot < 0) {
        if (f.tick <= 0) return;   // no recorded update has run yet
        const uint32_t *prev = f.host_ring + (int64_t)((f.tick - 1) % f.n_slots) * f.slot_words;
        slot = (int)prev[f.log_slot_word];
    }
    float sl = 0.f, se = 0.f, ss = 0.f;
    for (int i = lane; i < d.n_nets; i += 64) {
        sl += d.partials[2 * i];
        if (i == d.n_nets - 1) se = d.partials[2 * i + 1];
    }
    for (int i = lane; i < d.n_ss; i += 64) ss += d.sumsq[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sl += __shfl_xor(sl, o, 64); se += __shfl_xor(se, o, 64); ss += __shfl_xor(ss, o, 64);
    }
    const float l0 = sl / (d.denom * (float)d.n_rows), l1 = se / (float)d.n_rows, l2 = sqrtf(ss);
    float *dst = f.log_ring + (int64_t)slot * f.log_width;
    for (int i = lane; i < f.log_width; i += 64) {
        float v = i == 0 ? l0 : i == 1 ? l1 : i == 2 ? l2 : 0.0f;
        if (d.td_stats && i >= d.td_off && i < d.td_off + 3) v = d.td_stats[i - d.td_off];
        dst[i] = v;
    }
}
