// HBM/latency-bound pieces of the update path on gfx950: replay gather, tanh-normal head,
// TD target (+PopArt), loss gradients, temperature update, Polyak, Adam-from-grads, DrQ shifts.
//
// None of this is GEMM-shaped; it is byte/row work, so the rules that matter are coalesced
// row access and wavefront (64-lane) reductions -- no MFMA here.  Batch-wide statistics
// (TD-target mean/std for the logs and PopArt, loss means) are produced by single-workgroup
// kernels so their summation order is fixed and results are run-to-run deterministic.
#include <hip/hip_runtime.h>
#include <string.h>
#include <chrono>
#include <math.h>
#include <stdint.h>

#include "ssac_begin.h"
#include "ssac_internal.h"
#include "ssac_philox.h"

namespace {

constexpr int RED_THREADS = 1024;
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;
constexpr float LOG_2 = 0.69314718055994530942f;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide reductions for <=1024 threads; `scratch` is 16 floats of LDS; result broadcast.
template <int OP>  // 0 sum, 1 max, 2 min
__device__ float block_reduce(float v, float *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = OP == 0 ? wave_sum(v) : (OP == 1 ? wave_max(v) : wave_min(v));
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float r = scratch[0];
    for (int w = 1; w < nw; ++w) r = OP == 0 ? r + scratch[w] : (OP == 1 ? fmaxf(r, scratch[w]) : fminf(r, scratch[w]));
    return r;
}

__device__ __forceinline__ float softplus_f(float x) {
    // F.softplus, beta=1, threshold=20
    return x > 20.0f ? x : log1pf(expf(x));
}

__device__ __forceinline__ float popart_sigma(float mu, float nu) {
    // popart.py:22-23
    return fminf(fmaxf(sqrtf(nu - mu * mu) + 1e-5f, 1e-4f), 1e6f);
}

__global__ void philox_normal_kernel(float *out, int n_rows, int cols, RngArgs r) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * cols) return;
    const int b = i / cols, c = i - b * cols;
    out[i] = philox_normal(r.seed, rng_draw(r), b, c);
}

// ------------------------------------------------------------------ replay gather
template <typename T>
__global__ void gather_rows_kernel(const T *__restrict__ src, int64_t row_elems,
                                   const int64_t *__restrict__ idx, int n_rows,
                                   float *__restrict__ dst, int64_t ld, int64_t col0) {
    const int64_t total = (int64_t)n_rows * row_elems;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / row_elems, c = i - r * row_elems;
        dst[r * ld + col0 + c] = (float)src[idx[r] * row_elems + c];
    }
}

template <typename T>
__global__ void gather_transition_kernel(const T *__restrict__ s, const T *__restrict__ s1,
                                         int64_t s_elems, const float *__restrict__ act,
                                         int64_t a_elems, const float *__restrict__ rew,
                                         const uint8_t *__restrict__ done,
                                         const int64_t *__restrict__ idx, int n_rows,
                                         float *__restrict__ xsa, int64_t ld_x,
                                         float *__restrict__ x1sa, int64_t ld_x1,
                                         float *__restrict__ rew_out, float *__restrict__ done_out,
                                         const ssac_feed *feed, float *logs, int n_logs, ssac_adam_ctl *ctl) {
    if (feed) {
        // first launch of a captured update: the row indices are read straight from this update's slot of the
        // pinned host ring, and workgroup 0 also does what ssac_begin_update would (slot -> device block for
        // the later launches, log block cleared, optimizer step advanced) -- one launch fewer on the chain
        const ssac_feed f = *feed;
        idx = reinterpret_cast<const int64_t *>(feed_slot(f));
        if (blockIdx.x == 0) {
            feed_pull(f);
            if ((int)threadIdx.x < n_logs) logs[threadIdx.x] = 0.0f;
            if (threadIdx.x == 0 && ctl) adam_refresh(ctl, ctl->step + 1);
        }
    }
    // one row of work = 2*s_elems + a_elems + 2 elements; consecutive threads walk one row
    const int64_t per_row = 2 * s_elems + a_elems + 2;
    const int64_t total = (int64_t)n_rows * per_row;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row;
        int64_t c = i - r * per_row;
        const int64_t src = idx[r];
        if (c < s_elems) {
            xsa[r * ld_x + c] = (float)s[src * s_elems + c];
        } else if ((c -= s_elems) < a_elems) {
            xsa[r * ld_x + s_elems + c] = act[src * a_elems + c];
        } else if ((c -= a_elems) < s_elems) {
            x1sa[r * ld_x1 + c] = (float)s1[src * s_elems + c];
        } else if (c == s_elems) {
            rew_out[r] = rew[src];
        } else {
            done_out[r] = (float)done[src];
        }
    }
}

// ------------------------------------------------------------------ tanh-normal head
__global__ void tanh_normal_fwd_kernel(const float *__restrict__ out, int64_t ld_out,
                                       const float *__restrict__ eps, int n_rows, int A, float lo,
                                       float hi, float *__restrict__ act, int64_t ld_act,
                                       int64_t col0, float *__restrict__ logp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_rows) return;
    float lp = 0.0f;
    for (int i = 0; i < A; ++i) {
        const float mu = out[b * ld_out + i];
        const float raw = out[b * ld_out + A + i];
        const float log_std = lo + 0.5f * (hi - lo) * (tanhf(raw) + 1.0f);
        const float sd = expf(log_std);
        const float u = mu + sd * eps[(int64_t)b * A + i];
        const float a = tanhf(u);
        const float dlt = u - mu;
        const float base = -(dlt * dlt) / (2.0f * sd * sd) - logf(sd) - LOG_SQRT_2PI;
        const float ladj = 2.0f * (LOG_2 - u - softplus_f(-2.0f * u));
        lp += base - ladj;
        act[b * ld_act + col0 + i] = a;
    }
    if (logp) logp[b] = lp;
}

__global__ void tanh_normal_bwd_kernel(const float *__restrict__ dX, int n_nets, int64_t ldx,
                                       int64_t sX, int64_t col0, const float *__restrict__ out,
                                       int64_t ld_out, const float *__restrict__ eps, int n_rows,
                                       int A, float lo, float hi, const float *__restrict__ log_alpha,
                                       int use_entropy, float inv_members, float *__restrict__ d_out,
                                       int64_t ld_dout) {
    // one thread per (row, action dimension): the loads of a row's A x n_nets input gradients are spread over A lanes
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * A) return;
    const int b = idx / A, i = idx - b * A;
    const float c = use_entropy ? expf(log_alpha[0]) * inv_members / (float)n_rows : 0.0f;
    float gsum = 0.0f;
    for (int j = 0; j < n_nets; ++j) gsum += dX[j * sX + b * ldx + col0 + i];
    const float mu = out[b * ld_out + i];
    const float raw = out[b * ld_out + A + i];
    const float t = tanhf(raw);
    const float log_std = lo + 0.5f * (hi - lo) * (t + 1.0f);
    const float sd = expf(log_std);
    const float e = eps[(int64_t)b * A + i];
    const float a = tanhf(mu + sd * e);
    const float gu = gsum * (1.0f - a * a);
    const float d_mu = gu + c * 2.0f * a;
    const float d_ls = gu * sd * e + c * (-1.0f + 2.0f * a * sd * e);
    d_out[b * ld_dout + i] = d_mu;
    d_out[b * ld_dout + A + i] = d_ls * 0.5f * (hi - lo) * (1.0f - t * t);
}

__global__ void det_action_fwd_kernel(const float *__restrict__ out, int64_t ld_out,
                                      const float *__restrict__ eps, float sample_std,
                                      const float *__restrict__ noise, float scale, float clip,
                                      int n_rows, int A, float *__restrict__ act, int64_t ld_act,
                                      int64_t col0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * A) return;
    const int b = i / A, k = i - b * A;
    float a = tanhf(out[b * ld_out + k]);
    if (eps) a = a + sample_std * eps[i];
    if (noise) {
        float nz = scale * noise[i];
        if (clip > 0.0f) nz = fminf(fmaxf(nz, -clip), clip);
        a = fminf(fmaxf(a + nz, -1.0f + 1e-6f), 1.0f - 1e-6f);
    }
    act[b * ld_act + col0 + k] = a;
}

// GaussianExplorationNoise.sample on a device action (learning_utils.py:48-60), in place: a <- clamp(a + clamp(scale *
// noise, +-clip), -1 + 1e-6, 1 - 1e-6); the reference's "gradient preservation" passes gradients through unchanged
__global__ void exploration_noise_kernel(float *__restrict__ act, int64_t ld_act, int64_t col0,
                                         const float *__restrict__ noise, float scale, float clip, int n_rows, int A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * A) return;
    const int b = i / A, k = i - b * A;
    float nz = scale * noise[i];
    if (clip > 0.0f) nz = fminf(fmaxf(nz, -clip), clip);
    float *p = act + b * ld_act + col0 + k;
    *p = fminf(fmaxf(*p + nz, -1.0f + 1e-6f), 1.0f - 1e-6f);
}

// log-probability of a ContinuousDeterministic action under its own Normal(loc, 1e-4) (distributions.py:107-114),
// summed over the action dimensions: a = loc + 1e-4 eps (rsample) -> sum_d(-eps_d^2 / 2) + A (-log 1e-4 - log sqrt(2 pi));
// eps == NULL: a = loc (sample)
__global__ void det_logprob_kernel(const float *__restrict__ eps, int n_rows, int A, float *__restrict__ logp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_rows) return;
    float lp = 0.0f;
    for (int k = 0; k < A; ++k) {
        const float e = eps ? eps[b * A + k] : 0.0f;
        lp += -(e * e) / 2.0f - logf(1e-4f) - LOG_SQRT_2PI;
    }
    logp[b] = lp;
}

__global__ void det_action_bwd_kernel(const float *__restrict__ dX, int n_nets, int64_t ldx, int64_t sX,
                                      int64_t col0, const float *__restrict__ out, int64_t ld_out,
                                      int n_rows, int A, float *__restrict__ d_out, int64_t ld_dout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * A) return;
    const int b = i / A, k = i - b * A;
    float gsum = 0.0f;
    for (int j = 0; j < n_nets; ++j) gsum += dX[j * sX + b * ldx + col0 + k];
    const float t = tanhf(out[b * ld_out + k]);
    d_out[b * ld_dout + k] = gsum * (1.0f - t * t);
}

// ------------------------------------------------------------------ TD target (+PopArt)
__global__ __launch_bounds__(RED_THREADS) void td_target_kernel(
    const float *__restrict__ q_t, int n_sel, int n_rows, int qd, const float *__restrict__ lp,
    const float *__restrict__ rew, const float *__restrict__ done, const float *__restrict__ log_alpha,
    int use_entropy, float gamma, ssac_popart *popart, int pop, float *__restrict__ td,
    float *__restrict__ logs) {
    __shared__ float scratch[16];
    const float alpha = (use_entropy || qd > 1) ? expf(log_alpha[0]) : 0.0f;
    float pmu = 0.f, pnu = 0.f, pw = 1.f, pb = 0.f, psig = 1.f;
    if (popart) {
        pmu = popart->mu; pnu = popart->nu; pw = popart->w; pb = popart->b;
        psig = popart_sigma(pmu, pnu);
    }
    float s_td = 0.f, s_td2 = 0.f, s_bonus = 0.f;
    for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
        float val, bonus_acc;
        if (qd == 1) {
            float mq = q_t[b];
            for (int j = 1; j < n_sel; ++j) mq = fminf(mq, q_t[(int64_t)j * n_rows + b]);
            const float bonus = use_entropy ? alpha * lp[b] : 0.0f;
            val = mq - bonus;
            bonus_acc = bonus;
        } else {
            // SAC-Discrete: v = sum_a pi_a (minq_a - alpha log pi_a)   (learning_utils.py:322-328)
            const float *x = lp + (int64_t)b * qd;
            float mx = x[0];
            for (int a = 1; a < qd; ++a) mx = fmaxf(mx, x[a]);
            float se = 0.f;
            for (int a = 0; a < qd; ++a) se += expf(x[a] - mx);
            const float lse = mx + logf(se);
            val = 0.f;
            bonus_acc = 0.f;
            for (int a = 0; a < qd; ++a) {
                const float logpa = x[a] - lse;
                const float pa = expf(logpa);
                float mq = q_t[(int64_t)b * qd + a];
                for (int j = 1; j < n_sel; ++j)
                    mq = fminf(mq, q_t[((int64_t)j * n_rows + b) * qd + a]);
                const float bonus = alpha * logpa;
                val += pa * (mq - bonus);
                bonus_acc += bonus;
            }
            bonus_acc /= (float)qd;  // entropy_bonus.mean() runs over (B, A)
        }
        if (popart && pop) val = psig * (pw * val + pb) + pmu;  // popart(val, normalized=False)
        const float t = rew[b] + gamma * (1.0f - done[b]) * val;
        td[b] = t;
        s_td += t;
        s_td2 += t * t;
        s_bonus += bonus_acc;
    }
    const float inv_n = 1.0f / (float)n_rows;
    float mean = block_reduce<0>(s_td, scratch) * inv_n;
    const float mean2 = block_reduce<0>(s_td2, scratch) * inv_n;
    const float mbonus = block_reduce<0>(s_bonus, scratch) * inv_n;
    if (popart) {
        // popart.update_stats (popart.py:35-52), then normalize_values (popart.py:25-26)
        const int t = popart->t + 1;
        const double beta = popart->beta;
        const float beta_t = (float)(beta / (1.0 - pow(1.0 - beta, (double)t)));
        const float omb = (float)(1.0 - (beta / (1.0 - pow(1.0 - beta, (double)t))));
        const float nmu = omb * pmu + beta_t * mean;
        const float nnu = omb * pnu + beta_t * mean2;
        const float nsig = popart_sigma(nmu, nnu);
        const bool stable = (t > popart->min_steps) && (((1.0f - psig) / nsig) <= 0.1f);
        __syncthreads();
        if (threadIdx.x == 0) {
            popart->t = t;
            popart->mu = nmu;
            popart->nu = nnu;
            popart->stable = stable ? 1 : 0;
            if (stable) {
                popart->w = pw * (psig / nsig);
                popart->b = (psig * pb + pmu - nmu) / nsig;
            }
        }
        float s1 = 0.f;
        for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
            const float t2 = (td[b] - nmu) / nsig;
            td[b] = t2;
            s1 += t2;
        }
        mean = block_reduce<0>(s1, scratch) * inv_n;
    }
    float sv = 0.f;
    for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
        const float dlt = td[b] - mean;
        sv += dlt * dlt;
    }
    const float var = block_reduce<0>(sv, scratch) / (float)(n_rows > 1 ? n_rows - 1 : 1);
    if (threadIdx.x == 0 && logs) {
        logs[0] = mean;
        logs[1] = sqrtf(var);
        logs[2] = mbonus;
    }
}

// ------------------------------------------------------------------ critic loss gradient
__global__ __launch_bounds__(RED_THREADS) void critic_loss_bwd_kernel(
    const float *__restrict__ q, int n_nets, int n_rows, int qd, const float *__restrict__ act,
    int64_t ld_act, const float *td, const float *__restrict__ weight,
    const ssac_popart *popart, int pop, float denom, float *__restrict__ dq, float *__restrict__ logs,
    ssac_td_spec tds) {
    __shared__ float scratch[16];
    const float pw = (popart && pop) ? popart->w : 1.0f;
    const float pb = (popart && pop) ? popart->b : 0.0f;
    const float gscale = -2.0f * pw / (denom * (float)n_rows);
    if (tds.q_t) {
        // the TD targets are evaluated here (ssac_td_spec: same arithmetic and order as td_target_kernel) ...
        const float alpha = tds.use_entropy ? expf(tds.log_alpha[0]) : 0.0f;
        for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
            const float mq = ssac_td_min_q(tds, b, n_rows);
            const float bonus = tds.use_entropy ? alpha * tds.logp[b] : 0.0f;
            const float val = mq - bonus;
            tds.td_out[b] = tds.rew[b] + tds.gamma * (1.0f - tds.done[b]) * val;
        }
        __syncthreads();  // ... and read back below by other threads of this (single) workgroup
        td = tds.td_out;
    }
    float s_loss = 0.f, s_err_last = 0.f;
    const int total = n_nets * n_rows;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int j = i / n_rows, b = i - j * n_rows;
        const float w = weight ? weight[b] : 1.0f;
        float qv;
        int ai = 0;
        if (qd == 1) {
            qv = q[i];
        } else {
            ai = (int)act[b * ld_act];  // a.long() (learning.py:92)
            qv = q[(int64_t)i * qd + ai];
        }
        const float err = td[b] - (pw * qv + pb);
        s_loss += w * err * err;
        if (j == n_nets - 1) s_err_last += err;
        if (qd == 1) {
            dq[i] = gscale * w * err;
        } else {
            for (int a = 0; a < qd; ++a) dq[(int64_t)i * qd + a] = (a == ai) ? gscale * w * err : 0.0f;
        }
    }
    const float loss = block_reduce<0>(s_loss, scratch) / (denom * (float)n_rows);
    const float errm = block_reduce<0>(s_err_last, scratch) / (float)n_rows;
    if (threadIdx.x == 0 && logs) {
        logs[0] += loss;
        logs[1] = errm;
    }
}

// ------------------------------------------------------------------ advantage filter + filtered BC (AFBC)
// adv_estimator.py:58-79 (continuous): A(s,a) = Q(s,a) - V(s), Q = min over ALL critics (then popart(q), the
// normalised-space affine map, when the member has a PopArt layer), V = mean (or max) of Q over n_samp
// sampled policy actions.  q holds the critics' outputs on the stacked batch [data | sample 1 | ... | sample n]:
// (n_nets x (1+n_samp)*n_rows).  Also the binary filter (learning_utils.py:254-256), its mean
// ("losses/adv_weights_mean") and the PER priorities relu(A) + 1e-4 (learning_utils.py:293).
__global__ __launch_bounds__(RED_THREADS) void adv_filter_kernel(
    const float *__restrict__ q, int n_nets, int n_rows, int n_samp, const ssac_popart *popart, int use_max,
    float *__restrict__ adv, float *__restrict__ mask, float *__restrict__ prio, float *__restrict__ logs) {
    __shared__ float scratch[16];
    const float pw = popart ? popart->w : 1.0f, pb = popart ? popart->b : 0.0f;
    const int64_t ldq = (int64_t)(1 + n_samp) * n_rows;
    float s_mask = 0.f;
    for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
        float qd = 0.f, acc = 0.f;
        for (int k = 0; k <= n_samp; ++k) {
            float mq = q[(int64_t)k * n_rows + b];
            for (int j = 1; j < n_nets; ++j) mq = fminf(mq, q[j * ldq + (int64_t)k * n_rows + b]);
            if (popart) mq = pw * mq + pb;
            if (k == 0) qd = mq;
            else if (k == 1) acc = mq;
            else acc = use_max ? fmaxf(acc, mq) : acc + mq;
        }
        const float value = use_max ? acc : acc / (float)n_samp;
        const float a = qd - value;
        const float m = a >= 0.0f ? 1.0f : 0.0f;
        if (adv) adv[b] = a;
        if (mask) mask[b] = m;
        if (prio) prio[b] = fmaxf(a, 0.0f) + 1e-4f;
        s_mask += m;
    }
    const float tot = block_reduce<0>(s_mask, scratch);
    if (threadIdx.x == 0 && logs) logs[0] = tot / (float)n_rows;
}

// learning_utils.py:241-269 (continuous): loss_i = -mean(log pi(a_data | s) * mask); the data action misses
// the TanhTransform cache, so its pre-tanh value is atanh(clamp(a, +-0.99)) (distributions.py:74-84).
// Writes dL/d(actor output) for L = sum_i loss_i / E, logs_member[0] = loss_i, logs_total[0] += loss_i / E.
__global__ __launch_bounds__(RED_THREADS) void bc_logprob_bwd_kernel(
    const float *__restrict__ out, int64_t ld_out, const float *__restrict__ act, int64_t ld_act,
    const float *__restrict__ mask, int n_rows, int A, float lo, float hi, float inv_members,
    float *__restrict__ d_out, int64_t ld_dout, float *__restrict__ logs_member, float *__restrict__ logs_total) {
    __shared__ float scratch[16];
    float s = 0.f;
    for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
        const float w = mask ? mask[b] : 1.0f;
        const float coef = -w * inv_members / (float)n_rows;
        float lp = 0.f;
        for (int i = 0; i < A; ++i) {
            const float mu = out[b * ld_out + i], raw = out[b * ld_out + A + i];
            const float t = tanhf(raw);
            const float log_std = lo + 0.5f * (hi - lo) * (t + 1.0f);
            const float sd = expf(log_std);
            const float y = fminf(fmaxf(act[b * ld_act + i], -0.99f), 0.99f);
            const float x = 0.5f * (log1pf(y) - log1pf(-y));
            const float dlt = x - mu;
            const float r2 = (dlt * dlt) / (sd * sd);
            lp += (-(dlt * dlt) / (2.0f * sd * sd) - log_std - LOG_SQRT_2PI) - 2.0f * (LOG_2 - x - softplus_f(-2.0f * x));
            d_out[b * ld_dout + i] = coef * (dlt / (sd * sd));
            d_out[b * ld_dout + A + i] = coef * (r2 - 1.0f) * 0.5f * (hi - lo) * (1.0f - t * t);
        }
        s += lp * w;
    }
    const float tot = block_reduce<0>(s, scratch);
    if (threadIdx.x == 0) {
        const float loss = -tot / (float)n_rows;
        if (logs_member) logs_member[0] = loss;
        if (logs_total) logs_total[0] += loss * inv_members;
    }
}

// discrete (indirect) advantage, adv_estimator.py:41-56: V(s) = sum_a mean_actors(pi)(a) * Q(s)_a with Q = min over
// the member's critics (elementwise, then popart's w*q+b when present); A = Q(s)[a_data] - V(s).
// q: (n_nets x n_rows x A); logits: (n_actors x n_rows x A), actor stride n_rows*A; act: the action index as float.
__global__ __launch_bounds__(RED_THREADS) void adv_filter_discrete_kernel(
    const float *__restrict__ q, int n_nets, int n_rows, int A, const float *__restrict__ logits, int n_actors,
    const float *__restrict__ act, int64_t ld_act, const ssac_popart *popart, float *__restrict__ adv,
    float *__restrict__ mask, float *__restrict__ prio, float *__restrict__ logs) {
    __shared__ float scratch[16];
    const float pw = popart ? popart->w : 1.0f, pb = popart ? popart->b : 0.0f;
    float s_mask = 0.f;
    for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
        const int ai = (int)act[b * ld_act];
        float value = 0.f, qd = 0.f;
        for (int k = 0; k < A; ++k) {
            float mq = q[(int64_t)b * A + k];
            for (int j = 1; j < n_nets; ++j) mq = fminf(mq, q[((int64_t)j * n_rows + b) * A + k]);
            if (popart) mq = pw * mq + pb;
            float pk = 0.f;  // mean over the actors of softmax(logits)_k
            for (int m = 0; m < n_actors; ++m) {
                const float *x = logits + ((int64_t)m * n_rows + b) * A;
                float mx = x[0];
                for (int t = 1; t < A; ++t) mx = fmaxf(mx, x[t]);
                float se = 0.f;
                for (int t = 0; t < A; ++t) se += expf(x[t] - mx);
                pk += expf(x[k] - mx) / se;
            }
            pk /= (float)n_actors;
            value += pk * mq;
            if (k == ai) qd = mq;
        }
        const float a = qd - value;
        const float m = a >= 0.0f ? 1.0f : 0.0f;
        if (adv) adv[b] = a;
        if (mask) mask[b] = m;
        if (prio) prio[b] = fmaxf(a, 0.0f) + 1e-4f;
        s_mask += m;
    }
    const float tot = block_reduce<0>(s_mask, scratch);
    if (threadIdx.x == 0 && logs) logs[0] = tot / (float)n_rows;
}

// discrete filtered BC (learning_utils.py:257-268): loss_i = -mean(log_softmax(logits)[a_data] * mask)
__global__ __launch_bounds__(RED_THREADS) void bc_discrete_bwd_kernel(
    const float *__restrict__ logits, const float *__restrict__ act, int64_t ld_act, const float *__restrict__ mask,
    int n_rows, int A, float inv_members, float *__restrict__ d_logits, float *__restrict__ logs_member,
    float *__restrict__ logs_total) {
    __shared__ float scratch[16];
    float s = 0.f;
    for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
        const float w = mask ? mask[b] : 1.0f;
        const float coef = -w * inv_members / (float)n_rows;
        const float *x = logits + (int64_t)b * A;
        const int ai = (int)act[b * ld_act];
        float mx = x[0];
        for (int t = 1; t < A; ++t) mx = fmaxf(mx, x[t]);
        float se = 0.f;
        for (int t = 0; t < A; ++t) se += expf(x[t] - mx);
        const float lse = mx + logf(se);
        s += (x[ai] - lse) * w;
        for (int t = 0; t < A; ++t) d_logits[(int64_t)b * A + t] = coef * ((t == ai ? 1.0f : 0.0f) - expf(x[t] - lse));
    }
    const float tot = block_reduce<0>(s, scratch);
    if (threadIdx.x == 0) {
        const float loss = -tot / (float)n_rows;
        if (logs_member) logs_member[0] = loss;
        if (logs_total) logs_total[0] += loss * inv_members;
    }
}

// DR3 feature co-adaptation term (learning.py:100-108) on a stacked batch: rows [0,B) are (s,a), rows [B,2B) are
// (s',a'); h2 (n_nets x 2B x H) are the critics' fc2 features, dz2 their pre-activation gradients (ReLU mask of
// the TD loss already applied).  d/dh2 of coef * sum_h h2[b][h] h2[B+b][h] is added through the ReLU mask, and the
// per-block partial sums of the dot products are written for the "dr3_dotproduct" log.
__global__ void dr3_add_kernel(float *__restrict__ dz2, const float *__restrict__ h2, int n_nets, int B, int H,
                               float coef, float *__restrict__ partial) {
    __shared__ float red[4];
    const int64_t per_net = (int64_t)B * H, total = per_net * n_nets;
    float dot = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = i / per_net, r = i - j * per_net;
        const int64_t a = j * 2 * per_net + r, b = a + per_net;
        const float f = h2[a], f1 = h2[b];
        dz2[a] += f > 0.0f ? coef * f1 : 0.0f;
        dz2[b] += f1 > 0.0f ? coef * f : 0.0f;
        dot += f * f1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------ actor loss gradients
__global__ __launch_bounds__(RED_THREADS) void actor_loss_bwd_kernel(
    const float *__restrict__ q, int n_nets, int n_rows, const float *__restrict__ logp,
    const float *__restrict__ log_alpha, int use_entropy, const ssac_popart *popart, int pop,
    float inv_members, const float *__restrict__ qmin_global, float *__restrict__ dq,
    float *__restrict__ logs, const float *__restrict__ adv) {
    __shared__ float scratch[16];
    const float pw = (popart && pop) ? popart->w : 1.0f;
    const float pb = (popart && pop) ? popart->b : 0.0f;
    const float alpha = use_entropy ? expf(log_alpha[0]) : 0.0f;
    const float gq = -pw * inv_members / (float)n_rows;
    float s = 0.f;
    for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
        float mq = q[b];
        int am = 0;
        for (int j = 1; j < n_nets; ++j) {
            const float v = q[(int64_t)j * n_rows + b];
            if (v < mq) { mq = v; am = j; }
        }
        if (qmin_global) {
            // sharded ensemble: the min over ALL critics came from an all-reduce; the gradient is
            // routed only on the rank whose local min IS the global min
            const bool mine = mq == qmin_global[b];
            mq = qmin_global[b];
            if (!mine) am = -1;
        }
        for (int j = 0; j < n_nets; ++j) dq[(int64_t)j * n_rows + b] = (j == am) ? gq : 0.0f;
        const float bonus = use_entropy ? alpha * logp[b] : 0.0f;
        // use_baseline (learning.py:401): the objective is the advantage A = Q' - V(s); V carries no gradient
        s += (adv ? adv[b] : (pw * mq + pb)) - bonus;
    }
    const float tot = block_reduce<0>(s, scratch);
    if (threadIdx.x == 0 && logs) logs[0] += -inv_members * tot / (float)n_rows;
}

__global__ __launch_bounds__(RED_THREADS) void discrete_actor_loss_bwd_kernel(
    const float *__restrict__ logits, const float *__restrict__ q, int n_nets, int n_rows, int A,
    const float *__restrict__ log_alpha, const ssac_popart *popart, int pop, float inv_members,
    float *__restrict__ d_logits, float *__restrict__ logs) {
    __shared__ float scratch[16];
    const float pw = (popart && pop) ? popart->w : 1.0f;
    const float pb = (popart && pop) ? popart->b : 0.0f;
    const float alpha = expf(log_alpha[0]);
    const float scale = -inv_members / (float)n_rows;
    float s = 0.f;
    for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
        const float *x = logits + (int64_t)b * A;
        float mx = x[0];
        for (int a = 1; a < A; ++a) mx = fmaxf(mx, x[a]);
        float se = 0.f;
        for (int a = 0; a < A; ++a) se += expf(x[a] - mx);
        const float lse = mx + logf(se);
        float S = 0.f;
        for (int a = 0; a < A; ++a) {
            const float lpa = x[a] - lse, pa = expf(lpa);
            float mq = q[(int64_t)b * A + a];
            for (int j = 1; j < n_nets; ++j) mq = fminf(mq, q[((int64_t)j * n_rows + b) * A + a]);
            S += pa * ((pw * mq + pb) - alpha * lpa);
        }
        for (int a = 0; a < A; ++a) {
            const float lpa = x[a] - lse, pa = expf(lpa);
            float mq = q[(int64_t)b * A + a];
            for (int j = 1; j < n_nets; ++j) mq = fminf(mq, q[((int64_t)j * n_rows + b) * A + a]);
            const float f = (pw * mq + pb) - alpha * lpa;
            d_logits[(int64_t)b * A + a] = scale * pa * (f - S);
        }
        s += S;
    }
    const float tot = block_reduce<0>(s, scratch);
    if (threadIdx.x == 0 && logs) logs[0] += scale * tot;
}

// ------------------------------------------------------------------ temperature update

__global__ __launch_bounds__(RED_THREADS) void alpha_update_kernel(
    float *log_alpha, float *am, float *av, ssac_adam_ctl *ctl, const float *__restrict__ lp,
    int n_rows, int A, float target_entropy, float *__restrict__ logs) {
    __shared__ float scratch[16];
    float s = 0.f;
    for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
        float v;
        if (A <= 1) {
            v = lp[b];
        } else {  // sum_a pi log pi (learning.py:253)
            const float *x = lp + (int64_t)b * A;
            float mx = x[0];
            for (int a = 1; a < A; ++a) mx = fmaxf(mx, x[a]);
            float se = 0.f;
            for (int a = 0; a < A; ++a) se += expf(x[a] - mx);
            const float lse = mx + logf(se);
            v = 0.f;
            for (int a = 0; a < A; ++a) v += expf(x[a] - lse) * (x[a] - lse);
        }
        s += v + target_entropy;
    }
    const float mean = block_reduce<0>(s, scratch) / (float)n_rows;
    if (threadIdx.x == 0) {
        const float la = log_alpha[0];
        const float loss = -(la * mean);
        const float g = -mean;
        adam_refresh(ctl, ctl->step + 1);
        float m = am[0], v = av[0];
        m = m + (1.0f - ctl->beta1) * (g - m);
        v = v * ctl->beta2 + (1.0f - ctl->beta2) * g * g;
        const float denom = sqrtf(v) / ctl->bc2_sqrt + ctl->eps;
        const float nla = la - ctl->step_size * (m / denom);
        am[0] = m;
        av[0] = v;
        log_alpha[0] = nla;
        if (logs) {
            logs[0] = loss;
            logs[1] = expf(nla);
        }
    }
}

__global__ void adam_advance_kernel(ssac_adam_ctl *ctl) { adam_refresh(ctl, ctl->step + 1); }

// start of an update: clear the log block and advance the optimizer's step in one launch
__global__ void begin_update_kernel(float *logs, int n, ssac_adam_ctl *ctl, const ssac_feed *feed) {
    if ((int)threadIdx.x < n) logs[threadIdx.x] = 0.0f;
    if (threadIdx.x == 0 && ctl) adam_refresh(ctl, ctl->step + 1);
    if (feed) {  // this update's host inputs: pinned ring slot -> fixed device block (one PCIe round trip)
        feed_pull(*feed);
    }
}

__global__ void publish_logs_kernel(const float *logs, ssac_feed *feed) {
    const int slot = (int)feed->dst[feed->log_slot_word];
    const int w = feed->log_width;
    if ((int)threadIdx.x < w) feed->log_ring[(int64_t)slot * w + threadIdx.x] = logs[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) feed->tick += 1;
}

__global__ __launch_bounds__(RED_THREADS) void clip_coef_kernel(ssac_adam_ctl *ctl,
                                                               const float *__restrict__ sumsq, int n,
                                                               float max_norm, float *norm_out) {
    __shared__ float scratch[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += sumsq[i];
    const float tot = sqrtf(block_reduce<0>(s, scratch));
    if (threadIdx.x == 0) {
        // clip_grad_norm_: coef = clamp(max_norm / (total + 1e-6), max=1)
        if (ctl) ctl->clip_coef = max_norm > 0.0f ? fminf(max_norm / (tot + 1e-6f), 1.0f) : 1.0f;
        if (norm_out) norm_out[0] = tot;
    }
}

__global__ void group_norms_kernel(const float *__restrict__ sumsq, int group_size,
                                   const ssac_adam_ctl *scale, float *out) {
    __shared__ float scratch[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < group_size; i += blockDim.x) s += sumsq[(int64_t)blockIdx.x * group_size + i];
    const float tot = block_reduce<0>(s, scratch);
    if (threadIdx.x == 0) out[blockIdx.x] = sqrtf(tot) * (scale ? scale->clip_coef : 1.0f);
}

__global__ void adam_step_kernel(float *__restrict__ p, float *__restrict__ am, float *__restrict__ av,
                                 const float *__restrict__ g, int64_t n, const ssac_adam_ctl *ctl) {
    const ssac_adam_ctl c = *ctl;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float gr = g[i] * c.clip_coef;
        const float pv = p[i];
        if (c.weight_decay != 0.0f) gr = gr + c.weight_decay * pv;
        float m = am[i], v = av[i];
        m = m + (1.0f - c.beta1) * (gr - m);
        v = v * c.beta2 + (1.0f - c.beta2) * gr * gr;
        const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
        am[i] = m;
        av[i] = v;
        p[i] = pv - c.step_size * (m / denom);
    }
}

__global__ void polyak_kernel(float *__restrict__ t, const float *__restrict__ s, int64_t n, float tau) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        t[i] = t[i] * (1.0f - tau) + s[i] * tau;
}

// 16-byte-aligned arenas (the packed MLP arenas are): 4 elements per lane, two independent quads in flight
__global__ __launch_bounds__(256) void polyak4_kernel(float4 *__restrict__ t, const float4 *__restrict__ s,
                                                       int64_t n4, float tau) {
    const float k = 1.0f - tau;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const float4 a0 = t[i], b0 = s[i], a1 = t[i + stride], b1 = s[i + stride];
        t[i] = make_float4(a0.x * k + b0.x * tau, a0.y * k + b0.y * tau, a0.z * k + b0.z * tau, a0.w * k + b0.w * tau);
        t[i + stride] = make_float4(a1.x * k + b1.x * tau, a1.y * k + b1.y * tau, a1.z * k + b1.z * tau,
                                    a1.w * k + b1.w * tau);
    }
    if (i < n4) {
        const float4 a0 = t[i], b0 = s[i];
        t[i] = make_float4(a0.x * k + b0.x * tau, a0.y * k + b0.y * tau, a0.z * k + b0.z * tau, a0.w * k + b0.w * tau);
    }
}

// several tensors in ONE launch (the parameters of a module that is not a packed arena: a pixel encoder's conv / fc / norm
// tensors -- 12 launches of ~4.6 us each per soft_update of the DrQv2 encoder before).  Same arithmetic per element.
struct PolyakSegs { float *t[SSAC_MAX_POLYAK_SEGS]; const float *s[SSAC_MAX_POLYAK_SEGS]; int64_t n[SSAC_MAX_POLYAK_SEGS]; int count; };
__global__ __launch_bounds__(256) void polyak_multi_kernel(PolyakSegs a, float tau) {
    const float k = 1.0f - tau;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, first = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (int g = 0; g < a.count; ++g) {
        float *t = a.t[g];
        const float *s = a.s[g];
        const int64_t n = a.n[g];
        if ((n & 3) == 0 && (((uintptr_t)t | (uintptr_t)s) & 15) == 0) {
            float4 *t4 = reinterpret_cast<float4 *>(t);
            const float4 *s4 = reinterpret_cast<const float4 *>(s);
            const int64_t n4 = n >> 2;
            int64_t i = first;
            for (; i + stride < n4; i += 2 * stride) {   // (two elements' loads in flight)
                const float4 x0 = t4[i], y0 = s4[i], x1 = t4[i + stride], y1 = s4[i + stride];
                t4[i] = make_float4(x0.x * k + y0.x * tau, x0.y * k + y0.y * tau, x0.z * k + y0.z * tau, x0.w * k + y0.w * tau);
                t4[i + stride] = make_float4(x1.x * k + y1.x * tau, x1.y * k + y1.y * tau, x1.z * k + y1.z * tau, x1.w * k + y1.w * tau);
            }
            if (i < n4) {
                const float4 x = t4[i], y = s4[i];
                t4[i] = make_float4(x.x * k + y.x * tau, x.y * k + y.y * tau, x.z * k + y.z * tau, x.w * k + y.w * tau);
            }
        } else {
            for (int64_t i = first; i < n; i += stride) t[i] = t[i] * k + s[i] * tau;
        }
    }
}

__global__ void zero_kernel(float *p, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        p[i] = 0.0f;
}

// ------------------------------------------------------------------ SUNRISE weights
__global__ __launch_bounds__(RED_THREADS) void sunrise_weights_kernel(const float *__restrict__ q,
                                                                     int E, int n_rows, float temp,
                                                                     float *__restrict__ w,
                                                                     float *__restrict__ logs) {
    __shared__ float scratch[16];
    float s = 0.f, mx = -INFINITY, mn = INFINITY;
    for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
        float m = 0.f;
        for (int k = 0; k < E; ++k) m += q[(int64_t)k * n_rows + b];
        m /= (float)E;
        float var = 0.f;
        for (int k = 0; k < E; ++k) {
            const float d = q[(int64_t)k * n_rows + b] - m;
            var += d * d;
        }
        const float sd = sqrtf(var / (float)(E - 1));
        const float wv = 1.0f / (1.0f + expf(sd * temp)) + 0.5f;  // sigmoid(-sd*temp) + 0.5
        w[b] = wv;
        s += wv;
        mx = fmaxf(mx, wv);
        mn = fminf(mn, wv);
    }
    const float mean = block_reduce<0>(s, scratch) / (float)n_rows;
    mx = block_reduce<1>(mx, scratch);
    mn = block_reduce<2>(mn, scratch);
    float sv = 0.f;
    for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
        const float d = w[b] - mean;
        sv += d * d;
    }
    const float var = block_reduce<0>(sv, scratch) / (float)(n_rows > 1 ? n_rows - 1 : 1);
    if (threadIdx.x == 0 && logs) {
        logs[0] = mean; logs[1] = mx; logs[2] = mn; logs[3] = sqrtf(var);
    }
}

// "softmax" backup weights (learning_utils.py:383-393): w = n_rows * softmax_b(-std_k(q[k][b]) * temp)  (dim 0 = batch)
__global__ __launch_bounds__(RED_THREADS) void softmax_weights_kernel(const float *__restrict__ q, int E, int n_rows,
                                                                      float temp, float *__restrict__ w,
                                                                      float *__restrict__ logs) {
    __shared__ float scratch[16];
    float xm = -INFINITY;
    for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
        float m = 0.f;
        for (int k = 0; k < E; ++k) m += q[(int64_t)k * n_rows + b];
        m /= (float)E;
        float var = 0.f;
        for (int k = 0; k < E; ++k) {
            const float d = q[(int64_t)k * n_rows + b] - m;
            var += d * d;
        }
        const float x = -sqrtf(var / (float)(E - 1)) * temp;
        w[b] = x;  // logits, normalised below
        xm = fmaxf(xm, x);
    }
    xm = block_reduce<1>(xm, scratch);
    float se = 0.f;
    for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
        const float ev = expf(w[b] - xm);
        w[b] = ev;
        se += ev;
    }
    se = block_reduce<0>(se, scratch);
    float s = 0.f, mx = -INFINITY, mn = INFINITY;
    for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
        const float wv = (float)n_rows * (w[b] / se);
        w[b] = wv;
        s += wv;
        mx = fmaxf(mx, wv);
        mn = fminf(mn, wv);
    }
    const float mean = block_reduce<0>(s, scratch) / (float)n_rows;
    mx = block_reduce<1>(mx, scratch);
    mn = block_reduce<2>(mn, scratch);
    float sv = 0.f;
    for (int b = threadIdx.x; b < n_rows; b += blockDim.x) {
        const float d = w[b] - mean;
        sv += d * d;
    }
    const float var = block_reduce<0>(sv, scratch) / (float)(n_rows > 1 ? n_rows - 1 : 1);
    if (threadIdx.x == 0 && logs) {
        logs[0] = mean; logs[1] = mx; logs[2] = mn; logs[3] = sqrtf(var);
    }
}

// out[b] = min_j q[j][b][sel_b], sel_b = (int)act[b*ld_act] (q_dim > 1 with act) -- agent.Critic.forward(return_min=True)
// (agent.py:37-38) followed by .gather(-1, a.long()) (learning_utils.py:375, 389); act == null: out is (n_rows x q_dim)
__global__ void ensemble_min_select_kernel(const float *__restrict__ q, int n_nets, int n_rows, int q_dim,
                                           const float *__restrict__ act, int64_t ld_act, float *__restrict__ out) {
    const int n_out = act ? n_rows : n_rows * q_dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += gridDim.x * blockDim.x) {
        const int64_t e = act ? (int64_t)i * q_dim + (q_dim > 1 ? (int)act[(int64_t)i * ld_act] : 0) : i;
        float m = q[e];
        for (int j = 1; j < n_nets; ++j) m = fminf(m, q[(int64_t)j * n_rows * q_dim + e]);
        out[i] = m;
    }
}

// ------------------------------------------------------------------ DrQ shifts
// (linspace_f32 / drqv2_shift_axis: ssac_internal.h)

template <typename T>
__global__ void drq_shift_kernel(const T *__restrict__ src, const int64_t *__restrict__ idx, int n, int c,
                                 int h, int pad, const int64_t *__restrict__ shift, int mode,
                                 const float *__restrict__ noise, int n_aug, float *__restrict__ dst) {
    // The sampling grid must round like ATen's separate fp32 ops: HIP's __fmul_rn/__fadd_rn are
    // plain * and + and would be contracted into FMAs under the default -ffp-contract=fast.
#pragma clang fp contract(off)
    const int64_t per_img = (int64_t)c * h * h;
    const int64_t total = (int64_t)n * per_img;
    const int hp = h + 2 * pad;
    // Drqv2Aug.random_crop constants (augmentations.py:242-255)
    const float start = (float)(-1.0 + 1.0 / (double)hp), end = (float)(1.0 - 1.0 / (double)hp);
    const float step = __fdiv_rn(__fsub_rn(end, start), (float)(hp - 1));
    const float sscale = (float)(2.0 / (double)hp);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / per_img);
        const int64_t rem = i - (int64_t)b * per_img;
        const int ch = (int)(rem / (h * h));
        const int y = (int)((rem / h) % h), x = (int)(rem % h);
        const T *img = src + (idx ? idx[b] : (int64_t)b) * per_img + (int64_t)ch * h * h;
        float v;
        if (b >= n_aug) {
            v = (float)img[y * h + x];
        } else if (mode == 0) {
            const float gx = __fadd_rn(linspace_f32(start, end, step, hp, x),
                                       __fmul_rn((float)shift[2 * b + 0], sscale));
            const float gy = __fadd_rn(linspace_f32(start, end, step, hp, y),
                                       __fmul_rn((float)shift[2 * b + 1], sscale));
            // grid_sample un-normalise, align_corners=False: ((g + 1) * size - 1) / 2
            const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), (float)hp), 1.0f), 2.0f);
            const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), (float)hp), 1.0f), 2.0f);
            const float fx = floorf(ix), fy = floorf(iy);
            const float wx1 = __fsub_rn(ix, fx), wy1 = __fsub_rn(iy, fy);
            const float wx0 = __fsub_rn(1.0f, wx1), wy0 = __fsub_rn(1.0f, wy1);
            const int x0 = (int)fx, y0 = (int)fy;
            float acc = 0.0f;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int yy = y0 + dy;
                const float wy = dy ? wy1 : wy0;
                const bool vy = yy >= 0 && yy < hp;
                const int sy = min(max(yy - pad, 0), h - 1);  // replicate pad
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int xx = x0 + dx;
                    const float wx = dx ? wx1 : wx0;
                    const bool vx = xx >= 0 && xx < hp;
                    const int sx = min(max(xx - pad, 0), h - 1);
                    const float wgt = (vy && vx) ? __fmul_rn(wy, wx) : 0.0f;
                    acc = __fadd_rn(acc, __fmul_rn((float)img[sy * h + sx], wgt));
                }
            }
            v = fminf(fmaxf(acc, 0.0f), 255.0f);
        } else {
            // DrqAug: ReflectionPad2d(pad) then crop at (h1, w1) (augmentations.py:188-204)
            int sy = y + (int)shift[2 * b + 1] - pad, sx = x + (int)shift[2 * b + 0] - pad;
            sy = sy < 0 ? -sy : (sy >= h ? 2 * (h - 1) - sy : sy);
            sx = sx < 0 ? -sx : (sx >= h ? 2 * (h - 1) - sx : sx);
            v = (float)img[sy * h + sx];
            if (noise) v += noise[i];
            v = fminf(fmaxf(v, 0.0f), 255.0f);
        }
        dst[i] = v;
    }
}

// DrQv2 shift (mode 0), one workgroup per image plane (b, channel): the sampling positions and bilinear weights depend
// only on (b, x) and (b, y), so they are evaluated once per row / column into LDS -- with exactly the operations of the
// per-element kernel above -- instead of once per output element (~40 flops and two 64-bit div/mod chains each), and
// the plane itself (h x h source pixels) is staged in LDS, so the four taps of an output are LDS reads.  Same bits.
constexpr int SHIFT_MAX_H = 128;
template <typename T>
__global__ __launch_bounds__(256) void drqv2_shift_plane_kernel(const T *__restrict__ src, const int64_t *__restrict__ idx,
                                                                int c, int h, int pad, const int64_t *__restrict__ shift,
                                                                int n_aug, float *__restrict__ dst) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float pl[];   // [h * h] the source plane as float
    __shared__ float w0x[SHIFT_MAX_H], w1x[SHIFT_MAX_H], w0y[SHIFT_MAX_H], w1y[SHIFT_MAX_H];
    __shared__ int p0x[SHIFT_MAX_H], p0y[SHIFT_MAX_H];
    const int b = blockIdx.x / c, ch = blockIdx.x - b * c, tid = threadIdx.x;
    const int hh = h * h, hp = h + 2 * pad;
    const T *img = src + ((idx ? idx[b] : (int64_t)b) * c + ch) * (int64_t)hh;
    float *out = dst + ((int64_t)b * c + ch) * hh;
    if (b >= n_aug) {   // (rows beyond the augmented part of the mix: plain copy)
        for (int i = tid; i < hh; i += 256) out[i] = (float)img[i];
        return;
    }
    if (sizeof(T) == 1 && (hh & 3) == 0 && ((uintptr_t)img & 3) == 0) {   // uint8 plane: four pixels per load
        const uint32_t *w = reinterpret_cast<const uint32_t *>(img);
        const int nw = hh >> 2;
        // (8 words per thread requested together: the plane arrives in one round trip, not one per 256 words)
        for (int base = tid; base < nw; base += 8 * 256) {
            uint32_t u[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) u[k] = base + k * 256 < nw ? w[base + k * 256] : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (base + k * 256 < nw)
                    *reinterpret_cast<float4 *>(pl + 4 * (base + k * 256)) =
                        make_float4((float)(u[k] & 255u), (float)((u[k] >> 8) & 255u), (float)((u[k] >> 16) & 255u), (float)(u[k] >> 24));
        }
    } else {
        for (int i = tid; i < hh; i += 256) pl[i] = (float)img[i];
    }
    if (tid < 2 * h) {
        const int ax = tid >= h, i = ax ? tid - h : tid;   // ax 0: x / columns, 1: y / rows
        const ShiftAxis sa = drqv2_shift_axis(i, shift[2 * b + ax], hp);
        // a tap outside the padded image weighs 0 (grid_sample's zero padding): the mask goes into the axis weights --
        // all weights are >= 0, so w * 0 = +0, the value the product is replaced by otherwise: the same bits
        (ax ? w1y : w1x)[i] = (sa.p0 + 1 >= 0 && sa.p0 + 1 < hp) ? sa.w1 : 0.0f;
        (ax ? w0y : w0x)[i] = (sa.p0 >= 0 && sa.p0 < hp) ? sa.w0 : 0.0f;
        (ax ? p0y : p0x)[i] = sa.p0;
    }
    __syncthreads();
    // a thread owns output columns (their source columns and weights stay in registers) and walks down the rows: no
    // per-element index division, row terms are LDS broadcasts.  With h % 4 == 0 it owns FOUR adjacent columns and stores
    // 16 bytes per row (256 / (h/4) rows of the plane per pass), else one column.  Two columns per packed fp32 operation
    // (v_pk_mul_f32 / v_pk_add_f32: the same IEEE operations per column, in the per-element kernel's order).
    if ((h & 3) == 0 && ((uintptr_t)out & 15) == 0) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const int tpr = h >> 2, rpp = 256 / tpr, ty = tid / tpr, tx4 = tid - ty * tpr;
        if (ty >= rpp) return;
        f2 wxa[2], wxb[2];
        int sx0[4], sx1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = 4 * tx4 + j, x0 = p0x[x];
            wxa[j >> 1][j & 1] = w0x[x]; wxb[j >> 1][j & 1] = w1x[x];
            sx0[j] = min(max(x0 - pad, 0), h - 1); sx1[j] = min(max(x0 + 1 - pad, 0), h - 1);   // replicate pad
        }
        for (int y = ty; y < h; y += rpp) {
            const int y0 = p0y[y];
            f2 acc[2] = {f2{0.0f, 0.0f}, f2{0.0f, 0.0f}};
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const float wy = dy ? w1y[y] : w0y[y];
                const float *row = pl + min(max(y0 + dy - pad, 0), h - 1) * h;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f2 pa = f2{row[sx0[2 * q]], row[sx0[2 * q + 1]]}, pb = f2{row[sx1[2 * q]], row[sx1[2 * q + 1]]};
                    acc[q] = acc[q] + pa * (wxa[q] * wy);
                    acc[q] = acc[q] + pb * (wxb[q] * wy);
                }
            }
            *reinterpret_cast<float4 *>(out + y * h + 4 * tx4) =
                make_float4(__builtin_amdgcn_fmed3f(acc[0][0], 0.0f, 255.0f), __builtin_amdgcn_fmed3f(acc[0][1], 0.0f, 255.0f),
                            __builtin_amdgcn_fmed3f(acc[1][0], 0.0f, 255.0f), __builtin_amdgcn_fmed3f(acc[1][1], 0.0f, 255.0f));
        }
        return;
    }
    const int rpp = 256 / h, ty = tid / h, tx = tid - ty * h;
    if (ty >= rpp) return;
    const int x0 = p0x[tx];
    const float wx0 = w0x[tx], wx1 = w1x[tx];
    const bool vx0 = x0 >= 0 && x0 < hp, vx1 = x0 + 1 >= 0 && x0 + 1 < hp;
    const int sx0 = min(max(x0 - pad, 0), h - 1), sx1 = min(max(x0 + 1 - pad, 0), h - 1);   // replicate pad
    for (int y = ty; y < h; y += rpp) {
        const int y0 = p0y[y];
        float acc = 0.0f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int yy = y0 + dy;
            const float wy = dy ? w1y[y] : w0y[y];
            const bool vy = yy >= 0 && yy < hp;
            const float *row = pl + min(max(yy - pad, 0), h - 1) * h;
            acc = __fadd_rn(acc, __fmul_rn(row[sx0], (vy && vx0) ? __fmul_rn(wy, wx0) : 0.0f));
            acc = __fadd_rn(acc, __fmul_rn(row[sx1], (vy && vx1) ? __fmul_rn(wy, wx1) : 0.0f));
        }
        out[y * h + tx] = fminf(fmaxf(acc, 0.0f), 255.0f);
    }
}

inline int grid_for(int64_t n, int block = 256, int cap = 2048) {
    int64_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int ssac_philox_normal(float *out, int n_rows, int cols, const ssac_rng *rng, void *stream) {
    if (!rng || n_rows <= 0 || cols <= 0) return ssac_fail("ssac_philox_normal: bad arguments");
    const int n = n_rows * cols;
    SSAC_LAUNCH(philox_normal_kernel, dim3((n + 255) / 256), dim3(256), 0, ST, out, n_rows, cols,
                RngArgs{rng->seed, rng->counter, rng->offset});
    return ssac_check_launch("philox_normal");
}

extern "C" int ssac_gather_rows(const void *src, int src_dtype, int64_t row_elems, const int64_t *idx,
                                int n_rows, float *dst, int64_t ld_dst, int64_t dst_col0, void *stream) {
    if (n_rows <= 0 || row_elems <= 0) return 0;
    const int64_t total = (int64_t)n_rows * row_elems;
    if (src_dtype == 0)
        SSAC_LAUNCH(gather_rows_kernel<float>, dim3(grid_for(total)), dim3(256), 0, ST,
                           (const float *)src, row_elems, idx, n_rows, dst, ld_dst, dst_col0);
    else if (src_dtype == 1)
        SSAC_LAUNCH(gather_rows_kernel<uint8_t>, dim3(grid_for(total)), dim3(256), 0, ST,
                           (const uint8_t *)src, row_elems, idx, n_rows, dst, ld_dst, dst_col0);
    else
        return ssac_fail("ssac_gather_rows: unsupported src_dtype");
    return ssac_check_launch("gather_rows");
}

static int gather_transition_impl(const void *s, const void *s1, int s_dtype, int64_t s_elems, const float *act,
                                  int64_t a_elems, const float *rew, const uint8_t *done, const int64_t *idx,
                                  int n_rows, float *xsa, int64_t ld_x, float *x1sa, int64_t ld_x1,
                                  float *rew_out, float *done_out, const ssac_feed *feed, float *logs,
                                  int n_logs, ssac_adam_ctl *ctl, void *stream) {
    if (n_rows <= 0) return 0;
    if (n_logs > 256) return ssac_fail("ssac_gather_transition: log block too large");
    const int64_t total = (int64_t)n_rows * (2 * s_elems + a_elems + 2);
    if (s_dtype == 0)
        SSAC_LAUNCH(gather_transition_kernel<float>, dim3(grid_for(total)), dim3(256), 0, ST,
                    (const float *)s, (const float *)s1, s_elems, act, a_elems, rew, done, idx, n_rows, xsa, ld_x,
                    x1sa, ld_x1, rew_out, done_out, feed, logs, n_logs, ctl);
    else if (s_dtype == 1)
        SSAC_LAUNCH(gather_transition_kernel<uint8_t>, dim3(grid_for(total)), dim3(256), 0, ST,
                    (const uint8_t *)s, (const uint8_t *)s1, s_elems, act, a_elems, rew, done, idx, n_rows, xsa,
                    ld_x, x1sa, ld_x1, rew_out, done_out, feed, logs, n_logs, ctl);
    else
        return ssac_fail("ssac_gather_transition: unsupported s_dtype");
    return ssac_check_launch("gather_transition");
}

extern "C" int ssac_gather_transition(const void *s, const void *s1, int s_dtype, int64_t s_elems,
                                      const float *act, int64_t a_elems, const float *rew,
                                      const uint8_t *done, const int64_t *idx, int n_rows, float *xsa,
                                      int64_t ld_x, float *x1sa, int64_t ld_x1, float *rew_out,
                                      float *done_out, void *stream) {
    return gather_transition_impl(s, s1, s_dtype, s_elems, act, a_elems, rew, done, idx, n_rows, xsa, ld_x, x1sa,
                                  ld_x1, rew_out, done_out, nullptr, nullptr, 0, nullptr, stream);
}

extern "C" int ssac_gather_transition_begin(const void *s, const void *s1, int s_dtype, int64_t s_elems,
                                            const float *act, int64_t a_elems, const float *rew,
                                            const uint8_t *done, int n_rows, float *xsa, int64_t ld_x,
                                            float *x1sa, int64_t ld_x1, float *rew_out, float *done_out,
                                            const ssac_feed *feed, float *logs, int n_logs, ssac_adam_ctl *ctl,
                                            void *stream) {
    if (!feed) return ssac_fail("ssac_gather_transition_begin: needs a feed");
    return gather_transition_impl(s, s1, s_dtype, s_elems, act, a_elems, rew, done, nullptr, n_rows, xsa, ld_x,
                                  x1sa, ld_x1, rew_out, done_out, feed, logs, n_logs, ctl, stream);
}

extern "C" int ssac_tanh_normal_fwd(const float *out, int64_t ld_out, const float *eps, int n_rows,
                                    int act_dim, float lo, float hi, float *act_dst, int64_t ld_act,
                                    int64_t act_col0, float *logp, void *stream) {
    if (n_rows <= 0) return 0;
    SSAC_LAUNCH(tanh_normal_fwd_kernel, dim3((n_rows + 63) / 64), dim3(64), 0, ST, out, ld_out,
                       eps, n_rows, act_dim, lo, hi, act_dst, ld_act, act_col0, logp);
    return ssac_check_launch("tanh_normal_fwd");
}

extern "C" int ssac_det_action_fwd(const float *out, int64_t ld_out, const float *eps, float sample_std,
                                   const float *noise, float noise_scale, float noise_clip, int n_rows,
                                   int act_dim, float *act_dst, int64_t ld_act, int64_t act_col0,
                                   void *stream) {
    if (n_rows <= 0) return 0;
    const int total = n_rows * act_dim;
    SSAC_LAUNCH(det_action_fwd_kernel, dim3((total + 255) / 256), dim3(256), 0, ST, out, ld_out,
                       eps, sample_std, noise, noise_scale, noise_clip, n_rows, act_dim, act_dst, ld_act,
                       act_col0);
    return ssac_check_launch("det_action_fwd");
}

extern "C" int ssac_exploration_noise(float *act, int64_t ld_act, int64_t act_col0, const float *noise, float noise_scale,
                                      float noise_clip, int n_rows, int act_dim, void *stream) {
    if (!act || !noise) return ssac_fail("ssac_exploration_noise: null argument");
    if (n_rows <= 0) return 0;
    const int total = n_rows * act_dim;
    SSAC_LAUNCH(exploration_noise_kernel, dim3((total + 255) / 256), dim3(256), 0, ST, act, ld_act, act_col0, noise,
                noise_scale, noise_clip, n_rows, act_dim);
    return ssac_check_launch("exploration_noise");
}

extern "C" int ssac_det_logprob(const float *eps, int n_rows, int act_dim, float *logp, void *stream) {
    if (!logp) return ssac_fail("ssac_det_logprob: null argument");
    if (n_rows <= 0) return 0;
    SSAC_LAUNCH(det_logprob_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, ST, eps, n_rows, act_dim, logp);
    return ssac_check_launch("det_logprob");
}

extern "C" int ssac_td_target(const float *q_t, int n_sel, int n_rows, int q_dim,
                              const float *logp_or_logits, const float *rew, const float *done,
                              const float *log_alpha, int use_entropy, float gamma, ssac_popart *popart,
                              int pop, float *td, float *logs, void *stream) {
    if (n_sel < 1 || n_rows < 1 || q_dim < 1) return ssac_fail("ssac_td_target: bad sizes");
    SSAC_LAUNCH(td_target_kernel, dim3(1), dim3(RED_THREADS), 0, ST, q_t, n_sel, n_rows, q_dim,
                       logp_or_logits, rew, done, log_alpha, use_entropy, gamma, popart, pop, td, logs);
    return ssac_check_launch("td_target");
}

extern "C" int ssac_critic_loss_bwd(const float *q, int n_nets, int n_rows, int q_dim, const float *act,
                                    int64_t ld_act, const float *td, const float *weight,
                                    const ssac_popart *popart, int pop, float denom, float *dq,
                                    float *logs, void *stream) {
    if (n_nets < 1 || n_rows < 1 || q_dim < 1) return ssac_fail("ssac_critic_loss_bwd: bad sizes");
    if (q_dim > 1 && !act) return ssac_fail("ssac_critic_loss_bwd: discrete needs actions");
    SSAC_LAUNCH(critic_loss_bwd_kernel, dim3(1), dim3(RED_THREADS), 0, ST, q, n_nets, n_rows,
                       q_dim, act, ld_act, td, weight, popart, pop, denom, dq, logs, ssac_td_spec{});
    return ssac_check_launch("critic_loss_bwd");
}

extern "C" int ssac_critic_loss_bwd_lazy(const float *q, int n_nets, int n_rows, int q_dim, const float *act,
                                         int64_t ld_act, const ssac_td_spec *lazy_td, const float *weight,
                                         const ssac_popart *popart, int pop, float denom, float *dq, float *logs,
                                         void *stream) {
    if (n_nets < 1 || n_rows < 1 || q_dim < 1 || !lazy_td) return ssac_fail("ssac_critic_loss_bwd_lazy: bad arguments");
    if (q_dim > 1 && !act) return ssac_fail("ssac_critic_loss_bwd_lazy: discrete needs actions");
    SSAC_LAUNCH(critic_loss_bwd_kernel, dim3(1), dim3(RED_THREADS), 0, ST, q, n_nets, n_rows, q_dim, act, ld_act,
                (const float *)nullptr, weight, popart, pop, denom, dq, logs, *lazy_td);
    return ssac_check_launch("critic_loss_bwd_lazy");
}

extern "C" int ssac_dr3_blocks(void) { return 256; }

extern "C" int ssac_dr3_add(float *dz2, const float *h2, int n_nets, int batch, int hidden, float coef,
                            float *partial, void *stream) {
    if (n_nets < 1 || batch < 1 || hidden < 1) return ssac_fail("ssac_dr3_add: bad sizes");
    SSAC_LAUNCH(dr3_add_kernel, dim3(256), dim3(256), 0, ST, dz2, h2, n_nets, batch, hidden, coef, partial);
    return ssac_check_launch("dr3_add");
}

extern "C" int ssac_adv_filter_discrete(const float *q, int n_nets, int n_rows, int n_actions, const float *logits,
                                        int n_actors, const float *act, int64_t ld_act, const ssac_popart *popart,
                                        float *adv, float *mask, float *prio, float *logs, void *stream) {
    if (n_nets < 1 || n_rows < 1 || n_actions < 1 || n_actors < 1) return ssac_fail("ssac_adv_filter_discrete: bad sizes");
    SSAC_LAUNCH(adv_filter_discrete_kernel, dim3(1), dim3(RED_THREADS), 0, ST, q, n_nets, n_rows, n_actions, logits,
                n_actors, act, ld_act, popart, adv, mask, prio, logs);
    return ssac_check_launch("adv_filter_discrete");
}

extern "C" int ssac_bc_discrete_bwd(const float *logits, const float *act, int64_t ld_act, const float *mask,
                                    int n_rows, int n_actions, float inv_members, float *d_logits,
                                    float *logs_member, float *logs_total, void *stream) {
    if (n_rows < 1 || n_actions < 1) return ssac_fail("ssac_bc_discrete_bwd: bad sizes");
    SSAC_LAUNCH(bc_discrete_bwd_kernel, dim3(1), dim3(RED_THREADS), 0, ST, logits, act, ld_act, mask, n_rows, n_actions,
                inv_members, d_logits, logs_member, logs_total);
    return ssac_check_launch("bc_discrete_bwd");
}

extern "C" int ssac_adv_filter(const float *q, int n_nets, int n_rows, int n_samples, const ssac_popart *popart,
                               int use_max, float *adv, float *mask, float *prio, float *logs, void *stream) {
    if (n_nets < 1 || n_rows < 1 || n_samples < 1) return ssac_fail("ssac_adv_filter: bad sizes");
    SSAC_LAUNCH(adv_filter_kernel, dim3(1), dim3(RED_THREADS), 0, ST, q, n_nets, n_rows, n_samples, popart,
                use_max, adv, mask, prio, logs);
    return ssac_check_launch("adv_filter");
}

extern "C" int ssac_bc_logprob_bwd(const float *out, int64_t ld_out, const float *act, int64_t ld_act,
                                   const float *mask, int n_rows, int act_dim, float log_std_lo,
                                   float log_std_hi, float inv_members, float *d_out, int64_t ld_dout,
                                   float *logs_member, float *logs_total, void *stream) {
    if (n_rows < 1 || act_dim < 1) return ssac_fail("ssac_bc_logprob_bwd: bad sizes");
    SSAC_LAUNCH(bc_logprob_bwd_kernel, dim3(1), dim3(RED_THREADS), 0, ST, out, ld_out, act, ld_act, mask, n_rows,
                act_dim, log_std_lo, log_std_hi, inv_members, d_out, ld_dout, logs_member, logs_total);
    return ssac_check_launch("bc_logprob_bwd");
}

extern "C" int ssac_actor_loss_bwd(const float *q, int n_nets, int n_rows, const float *logp,
                                   const float *log_alpha, int use_entropy, const ssac_popart *popart,
                                   int pop, float inv_members, const float *qmin_global, float *dq,
                                   float *logs, void *stream) {
    if (n_nets < 1 || n_rows < 1) return ssac_fail("ssac_actor_loss_bwd: bad sizes");
    SSAC_LAUNCH(actor_loss_bwd_kernel, dim3(1), dim3(RED_THREADS), 0, ST, q, n_nets, n_rows, logp,
                       log_alpha, use_entropy, popart, pop, inv_members, qmin_global, dq, logs,
                       (const float *)nullptr);
    return ssac_check_launch("actor_loss_bwd");
}

extern "C" int ssac_actor_loss_bwd_adv(const float *q, int n_nets, int n_rows, const float *logp,
                                       const float *log_alpha, int use_entropy, const ssac_popart *popart,
                                       int pop, float inv_members, const float *adv, float *dq, float *logs,
                                       void *stream) {
    if (n_nets < 1 || n_rows < 1 || !adv) return ssac_fail("ssac_actor_loss_bwd_adv: bad arguments");
    SSAC_LAUNCH(actor_loss_bwd_kernel, dim3(1), dim3(RED_THREADS), 0, ST, q, n_nets, n_rows, logp,
                       log_alpha, use_entropy, popart, pop, inv_members, (const float *)nullptr, dq, logs, adv);
    return ssac_check_launch("actor_loss_bwd_adv");
}

extern "C" int ssac_tanh_normal_bwd(const float *dX, int n_nets, int64_t ldx, int64_t x_net_stride,
                                    int64_t act_col0, const float *out, int64_t ld_out, const float *eps,
                                    int n_rows, int act_dim, float lo, float hi, const float *log_alpha,
                                    int use_entropy, float inv_members, float *d_out, int64_t ld_dout,
                                    void *stream) {
    if (n_rows <= 0) return 0;
    SSAC_LAUNCH(tanh_normal_bwd_kernel, dim3((n_rows * act_dim + 255) / 256), dim3(256), 0, ST, dX, n_nets, ldx,
                       x_net_stride, act_col0, out, ld_out, eps, n_rows, act_dim, lo, hi, log_alpha,
                       use_entropy, inv_members, d_out, ld_dout);
    return ssac_check_launch("tanh_normal_bwd");
}

extern "C" int ssac_det_action_bwd(const float *dX, int n_nets, int64_t ldx, int64_t x_net_stride,
                                   int64_t act_col0, const float *out, int64_t ld_out, int n_rows,
                                   int act_dim, float *d_out, int64_t ld_dout, void *stream) {
    if (n_rows <= 0) return 0;
    const int total = n_rows * act_dim;
    SSAC_LAUNCH(det_action_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, ST, dX, n_nets, ldx,
                       x_net_stride, act_col0, out, ld_out, n_rows, act_dim, d_out, ld_dout);
    return ssac_check_launch("det_action_bwd");
}

extern "C" int ssac_discrete_actor_loss_bwd(const float *logits, const float *q, int n_nets, int n_rows,
                                            int n_act, const float *log_alpha, const ssac_popart *popart,
                                            int pop, float inv_members, float *d_logits, float *logs,
                                            void *stream) {
    if (n_nets < 1 || n_rows < 1 || n_act < 2) return ssac_fail("ssac_discrete_actor_loss_bwd: bad sizes");
    SSAC_LAUNCH(discrete_actor_loss_bwd_kernel, dim3(1), dim3(RED_THREADS), 0, ST, logits, q,
                       n_nets, n_rows, n_act, log_alpha, popart, pop, inv_members, d_logits, logs);
    return ssac_check_launch("discrete_actor_loss_bwd");
}

extern "C" int ssac_alpha_update(float *log_alpha, float *adam_m, float *adam_v, ssac_adam_ctl *ctl,
                                 const float *logp_or_logits, int n_rows, int n_act, float target_entropy,
                                 float *logs, void *stream) {
    if (n_rows < 1) return ssac_fail("ssac_alpha_update: bad sizes");
    SSAC_LAUNCH(alpha_update_kernel, dim3(1), dim3(RED_THREADS), 0, ST, log_alpha, adam_m, adam_v,
                       ctl, logp_or_logits, n_rows, n_act, target_entropy, logs);
    return ssac_check_launch("alpha_update");
}

extern "C" int ssac_adam_advance(ssac_adam_ctl *ctl, void *stream) {
    SSAC_LAUNCH(adam_advance_kernel, dim3(1), dim3(1), 0, ST, ctl);
    return ssac_check_launch("adam_advance");
}

extern "C" int ssac_begin_update(float *logs, int n_logs, ssac_adam_ctl *ctl, const ssac_feed *feed,
                                 void *stream) {
    if (n_logs > 256) return ssac_fail("ssac_begin_update: log block too large");
    SSAC_LAUNCH(begin_update_kernel, dim3(1), dim3(256), 0, ST, logs, n_logs, ctl, feed);
    return ssac_check_launch("begin_update");
}

extern "C" int ssac_publish_logs(const float *logs, ssac_feed *feed, void *stream) {
    if (!logs || !feed) return ssac_fail("ssac_publish_logs: null argument");
    SSAC_LAUNCH(publish_logs_kernel, dim3(1), dim3(256), 0, ST, logs, feed);
    return ssac_check_launch("publish_logs");
}

extern "C" int ssac_clip_coef(ssac_adam_ctl *ctl, const float *sumsq, int n, float max_norm,
                              float *norm_out, void *stream) {
    SSAC_LAUNCH(clip_coef_kernel, dim3(1), dim3(RED_THREADS), 0, ST, ctl, sumsq, n, max_norm,
                       norm_out);
    return ssac_check_launch("clip_coef");
}

extern "C" int ssac_group_norms(const float *sumsq, int n_groups, int group_size,
                                const ssac_adam_ctl *scale_by_clip, float *out, void *stream) {
    if (n_groups <= 0 || group_size <= 0) return 0;
    SSAC_LAUNCH(group_norms_kernel, dim3(n_groups), dim3(64), 0, ST, sumsq, group_size,
                       scale_by_clip, out);
    return ssac_check_launch("group_norms");
}

extern "C" int ssac_adam_step(float *params, float *adam_m, float *adam_v, const float *grads, int64_t n,
                              const ssac_adam_ctl *ctl, void *stream) {
    if (n <= 0) return 0;
    SSAC_LAUNCH(adam_step_kernel, dim3(grid_for(n)), dim3(256), 0, ST, params, adam_m, adam_v,
                       grads, n, ctl);
    return ssac_check_launch("adam_step");
}

extern "C" int ssac_polyak(float *target, const float *source, int64_t n, float tau, void *stream) {
    if (n <= 0) return 0;
    if ((n & 3) == 0 && (((uintptr_t)target | (uintptr_t)source) & 15) == 0) {
        const int64_t n4 = n >> 2;
        const int grid = (int)((n4 + 511) / 512 < 1024 ? (n4 + 511) / 512 : 1024);  // two quads per thread per trip
        SSAC_LAUNCH(polyak4_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, ST, (float4 *)target,
                    (const float4 *)source, n4, tau);
    } else {
        SSAC_LAUNCH(polyak_kernel, dim3(grid_for(n)), dim3(256), 0, ST, target, source, n, tau);
    }
    return ssac_check_launch("polyak");
}

extern "C" int ssac_polyak_multi(float *const *targets, const float *const *sources, const int64_t *counts, int n_tensors,
                                 float tau, void *stream) {
    if (n_tensors <= 0) return 0;
    if (!targets || !sources || !counts) return ssac_fail("ssac_polyak_multi: null argument");
    for (int i0 = 0; i0 < n_tensors; i0 += SSAC_MAX_POLYAK_SEGS) {
        PolyakSegs a{};
        int64_t most = 0;
        for (int i = i0; i < n_tensors && i < i0 + SSAC_MAX_POLYAK_SEGS; ++i) {
            if (!targets[i] || !sources[i] || counts[i] < 0) return ssac_fail("ssac_polyak_multi: bad tensor");
            if (counts[i] == 0) continue;
            a.t[a.count] = targets[i]; a.s[a.count] = sources[i]; a.n[a.count] = counts[i];
            if (counts[i] > most) most = counts[i];
            ++a.count;
        }
        if (a.count == 0) continue;
        const int64_t want = (most / 4 + 255) / 256;   // one quad per thread over the largest tensor, at most 1024 workgroups
        SSAC_LAUNCH(polyak_multi_kernel, dim3((unsigned)(want < 1 ? 1 : (want > 1024 ? 1024 : want))), dim3(256), 0, ST, a, tau);
    }
    return ssac_check_launch("polyak_multi");
}

extern "C" int ssac_zero(float *p, int64_t n, void *stream) {
    if (n <= 0) return 0;
    SSAC_LAUNCH(zero_kernel, dim3(grid_for(n)), dim3(256), 0, ST, p, n);
    return ssac_check_launch("zero");
}

extern "C" int ssac_sunrise_weights(const float *q, int n_members, int n_rows, float temp, float *w,
                                    float *logs, void *stream) {
    if (n_members < 2 || n_rows < 1) return ssac_fail("ssac_sunrise_weights: bad sizes");
    SSAC_LAUNCH(sunrise_weights_kernel, dim3(1), dim3(RED_THREADS), 0, ST, q, n_members, n_rows,
                       temp, w, logs);
    return ssac_check_launch("sunrise_weights");
}

extern "C" int ssac_softmax_weights(const float *q, int n_members, int n_rows, float temp, float *w,
                                    float *logs, void *stream) {
    if (n_members < 2 || n_rows < 1) return ssac_fail("ssac_softmax_weights: bad sizes");
    SSAC_LAUNCH(softmax_weights_kernel, dim3(1), dim3(RED_THREADS), 0, ST, q, n_members, n_rows, temp, w, logs);
    return ssac_check_launch("softmax_weights");
}

extern "C" int ssac_ensemble_min_select(const float *q, int n_nets, int n_rows, int q_dim, const float *act,
                                        int64_t ld_act, float *out, void *stream) {
    if (!q || !out || n_nets < 1 || q_dim < 1) return ssac_fail("ssac_ensemble_min_select: bad arguments");
    if (n_rows <= 0) return 0;
    const int64_t n_out = act ? n_rows : (int64_t)n_rows * q_dim;
    SSAC_LAUNCH(ensemble_min_select_kernel, dim3(grid_for(n_out, 256, 1024)), dim3(256), 0, ST, q, n_nets, n_rows,
                q_dim, act, ld_act, out);
    return ssac_check_launch("ensemble_min_select");
}

extern "C" int ssac_drq_shift(const void *src, int src_dtype, const int64_t *idx, int n, int c, int h,
                              int pad, const int64_t *shift, int mode, const float *noise, int n_aug,
                              float *dst, void *stream) {
    if (n <= 0) return 0;
    if (mode != 0 && mode != 1) return ssac_fail("ssac_drq_shift: bad mode");
    const int64_t total = (int64_t)n * c * h * h;
    if (mode == 0 && h <= SHIFT_MAX_H && (src_dtype == 0 || src_dtype == 1) && (int64_t)n * c < (1ll << 30)) {
        const size_t lds = sizeof(float) * (size_t)h * h;   // 28 KB at 84 x 84
        if (src_dtype == 0)
            SSAC_LAUNCH(drqv2_shift_plane_kernel<float>, dim3(n * c), dim3(256), lds, ST, (const float *)src, idx, c, h, pad,
                        shift, n_aug, dst);
        else
            SSAC_LAUNCH(drqv2_shift_plane_kernel<uint8_t>, dim3(n * c), dim3(256), lds, ST, (const uint8_t *)src, idx, c, h,
                        pad, shift, n_aug, dst);
        return ssac_check_launch("drq_shift");
    }
    if (src_dtype == 0)
        SSAC_LAUNCH(drq_shift_kernel<float>, dim3(grid_for(total, 256, 8192)), dim3(256), 0, ST,
                           (const float *)src, idx, n, c, h, pad, shift, mode, noise, n_aug, dst);
    else if (src_dtype == 1)
        SSAC_LAUNCH(drq_shift_kernel<uint8_t>, dim3(grid_for(total, 256, 8192)), dim3(256), 0, ST,
                           (const uint8_t *)src, idx, n, c, h, pad, shift, mode, noise, n_aug, dst);
    else
        return ssac_fail("ssac_drq_shift: unsupported src_dtype");
    return ssac_check_launch("drq_shift");
}

// ------------------------------------------------------------------ arg-min routing on a critic-sharded rank
// The online actor update routes dL/dQ through the arg-min critic of every row (learning.py:402: min over ALL critics).  A
// rank that holds n_loc of them reduces locally first: q_loc[b] = min_j Q_j[b], d_sel[b][:] = dQ_am/da of the local arg-min
// (first index on ties, as torch.min); after the MIN all-reduce of q_loc only the rank whose local minimum IS the global one
// keeps its d_sel row (the others zero it), and the SUM all-reduce that follows hands every rank the routed gradient.
__global__ void actor_route_local_kernel(const float *__restrict__ q, const float *__restrict__ dxu, int n_loc, int n_rows,
                                         int A, float *__restrict__ q_loc, float *__restrict__ q_red,
                                         float *__restrict__ d_sel) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_rows) return;
    float mq = q[b];
    int am = 0;
    for (int j = 1; j < n_loc; ++j) {
        const float v = q[(int64_t)j * n_rows + b];
        if (v < mq) { mq = v; am = j; }
    }
    q_loc[b] = mq;
    q_red[b] = mq;
    for (int i = 0; i < A; ++i) d_sel[(int64_t)b * A + i] = dxu[((int64_t)am * n_rows + b) * A + i];
}
// Ties ACROSS ranks (bit-equal minima on two ranks -- saturated Q values, identical critics): torch.min routes the gradient
// to exactly one index, the first.  Every rank whose local minimum is the global one CLAIMS the row with its rank number
// (+inf otherwise); after a MIN all-reduce of the claims the lowest claiming rank -- the owner of the first arg-min index,
// ranks hold ascending critic ranges -- keeps its row, everybody else zeroes it.
__global__ void actor_route_claim_kernel(const float *__restrict__ q_loc, const float *__restrict__ q_glob, int n_rows,
                                         int rank, float *__restrict__ claim) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_rows) return;
    claim[b] = q_loc[b] == q_glob[b] ? (float)rank : __builtin_inff();
}
__global__ void actor_route_mask_kernel(const float *__restrict__ claim, int rank, int n_rows, int A,
                                        float *__restrict__ d_sel) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * A) return;
    const int b = t / A;
    if (claim[b] != (float)rank) d_sel[t] = 0.0f;
}

extern "C" int ssac_actor_route_local(const float *q, const float *dxu, int n_local, int n_rows, int action_dim,
                                      float *q_local, float *q_reduce, float *d_sel, void *stream) {
    if (!q || !dxu || !q_local || !q_reduce || !d_sel || n_local < 1 || action_dim < 1)
        return ssac_fail("ssac_actor_route_local: bad arguments");
    if (n_rows <= 0) return 0;
    SSAC_LAUNCH(actor_route_local_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, ST, q, dxu, n_local, n_rows, action_dim,
                q_local, q_reduce, d_sel);
    return ssac_check_launch("actor_route_local");
}

extern "C" int ssac_actor_route_claim(const float *q_local, const float *q_global, int n_rows, int rank, float *claim,
                                      void *stream) {
    if (!q_local || !q_global || !claim || rank < 0) return ssac_fail("ssac_actor_route_claim: bad arguments");
    if (n_rows <= 0) return 0;
    SSAC_LAUNCH(actor_route_claim_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, ST, q_local, q_global, n_rows, rank, claim);
    return ssac_check_launch("actor_route_claim");
}

extern "C" int ssac_actor_route_mask(const float *claim, int rank, int n_rows, int action_dim, float *d_sel, void *stream) {
    if (!claim || !d_sel || action_dim < 1 || rank < 0) return ssac_fail("ssac_actor_route_mask: bad arguments");
    if (n_rows <= 0) return 0;
    SSAC_LAUNCH(actor_route_mask_kernel, dim3((n_rows * action_dim + 255) / 256), dim3(256), 0, ST, claim, rank, n_rows,
                action_dim, d_sel);
    return ssac_check_launch("actor_route_mask");
}

// ------------------------------------------------------------------ logs of the fused online actor update
// logs_loss[0] += -inv_members * sum(partials) / n_rows  (losses/actor_pg_loss accumulates over the members);
// logs_gn[0] = sqrt(sum(sumsq))  (gradients/random_actor_online_grad of the picked member)
__global__ __launch_bounds__(RED_THREADS) void actor_logs_kernel(const float *__restrict__ partials, int n_tiles,
                                                                int n_rows, float inv_members,
                                                                const float *__restrict__ sumsq, int n_ss,
                                                                float *logs_loss, float *logs_gn, const float *block,
                                                                int width, float *publish_dst) {
    __shared__ float scratch[16];
    __shared__ float fin[2];
    float s = 0.f, ss = 0.f;
    for (int i = threadIdx.x; i < n_tiles; i += blockDim.x) s += partials[i];
    for (int i = threadIdx.x; i < n_ss; i += blockDim.x) ss += sumsq[i];
    s = block_reduce<0>(s, scratch);
    ss = block_reduce<0>(ss, scratch);
    if (threadIdx.x == 0) {
        const float loss = (logs_loss ? logs_loss[0] : 0.0f) + -inv_members * s / (float)n_rows, gn = sqrtf(ss);
        if (logs_loss) logs_loss[0] = loss;
        if (logs_gn) logs_gn[0] = gn;
        fin[0] = loss; fin[1] = gn;
    }
    // publish_dst (recorded update of a single-member agent): the finished log block -> its slot of the log ring in THIS
    // launch -- the two values just computed from LDS, the rest of the block as it stands -- instead of a copy launch behind
    // every replay
    if (publish_dst) {
        __syncthreads();
        for (int i = threadIdx.x; i < width; i += blockDim.x) {
            float v = block[i];
            if (logs_loss && block + i == logs_loss) v = fin[0];
            if (logs_gn && block + i == logs_gn) v = fin[1];
            publish_dst[i] = v;
        }
    }
}

extern "C" int ssac_actor_logs(const float *partials, int n_tiles, int n_rows, float inv_members, const float *sumsq,
                               int n_sumsq, float *logs_loss, float *logs_gn, const float *block, int width, float *ring,
                               long long ring_slot, void *stream) {
    if (!partials || n_tiles <= 0 || n_rows <= 0) return ssac_fail("ssac_actor_logs: bad arguments");
    if (ring && (!block || width <= 0 || ring_slot < 0)) return ssac_fail("ssac_actor_logs: a ring needs the block, its width and a slot");
    float *dst = ring ? ring + ring_slot * width : nullptr;
    SSAC_LAUNCH(actor_logs_kernel, dim3(1), dim3(RED_THREADS), 0, ST, partials, n_tiles, n_rows, inv_members, sumsq,
                sumsq ? n_sumsq : 0, logs_loss, logs_gn, block, width, dst);
    // recorded: every replay names its own slot (ssac_replay_value2's second number)
    if (ring) ssac_record_value_patch(10, 0, 2, (long long)(uintptr_t)ring, (long long)width * 4);
    return ssac_check_launch("actor_logs");
}

// ------------------------------------------------------------------ replay push (ring-buffer add, replay.py:48-60)
// ONE launch scatters every field of n freshly collected transitions from a packed staging buffer (one async H2D copy
// of everything) into the SoA ring: rows (start + i) % capacity.  Payload bytes are copied verbatim.
struct PushField { unsigned char *dst; int64_t row_bytes; int64_t src_off; };
struct PushArgs { PushField f[SSAC_MAX_PUSH_FIELDS]; int n_fields; const unsigned char *src; int n; int64_t start, capacity; };

__global__ void replay_push_kernel(PushArgs a) {
    for (int fi = 0; fi < a.n_fields; ++fi) {
        const PushField f = a.f[fi];
        const int64_t total = (int64_t)a.n * f.row_bytes;
        const unsigned char *src = a.src + f.src_off;
        if ((f.row_bytes & 3) == 0 && (((uintptr_t)src | (uintptr_t)f.dst) & 3) == 0) {
            const int64_t rw = f.row_bytes >> 2, tw = total >> 2;
            for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < tw; i += (int64_t)gridDim.x * blockDim.x) {
                const int64_t r = i / rw, c = i - r * rw;
                reinterpret_cast<uint32_t *>(f.dst + ((a.start + r) % a.capacity) * f.row_bytes)[c] =
                    reinterpret_cast<const uint32_t *>(src)[i];
            }
        } else {
            for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
                const int64_t r = i / f.row_bytes, c = i - r * f.row_bytes;
                f.dst[((a.start + r) % a.capacity) * f.row_bytes + c] = src[i];
            }
        }
    }
}

extern "C" int ssac_replay_push(const ssac_push_field *fields, int n_fields, const void *packed, int n_rows,
                                int64_t start_row, int64_t capacity, void *stream) {
    if (!fields || !packed || n_fields <= 0 || n_fields > SSAC_MAX_PUSH_FIELDS || capacity <= 0 || start_row < 0)
        return ssac_fail("ssac_replay_push: bad arguments");
    if (n_rows <= 0) return 0;
    PushArgs a{};
    int64_t most = 0;
    for (int i = 0; i < n_fields; ++i) {
        if (!fields[i].dst || fields[i].row_bytes <= 0) return ssac_fail("ssac_replay_push: bad field");
        a.f[i] = PushField{(unsigned char *)fields[i].dst, fields[i].row_bytes, fields[i].src_offset};
        if (fields[i].row_bytes > most) most = fields[i].row_bytes;
    }
    a.n_fields = n_fields; a.src = (const unsigned char *)packed; a.n = n_rows; a.start = start_row; a.capacity = capacity;
    SSAC_LAUNCH(replay_push_kernel, dim3(grid_for((int64_t)n_rows * most / 4 + 1, 256, 2048)), dim3(256), 0, ST, a);
    return ssac_check_launch("replay_push");
}

// ------------------------------------------------------------------ error plumbing
static thread_local char g_err[256] = "";

int ssac_fail(const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return 1;
}

int ssac_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return 2;
    }
    return 0;
}

extern "C" const char *ssac_last_error(void) { return g_err; }
extern "C" int ssac_abi_version(void) { return SSAC_ABI_VERSION; }

// ---------------------------------------------------------------------------------------------
// launch lists (see ssac_internal.h)
// ---------------------------------------------------------------------------------------------
thread_local std::vector<SsacLaunchRec> *g_ssac_recording = nullptr;

struct ssac_launch_list {
    std::vector<SsacLaunchRec> recs;
};

// ---- input ring of ssac_feed (see include/ssac_hip.h)
int g_ssac_feed_device = 1;  // ssac_feed_ring_mode(0): always pinned host memory
extern "C" int ssac_feed_ring_mode(int device_ok) { g_ssac_feed_device = device_ok ? 1 : 0; return 0; }

extern "C" int ssac_feed_ring_alloc(size_t bytes, void **ring, int *device_resident) {
    if (!ring || !device_resident || bytes == 0) return ssac_fail("ssac_feed_ring_alloc: bad arguments");
    int dev = 0, large_bar = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ssac_fail("ssac_feed_ring_alloc: no device");
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) != hipSuccess) large_bar = 0;
    *ring = nullptr;
    // `bytes` of slots + the zero-initialised ring tail (SSAC_FEED_TAIL_BYTES: begun counter, Polyak requests)
    const size_t total = bytes + SSAC_FEED_TAIL_BYTES;
    if (large_bar && g_ssac_feed_device &&
        hipExtMallocWithFlags(ring, total, hipDeviceMallocUncached) == hipSuccess && *ring) {
        *device_resident = 1;
        if (hipMemset((char *)*ring + bytes, 0, SSAC_FEED_TAIL_BYTES) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
            return ssac_fail("ssac_feed_ring_alloc: cannot clear the ring tail");
        return 0;
    }
    (void)hipGetLastError();
    if (hipHostMalloc(ring, total, hipHostMallocDefault) != hipSuccess || !*ring)
        return ssac_fail("ssac_feed_ring_alloc: hipHostMalloc failed");
    memset((char *)*ring + bytes, 0, SSAC_FEED_TAIL_BYTES);
    *device_resident = 0;
    return 0;
}

extern "C" int ssac_feed_ring_free(void *ring, int device_resident) {
    if (!ring) return 0;
    const hipError_t e = device_resident ? hipFree(ring) : hipHostFree(ring);
    return e == hipSuccess ? 0 : ssac_fail("ssac_feed_ring_free: free failed");
}

extern "C" int ssac_feed_write(void *ring_slot, const void *src, size_t bytes) {
    if (!ring_slot || !src) return ssac_fail("ssac_feed_write: null pointer");
    memcpy(ring_slot, src, bytes);
    __builtin_ia32_sfence();  // drain the write-combining buffers (BAR-mapped ring) before the launch is submitted
    return 0;
}

// bit 0: fused MLP kernels, bit 1: GEMM / weight-gradient kernels; bit 2: the merged weight-gradient launch NOT per
// workgroup class, bit 3: the chained launch NOT XCD-contiguous (both are on by default)
int g_ssac_xcd = 2;
extern "C" int ssac_xcd_order(int mask) { g_ssac_xcd = mask & 15; return 0; }
long long *g_ssac_timeline = nullptr;   // [2048]: (start, end) per workgroup of the chained launch, then (from 1024) of the merged weight-gradient launch
#ifdef SSAC_LAB   // (ssac_hip_test.h, lab hooks: the product library does not define the symbol)
extern "C" int ssac_debug_timeline(long long *dev_buf) {
    g_ssac_timeline = dev_buf;
    return 0;
}
#endif

// ssac_slot_by_value(0): ssac_step_run replays without the slot pointer (the kernels go through the feed block) -- the A/B
// switch of tests/test_hip_cases.py::test_graph_replay_equals_eager_launches and tools/one_config.py
int g_ssac_slot_by_value = 1;
extern "C" int ssac_slot_by_value(int on) {
    g_ssac_slot_by_value = on ? 1 : 0;
    return 0;
}

extern "C" int ssac_record_begin(void) {
    if (g_ssac_recording) return ssac_fail("ssac_record_begin: a recording is already open on this thread");
    g_ssac_recording = new std::vector<SsacLaunchRec>();
    return 0;
}

extern "C" ssac_launch_list *ssac_record_end(void) {
    if (!g_ssac_recording) {
        ssac_fail("ssac_record_end: no recording is open on this thread");
        return nullptr;
    }
    ssac_launch_list *l = new ssac_launch_list();
    l->recs.swap(*g_ssac_recording);
    delete g_ssac_recording;
    g_ssac_recording = nullptr;
    return l;
}

extern "C" int ssac_launch_list_size(const ssac_launch_list *list) { return list ? (int)list->recs.size() : -1; }

// slot_now: the address of this update's slot of the input ring for the argument members registered with
// ssac_record_slot_patch (ssac_step_run), or null: the kernels find the slot through the feed block.
// value: this update's number for the members registered with ssac_record_value_patch (has_value: ssac_replay_value)
static int replay_list(ssac_launch_list *list, void *stream, const void *slot_now, bool has_value, long long value,
                       long long value2 = -1) {
    void *argv[64];
    for (SsacLaunchRec &r : list->recs) {
        if (r.offsets.size() > 64) return ssac_fail("ssac_replay: too many kernel arguments");
        if (!r.value_patches.empty() && !has_value)
            return ssac_fail("ssac_replay: this list holds launches that take a per-update value: replay it with ssac_replay_value");
        for (size_t off : r.slot_patches) memcpy(r.blob.data() + off, &slot_now, sizeof(void *));
        for (const SsacLaunchRec::ValuePatch &vp : r.value_patches) {
            if (vp.kind == 0) {
                const uint32_t v = (uint32_t)((value + vp.addend) & 0x7fffffff);
                memcpy(r.blob.data() + vp.off, &v, 4);
            } else if (vp.kind == 2) {
                if (value2 < 0) return ssac_fail("ssac_replay: this list holds a launch that writes into a ring slot: replay it with ssac_replay_value2");
                const long long v = vp.addend + value2 * vp.stride;
                memcpy(r.blob.data() + vp.off, &v, 8);
            } else {
                const long long v = value + vp.addend;
                memcpy(r.blob.data() + vp.off, &v, 8);
            }
        }
        for (size_t i = 0; i < r.offsets.size(); ++i) argv[i] = r.blob.data() + r.offsets[i];
        hipError_t e = hipLaunchKernel(r.func, r.grid, r.block, argv, r.lds, ST);
        if (e != hipSuccess) return ssac_check_launch("ssac_replay");
    }
    return 0;
}

extern "C" int ssac_replay(ssac_launch_list *list, void *stream) {
    if (!list) return ssac_fail("ssac_replay: null launch list");
    return replay_list(list, stream, nullptr, false, 0);
}

extern "C" int ssac_replay_value(ssac_launch_list *list, void *stream, long long value) {
    if (!list) return ssac_fail("ssac_replay_value: null launch list");
    if (value < 0) return ssac_fail("ssac_replay_value: the per-update value must not be negative");
    return replay_list(list, stream, nullptr, true, value);
}

extern "C" int ssac_replay_value2(ssac_launch_list *list, void *stream, long long value, long long value2) {
    if (!list) return ssac_fail("ssac_replay_value2: null launch list");
    if (value < 0 || value2 < 0) return ssac_fail("ssac_replay_value2: the per-update values must not be negative");
    return replay_list(list, stream, nullptr, true, value, value2);
}

extern "C" void ssac_launch_list_free(ssac_launch_list *list) { delete list; }

// ---------------------------------------------------------------------------------------------
// ssac_step: ONE host call per recorded update (include/ssac_hip.h).  Composes the update's input slot in host
// memory, copies it into the input ring (BAR-mapped device memory or pinned host memory), re-issues the recorded launch
// segments and keeps the slot-reuse events, so the per-update host work above the C ABI is the three host-RNG draws.
// ---------------------------------------------------------------------------------------------
struct ssac_step {
    char *ring; int n_slots, slot_bytes, n_rows, n_ids, ids_off, logslot_off, draw_off, event_every;
    std::vector<char> stage;
    std::vector<ssac_launch_list *> lists;
    std::vector<hipEvent_t> events;
    int64_t k;
};

extern "C" ssac_step *ssac_step_create(void *ring, int n_slots, int slot_bytes, int n_rows, int n_ids, int ids_off,
                                       int logslot_off, int draw_off, int event_every) {
    if (!ring || n_slots <= 0 || slot_bytes <= 0 || (slot_bytes & 15) || n_rows <= 0 || n_ids < 0 || event_every <= 0 ||
        n_slots % event_every != 0 || 8 * n_rows > ids_off || ids_off + 4 * n_ids > logslot_off ||
        logslot_off + 4 > slot_bytes || (draw_off >= 0 && (draw_off + 8 > slot_bytes || (draw_off & 7)))) {
        ssac_fail("ssac_step_create: bad slot geometry");
        return nullptr;
    }
    ssac_step *s = new ssac_step();
    s->ring = (char *)ring; s->n_slots = n_slots; s->slot_bytes = slot_bytes; s->n_rows = n_rows; s->n_ids = n_ids;
    s->ids_off = ids_off; s->logslot_off = logslot_off; s->draw_off = draw_off; s->event_every = event_every;
    s->stage.assign((size_t)slot_bytes, 0);
    s->events.resize(n_slots / event_every);
    for (hipEvent_t &e : s->events)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            ssac_fail("ssac_step_create: hipEventCreate failed");
            delete s;
            return nullptr;
        }
    s->k = 0;
    return s;
}

extern "C" int ssac_step_add_list(ssac_step *s, ssac_launch_list *list) {
    if (!s || !list) return ssac_fail("ssac_step_add_list: null argument");
    s->lists.push_back(list);
    return 0;
}

extern "C" int64_t ssac_step_count(const ssac_step *s) { return s ? s->k : -1; }
extern "C" int ssac_step_seek(ssac_step *s, int64_t k) {
    if (!s || k < 0) return ssac_fail("ssac_step_seek: bad argument");
    s->k = k;
    return 0;
}

extern "C" int ssac_step_run(ssac_step *s, const int64_t *idx_host, const int32_t *ids_host, int32_t log_slot,
                             int64_t draw, void *stream) {
    if (!s || !idx_host || (s->n_ids > 0 && !ids_host)) return ssac_fail("ssac_step_run: null argument");
    const int64_t k = s->k;
    const int slot = (int)(k % s->n_slots);
    const int groups = s->n_slots / s->event_every;
    // slot reuse: the update that read this slot n_slots updates ago must have finished (one event per group of
    // event_every updates, recorded behind the group's last update)
    if (k >= s->n_slots && k % s->event_every == 0) {
        if (hipEventSynchronize(s->events[((k - s->n_slots) / s->event_every) % groups]) != hipSuccess)
            return ssac_check_launch("ssac_step_run: event wait");
    }
    char *st = s->stage.data();
    memcpy(st, idx_host, 8 * (size_t)s->n_rows);
    if (s->n_ids > 0) memcpy(st + s->ids_off, ids_host, 4 * (size_t)s->n_ids);
    memcpy(st + s->logslot_off, &log_slot, 4);
    if (s->draw_off >= 0) memcpy(st + s->draw_off, &draw, 8);
    memcpy(s->ring + (size_t)slot * s->slot_bytes, st, (size_t)s->slot_bytes);
    __builtin_ia32_sfence();
    for (ssac_launch_list *l : s->lists) {
        const int rc = replay_list(l, stream, g_ssac_slot_by_value ? s->ring + (size_t)slot * s->slot_bytes : nullptr, false, 0);
        if (rc) return rc;
    }
    if (k % s->event_every == s->event_every - 1) {
        if (hipEventRecord(s->events[(k / s->event_every) % groups], ST) != hipSuccess)
            return ssac_check_launch("ssac_step_run: event record");
    }
    s->k = k + 1;
    return 0;
}

// ---- late-bound Polyak (include/ssac_hip.h).  Ring tail: int64 begun | int64 decided | uint32 last_served | pad |
//      (byte 32) n_slots x {tag, tau bits}.  The request is a posted store into the (BAR-mapped or pinned) ring; the
//      read of `begun` that follows cannot pass it (a full fence on the host side, then PCIe ordering: a read does not
//      overtake the requester's earlier posted writes), and the device publishes `begun` BEFORE it looks for the request, so "begun <= k at that read"
//      proves the request will be seen.
extern "C" int ssac_step_polyak(ssac_step *s, float tau) {
    if (!s || s->k <= 0) return 0;
    const int64_t k = s->k - 1;   // the update issued last
    char *tail = s->ring + (size_t)s->n_slots * s->slot_bytes;
    volatile uint32_t *req = reinterpret_cast<volatile uint32_t *>(tail + 32) + 2 * (k % s->n_slots);
    const uint32_t tag = (uint32_t)(k & 0x7fffffff) + 1u;
    uint32_t bits;
    memcpy(&bits, &tau, 4);
    req[1] = bits;
    __builtin_ia32_sfence();
    req[0] = tag;   // (tag last: a launch that sees the tag sees the tau)
    // FULL fence: sfence orders stores only, and on a pinned-host ring (no large BAR) the load of `begun` below could
    // pass the tag store (store-buffer litmus) -- the host would read a stale begun <= k while the device decider has
    // already looked for the tag and missed it.  mfence also drains the write-combining buffers of a BAR-mapped ring.
    __builtin_ia32_mfence();
    if (*reinterpret_cast<volatile int64_t *>(tail) <= k) return 1;
    // the device had already begun update k (it is keeping up with the host): its one decider takes a few microseconds
    const auto t0 = std::chrono::steady_clock::now();
    while (*reinterpret_cast<volatile int64_t *>(tail + 8) <= k) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) return -1;
        __builtin_ia32_pause();
    }
    return *reinterpret_cast<volatile uint32_t *>(tail + 16) == tag ? 1 : 0;
}

// after a stream synchronisation: was the last request served?
extern "C" int ssac_step_polyak_done(ssac_step *s) {
    if (!s || s->k <= 0) return 0;
    const int64_t k = s->k - 1;
    char *tail = s->ring + (size_t)s->n_slots * s->slot_bytes;
    return *reinterpret_cast<volatile uint32_t *>(tail + 16) == (uint32_t)(k & 0x7fffffff) + 1u ? 1 : 0;
}

extern "C" void ssac_step_destroy(ssac_step *s) {
    if (!s) return;
    for (hipEvent_t &e : s->events) (void)hipEventDestroy(e);
    delete s;
}
