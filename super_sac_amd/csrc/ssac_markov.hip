// Markov state-abstraction update (reference learning.py:266-341): the loss heads that are
// not plain MLP arithmetic.  The two MLPs (inverse model, contrastive model) and the encoder run on the engine's own
// forward / backward kernels; the inverse-model head is the behavioural-cloning log-probability kernel
// (ssac_bc_logprob_bwd / ssac_bc_discrete_bwd).  What is left is row work:
//
//   contrastive head   F.binary_cross_entropy(sigmoid(z), labels) over 2B rows, first B labelled 1   (learning.py:300-308)
//   smoothness term    mean_b relu(||s'_b - s_b|| / sqrt(D) - max_dist)^2                             (learning.py:310-311)
//   log block          inverse / contrastive / smoothness / total loss                                (learning.py:337-340)
//
// Batch-wide means are taken by ONE workgroup in a fixed order (run-to-run deterministic), like every other loss
// statistic of the update path.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ssac_internal.h"

namespace {

constexpr int MK_THREADS = 1024;

__device__ __forceinline__ float mk_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the workgroup, fixed order; scratch: 16 floats of LDS; result broadcast
__device__ float mk_block_sum(float v, float *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = mk_wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float r = scratch[0];
    for (int w = 1; w < nw; ++w) r += scratch[w];
    return r;
}

// p = sigmoid(z); loss = -[y log p + (1-y) log(1-p)] with the logs clamped at -100 (torch's binary_cross_entropy);
// dL/dz = dL/dp * p(1-p), dL/dp = (p - y) / max(p(1-p), 1e-12) / n  (torch's binary_cross_entropy_backward).
__global__ __launch_bounds__(MK_THREADS) void bce_sigmoid_bwd_kernel(const float *__restrict__ z, int n_pos, int n,
                                                                      float coeff, float *__restrict__ dz,
                                                                      float *__restrict__ loss_out) {
    __shared__ float scratch[16];
    float acc = 0.0f;
    const float inv_n = 1.0f / (float)n;
    for (int b = threadIdx.x; b < n; b += MK_THREADS) {
        const float y = b < n_pos ? 1.0f : 0.0f;
        const float p = 1.0f / (1.0f + expf(-z[b]));
        const float lp = fmaxf(logf(p), -100.0f), l1p = fmaxf(logf(1.0f - p), -100.0f);
        acc += -(y * lp + (1.0f - y) * l1p);
        const float pq = p * (1.0f - p);
        dz[b] = coeff * ((p - y) / fmaxf(pq, 1e-12f)) * inv_n * pq;
    }
    const float tot = mk_block_sum(acc, scratch);
    if (threadIdx.x == 0) loss_out[0] = tot * inv_n;
}

// one wave per row: dist_b = ||s1_b - s_b|| / sqrt(D); l_b = relu(dist_b - max_dist)^2; loss = mean_b l_b.
// d loss / d s1_b = coeff * 2 relu(dist_b - max_dist) / B * (s1_b - s_b) / (||s1_b - s_b|| sqrt(D))  (0 where the norm
// is 0, as torch's norm backward), d / d s_b = its negative.  ds / ds1: written (accumulate = 0) or added to.
__global__ __launch_bounds__(MK_THREADS) void smoothness_bwd_kernel(const float *__restrict__ s, int64_t lds,
                                                                     const float *__restrict__ s1, int64_t lds1,
                                                                     int n_rows, int dim, float max_dist, float coeff,
                                                                     float *__restrict__ ds, int64_t ldds,
                                                                     float *__restrict__ ds1, int64_t ldds1,
                                                                     int accumulate, float *__restrict__ loss_out) {
    __shared__ float scratch[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = MK_THREADS / 64;
    const float inv_sqrt_d = 1.0f / sqrtf((float)dim);
    float acc = 0.0f;   // lane 0 of every wave: sum of its rows' losses, rows in increasing order
    for (int b = wave; b < n_rows; b += nw) {
        float ss = 0.0f;
        for (int j = lane; j < dim; j += 64) {
            const float d = s1[(int64_t)b * lds1 + j] - s[(int64_t)b * lds + j];
            ss += d * d;
        }
        ss = mk_wave_sum(ss);
        const float nrm = sqrtf(ss);
        const float ex = fmaxf(nrm * inv_sqrt_d - max_dist, 0.0f);
        acc += ex * ex;
        if (ds || ds1) {
            const float g = nrm > 0.0f ? coeff * 2.0f * ex / (float)n_rows * inv_sqrt_d / nrm : 0.0f;
            for (int j = lane; j < dim; j += 64) {
                const float d = s1[(int64_t)b * lds1 + j] - s[(int64_t)b * lds + j];
                if (ds1) { float *p = ds1 + (int64_t)b * ldds1 + j; *p = (accumulate ? *p : 0.0f) + g * d; }
                if (ds) { float *p = ds + (int64_t)b * ldds + j; *p = (accumulate ? *p : 0.0f) - g * d; }
            }
        }
    }
    // per-wave partial sums live in lane 0; combine them in wave order
    __syncthreads();
    if (lane == 0) scratch[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.0f;
        for (int w = 0; w < nw; ++w) tot += scratch[w];
        loss_out[0] = tot / (float)n_rows;
    }
}

// Encoder invariance constraint (learning_utils.py:401-409): loss = ||A - Bm||_F over the whole (n_rows x dim) batch;
// d loss / d A = (A - Bm) / ||A - Bm||  (0 where the norm is 0, as torch.norm's backward).  One workgroup: the norm
// is a fixed-order sum.  d_a (nullable) receives coeff * gradient (written or added), loss_out[0] the norm, and
// add_to[0] (nullable) += coeff * norm (the critic update's overall-loss log includes the regulariser).
__global__ __launch_bounds__(MK_THREADS) void frobenius_diff_bwd_kernel(const float *__restrict__ A, int64_t lda,
                                                                         const float *__restrict__ Bm, int64_t ldb,
                                                                         int n_rows, int dim, float coeff,
                                                                         float *__restrict__ d_a, int64_t ldd,
                                                                         int accumulate, float *__restrict__ loss_out,
                                                                         float *__restrict__ add_to) {
    __shared__ float scratch[16];
    const int n = n_rows * dim;
    float acc = 0.0f;
    for (int i = threadIdx.x; i < n; i += MK_THREADS) {
        const int r = i / dim, c = i - r * dim;
        const float d = A[(int64_t)r * lda + c] - Bm[(int64_t)r * ldb + c];
        acc += d * d;
    }
    const float nrm = sqrtf(mk_block_sum(acc, scratch));
    if (d_a) {
        const float g = nrm > 0.0f ? coeff / nrm : 0.0f;
        for (int i = threadIdx.x; i < n; i += MK_THREADS) {
            const int r = i / dim, c = i - r * dim;
            const float d = A[(int64_t)r * lda + c] - Bm[(int64_t)r * ldb + c];
            float *p = d_a + (int64_t)r * ldd + c;
            *p = (accumulate ? *p : 0.0f) + g * d;
        }
    }
    if (threadIdx.x == 0) {
        loss_out[0] = nrm;
        if (add_to) add_to[0] += coeff * nrm;
    }
}

// Action invariance constraint (learning_utils.py:272-285), tanh-normal actor.  act: the action sampled from the actor's
// distribution at the ORIGINAL observation (no gradient), olp its log-probability there (cached pre-tanh value);
// out_a: the actor's output at the AUGMENTED observation.  alp_b = log pi_a(act_b) goes through the TanhTransform's
// inverse, atanh(clamp(a, +-0.99)) (a fresh distribution object: no cache; distributions.py:74-84), summed over the
// action dimensions; loss = F.mse_loss(olp, alp) = mean_b (olp_b - alp_b)^2; d_out = coeff * d loss / d out_a.
__global__ __launch_bounds__(MK_THREADS) void action_invariance_bwd_kernel(
    const float *__restrict__ out_a, int64_t ld_out, const float *__restrict__ act, int64_t ld_act,
    const float *__restrict__ olp, int n_rows, int A, float lo, float hi, float coeff, float *__restrict__ d_out,
    int64_t ld_dout, float *__restrict__ loss_out, float *__restrict__ add_to) {
    __shared__ float scratch[16];
    constexpr float LOG_SQRT_2PI_ = 0.91893853320467274178f, LOG_2_ = 0.69314718055994530942f;
    float acc = 0.0f;
    for (int b = threadIdx.x; b < n_rows; b += MK_THREADS) {
        float lp = 0.0f;
        for (int i = 0; i < A; ++i) {
            const float mu = out_a[b * ld_out + i], raw = out_a[b * ld_out + A + i];
            const float log_std = lo + 0.5f * (hi - lo) * (tanhf(raw) + 1.0f);
            const float sd = expf(log_std);
            const float y = fminf(fmaxf(act[b * ld_act + i], -0.99f), 0.99f);
            const float x = 0.5f * (log1pf(y) - log1pf(-y));
            const float dlt = x - mu;
            const float sp = -2.0f * x > 20.0f ? -2.0f * x : log1pf(expf(-2.0f * x));   // F.softplus
            lp += (-(dlt * dlt) / (2.0f * sd * sd) - log_std - LOG_SQRT_2PI_) - 2.0f * (LOG_2_ - x - sp);
        }
        const float diff = lp - olp[b];
        acc += diff * diff;
        const float coef = coeff * 2.0f * diff / (float)n_rows;   // d loss / d alp_b
        for (int i = 0; i < A; ++i) {
            const float mu = out_a[b * ld_out + i], raw = out_a[b * ld_out + A + i];
            const float t = tanhf(raw);
            const float sd = expf(lo + 0.5f * (hi - lo) * (t + 1.0f));
            const float y = fminf(fmaxf(act[b * ld_act + i], -0.99f), 0.99f);
            const float dlt = 0.5f * (log1pf(y) - log1pf(-y)) - mu;
            const float r2 = (dlt * dlt) / (sd * sd);
            d_out[b * ld_dout + i] = coef * (dlt / (sd * sd));
            d_out[b * ld_dout + A + i] = coef * (r2 - 1.0f) * 0.5f * (hi - lo) * (1.0f - t * t);
        }
    }
    const float tot = mk_block_sum(acc, scratch);
    if (threadIdx.x == 0) {
        loss_out[0] = tot / (float)n_rows;
        if (add_to) add_to[0] += coeff * tot / (float)n_rows;
    }
}

// ... categorical actor.  The reference sums the (B,) log-probabilities over dim -1 before the mse
// (learning_utils.py:280-285: `.sum(-1, keepdim=True)` of a Categorical's log_prob), so the loss is
// (sum_b olp_b - sum_b alp_b)^2 -- reproduced as is.  logits_o / logits_a: actor outputs at the original / augmented obs.
__global__ __launch_bounds__(MK_THREADS) void action_invariance_discrete_bwd_kernel(
    const float *__restrict__ logits_o, const float *__restrict__ logits_a, const float *__restrict__ act, int n_rows,
    int A, float coeff, float *__restrict__ d_logits, float *__restrict__ loss_out, float *__restrict__ add_to) {
    __shared__ float scratch[16];
    auto logp_of = [&](const float *x, int ai, float &lse) {
        float mx = x[0];
        for (int t = 1; t < A; ++t) mx = fmaxf(mx, x[t]);
        float se = 0.0f;
        for (int t = 0; t < A; ++t) se += expf(x[t] - mx);
        lse = mx + logf(se);
        return x[ai] - lse;
    };
    float acc = 0.0f;
    for (int b = threadIdx.x; b < n_rows; b += MK_THREADS) {
        const int ai = (int)act[b];
        float l0, l1;
        acc += logp_of(logits_a + (int64_t)b * A, ai, l1) - logp_of(logits_o + (int64_t)b * A, ai, l0);
    }
    const float S = mk_block_sum(acc, scratch);   // sum_b alp_b - sum_b olp_b
    for (int b = threadIdx.x; b < n_rows; b += MK_THREADS) {
        const int ai = (int)act[b];
        const float *x = logits_a + (int64_t)b * A;
        float lse;
        (void)logp_of(x, ai, lse);
        for (int t = 0; t < A; ++t)
            d_logits[(int64_t)b * A + t] = coeff * 2.0f * S * ((t == ai ? 1.0f : 0.0f) - expf(x[t] - lse));
    }
    if (threadIdx.x == 0) {
        loss_out[0] = S * S;
        if (add_to) add_to[0] += coeff * S * S;
    }
}

// logs[0..4) = inverse loss, contrastive loss, smoothness loss, total (learning.py:313-317, 337-340)
__global__ void markov_logs_kernel(const float *inv_raw, float inv_scale, const float *con, const float *smooth,
                                   float ic, float cc, float sc, float *logs) {
    if (threadIdx.x != 0) return;
    const float li = inv_raw[0] * inv_scale, lc = con[0], ls = smooth[0];
    logs[0] = li;
    logs[1] = lc;
    logs[2] = ls;
    logs[3] = ic * li + cc * lc + sc * ls;
}

}  // namespace

extern "C" int ssac_bce_sigmoid_bwd(const float *z, int n_pos, int n, float coeff, float *dz, float *loss_out,
                                    void *stream) {
    if (!z || !dz || !loss_out || n <= 0 || n_pos < 0 || n_pos > n) return ssac_fail("ssac_bce_sigmoid_bwd: bad arguments");
    SSAC_LAUNCH(bce_sigmoid_bwd_kernel, dim3(1), dim3(MK_THREADS), 0, (hipStream_t)stream, z, n_pos, n, coeff, dz,
                loss_out);
    return ssac_check_launch("bce_sigmoid_bwd");
}

extern "C" int ssac_markov_smoothness_bwd(const float *s, int64_t lds, const float *s1, int64_t lds1, int n_rows,
                                          int dim, float max_dist, float coeff, float *ds, int64_t ldds, float *ds1,
                                          int64_t ldds1, int accumulate, float *loss_out, void *stream) {
    if (!s || !s1 || !loss_out || n_rows <= 0 || dim <= 0) return ssac_fail("ssac_markov_smoothness_bwd: bad arguments");
    SSAC_LAUNCH(smoothness_bwd_kernel, dim3(1), dim3(MK_THREADS), 0, (hipStream_t)stream, s, lds, s1, lds1, n_rows, dim,
                max_dist, coeff, ds, ldds, ds1, ldds1, accumulate, loss_out);
    return ssac_check_launch("markov_smoothness_bwd");
}

extern "C" int ssac_markov_logs(const float *inverse_raw, float inverse_scale, const float *contrastive,
                                const float *smoothness, float inverse_coeff, float contrastive_coeff,
                                float smoothness_coeff, float *logs, void *stream) {
    if (!inverse_raw || !contrastive || !smoothness || !logs) return ssac_fail("ssac_markov_logs: null argument");
    SSAC_LAUNCH(markov_logs_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, inverse_raw, inverse_scale, contrastive,
                smoothness, inverse_coeff, contrastive_coeff, smoothness_coeff, logs);
    return ssac_check_launch("markov_logs");
}

extern "C" int ssac_frobenius_diff_bwd(const float *a, int64_t lda, const float *b, int64_t ldb, int n_rows, int dim,
                                       float coeff, float *d_a, int64_t ldd, int accumulate, float *loss_out,
                                       float *add_to, void *stream) {
    if (!a || !b || !loss_out || n_rows <= 0 || dim <= 0) return ssac_fail("ssac_frobenius_diff_bwd: bad arguments");
    SSAC_LAUNCH(frobenius_diff_bwd_kernel, dim3(1), dim3(MK_THREADS), 0, (hipStream_t)stream, a, lda, b, ldb, n_rows, dim,
                coeff, d_a, ldd, accumulate, loss_out, add_to);
    return ssac_check_launch("frobenius_diff_bwd");
}

extern "C" int ssac_action_invariance_bwd(const float *out_a, int64_t ld_out, const float *act, int64_t ld_act,
                                          const float *olp, int n_rows, int act_dim, float log_std_lo, float log_std_hi,
                                          float coeff, float *d_out, int64_t ld_dout, float *loss_out, float *add_to,
                                          void *stream) {
    if (!out_a || !act || !olp || !d_out || !loss_out || n_rows <= 0 || act_dim <= 0)
        return ssac_fail("ssac_action_invariance_bwd: bad arguments");
    SSAC_LAUNCH(action_invariance_bwd_kernel, dim3(1), dim3(MK_THREADS), 0, (hipStream_t)stream, out_a, ld_out, act, ld_act,
                olp, n_rows, act_dim, log_std_lo, log_std_hi, coeff, d_out, ld_dout, loss_out, add_to);
    return ssac_check_launch("action_invariance_bwd");
}

// ---- deterministic actors (nets/mlps.py:78-93: loc = tanh(out)) in the offline actor update.  The reference treats
// them as Normal(loc, 1e-4) (distributions.py:107-114), so log pi(a) = sum_k [-(a_k - loc_k)^2 / (2 var) - log sd -
// log sqrt(2 pi)] with sd = 1e-4f, var = sd * sd in fp32 (torch.distributions.Normal.log_prob, operation by operation).
constexpr float DET_SD = 1e-4f;

// filtered behaviour cloning (learning_utils.py:241-269): loss_i = -mean_b(mask_b log pi(a_b)); d_out = d(loss_i / E)/d out
__global__ __launch_bounds__(MK_THREADS) void bc_det_logprob_bwd_kernel(
    const float *__restrict__ out, int64_t ld_out, const float *__restrict__ act, int64_t ld_act,
    const float *__restrict__ mask, int n_rows, int A, float inv_members, float *__restrict__ d_out, int64_t ld_dout,
    float *__restrict__ logs_member, float *__restrict__ logs_total) {
    __shared__ float scratch[16];
    constexpr float LOG_SQRT_2PI_ = 0.91893853320467274178f;
    const float var = DET_SD * DET_SD, log_sd = logf(DET_SD);
    float acc = 0.0f;
    for (int b = threadIdx.x; b < n_rows; b += MK_THREADS) {
        const float w = mask ? mask[b] : 1.0f;
        const float coef = -w * inv_members / (float)n_rows;
        float lp = 0.0f;
        for (int i = 0; i < A; ++i) {
            const float loc = tanhf(out[b * ld_out + i]);
            const float dlt = act[b * ld_act + i] - loc;
            lp += -(dlt * dlt) / (2.0f * var) - log_sd - LOG_SQRT_2PI_;
            d_out[b * ld_dout + i] = coef * (dlt / var) * (1.0f - loc * loc);
        }
        acc += lp * w;
    }
    const float tot = mk_block_sum(acc, scratch);
    if (threadIdx.x == 0) {
        const float loss = -tot / (float)n_rows;
        if (logs_member) logs_member[0] = loss;
        if (logs_total) logs_total[0] += loss * inv_members;
    }
}

// action invariance (learning_utils.py:272-285): a = o_dist.sample() = loc_o (distributions.py:113-114), so
// olp_b = A (-log sd - log sqrt(2 pi)) exactly and alp_b = sum_k [-(loc_o - loc_a)^2 / (2 var) - log sd - log sqrt(2 pi)];
// loss = mean_b (olp_b - alp_b)^2; d_out = coeff * d loss / d out_a (nothing flows to out_o: no_grad in the reference).
__global__ __launch_bounds__(MK_THREADS) void action_invariance_det_bwd_kernel(
    const float *__restrict__ out_o, int64_t ld_o, const float *__restrict__ out_a, int64_t ld_a, int n_rows, int A,
    float coeff, float *__restrict__ d_out, int64_t ld_dout, float *__restrict__ loss_out, float *__restrict__ add_to) {
    __shared__ float scratch[16];
    constexpr float LOG_SQRT_2PI_ = 0.91893853320467274178f;
    const float var = DET_SD * DET_SD, log_sd = logf(DET_SD);
    float acc = 0.0f;
    for (int b = threadIdx.x; b < n_rows; b += MK_THREADS) {
        float olp = 0.0f, alp = 0.0f;
        for (int i = 0; i < A; ++i) {
            const float a = tanhf(out_o[b * ld_o + i]), loc = tanhf(out_a[b * ld_a + i]);
            const float z = a - a;   // (value - loc of the distribution the action was drawn from)
            olp += -(z * z) / (2.0f * var) - log_sd - LOG_SQRT_2PI_;
            const float dlt = a - loc;
            alp += -(dlt * dlt) / (2.0f * var) - log_sd - LOG_SQRT_2PI_;
        }
        const float diff = olp - alp;
        acc += diff * diff;
        const float coef = coeff * 2.0f * diff / (float)n_rows;   // d loss / d (olp - alp); d(-alp)/d loc = -(a - loc)/var
        for (int i = 0; i < A; ++i) {
            const float a = tanhf(out_o[b * ld_o + i]), loc = tanhf(out_a[b * ld_a + i]);
            d_out[b * ld_dout + i] = coef * (-(a - loc) / var) * (1.0f - loc * loc);
        }
    }
    const float tot = mk_block_sum(acc, scratch);
    if (threadIdx.x == 0) {
        loss_out[0] = tot / (float)n_rows;
        if (add_to) add_to[0] += coeff * tot / (float)n_rows;
    }
}

extern "C" int ssac_bc_det_logprob_bwd(const float *out, int64_t ld_out, const float *act, int64_t ld_act, const float *mask,
                                       int n_rows, int act_dim, float inv_members, float *d_out, int64_t ld_dout,
                                       float *logs_member, float *logs_total, void *stream) {
    if (!out || !act || !d_out || n_rows <= 0 || act_dim <= 0) return ssac_fail("ssac_bc_det_logprob_bwd: bad arguments");
    SSAC_LAUNCH(bc_det_logprob_bwd_kernel, dim3(1), dim3(MK_THREADS), 0, (hipStream_t)stream, out, ld_out, act, ld_act, mask,
                n_rows, act_dim, inv_members, d_out, ld_dout, logs_member, logs_total);
    return ssac_check_launch("bc_det_logprob_bwd");
}

extern "C" int ssac_action_invariance_det_bwd(const float *out_o, int64_t ld_o, const float *out_a, int64_t ld_a, int n_rows,
                                              int act_dim, float coeff, float *d_out, int64_t ld_dout, float *loss_out,
                                              float *add_to, void *stream) {
    if (!out_o || !out_a || !d_out || !loss_out || n_rows <= 0 || act_dim <= 0)
        return ssac_fail("ssac_action_invariance_det_bwd: bad arguments");
    SSAC_LAUNCH(action_invariance_det_bwd_kernel, dim3(1), dim3(MK_THREADS), 0, (hipStream_t)stream, out_o, ld_o, out_a, ld_a,
                n_rows, act_dim, coeff, d_out, ld_dout, loss_out, add_to);
    return ssac_check_launch("action_invariance_det_bwd");
}

extern "C" int ssac_action_invariance_discrete_bwd(const float *logits_o, const float *logits_a, const float *act,
                                                   int n_rows, int n_actions, float coeff, float *d_logits,
                                                   float *loss_out, float *add_to, void *stream) {
    if (!logits_o || !logits_a || !act || !d_logits || !loss_out || n_rows <= 0 || n_actions <= 0)
        return ssac_fail("ssac_action_invariance_discrete_bwd: bad arguments");
    SSAC_LAUNCH(action_invariance_discrete_bwd_kernel, dim3(1), dim3(MK_THREADS), 0, (hipStream_t)stream, logits_o, logits_a,
                act, n_rows, n_actions, coeff, d_logits, loss_out, add_to);
    return ssac_check_launch("action_invariance_discrete_bwd");
}
