// The ACTING path's device side (agent.py:204-315): what an environment step pays before any update.
//
// The collection loop hands over ONE observation per environment (a few dozen floats) and waits for ONE action: the
// arithmetic is two or three tiny MLP passes, so the cost of a call is its launches and its two trips over PCIe.  Round 5
// paid a pageable H2D copy, 3-25 torch / library launches each submitted from Python, and a synchronising D2H copy
// (69 us for a SAC / REDQ agent, 441 us for SUNRISE's UCB rule on five members: bench.py `secondary.acting`).  Here:
//
//   ssac_act       an observation buffer the HOST writes directly (uncached device memory behind the large BAR, or pinned
//                  host memory), a result buffer the DEVICE writes directly (pinned host memory) with a sequence word behind
//                  it, a device-resident call counter (the draw number of the engine's Philox stream: ssac_rng.counter),
//                  and one recorded launch list.  ssac_act_run = memcpy + sfence, re-issue the list, spin on the sequence
//                  word, memcpy: no hipMemcpy, no stream synchronisation, one C call.
//   the reductions of the acting rules as kernels of their own (they were torch ops): SUNRISE's UCB rule (min over a
//                  member's critics, mean + bonus * unbiased std over the members, arg-max over the candidates, gather),
//                  the mean of the actors' mean actions, the categorical draw and the arg-max of the mean probabilities.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <chrono>
#include <vector>

#include "ssac_internal.h"
#include "ssac_philox.h"

#define ST ((hipStream_t)stream)

namespace {

constexpr int ACT_MAX_E = 8;

struct PtrList { const float *p[ACT_MAX_E]; };

// ---- publish: result -> pinned host memory, then the sequence word (what the host spins on), counter += 1
__global__ __launch_bounds__(256) void act_publish_kernel(const float *__restrict__ src, int n, float *dst,
                                                         unsigned long long *seq, long long *counter) {
    for (int i = threadIdx.x; i < n; i += 256) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: the payload before the word
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long c = *counter + 1;
        *counter = c;
        __hip_atomic_store(seq, (unsigned long long)c, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- SUNRISE's UCB rule (agent.py:262-300) on the stacked candidates.  q[c]: member c's critics on row (a n_rows + b) of X,
// (n_nets x n_cand n_rows); value of candidate a for member c = min over its nets (agent.Critic.forward, return_min);
// score = mean_c + bonus * std_c (unbiased, torch.std); best = first arg-max over a (torch.argmax); the action = columns
// [col0, col0 + A) of X's row (best n_rows + b), clamped to [-1, 1] (_process_act).
__global__ void ucb_select_kernel(PtrList q, int n_members, int n_nets, int n_cand, int n_rows, float bonus,
                                  const float *__restrict__ X, int64_t ldx, int col0, int A, float *__restrict__ act) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_rows) return;
    const int64_t per_net = (int64_t)n_cand * n_rows;
    int best = 0;
    float best_v = 0.0f;
    for (int a = 0; a < n_cand; ++a) {
        float v[ACT_MAX_E];   // (constant trip counts + guards: the array stays in registers)
        float sum = 0.0f;
#pragma unroll
        for (int c = 0; c < ACT_MAX_E; ++c) {
            v[c] = 0.0f;
            if (c < n_members) {
                float m = q.p[c][(int64_t)a * n_rows + b];
                for (int j = 1; j < n_nets; ++j) m = fminf(m, q.p[c][j * per_net + (int64_t)a * n_rows + b]);
                v[c] = m;
                sum += m;
            }
        }
        const float mean = sum / (float)n_members;
        float ss = 0.0f;
#pragma unroll
        for (int c = 0; c < ACT_MAX_E; ++c)
            if (c < n_members) ss += (v[c] - mean) * (v[c] - mean);
        const float score = mean + bonus * sqrtf(ss / (float)(n_members - 1));
        if (a == 0 || score > best_v) { best = a; best_v = score; }
    }
    for (int i = 0; i < A; ++i)
        act[(int64_t)b * A + i] = fminf(fmaxf(X[((int64_t)best * n_rows + b) * ldx + col0 + i], -1.0f), 1.0f);
}

// ---- the candidates of the UCB rule from the PACKED actors' head outputs: row (e n_rows + b) of X = [s_b | a_e,b] with
// a = tanh(mu + sd eps) (distributions.py:9-15, 64-104; tanh_normal_fwd_kernel's arithmetic), eps = element (b, i) of the
// engine's Philox stream at draw rng.offset + e * member_stride (+ *rng.counter): member e draws what a launch of its own
// (ssac_actor_sample_concat_fused with that offset) would draw
__global__ void ucb_candidates_kernel(const float *__restrict__ outs, int n_actors, int n_rows, int A,
                                      const float *__restrict__ S_rows, int64_t lds, int S, float lo, float hi, RngArgs rng,
                                      long long member_stride, float *__restrict__ X, int64_t ldx) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int W = S + A;
    if (idx >= n_actors * n_rows * W) return;
    const int r = idx / W, c = idx - r * W;      // r = e n_rows + b
    const int e = r / n_rows, b = r - e * n_rows;
    if (c < S) { X[(int64_t)r * ldx + c] = S_rows[(int64_t)b * lds + c]; return; }
    const int i = c - S;
    const float *o = outs + ((int64_t)e * n_rows + b) * (2 * A);
    const float mu = o[i], raw = o[A + i];
    const float log_std = lo + 0.5f * (hi - lo) * (tanhf(raw) + 1.0f);
    const float sd = expf(log_std);
    const float eps = philox_normal(rng.seed, rng_draw(rng) + (long long)e * member_stride, b, i);
    X[(int64_t)r * ldx + c] = tanhf(mu + sd * eps);
}

// ---- greedy continuous action (agent.py:204-246): mean over the actors of dist.mean = tanh(mu) (SquashedNormal.mean /
// the deterministic actor's tanh(out)), clamped
__global__ void mean_tanh_kernel(PtrList outs, int n_actors, int64_t ld_out, int n_rows, int A, float *__restrict__ act) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * A) return;
    const int b = i / A, k = i - b * A;
    float s = 0.0f;
    for (int e = 0; e < n_actors; ++e) s += tanhf(outs.p[e][b * ld_out + k]);
    act[i] = fminf(fmaxf(s / (float)n_actors, -1.0f), 1.0f);
}

// ---- one row's copy + clamp (the sampled action of a single actor: columns [col0, col0 + A) of a row-major buffer)
__global__ void take_clamp_kernel(const float *__restrict__ src, int64_t ld, int col0, int n_rows, int A, float lo, float hi,
                                  float *__restrict__ act) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * A) return;
    const int b = i / A, k = i - b * A;
    act[i] = fminf(fmaxf(src[b * ld + col0 + k], lo), hi);
}

// ---- discrete actors.  greedy: arg-max of the mean over the actors of softmax(logits) (agent.py:218-226); sample:
// Categorical(logits).sample() of ONE actor (agent.py:301-309) by inversion of the cumulative distribution with one
// uniform per row from the engine's Philox stream (element (row, 0) of the draw's first word).  The index leaves as a float.
__global__ void discrete_act_kernel(PtrList outs, int n_actors, int64_t ld_out, int n_rows, int A, int sample, RngArgs rng,
                                    float *__restrict__ act) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_rows) return;
    // per-actor softmax statistics in registers (constant trip counts + guards), the mean probability of an action recomputed
    // where it is needed: no per-thread array indexed at run time, no scratch memory
    float mx[ACT_MAX_E], rz[ACT_MAX_E];
#pragma unroll
    for (int e = 0; e < ACT_MAX_E; ++e) {
        mx[e] = 0.0f; rz[e] = 0.0f;
        if (e < n_actors) {
            const float *o = outs.p[e] + b * ld_out;
            float m = o[0];
            for (int k = 1; k < A; ++k) m = fmaxf(m, o[k]);
            float z = 0.0f;
            for (int k = 0; k < A; ++k) z += expf(o[k] - m);
            mx[e] = m; rz[e] = 1.0f / z;
        }
    }
    auto prob = [&](int k) {
        float pk = 0.0f;
#pragma unroll
        for (int e = 0; e < ACT_MAX_E; ++e)
            if (e < n_actors) pk += expf(outs.p[e][b * ld_out + k] - mx[e]) * rz[e];
        return pk;
    };
    int pick = 0;
    if (sample) {
        uint32_t c[4] = {(uint32_t)b, 0u, 0u, 0u};
        const int64_t draw = rng_draw(rng);
        c[2] = (uint32_t)draw; c[3] = (uint32_t)((uint64_t)draw >> 32);
        philox4x32_10(c, (uint32_t)rng.seed, (uint32_t)(rng.seed >> 32));
        float tot = 0.0f;
        for (int k = 0; k < A; ++k) tot += prob(k);
        const float u = (float)c[0] * 2.3283064365386963e-10f * tot;   // [0, tot)
        float cum = 0.0f;
        pick = A - 1;
        for (int k = 0; k < A; ++k) {
            cum += prob(k);
            if (u < cum) { pick = k; break; }
        }
    } else {
        float best = prob(0);
        for (int k = 1; k < A; ++k) {
            const float pk = prob(k);
            if (pk > best) { best = pk; pick = k; }   // (first maximum: torch.argmax)
        }
    }
    act[b] = (float)pick;
}

}  // namespace

struct ssac_act {
    void *obs; int obs_device; size_t obs_bytes;   // host-writable observation buffer (+ its kind, for the free)
    float *out_host, *out_dev; size_t out_bytes;   // pinned result buffer: host view, device view; the sequence word sits behind it
    long long *counter;                            // device-resident call counter
    unsigned long long calls;                      // host mirror
    ssac_launch_list *list;
    std::vector<ssac_launch_list *> lists;         // (one per actor for agents that draw a random actor per call)
};

extern "C" ssac_act *ssac_act_create(int obs_bytes, int out_floats) {
    if (obs_bytes <= 0 || out_floats <= 0) { ssac_fail("ssac_act_create: bad sizes"); return nullptr; }
    ssac_act *a = new ssac_act();
    a->obs_bytes = ((size_t)obs_bytes + 63) & ~(size_t)63;
    a->out_bytes = (((size_t)out_floats * 4 + 63) & ~(size_t)63) + 64;   // + the sequence word's line
    a->list = nullptr;
    a->calls = 0;
    int dev = 0, large_bar = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) != hipSuccess) large_bar = 0;
    a->obs = nullptr;
    a->obs_device = 0;
    if (large_bar && hipExtMallocWithFlags(&a->obs, a->obs_bytes, hipDeviceMallocUncached) == hipSuccess && a->obs) {
        a->obs_device = 1;
    } else {
        (void)hipGetLastError();
        if (hipHostMalloc(&a->obs, a->obs_bytes, hipHostMallocMapped) != hipSuccess) { delete a; ssac_fail("ssac_act_create: no observation buffer"); return nullptr; }
    }
    if (hipHostMalloc((void **)&a->out_host, a->out_bytes, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&a->out_dev, a->out_host, 0) != hipSuccess ||
        hipMalloc((void **)&a->counter, 64) != hipSuccess || hipMemset(a->counter, 0, 64) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {
        ssac_fail("ssac_act_create: allocation failed");
        delete a;
        return nullptr;
    }
    memset(a->out_host, 0, a->out_bytes);
    return a;
}

static void *act_obs_dev(ssac_act *a) {
    if (a->obs_device) return a->obs;
    void *d = nullptr;
    (void)hipHostGetDevicePointer(&d, a->obs, 0);
    return d;
}

extern "C" void *ssac_act_obs(ssac_act *a) { return a ? act_obs_dev(a) : nullptr; }
extern "C" const int64_t *ssac_act_counter(ssac_act *a) { return a ? reinterpret_cast<const int64_t *>(a->counter) : nullptr; }

extern "C" int ssac_act_publish(ssac_act *a, const float *src, int n, void *stream) {
    if (!a || !src || n <= 0 || (size_t)n * 4 + 64 > a->out_bytes) return ssac_fail("ssac_act_publish: bad arguments");
    unsigned long long *seq = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a->out_dev) + a->out_bytes - 64);
    SSAC_LAUNCH(act_publish_kernel, dim3(1), dim3(256), 0, ST, src, n, a->out_dev, seq, a->counter);
    return ssac_check_launch("ssac_act_publish");
}

// the plan's launch lists: index 0 for agents with one list; an agent that draws a random actor per call records one list
// per actor and names it at run time
extern "C" int ssac_act_add_list(ssac_act *a, ssac_launch_list *list) {
    if (!a || !list) { ssac_fail("ssac_act_add_list: null argument"); return -1; }
    // the recording pass ISSUED its launches too (on whatever the observation buffer held), the publish step included: the
    // device-side call counter has advanced.  Drain the device and take the host's count from the sequence word, so that the
    // next ssac_act_run waits for the number its own publish step will write.
    if (hipDeviceSynchronize() != hipSuccess) { ssac_check_launch("ssac_act_add_list"); return -1; }
    a->calls = __atomic_load_n(reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a->out_host) + a->out_bytes - 64),
                               __ATOMIC_ACQUIRE);
    a->lists.push_back(list);
    return (int)a->lists.size() - 1;
}

extern "C" int ssac_act_run(ssac_act *a, int which, const void *obs_host, int obs_bytes, float *out_host, int out_floats,
                            void *stream) {
    if (!a || which < 0 || which >= (int)a->lists.size() || !obs_host || !out_host || obs_bytes < 0 ||
        (size_t)obs_bytes > a->obs_bytes || (size_t)out_floats * 4 + 64 > a->out_bytes)
        return ssac_fail("ssac_act_run: bad arguments");
    memcpy(a->obs, obs_host, (size_t)obs_bytes);
    __builtin_ia32_sfence();   // (write-combining buffers of the BAR mapping drained before the launches are submitted)
    const int rc = ssac_replay(a->lists[which], stream);
    if (rc) return rc;
    const unsigned long long want = a->calls + 1;
    volatile unsigned long long *seq =
        reinterpret_cast<volatile unsigned long long *>(reinterpret_cast<char *>(a->out_host) + a->out_bytes - 64);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != want) {
        __builtin_ia32_pause();
        if ((++spins & 0x3fff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
            (void)hipStreamSynchronize(ST);   // (surface a launch failure if that is what happened)
            if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == want) break;
            return ssac_check_launch("ssac_act_run") ? 1 : ssac_fail("ssac_act_run: the result did not arrive within 5 s");
        }
    }
    a->calls = want;
    memcpy(out_host, a->out_host, (size_t)out_floats * 4);
    return 0;
}

extern "C" long long ssac_act_calls(const ssac_act *a) { return a ? (long long)a->calls : -1; }

extern "C" void ssac_act_destroy(ssac_act *a) {
    if (!a) return;
    (void)hipDeviceSynchronize();
    if (a->obs) (void)(a->obs_device ? hipFree(a->obs) : hipHostFree(a->obs));
    if (a->out_host) (void)hipHostFree(a->out_host);
    if (a->counter) (void)hipFree(a->counter);
    for (ssac_launch_list *l : a->lists) ssac_launch_list_free(l);
    delete a;
}

// ---- the acting rules' reductions (launchable on their own, recordable)
static int fill_ptrs(PtrList &pl, const float *const *ptrs, int n, const char *who) {
    if (!ptrs || n <= 0 || n > ACT_MAX_E) return ssac_fail(who);
    for (int i = 0; i < ACT_MAX_E; ++i) pl.p[i] = ptrs[i < n ? i : 0];
    return 0;
}

extern "C" int ssac_ucb_select(const float *const *q_members, int n_members, int n_nets, int n_cand, int n_rows, float bonus,
                               const float *X, int64_t ldx, int col0, int act_dim, float *act, void *stream) {
    PtrList pl;
    if (fill_ptrs(pl, q_members, n_members, "ssac_ucb_select: 1..8 members")) return 1;
    if (n_members < 2 || n_nets <= 0 || n_cand <= 0 || n_rows <= 0 || !X || !act || act_dim <= 0)
        return ssac_fail("ssac_ucb_select: bad arguments (the unbiased std needs >= 2 members)");
    SSAC_LAUNCH(ucb_select_kernel, dim3((n_rows + 63) / 64), dim3(64), 0, ST, pl, n_members, n_nets, n_cand, n_rows, bonus, X,
                ldx, col0, act_dim, act);
    return ssac_check_launch("ssac_ucb_select");
}

extern "C" int ssac_act_candidates(const float *outs, int n_actors, int n_rows, int act_dim, const float *S_rows, int64_t lds,
                                   int state_dim, float log_std_lo, float log_std_hi, const ssac_rng *rng, long long member_stride,
                                   float *X, int64_t ldx, void *stream) {
    if (!outs || !S_rows || !X || !rng || n_actors <= 0 || n_rows <= 0 || act_dim <= 0 || state_dim <= 0 || ldx < state_dim + act_dim)
        return ssac_fail("ssac_act_candidates: bad arguments");
    const int n = n_actors * n_rows * (state_dim + act_dim);
    SSAC_LAUNCH(ucb_candidates_kernel, dim3((n + 255) / 256), dim3(256), 0, ST, outs, n_actors, n_rows, act_dim, S_rows, lds,
                state_dim, log_std_lo, log_std_hi, RngArgs{rng->seed, rng->counter, rng->offset}, member_stride, X, ldx);
    return ssac_check_launch("ssac_act_candidates");
}

extern "C" int ssac_act_mean_tanh(const float *const *outs, int n_actors, int64_t ld_out, int n_rows, int act_dim, float *act,
                                  void *stream) {
    PtrList pl;
    if (fill_ptrs(pl, outs, n_actors, "ssac_act_mean_tanh: 1..8 actors")) return 1;
    if (n_rows <= 0 || act_dim <= 0 || !act) return ssac_fail("ssac_act_mean_tanh: bad arguments");
    const int n = n_rows * act_dim;
    SSAC_LAUNCH(mean_tanh_kernel, dim3((n + 255) / 256), dim3(256), 0, ST, pl, n_actors, ld_out, n_rows, act_dim, act);
    return ssac_check_launch("ssac_act_mean_tanh");
}

extern "C" int ssac_act_take_clamp(const float *src, int64_t ld, int col0, int n_rows, int act_dim, float lo, float hi,
                                   float *act, void *stream) {
    if (!src || !act || n_rows <= 0 || act_dim <= 0) return ssac_fail("ssac_act_take_clamp: bad arguments");
    const int n = n_rows * act_dim;
    SSAC_LAUNCH(take_clamp_kernel, dim3((n + 255) / 256), dim3(256), 0, ST, src, ld, col0, n_rows, act_dim, lo, hi, act);
    return ssac_check_launch("ssac_act_take_clamp");
}

extern "C" int ssac_act_discrete(const float *const *outs, int n_actors, int64_t ld_out, int n_rows, int n_actions, int sample,
                                 const ssac_rng *rng, float *act, void *stream) {
    PtrList pl;
    if (fill_ptrs(pl, outs, n_actors, "ssac_act_discrete: 1..8 actors")) return 1;
    if (n_rows <= 0 || n_actions <= 0 || n_actions > 64 || !act || (sample && (!rng || n_actors != 1)))
        return ssac_fail("ssac_act_discrete: bad arguments (<= 64 actions; a sample is one actor's and needs an rng stream)");
    RngArgs r{0, nullptr, 0};
    if (rng) r = RngArgs{rng->seed, rng->counter, rng->offset};
    SSAC_LAUNCH(discrete_act_kernel, dim3((n_rows + 63) / 64), dim3(64), 0, ST, pl, n_actors, ld_out, n_rows, n_actions, sample,
                r, act);
    return ssac_check_launch("ssac_act_discrete");
}
