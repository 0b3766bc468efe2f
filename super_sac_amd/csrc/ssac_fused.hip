// Fused 3-layer ensemble-MLP kernels for gfx950: one launch runs fc1 -> ReLU -> fc2 -> ReLU ->
// head for a 32-row tile of one net with the activations resident in LDS, and optionally
// continues with an epilogue that would otherwise be 3-6 more launches:
//
//   MODE_PLAIN   store y (and h1/h2 when the caller needs them for a backward pass)
//   MODE_SAMPLE  tanh-normal policy head: a = tanh(mu + sigma*eps) written into the [s'|a'] batch
//                buffer and log pi (distributions.py:9-15, 64-104)
//   MODE_CRITIC  critic loss gradient (learning.py:90-98), head backward, and the backward-data
//                GEMM of fc2 -- i.e. forward + loss + the whole dL/d(activations) chain.  Only
//                what the weight-gradient GEMMs need (h1, h2, dz2, dz1, dq) is written to HBM.
//   MODE_CRITIC_BWD  the second half of MODE_CRITIC alone: h1, h2, q tiles are read back from HBM (written
//                by an earlier MODE_PLAIN launch), then loss gradient, head backward, fc2 backward-data.
//                Lets the critic FORWARD run concurrently with the actor/target-critic/TD-target chain
//                (it does not depend on the TD target); bit-identical to MODE_CRITIC.
//   MODE_CRITIC_BWDU the TD-INDEPENDENT part of the backward pass.  The loss gradient of row b is a scalar c_b (it
//                holds the TD error) times a fixed selector of the head output, so dz2[b,:] = c_b * (W3[a_b,:] (.)
//                [h2 > 0]) and dz1[b,:] = c_b * ((dz2u[b,:] W2) (.) [h1 > 0]): this mode computes the UNSCALED dz2u /
//                dz1u from the saved forward alone; c_b enters later as a per-row scale in the weight-gradient
//                reductions (ssac_mlp_wgrad_all_scaled).  So the backward GEMM can run beside the target critics.
//
// A workgroup is 512 threads = 8 waves (2 per SIMD, so one wave's LDS/barrier phases hide under
// the other's MFMAs); wave w owns output columns [32w, 32w+32) of the 32 x H activation tile as one
// v_mfma_f32_32x32x2_f32 accumulator (exact fp32).  The A operand (x, h1, dz2) is read from LDS
// with odd row strides (bank = (row + k) % 32, conflict free); weight chunks of 32 k-steps are
// staged through a 33 KB LDS buffer with the next chunk's global loads in flight during the MFMAs.
// Constraints (checked by ssac_fused_supported): hidden % 32 == 0, hidden <= 256, out_dim <= 64,
// and the LDS carve (depends on in_dim) must fit 160 KB; other shapes use the per-layer kernels.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "ssac_internal.h"
#include "ssac_head_wgrad.h"
#include "ssac_philox.h"
#include "ssac_critic_logs.h"
#include "ssac_begin.h"
#include "ssac_xchg_body.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TM = 32;          // rows per workgroup
constexpr int NTHR = 512;       // 8 waves
constexpr int WS_LD = 36;             // K-contiguous staging row stride: 16-byte rows, conflict-free b128
constexpr int WS_FLOATS = 256 * WS_LD;  // weight staging buffer (>= 32*256 for the row-contiguous image)
constexpr int APAD = 4;               // activation rows are padded by 4 floats (16-byte aligned rows)
constexpr int MAX_OUT = 16;   // head outputs per MFMA pass (one 16-wide B tile)
constexpr int CO_SCRATCH = 8 * 16 * 16;   // co-resident carve: the K-split partial head tiles [8 waves][16 rows][16] are all that "Ws" holds
constexpr int HEAD_MAX = 64;  // widest head the fused kernels take (several passes)
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;
constexpr float LOG_2 = 0.69314718055994530942f;

enum { MODE_PLAIN = 0, MODE_SAMPLE = 1, MODE_CRITIC = 2, MODE_CRITIC_BWD = 3, MODE_CRITIC_BWDU = 4,
       MODE_CRITIC_U = 5 /* forward + the TD-independent (unscaled) backward, activations never leave LDS */,
       MODE_ACTOR_BWD = 6 /* policy-gradient backward of the actor: arg-min routing over the critics' Q, tanh-normal
                             backward, head backward and fc2 backward-data on the saved actor forward */ };

// policy-gradient routing of the online actor update (learning.py:392-408), evaluated per row by MODE_ACTOR_BWD
struct ActorBwdArgs {
    const float *qc; int n_critics;     // (n_critics x n_rows) critics' Q(s, a_theta)
    const float *dxu;                   // (n_critics x n_rows x A) UNSCALED dQ_j/da (MODE_CRITIC_U's DXU)
    const float *aout;                  // (n_rows x 2A) actor head output of the forward pass
    const float *eps, *logp, *log_alpha; int use_entropy;
    float lo, hi, inv_members;
    float *partials;                    // [row tiles] sum over the tile's rows of (Q' - alpha log pi)
};

// (struct Handoff, handoff_poll / handoff_publish: ssac_internal.h -- shared with the bf16 family)
constexpr int HANDOFF_MAX_WA = 9;   // W1's action columns per thread (H * A <= 512 * 9: 256 x 18)

struct FusedArgs {
    const float *params; int64_t net_stride; int in_dim, hidden, out_dim;
    int64_t off[6];
    const int32_t *ids;
    const float *X; int64_t ldx, sX; int n_rows;
    float *H1, *H2;              // (n_sel, n_rows, hidden) or null
    float *Y;                    // (n_sel, n_rows, out) or null
    // MODE_SAMPLE
    const float *eps; float lo, hi; float *act_dst; int64_t ld_act, act_col0; float *logp;
    RngArgs rng;  // eps == null: the noise comes from the engine's Philox stream
    // MODE_CRITIC
    const float *td, *weight, *act; int64_t ld_a; const ssac_popart *popart; int pop; float denom;
    float *DQ, *DZ2, *DZ1; float *partials;  // partials[(e*tiles + tile)*2 + {loss, err}]
    float *W3S;   // MODE_CRITIC_U with DZ2 == null: (n_nets x hidden) copy of the head rows W3 as this launch saw them --
                  // the weight-gradient launch that rebuilds dz2u from h2 must not read W3 itself: its own head
                  // workgroups update W3 while its fc2 tiles run
    int xcd;                     // workgroups take their tile in XCD-contiguous order (ssac_internal.h)
    const uint32_t *slot_now;       // this update's slot of gth.feed's input ring, filled in by ssac_step_run's replay
                                    // (ssac_record_slot_patch, ssac_internal.h); null: found through the feed block
    ssac_gather gth; int gth_role;  // 1: actor half (s' rows, begin duties), 3: actor half without the begin duties,
                                    // 2: critic half ([s|a] rows), 4: rows from X, net ids from the input slot; 0: X;
                                    // 5: hand-off consumer (s' rows like the actor half, nothing written, ids from the slot)
    long long *dbg;  // optional phase timestamps (s_memtime) of workgroup (0,0), thread 0
    long long *tl;   // optional per-workgroup (start, end) stamps of the chained launch (s_memrealtime; ssac_debug_timeline)
    ssac_td_spec tds;  // tds.q_t != null: the TD target is computed here instead of read from `td`
    float *DXU; int dx_col0, dx_cols;  // MODE_CRITIC_U: also the unscaled input gradient of columns [dx_col0, +dx_cols)
    int copy_x;                        // MODE_SAMPLE: also copy the input tile into act_dst[:, 0:in_dim]
    ActorBwdArgs ab;                   // MODE_ACTOR_BWD
    float *begin_logs; int begin_n; ssac_adam_ctl *begin_ctl;   // MODE_SAMPLE, tile 0: ssac_begin_update's duties folded in
                                                                  // (log block cleared, optimizer step advanced)
    Handoff ho;                        // MODE_SAMPLE: publish a'; MODE_PLAIN (16-row tiles): take the action columns from it
    unsigned *xarrive;                 // hand-off consumers of a SHARDED rank whose exchange runs in the launch's tail workgroup
                                       // (fused_chain_pc_kernel): Y leaves as agent-scope stores and the workgroup counts itself in
};

// TD target of row b (see ssac_td_spec; same operation order as td_target_kernel in ssac_elementwise.hip)
__device__ __forceinline__ float td_of_row(const FusedArgs &g, int b, int e) {
    if (!g.tds.q_t) return g.td[b];
    const float mq = ssac_td_min_q(g.tds, b, g.n_rows);
    const float alpha = g.tds.use_entropy ? expf(g.tds.log_alpha[0]) : 0.0f;
    const float bonus = g.tds.use_entropy ? alpha * g.tds.logp[b] : 0.0f;
    const float val = mq - bonus;
    const float t = g.tds.rew[b] + g.tds.gamma * (1.0f - g.tds.done[b]) * val;
    if (e == 0) g.tds.td_out[b] = t;
    return t;
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// ---------------------------------------------------------------------------------------------
// Weight staging.  Two LDS buffers; chunk c+1 is written while chunk c is being multiplied and the
// global loads of chunk c+2 are in flight, so there is ONE barrier per 32-deep K chunk and the
// memory instructions of a wave sit between its MFMAs instead of in a separate phase.  Per-thread
// source pointers are computed once and advanced by a uniform step (no 64-bit address arithmetic in
// the loop).  All global loads are 16 bytes wide; the source only has to be dword aligned (gfx950
// global loads have no wider alignment requirement), so rows of any length -- e.g. the 393-float rows of
// a Humanoid critic's fc1 -- stage as fast as aligned ones.
// ---------------------------------------------------------------------------------------------
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

struct NoStage {
    __device__ __forceinline__ void load(int, int) {}
    __device__ __forceinline__ void load_full(int) {}
};

// W is (Nout x K), K contiguous.  LDS image Ws[n*WS_LD + k] for n < 256, k < 32.
struct KcStage {
    const float *p[4];
    unsigned okmask;
    int kk;
    f4 v[4];
    __device__ __forceinline__ void init(const float *W, int ldw, int Nout, int tid) {
        okmask = 0;
        kk = (tid & 7) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = (tid >> 3) + 64 * q;
            const bool ok = n < Nout;
            p[q] = W + (ok ? (int64_t)n * ldw + kk : 0);
            okmask |= (ok ? 1u : 0u) << q;
        }
    }
    __device__ __forceinline__ void load(int k0, int K) {
#if defined(SSAC_LAB) && defined(SSAC_EXP_NO_WLOAD)
        if (k0 > 0) return;
#endif
        const int left = K - (k0 + kk);  // valid k values at and after this thread's first one
        if (left >= 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f4u *>(p[q] + k0);
        } else {  // ragged tail of the last chunk (K % 32 != 0): element-wise, zero filled
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[q][i] = i < left ? p[q][k0 + i] : 0.0f;
        }
    }
    // a chunk that is known to lie fully inside K (every chunk but the last): no ragged-tail branch
    __device__ __forceinline__ void load_full(int k0) {
#if defined(SSAC_LAB) && defined(SSAC_EXP_NO_WLOAD)
        return;   // (experiment build: the K loops' weight chunks are never fetched)
#endif
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f4u *>(p[q] + k0);
    }
    // (weight rows beyond the layer width are NOT zeroed: they only ever reach accumulator columns beyond the layer
    // width, which no epilogue reads -- and the K loop loses 16 selects per chunk; their pointers read row 0)
    __device__ __forceinline__ void store(float *__restrict__ Ws, int tid) const {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f4 *>(Ws + ((tid >> 3) + 64 * q) * WS_LD + kk) = v[q];
    }
};

// W is (K x Ncols), rows contiguous, Ncols % 4 == 0; stage rows [n0, n0+32), columns [0, 256) -> Wt[n*256 + c].
struct RcStage {
    const float *p[4];
    bool cok;
    int c0;
    f4 v[4];
    int ldw_;
    __device__ __forceinline__ void init(const float *W, int ldw, int Ncols, int tid) {
        ldw_ = ldw;
        c0 = (tid & 63) * 4;
        cok = c0 < Ncols;
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = W + (cok ? (int64_t)((tid >> 6) + 8 * q) * ldw + c0 : 0);
    }
    // rows n0 + local row; K = number of rows of W
    __device__ __forceinline__ void load(int n0, int K) {
#if defined(SSAC_LAB) && defined(SSAC_EXP_NO_WLOAD)
        if (n0 > 0) return;
#endif
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool ok = cok && (n0 + (int)(threadIdx.x >> 6) + 8 * q) < K;
            const f4 x = *reinterpret_cast<const f4u *>(p[q] + (ok ? (int64_t)n0 * ldw_ : 0));
            v[q] = ok ? x : (f4){0.f, 0.f, 0.f, 0.f};
        }
    }
    // rows [n0, n0 + 32) known to lie inside K.  (columns beyond the matrix are not zeroed -- they feed accumulator
    // columns nobody reads -- so the steady loop has no per-lane predicate; their pointers read column 0)
    __device__ __forceinline__ void load_full(int n0) {
#if defined(SSAC_LAB) && defined(SSAC_EXP_NO_WLOAD)
        return;
#endif
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f4u *>(p[q] + (int64_t)n0 * ldw_);
    }
    __device__ __forceinline__ void store(float *__restrict__ Wt, int tid) const {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f4 *>(Wt + ((tid >> 6) + 8 * q) * 256 + c0) = v[q];
    }
};

// ---------------------------------------------------------------------------------------------
// Row-tile policies.  A workgroup (8 waves) owns TMR rows x 256 columns of an activation tile;
// wave w owns columns [32w, 32w+32):
//   Tile<32>: one v_mfma_f32_32x32x2_f32 accumulator (16 regs); MFMA step t of a 32-deep K chunk takes
//             k = 16*half + t from lane half `half`  -> 16 consecutive floats per lane (4 x ds_read_b128)
//   Tile<16>: two v_mfma_f32_16x16x4_f32 accumulators (4 regs each, column sub-tiles of 16); step t takes
//             k = 8*group + t from lane group `group` = lane>>4 -> 8 consecutive floats (2 x ds_read_b128)
// (a permutation of the chunk's k values, identical for A and B, so the sum is unchanged up to fp32
// association).  Tile<16> doubles the workgroup count for the same batch -- the actor / target-critic
// launches use 32 / 64 CUs instead of 16 / 32 and each CU carries half the matrix work -- at the
// same MFMA rate (both shapes are 64 FLOP/clk/SIMD).
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TMR> struct Tile;

template <> struct Tile<32> {
    typedef f32x16 Acc;
    struct Frag { f4 a[4]; f4 b[4]; float bs[16]; };
    static __device__ __forceinline__ void zero(Acc &a) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = 0.0f;
    }
    template <bool NN>
    static __device__ __forceinline__ void read(Frag &f, const float *As, int lda, const float *buf, int c,
                                                int lane, int col0) {
        const int li = lane & 31, lh = lane >> 5;
        const f4 *ap = reinterpret_cast<const f4 *>(As + li * lda + c * 32 + lh * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) f.a[q] = ap[q];
        if (NN) {
            const float *bp = buf + (lh * 16) * 256 + col0 + li;
#pragma unroll
            for (int t = 0; t < 16; ++t) f.bs[t] = bp[t * 256];
        } else {
            const f4 *bp = reinterpret_cast<const f4 *>(buf + (col0 + li) * WS_LD + lh * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) f.b[q] = bp[q];
        }
    }
    template <bool NN>
    static __device__ __forceinline__ void mfma_half(Acc &acc, const Frag &f, int half) {
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8) {
            const int t = half * 8 + t8;
            // the WEIGHT fragment is the MFMA's A operand and the activation fragment its B operand: the
            // accumulator is then the transposed tile D[n][row], i.e. a lane holds 4 CONSECUTIVE output columns of
            // ONE row per register quad, and the epilogues store 16 bytes at a time (LDS and global) instead of 4
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(NN ? f.bs[t] : f.b[t >> 2][t & 3], f.a[t >> 2][t & 3],
                                                      acc, 0, 0, 0);
        }
    }
    // f(row, column within the wave's 32, value)
    template <class F>
    static __device__ __forceinline__ void foreach(const Acc &acc, int lane, F fn) {
        const int li = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int r = 0; r < 16; ++r) fn(li, (r & 3) + 8 * (r >> 2) + 4 * lh, acc[r]);
    }
    // f(row, first of 4 consecutive columns within the wave's 32, the 4 values)
    template <class F>
    static __device__ __forceinline__ void foreach4(const Acc &acc, int lane, F fn) {
        const int li = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            fn(li, 8 * q + 4 * lh, (f4){acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
    }
};

template <> struct Tile<16> {
    struct Acc { f32x4 v[2]; };
    struct Frag { f4 a[2]; f4 b[2][2]; float bs[2][8]; };
    static __device__ __forceinline__ void zero(Acc &a) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) a.v[u][i] = 0.0f;
    }
    template <bool NN>
    static __device__ __forceinline__ void read(Frag &f, const float *As, int lda, const float *buf, int c,
                                                int lane, int col0) {
        const int li = lane & 15, lg = lane >> 4;
        const f4 *ap = reinterpret_cast<const f4 *>(As + li * lda + c * 32 + lg * 8);
        f.a[0] = ap[0]; f.a[1] = ap[1];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (NN) {
                const float *bp = buf + (lg * 8) * 256 + col0 + 16 * u + li;
#pragma unroll
                for (int t = 0; t < 8; ++t) f.bs[u][t] = bp[t * 256];
            } else {
                const f4 *bp = reinterpret_cast<const f4 *>(buf + (col0 + 16 * u + li) * WS_LD + lg * 8);
                f.b[u][0] = bp[0]; f.b[u][1] = bp[1];
            }
        }
    }
    template <bool NN>
    static __device__ __forceinline__ void mfma_half(Acc &acc, const Frag &f, int half) {
#if defined(SSAC_LAB) && defined(SSAC_EXP_NO_MFMA)
        acc.v[0][0] += f.a[0][0] + (NN ? f.bs[0][half] : f.b[0][0][0]);   // (experiment build: the operands stay live, no matrix work)
        return;
#endif
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const int t = half * 4 + t4;
#pragma unroll
            for (int u = 0; u < 2; ++u)
                acc.v[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(NN ? f.bs[u][t] : f.b[u][t >> 2][t & 3],
                                                                f.a[t >> 2][t & 3], acc.v[u], 0, 0, 0);
        }
    }
    template <class F>
    static __device__ __forceinline__ void foreach(const Acc &acc, int lane, F fn) {
        const int li = lane & 15, lg = lane >> 4;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) fn(li, 16 * u + 4 * lg + r, acc.v[u][r]);
    }
    template <class F>
    static __device__ __forceinline__ void foreach4(const Acc &acc, int lane, F fn) {
        const int li = lane & 15, lg = lane >> 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) fn(li, 16 * u + 4 * lg, (f4){acc.v[u][0], acc.v[u][1], acc.v[u][2], acc.v[u][3]});
    }
};

// ---------------------------------------------------------------------------------------------
// The K loop: one barrier per 32-deep chunk, placed in the MIDDLE of the chunk's MFMAs:
//   [LDS stores of chunk c+1, global loads of chunk c+2, first half of chunk c's MFMAs]
//   barrier   (chunk c+1 visible; everybody finished reading chunk c-1's buffer long ago)
//   [fragment reads of chunk c+1 into the other register set, second half of chunk c's MFMAs]
// so fragment-read latency hides under MFMAs and the barrier skew under the MFMAs already in the pipe.
// NN = false: acc += A[TMR x K] * W^T, W = (Nw x K) K-contiguous       (forward)
// NN = true : acc += A[TMR x K] * W,   W = (K x Nw) rows contiguous    (backward-data)
// Every wave runs the loop (waves whose columns lie beyond the layer width multiply staged zeros).
// ---------------------------------------------------------------------------------------------
template <int TMR, bool NN, typename Stage, typename Next>
__device__ __forceinline__ void pipe_step(typename Tile<TMR>::Acc &acc, Stage &st, int c, int nch, int K,
                                          const float *As, int lda, float *nxt,
                                          const typename Tile<TMR>::Frag &cur, typename Tile<TMR>::Frag &nf,
                                          int tid, int col0, Next &nx, int Knext) {
    const bool has1 = (c + 1) < nch, has2 = (c + 2) < nch;
    if (has1) st.store(nxt, tid);
    if (has2) st.load((c + 2) * 32, K);
    if (!has1) nx.load(0, Knext);  // last chunk: the NEXT phase's first weight chunk goes in flight
    Tile<TMR>::template mfma_half<NN>(acc, cur, 0);
    lds_barrier();  // (LDS hand-off only: the loads of chunk c+2 stay in flight)
    if (has1) Tile<TMR>::template read<NN>(nf, As, lda, nxt, c + 1, tid & 63, col0);
    Tile<TMR>::template mfma_half<NN>(acc, cur, 1);
}

// Steady-state step (chunks c+1 and c+2 exist and are full): straight-line code, so the staging stores and the next
// global loads can be interleaved BETWEEN the MFMAs of the first half and the fragment reads between those of the
// second half (a wave is issue-stalled behind its own dependent MFMA chain most of the time: those slots are free).
template <int TMR, bool NN, typename Stage>
__device__ __forceinline__ void pipe_steady(typename Tile<TMR>::Acc &acc, Stage &st, int c, const float *As, int lda,
                                            float *nxt, const typename Tile<TMR>::Frag &cur,
                                            typename Tile<TMR>::Frag &nf, int tid, int col0) {
    st.store(nxt, tid);
    st.load_full((c + 2) * 32);
    Tile<TMR>::template mfma_half<NN>(acc, cur, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // MFMA, LDS write, MFMA, global load
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    lds_barrier();  // (LDS hand-off only: the loads of chunk c+2 stay in flight)
    Tile<TMR>::template read<NN>(nf, As, lda, nxt, c + 1, tid & 63, col0);
    Tile<TMR>::template mfma_half<NN>(acc, cur, 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // MFMA, LDS read, MFMA, LDS read, ...
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NN ? 3 : 1, 0);
    }
}

// Contract: the caller has staged chunk 0 into B0 (stage_first below), and a barrier since then has made
// it and the A operand visible.  While the LAST chunk is being multiplied, nx.load(0, Knext) puts the next
// phase's first weight chunk in flight, so no phase starts by waiting a full HBM/L2 round trip; the caller
// stores it (stage_first) after this function's closing barrier, next to its epilogue.
template <typename Stage>
__device__ __forceinline__ void stage_first(Stage &st, float *B0, int K, int tid) {
    st.store(B0, tid);          // chunk 0 (already loaded) -> LDS
    if (K > 32) st.load(32, K);  // chunk 1 -> registers
}

template <int TMR, bool NN, bool DBUF, typename Stage, typename Next>
__device__ __forceinline__ void gemm_tile(typename Tile<TMR>::Acc &acc, Stage &st, const float *__restrict__ As,
                                          int lda, int K, float *B0, float *B1, int tid, int col0, Next &nx,
                                          int Knext) {
    const int nch = (K + 31) >> 5;
    if (!DBUF) {
        // single staging buffer, two barriers per chunk: half the LDS, so TWO workgroups fit a CU and
        // each one's barrier / staging phases hide under the other's MFMAs (4 waves per SIMD)
        typename Tile<TMR>::Frag f;
        for (int c = 0; c < nch; ++c) {
            if (c > 0) {
                lds_barrier();  // everybody is done with the previous chunk
                st.store(B0, tid);
                if (c + 1 < nch) st.load((c + 1) * 32, K);
                lds_barrier();
            }
            if (c + 1 == nch) nx.load(0, Knext);
            Tile<TMR>::template read<NN>(f, As, lda, B0, c, tid & 63, col0);
            Tile<TMR>::template mfma_half<NN>(acc, f, 0);
            Tile<TMR>::template mfma_half<NN>(acc, f, 1);
        }
        lds_barrier();
        return;
    }
    typename Tile<TMR>::Frag f0, f1;
    Tile<TMR>::template read<NN>(f0, As, lda, B0, 0, tid & 63, col0);
    int c = 0;
    const int full = (K & 31) ? nch - 1 : nch;  // chunks [0, full) hold 32 k values each
    for (; c + 3 < full; c += 2) {  // chunks c+2 and c+3 exist and are full
        pipe_steady<TMR, NN>(acc, st, c, As, lda, B1, f0, f1, tid, col0);
        pipe_steady<TMR, NN>(acc, st, c + 1, As, lda, B0, f1, f0, tid, col0);
    }
    for (; c < nch; c += 2) {
        pipe_step<TMR, NN>(acc, st, c, nch, K, As, lda, B1, f0, f1, tid, col0, nx, Knext);
        if (c + 1 < nch) pipe_step<TMR, NN>(acc, st, c + 1, nch, K, As, lda, B0, f1, f0, tid, col0, nx, Knext);
    }
    lds_barrier();  // all fragment reads done before the caller reuses As / the staging buffers
}

// ---------------------------------------------------------------------------------------------
// The K loop of the CO-RESIDENT carve (16-row tiles, two workgroups per CU; round 5): the weight fragments go STRAIGHT
// from L2 into the MFMA operand registers -- no LDS image of the weights, no staging stores, NO barrier inside the loop.
// Measured first (profiles/r5_chain_coresident.md): with LDS-staged weights two co-resident 16-row tiles cost MORE than one
// 32-row tile (35 vs 30 us per launch) -- per 32 rows they stage every weight chunk twice, read 1.5 x the fragments and
// hit twice the barriers, and without the MFMAs AND without the weight loads the pair still took 49 k clocks: the LDS /
// barrier skeleton, not L2 streaming and not the matrix pipe, was what two tiles on a CU competed for.  Here a wave's
// only LDS traffic in the loop is its activation fragment (2 x ds_read_b128 per chunk); its 32 columns' weights are 4
// (forward: 16 bytes, K-contiguous rows) or 16 (backward-data: dwords down a column block) global loads per chunk,
// requested one chunk ahead, the next PHASE's first chunk during the last one -- with 4 waves per SIMD from two
// independent workgroups the L2 round trip hides under the other waves' MFMAs.
// Same fragments, same MFMA order as Tile<16>: bit-identical to the LDS-staged 16-row tiles.
// ---------------------------------------------------------------------------------------------
template <bool NN>
struct DirectW {
    // NN = false: W is (Nw x K), K contiguous: lane (li, lg) reads W[col0 + 16u + li][32c + 8lg .. + 8) -- two 16-byte global
    //   loads per column sub-tile u through a per-lane pointer (2 address pairs in all).
    // NN = true : W is (K x Nw), rows contiguous: lane reads W[32c + 8lg + t][col0 + 16u + li], t = 0..7 -- 16 dword loads per
    //   chunk.  As flat loads with 64-bit lane addresses they held 16 address pairs (32 VGPRs: spills at the 128-VGPR budget
    //   of two workgroups per CU), so they are BUFFER loads: one descriptor in SGPRs (wave-uniform base), ONE 32-bit byte
    //   offset per lane and sub-tile, the row as a scalar offset (wgrad_small_pair_kernel's idiom, ssac_gemm.hip).
    //   (This toolchain lowers __builtin_amdgcn_raw_buffer_load_b128 / _b64 to a ONE-dword load whose value is replicated --
    //   checked in the ISA -- so the forward form cannot use the descriptor path for its 16-byte loads.)
    // Columns beyond the layer width read column / row 0: they only reach accumulator columns nobody reads.
    const float *p[2];
    __amdgpu_buffer_rsrc_t r;
    int off[2];          // NN: byte offset of this lane's first element of chunk 0, column sub-tiles u = 0, 1
    uint32_t ld4;        // NN: row stride in bytes (uniform)
    __device__ __forceinline__ void init(const float *W, int ldw, int Nw, int col0, int lane) {
        const int li = lane & 15, lg = lane >> 4;
        if (NN) {
            // (readfirstlane returns a SIGNED int: each half goes through uint32_t, or a base whose bit 31 is set comes back
            // sign-extended -- 0xffffffff in the upper word, a memory access fault on every other allocation layout)
            const uint64_t ub = (uint64_t)(uintptr_t)W;
            const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ub);
            const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ub >> 32));
            void *base = (void *)(uintptr_t)(((uint64_t)bhi << 32) | (uint64_t)blo);
            r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffffc, 0x00020000);
            ld4 = 4u * (uint32_t)__builtin_amdgcn_readfirstlane(ldw);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n = col0 + 16 * u + li;
            const uint32_t nn = n < Nw ? (uint32_t)n : 0u;
            if (NN) off[u] = (int)((uint32_t)(lg * 8) * ld4 + 4u * nn);
            else p[u] = W + (int64_t)nn * ldw + lg * 8;
        }
    }
    // chunk c (32 k values).  A ragged last chunk of the forward form (K % 32 != 0: fc1) reads up to 31 floats past the
    // row's end -- the next rows / the bias of the same parameter arena, finite values that meet the zero padding of the
    // activation tile in the MFMA.
    __device__ __forceinline__ void load(float (&v)[2][8], int c) const {
#if defined(SSAC_LAB) && defined(SSAC_EXP_NO_WLOAD)
        if (c > 0) return;   // (experiment build: only a phase's first chunk is ever fetched)
#endif
#if defined(SSAC_LAB) && defined(SSAC_EXP_NO_NN)
        if (NN) return;
#endif
#if defined(SSAC_LAB) && defined(SSAC_EXP_NO_FWD)
        if (!NN) return;
#endif
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (NN) {
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    v[u][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off[u], (int)((uint32_t)(c * 32 + t) * ld4), 0));
            } else {
                const f4 x0 = *reinterpret_cast<const f4u *>(p[u] + c * 32), x1 = *reinterpret_cast<const f4u *>(p[u] + c * 32 + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[u][i] = x0[i]; v[u][4 + i] = x1[i]; }
            }
        }
    }
};
struct NoDirect {
    __device__ __forceinline__ void load(float (&)[2][8], int) const {}
};

__device__ __forceinline__ void direct_chunk16(Tile<16>::Acc &acc, const float (&b)[2][8], const float *As, int lda, int c, int lane) {
    const int li = lane & 15, lg = lane >> 4;
    const f4 *ap = reinterpret_cast<const f4 *>(As + li * lda + c * 32 + lg * 8);
    const f4 a0 = ap[0], a1 = ap[1];
#if defined(SSAC_LAB) && defined(SSAC_EXP_NO_MFMA)
    acc.v[0][0] += a0[0] + b[0][0] + b[1][7] + a1[3];
    return;
#endif
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            acc.v[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u][t], t < 4 ? a0[t & 3] : a1[t & 3], acc.v[u], 0, 0, 0);
}

// b0: chunk 0 of W, already requested by the caller.  nb: receives chunk 0 of the NEXT phase's weights (nx), requested
// while the last chunk multiplies.  Ends with a barrier: every wave is done reading As.
template <bool NN, typename Next>
__device__ __forceinline__ void gemm_direct16(Tile<16>::Acc &acc, const DirectW<NN> &w, float (&b0)[2][8], const float *As,
                                              int lda, int K, int lane, const Next &nx, float (&nb)[2][8]) {
    const int nch = (K + 31) >> 5;
    // TWO chunks in flight ahead of the one that multiplies (three register sets): with one, a wave's 4 KB per chunk x 16
    // waves per CU is 64 KB in flight against a ~2 k-clock loaded L2 round trip, i.e. ~32 B/clk per CU -- the K loop ran at
    // the load rate, not at the MFMA rate (fc2 of a 16-row tile 24 k clocks against a floor of 8.2 k / 16.4 k shared).
    float b1[2][8], b2[2][8];
    if (nch > 1) w.load(b1, 1);
    int c = 0;
    for (; c + 3 < nch; c += 3) {   // steady state (chunks up to c + 3 + 1 exist or are guarded below)
        w.load(b2, c + 2);
        direct_chunk16(acc, b0, As, lda, c, lane);
        w.load(b0, c + 3);
        direct_chunk16(acc, b1, As, lda, c + 1, lane);
        if (c + 4 < nch) w.load(b1, c + 4);
        direct_chunk16(acc, b2, As, lda, c + 2, lane);
    }
    // tail: 1, 2 or 3 chunks left; b0 holds chunk c, b1 chunk c + 1 (if it exists) -- the next phase's first chunk goes in
    // flight as early as a register set is free
    const int left = nch - c;
    if (left == 3) {
        w.load(b2, c + 2);
        direct_chunk16(acc, b0, As, lda, c, lane);
        nx.load(nb, 0);
        direct_chunk16(acc, b1, As, lda, c + 1, lane);
        direct_chunk16(acc, b2, As, lda, c + 2, lane);
    } else if (left == 2) {
        nx.load(nb, 0);
        direct_chunk16(acc, b0, As, lda, c, lane);
        direct_chunk16(acc, b1, As, lda, c + 1, lane);
    } else {
        nx.load(nb, 0);
        direct_chunk16(acc, b0, As, lda, c, lane);
    }
    lds_barrier();
}

// ---------------------------------------------------------------------------------------------
// Backward-data K loop WITHOUT an LDS image of the weights, every tile size (round 5).  dz1 = dz2 W2 reads W2 (K x N, rows
// contiguous) down column blocks: the MFMA's weight fragment of step t is W2[k(t)][col0 + lane's column] -- for the 64
// lanes of a wave two (32-row tiles) or four (16-row tiles) fully used 128- / 64-byte runs, i.e. as a GLOBAL access the
// fragment load is perfectly coalesced, while as an LDS access it was 16 scalar ds_read_b32 per chunk per lane behind 4
// staging stores and a workgroup barrier.  So the fragments are buffer loads (one descriptor in SGPRs, one byte offset per
// lane, the row as a scalar offset: no address register per load), two chunks in flight ahead of the multiplying one, and the
// loop has no barrier and no LDS traffic but the activation fragment.  Measured inside the co-resident experiment
// (profiles/r5_chain_coresident.md: 10.9 k clocks per 16-row K loop against 19.2 - 24.4 k for the forward form, whose 16-byte
// row-wise loads touch 16 lines per instruction) and then taken for the product's tiles.  Same fragments, same MFMA
// order as the staged loop: bit-identical.  (The forward loops keep the LDS image: a lane needs 8 - 16 consecutive k of its
// OWN weight row, which only a staged, re-laid-out chunk serves with wide accesses.)
// ---------------------------------------------------------------------------------------------
template <int TMR>
struct DirectNN {
    static constexpr int NOFF = TMR == 32 ? 1 : 2;
    __amdgpu_buffer_rsrc_t r;
    int off[NOFF];       // byte offset of this lane's element of row (its k group's first), per column sub-tile
    uint32_t ld4;        // row stride in bytes (uniform)
    float b0[16];        // chunk 0, requested through the `Next` interface of the K loop in front (load(0, K))
    __device__ __forceinline__ void init(const float *W, int ldw, int Nw, int col0, int lane) {
        // (readfirstlane returns a signed int: each half through uint32_t, or a base with bit 31 set is sign-extended)
        const uint64_t ub = (uint64_t)(uintptr_t)W;
        const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ub);
        const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ub >> 32));
        r = __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)(((uint64_t)bhi << 32) | (uint64_t)blo), 0, 0x7ffffffc, 0x00020000);
        ld4 = 4u * (uint32_t)__builtin_amdgcn_readfirstlane(ldw);
        if (TMR == 32) {
            const int li = lane & 31, lh = lane >> 5, n = col0 + li;
            off[0] = (int)((uint32_t)(lh * 16) * ld4 + 4u * (uint32_t)(n < Nw ? n : 0));
        } else {
            const int li = lane & 15, lg = lane >> 4;
#pragma unroll
            for (int u = 0; u < NOFF; ++u) {
                const int n = col0 + 16 * u + li;
                off[u] = (int)((uint32_t)(lg * 8) * ld4 + 4u * (uint32_t)(n < Nw ? n : 0));
            }
        }
    }
    // chunk c: Tile<32>: v[t] = W[32c + 16 lh + t][column], t < 16;  Tile<16>: v[8u + t] = W[32c + 8 lg + t][column of u], t < 8
    __device__ __forceinline__ void load_chunk(float (&v)[16], int c) const {
#if defined(SSAC_LAB) && defined(SSAC_EXP_NO_WLOAD)
        if (c > 0) return;
#endif
        if (TMR == 32) {
#pragma unroll
            for (int t = 0; t < 16; ++t)
                v[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off[0], (int)((uint32_t)(c * 32 + t) * ld4), 0));
        } else {
#pragma unroll
            for (int u = 0; u < NOFF; ++u)
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    v[8 * u + t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off[u], (int)((uint32_t)(c * 32 + t) * ld4), 0));
        }
    }
    __device__ __forceinline__ void load(int k0, int) { load_chunk(b0, k0 >> 5); }   // (`Next` interface of gemm_tile)
};

template <int TMR>
__device__ __forceinline__ void direct_nn_chunk(typename Tile<TMR>::Acc &acc, const float (&b)[16], const float *As, int lda,
                                                int c, int lane) {
    if constexpr (TMR == 32) {
        const int li = lane & 31, lh = lane >> 5;
        const f4 *ap = reinterpret_cast<const f4 *>(As + li * lda + c * 32 + lh * 16);
        const f4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f4 &aq = t < 4 ? a0 : (t < 8 ? a1 : (t < 12 ? a2 : a3));
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t], aq[t & 3], acc, 0, 0, 0);
        }
    } else {
        const int li = lane & 15, lg = lane >> 4;
        const f4 *ap = reinterpret_cast<const f4 *>(As + li * lda + c * 32 + lg * 8);
        const f4 a0 = ap[0], a1 = ap[1];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                acc.v[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[8 * u + t], t < 4 ? a0[t & 3] : a1[t & 3], acc.v[u], 0, 0, 0);
    }
}

// w.b0 holds chunk 0 (requested by the caller a phase ahead).  Ends with a barrier: every wave is done reading As.
template <int TMR>
__device__ __forceinline__ void gemm_direct_nn(typename Tile<TMR>::Acc &acc, DirectNN<TMR> &w, const float *As, int lda, int K,
                                               int lane) {
    const int nch = (K + 31) >> 5;
    float b1[16], b2[16];
    if (nch > 1) w.load_chunk(b1, 1);
    int c = 0;
    for (; c + 3 < nch; c += 3) {
        w.load_chunk(b2, c + 2);
        direct_nn_chunk<TMR>(acc, w.b0, As, lda, c, lane);
        w.load_chunk(w.b0, c + 3);
        direct_nn_chunk<TMR>(acc, b1, As, lda, c + 1, lane);
        if (c + 4 < nch) w.load_chunk(b1, c + 4);
        direct_nn_chunk<TMR>(acc, b2, As, lda, c + 2, lane);
    }
    const int left = nch - c;
    if (left == 3) {
        w.load_chunk(b2, c + 2);
        direct_nn_chunk<TMR>(acc, w.b0, As, lda, c, lane);
        direct_nn_chunk<TMR>(acc, b1, As, lda, c + 1, lane);
        direct_nn_chunk<TMR>(acc, b2, As, lda, c + 2, lane);
    } else if (left == 2) {
        direct_nn_chunk<TMR>(acc, w.b0, As, lda, c, lane);
        direct_nn_chunk<TMR>(acc, b1, As, lda, c + 1, lane);
    } else {
        direct_nn_chunk<TMR>(acc, w.b0, As, lda, c, lane);
    }
    lds_barrier();
}

// W3 (OUT x H, rows contiguous) -> LDS rows of stride ldw3.  Four independent loads are issued before the
// first LDS store, so the copy costs one global round trip instead of one per head row.
__device__ __forceinline__ void stage_head_weights(float *w3s, const float *__restrict__ W3, int OUT, int H,
                                                   int ldw3, int tid, int start = 0) {
    const int n3 = OUT * H;
    for (int i0 = start + tid; i0 < n3; i0 += 4 * NTHR) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NTHR;
            v[u] = W3[i < n3 ? i : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NTHR;
            if (i < n3) {
                const int o = i / H;
                w3s[o * ldw3 + (i - o * H)] = v[u];
            }
        }
    }
}

#ifdef SSAC_LAB
#define STAMP(i) do { if (g.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
#ifdef SSAC_LAB
#define BSTAMP(i) do { if (g.dbg && dbg_off >= 0 && bx == 0 && e == 0 && threadIdx.x == 0) g.dbg[dbg_off + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BSTAMP(i) do { } while (0)
#endif

// DBUF = false: 2 workgroups per CU (needs <= 128 VGPRs and <= 80 KB of LDS each)
// (bx, e, grid_x) = tile index, net slot and number of row tiles of the launch this workgroup works for -- blockIdx /
// gridDim of a plain launch, or the position inside one half of a merged launch (fused_dual_kernel).
// dbg_off: slot offset of this role's phase stamps in the debug buffer (-1: none); the kernel parameters themselves are
// never modified (a by-value parameter that is written to is copied to scratch memory, all ~500 bytes of it)
// HO: this instantiation may be the CONSUMER of a hand-off (fused_chain_pc_kernel's target-critic workgroups only: the
// register-resident weight fragments of the column-split form must not weigh on the other kernels' register budgets)
// W3LATE (MODE_SAMPLE, double-buffered): the head's weight rows are NOT resident in LDS from the start -- a wide head over a
// wide input (Humanoid's actor: 376 -> 256 -> 256 -> 34, 35 KB of W3) leaves no room for the second staging buffer, and with
// a single buffer its 20 K chunks cost 2.4 k clocks each instead of 1.4 k.  They are requested into registers before fc2 and
// parked in staging buffer 1 behind fc2's K loop, where the head reads them.
// CO ("co-resident", round 5): the LDS carve of a workgroup that shares its CU with a second one (<= 80 KB, <= 128 VGPRs:
// fused_chain_co_kernel) -- 16-row tiles; h2 / dz2u take h1's place in LDS (fc2's epilogue runs behind the K loop's closing
// barrier, when nobody reads h1 any more), and the ReLU mask of h1 that the backward-data epilogue needs stays in a
// register: a lane's accumulator elements are the same (row, column) set in fc1 and in the backward-data GEMM.
template <int MODE, int TMR, bool DBUF, bool HO = false, bool W3LATE = false, bool CO = false>
__device__ __forceinline__ void fused_mlp_body(const FusedArgs &g, float *smem, const int bx, const int e,
                                               const int grid_x, const int dbg_off = 0, const int split = 0) {
    static_assert(!W3LATE || (MODE == MODE_SAMPLE && DBUF), "the late head image is the actor pass's, behind a double-buffered fc2");
    static_assert(!CO || (TMR == 16 && !W3LATE && (MODE == MODE_PLAIN || MODE == MODE_SAMPLE || MODE == MODE_CRITIC_U)),
                  "the co-resident carve: 16-row tiles of the chained launch's three roles");
    typedef Tile<TMR> T;
    const int H = g.hidden, IN = g.in_dim, OUT = g.out_dim;
    const int ldo = (OUT + 15) & ~15;  // row stride of the per-row head outputs / output gradients in LDS
    const int KP = (IN + 31) & ~31;
    const int ldx_s = KP + APAD, ldh = H + APAD;
    float *xs = smem;                       // [TMR][KP+4]
    float *h1s = xs + TMR * ldx_s;          // [TMR][H+4]
    float *h2s = CO ? h1s : h1s + TMR * ldh;   // [TMR][H+4]  (CO: in h1's place)
    float *Ws = h2s + TMR * ldh;            // staging buffer 0
    float *Ws1 = Ws + WS_FLOATS;            // staging buffer 1 (DBUF only)
    float *ys = Ws + (CO ? CO_SCRATCH : (DBUF ? 2 : 1) * WS_FLOATS);  // [TMR][ldo], ldo = out_dim rounded up to 16  (CO: no weight image, Ws is scratch)
    float *dqs = ys + TMR * ldo;            // [TMR][ldo]
    float *rowred = dqs + TMR * ldo;        // [64]
    // small operands fetched at kernel start, so later phases never begin with a global round trip:
    float *b1s = rowred + 64;               // [H]
    float *b2s = b1s + H;                   // [H]
    float *b3s = b2s + H;                   // [HEAD_MAX]
    float *w3s = W3LATE ? Ws1 : b3s + HEAD_MAX;   // [OUT][H+4]  (W3LATE: staging buffer 1, filled behind fc2)
    float *rowin = W3LATE ? b3s + HEAD_MAX : w3s + OUT * (H + APAD);  // [3][TMR]: td, weight, action index of this tile's rows
    float *wa_co = rowin + 3 * TMR;         // CO consumer: W1's action columns [H][A] (the other carves park them in staging buffer 1)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = bx * TMR;
    // replay gather folded into this launch (ssac_gather): where this tile's rows come from
    const int64_t *gidx = nullptr;
    const uint32_t *gslot = nullptr;
    const int32_t *idsp = g.ids;
    if (g.gth_role) {
        gidx = g.gth.idx;
        if (g.gth.feed) {
            if (g.slot_now) {
                gslot = g.slot_now;   // by value: no dependent load in front of the index vector
            } else {
                const ssac_feed f = *g.gth.feed;
                gslot = feed_slot(f);
            }
            gidx = reinterpret_cast<const int64_t *>(gslot);
            if (g.gth_role == 1 && bx == 0) {  // start-of-update duties (ssac_begin_update)
                const ssac_feed f = *g.gth.feed;
                feed_pull(f);
                if (tid < g.gth.n_logs) g.gth.logs[tid] = 0.0f;
                if (tid == 0 && g.gth.ctl) adam_refresh(g.gth.ctl, g.gth.ctl->step + 1);
            }
            // role 4: the subset ids of a target-critic pass that runs in the launch that mirrors the slot
            if ((g.gth_role == 4 || g.gth_role == 5) && g.gth.ids_word >= 0) idsp = reinterpret_cast<const int32_t *>(gslot + g.gth.ids_word);
        }
        if (g.gth_role == 4) gidx = nullptr;  // (its rows are read from X)
    }
    // (sharded rank, exchange inside this launch: a consumer's Q must be visible to the exchange's workgroup on another CU / XCD
    //  -- agent-scope stores -- and the workgroup counts itself in once its stores have left the CU)
    auto y_store = [&](float *p_, float v_) {
        if (HO && MODE == MODE_PLAIN && g.xarrive) __hip_atomic_store(p_, v_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *p_ = v_;
    };
    auto consumer_arrive = [&]() {
        if (HO && MODE == MODE_PLAIN && g.xarrive) {
            // every wave drains ITS OWN stores before the barrier: __syncthreads() is s_waitcnt lgkmcnt(0) + s_barrier on this
            // toolchain (no vmcnt), and thread 0's release below waits on wave 0's stores only -- the column-split consumers'
            // other seven waves store Qt partials too
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(g.xarrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    const int net = idsp ? idsp[e] : e;
    if (net < 0) {
        // slot without a net (a REDQ subset member another rank owns): its outputs are +inf, the neutral
        // element of the min that follows, so a sharded launch sequence is the same for every subset draw
        if (MODE == MODE_PLAIN && g.Y) {
            const int nsp = (HO && TMR == 16 && g.ho.pub && g.ho.nsplit > 1) ? g.ho.nsplit : 1;   // (split consumers: +inf in part 0, 0 elsewhere)
            for (int i = threadIdx.x; i < TMR * OUT; i += NTHR) {
                const int r = i / OUT, o = i - r * OUT;
                if ((m0 + r) < g.n_rows)
                    y_store(g.Y + (((int64_t)e * nsp + split) * g.n_rows + m0 + r) * OUT + o, split == 0 ? __builtin_inff() : 0.0f);
            }
        }
        consumer_arrive();
        return;
    }
    const float *P = g.params + (int64_t)net * g.net_stride;
    const float *X = g.X + (int64_t)e * g.sX;
    const int col0 = wave * 32;
    if (MODE == MODE_SAMPLE && bx == 0 && g.begin_ctl) {   // (the chained actor update: no begin launch in front)
        if (tid < g.begin_n) g.begin_logs[tid] = 0.0f;
        if (tid == 0) adam_refresh(g.begin_ctl, g.begin_ctl->step + 1);
    }

    BSTAMP(0);
    const int ldw3 = H + APAD;
    KcStage st1, st2;
    RcStage st3;
    DirectNN<TMR> dnn;   // backward-data without an LDS image of W2 (gemm_direct_nn); the staged loop (st3) stays for A/B builds
    // Which loop -- measured, A/B builds in one box (profiles/r5_kernel_stats.md, "backward-data loop"): 16-row tiles take the
    // DIRECT loop (Humanoid N 16, whose tiles are 16 rows because 32 do not fit the LDS: 135.8 -> 131.6 us per update; a
    // 2-of-16 shard 0 .. -1.6 us), 32-row tiles keep the STAGED loop (headline 53.0 vs 53.3 us, N 16 78.0 vs 78.3: a 32-row
    // tile's 16 dword loads per lane and chunk against 4 sixteen-byte loads per thread for the image).  Compile-time per tile
    // size: with BOTH loops behind a run-time flag in one kernel the headline lost the difference again (code size /
    // register allocation of the fc2 loop in front).
#if defined(SSAC_LAB) && defined(SSAC_EXP_STAGED_BWD)
    constexpr bool DIRECT_BWD = false;
#elif defined(SSAC_LAB) && defined(SSAC_EXP_DIRECT_BWD)
    constexpr bool DIRECT_BWD = !CO;
#else
    constexpr bool DIRECT_BWD = !CO && TMR == 16;   // (the co-resident carve has its own direct loops)
#endif
    NoStage none;
    DirectW<false> d1, d2;   // CO: weight fragments straight from memory (gemm_direct16)
    DirectW<true> d3;
    NoDirect nodirect;
    float bq1[2][8], bq2[2][8], bq3[2][8];   // CO: the first weight chunk of fc1 / fc2 / backward-data, requested a phase ahead
    constexpr bool UNSCALED = MODE == MODE_CRITIC_BWDU || MODE == MODE_CRITIC_U;
    constexpr bool ACTOR = MODE == MODE_ACTOR_BWD;
    constexpr bool BWD_ONLY = MODE == MODE_CRITIC_BWD || MODE == MODE_CRITIC_BWDU || ACTOR;
    constexpr bool FWD_BWD = MODE == MODE_CRITIC || MODE == MODE_CRITIC_U;  // forward, then backward in the same workgroup
    constexpr bool IS_CRITIC = FWD_BWD || BWD_ONLY;
    typename T::Acc acc;
    unsigned m1 = 0;   // CO: [h1 > 0] of this lane's accumulator elements, in foreach4's visiting order
    float *hpart = Ws;  // K-split partial head tiles [8][TMR][16] (staging buffer 0 is free after fc2)
    if (BWD_ONLY) {
        // ---- h1, h2, q tiles of an earlier forward launch -> LDS; W2's first chunk in flight meanwhile
        if (DIRECT_BWD) { dnn.init(P + g.off[2], H, H, col0, lane); dnn.load(0, H); }
        else { st3.init(P + g.off[2], H, H, tid); st3.load(0, H); }
        // all loads of the two activation tiles before anything is stored: one round trip for the prologue
        const int c = (tid & 63) * 4;  // one wave per row pass, 16 bytes per lane
        constexpr int RP = TMR / (NTHR / 64);
        f4 a1[RP], a2[RP];
        if (c < H) {
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const int r = (tid >> 6) + j * (NTHR / 64);
                const bool ok = (m0 + r) < g.n_rows;
                const int64_t src = ((int64_t)e * g.n_rows + (ok ? m0 + r : 0)) * H + c;
                a1[j] = *reinterpret_cast<const f4u *>(g.H1 + src);
                a2[j] = *reinterpret_cast<const f4u *>(g.H2 + src);
            }
        }
        stage_head_weights(w3s, P + g.off[4], OUT, H, ldw3, tid);
        if (!ACTOR && tid < TMR) {
            const int b = m0 + tid;
            const bool ok = b < g.n_rows;
            if (!UNSCALED) {
                rowin[tid] = ok ? td_of_row(g, b, e) : 0.0f;
                rowin[TMR + tid] = (ok && g.weight) ? g.weight[b] : 1.0f;
            }
            rowin[2 * TMR + tid] = (ok && OUT > 1) ? g.act[b * g.ld_a] : 0.0f;
        }
        if (c < H) {
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const int r = (tid >> 6) + j * (NTHR / 64);
                const bool ok = (m0 + r) < g.n_rows;
                const f4 z = {0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f4 *>(h1s + r * ldh + c) = ok ? a1[j] : z;
                *reinterpret_cast<f4 *>(h2s + r * ldh + c) = ok ? a2[j] : z;
            }
        }
        if (!UNSCALED && !ACTOR)
            for (int i = tid; i < TMR * OUT; i += NTHR) {
                const int r = i / OUT, o = i - r * OUT;
                ys[r * ldo + o] = (m0 + r) < g.n_rows ? g.Y[((int64_t)e * g.n_rows + m0 + r) * OUT + o] : 0.0f;
            }
    } else {
        // ---- first weight chunk of fc1 in flight before anything else
        if constexpr (CO) {
            d1.init(P + g.off[0], IN, H, col0, lane);
            d2.init(P + g.off[2], H, H, col0, lane);
            d1.load(bq1, 0);
        } else {
            st1.init(P + g.off[0], IN, H, tid);
            st1.load(0, IN);
            st2.init(P + g.off[2], H, H, tid);
        }
        // ---- Every global load of the prologue is issued before the first LDS store, so the prologue costs one
        //      round trip (two with the replay gather: index, then row) instead of one per operand: biases, head
        //      weights (first 4 per thread), the x tile's first 32 columns.  Wider inputs / heads loop afterwards.
        constexpr int XR = TMR / 16;  // x-tile rows per thread: half-wave per row, 16 rows a pass
        const int xk = tid & 31, xr0 = tid >> 5;
        const int Sg = gidx ? (int)g.gth.s_elems : 0;
        const bool actor_half = (g.gth_role & 1) != 0;  // roles 1, 3 (3: no start-of-update duties) and 5 (below)
        // CONSUMER of a hand-off (MODE_PLAIN, role 5 / no gather): the input's state columns are fetched here -- s' rows
        // from the replay arrays or from X -- the action columns [S, S + A) stay ZERO in the x tile: fc1 runs on the state
        // part while the actor workgroup of the tile is still sampling, a' W1[:, S:]^T is added when it arrives
        // (... or of the chained actor update: an online critic's forward + dQ/da pass, 16- or 32-row tiles)
        const bool CONS = HO && ((MODE == MODE_PLAIN && TMR == 16) || MODE == MODE_CRITIC_U) && g.ho.pub != nullptr;
        const int XC = CONS ? g.ho.S : IN;   // columns of the x tile that are loaded
        int64_t gsrc[XR];
        bool xrok[XR];
#pragma unroll
        for (int j = 0; j < XR; ++j) {
            xrok[j] = (m0 + xr0 + 16 * j) < g.n_rows;
#if defined(SSAC_LAB) && defined(SSAC_EXP_IDENTITY_IDX)
            // (measurement build only, WRONG rows: the bound of a batch whose rows are known without the index load -- one
            //  dependent load less in front of fc1: round-5 review, item 1(b))
            gsrc[j] = gidx ? (int64_t)(m0 + xr0 + 16 * j) : 0;
#else
            gsrc[j] = gidx ? gidx[xrok[j] ? m0 + xr0 + 16 * j : 0] : 0;
#endif
        }
        float bv = 0.0f;  // thread t < H: b1[t]; thread 256 + t: b2[t]   (H <= 256, 512 threads)
        if (tid < H) bv = P[g.off[1] + tid];
        else if (tid >= 256 && tid - 256 < H) bv = P[g.off[3] + tid - 256];
        const float b3v = tid < OUT ? P[g.off[5] + tid] : 0.0f;
        const float *W3 = P + g.off[4];
        const int n3 = OUT * H;
        float w3v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i3 = tid + u * NTHR;
            w3v[u] = W3LATE ? 0.0f : W3[i3 < n3 ? i3 : 0];
        }
        // (consumer) W1's action columns, H x A floats over the 512 threads: requested here, parked in staging buffer 1
        // behind fc1's K loop
        float war[HANDOFF_MAX_WA];
        if (CONS) {
            const float *W1 = P + g.off[0];
            const int na = H * g.ho.A;
#pragma unroll
            for (int u = 0; u < HANDOFF_MAX_WA; ++u) {
                const int ei = tid + u * NTHR;
                const int c = ei / g.ho.A, i = ei - c * g.ho.A;
                war[u] = W1[ei < na ? (int64_t)c * IN + g.ho.S + i : 0];
            }
        }
        // COLUMN-SPLIT consumer (ho.nsplit = 2 or 4 workgroups per (slot, tile), hidden 256): this workgroup computes
        // hidden / nsplit columns of fc2 -- the waves form CG column groups of 32 x KG K-groups -- and its weights never
        // touch LDS: a lane's MFMA fragments of W2[its 32 columns][its K range] are requested HERE, 16-byte loads
        // straight into registers, and wait there while fc1 runs and the hand-off is polled.  Behind a' the workgroup then
        // has no memory access left but its LDS: NCH x 16 MFMAs per wave, a sum over the K-groups, the head's partial dot
        // product -- q_t[(slot nsplit + split)][row], summed by the TD evaluation (ssac_td_spec.n_parts).
        const int NSPL = (!CO && CONS && MODE == MODE_PLAIN && g.ho.nsplit > 1) ? g.ho.nsplit : 1;
        constexpr int MAXCH = 4;            // K chunks of 32 per wave: 4 (nsplit 2) or 2 (nsplit 4)
        f4 wq[MAXCH][2][2];
        const int CG_ = 8 / NSPL;           // column groups of 32 inside the workgroup's hidden / nsplit columns
        const int KG_ = 8 / CG_, KK_ = H / KG_, NCH_ = KK_ >> 5;
        const int scg = wave % CG_, skg = wave / CG_;
        const int scb = split * (H / NSPL) + scg * 32, skb = skg * KK_;
        if (NSPL > 1) {
            const float *W2 = P + g.off[2];
            const int li_ = lane & 15, lg_ = lane >> 4;
#pragma unroll
            for (int c = 0; c < MAXCH; ++c)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f4 *bp = reinterpret_cast<const f4 *>(W2 + (int64_t)(scb + 16 * u + li_) * H + skb + 32 * (c < NCH_ ? c : 0) + lg_ * 8);
                    wq[c][u][0] = bp[0];
                    wq[c][u][1] = bp[1];
                }
        }
        const float *xrow[XR];
        float xv[XR];
#pragma unroll
        for (int j = 0; j < XR; ++j) {
            const bool ok = xrok[j] && xk < XC;
            if (gidx) {
                const float *sr = (actor_half ? g.gth.s1 : g.gth.s) + gsrc[j] * Sg;
                const float *ar = g.gth.act + gsrc[j] * g.gth.a_elems - Sg;
                xrow[j] = sr;
                xv[j] = (xk < Sg ? sr : ar)[ok ? xk : (xk < Sg ? 0 : Sg)];
            } else {
                xrow[j] = X + (xrok[j] ? (int64_t)(m0 + xr0 + 16 * j) * g.ldx : 0);
                xv[j] = xrow[j][ok ? xk : 0];
            }
        }
        // (MODE_SAMPLE with in-kernel noise: the standard normals of this tile -- Philox4x32-10 + Box-Muller, ~300 instructions
        // that depend on nothing but (seed, draw, row, dimension) -- are computed HERE, while the prologue's loads are in
        // flight, and parked in the upper half of the sample epilogue's scratch rows; in the epilogue, behind the head, they
        // were half of its ~5 k clocks on the critical path of every launch that waits for its actor workgroups)
        if (MODE == MODE_SAMPLE && !g.eps) {
            const int A = OUT >> 1;
            for (int t = tid; t < TMR * A; t += NTHR) {
                const int r = t / A, i = t - r * A, b = m0 + r;
                const int64_t draw = (gslot && g.gth.rng_word >= 0)
                                         ? g.rng.offset + *reinterpret_cast<const int64_t *>(gslot + g.gth.rng_word)
                                         : rng_draw(g.rng);
                dqs[r * ldo + A + i] = b < g.n_rows ? philox_normal(g.rng.seed, draw, b, i) : 0.0f;
            }
        }
        // ---- ... and now the LDS stores (visible after the barrier below)
        if (tid < H) b1s[tid] = bv;
        else if (tid >= 256 && tid - 256 < H) b2s[tid - 256] = bv;
        if (tid < OUT) b3s[tid] = b3v;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i3 = tid + u * NTHR;
            if (!W3LATE && i3 < n3) {
                const int o = i3 / H;
                w3s[o * ldw3 + (i3 - o * H)] = w3v[u];
            }
        }
        float *outp = gidx ? (actor_half ? g.gth.x1sa : (e == 0 ? g.gth.xsa : nullptr))
                           : ((MODE == MODE_SAMPLE && g.copy_x) ? g.act_dst : nullptr);   // [s | .] for the critics
        if (CONS) outp = nullptr;   // (the actor workgroup of the tile writes the [s' | a'] rows)
        const int64_t ldo_g = gidx ? (actor_half ? g.gth.ld_x1 : g.gth.ld_x) : g.ld_act;
#pragma unroll
        for (int j = 0; j < XR; ++j) {
            const int r = xr0 + 16 * j;
            const bool ok = xrok[j] && xk < XC;
            xs[r * ldx_s + xk] = ok ? xv[j] : 0.0f;
            if (ok && outp) outp[(int64_t)(m0 + r) * ldo_g + xk] = xv[j];
        }
        // inputs wider than 32 columns: four column passes of every row in flight at a time
        for (int k0 = xk + 32; k0 < KP; k0 += 128) {
            float wv[XR][4];
#pragma unroll
            for (int j = 0; j < XR; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + 32 * u;
                    const bool okk = xrok[j] && k < XC;
                    if (gidx) {
                        const float *ar = g.gth.act + gsrc[j] * g.gth.a_elems - Sg;
                        wv[j][u] = (k < Sg ? xrow[j] : ar)[okk ? k : (k < Sg ? 0 : Sg)];
                    } else {
                        wv[j][u] = xrow[j][okk ? k : 0];
                    }
                }
#pragma unroll
            for (int j = 0; j < XR; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + 32 * u, r = xr0 + 16 * j;
                    if (k < KP) {
                        const bool okk = xrok[j] && k < XC;
                        xs[r * ldx_s + k] = okk ? wv[j][u] : 0.0f;
                        if (okk && outp) outp[(int64_t)(m0 + r) * ldo_g + k] = wv[j][u];
                    }
                }
        }
        if (gidx && actor_half && !CONS && tid < TMR && (m0 + tid) < g.n_rows) {
            const int64_t src = gidx[m0 + tid];
            g.gth.rew_out[m0 + tid] = g.gth.rew[src];
            g.gth.done_out[m0 + tid] = (float)g.gth.done[src];
        }
        if (!W3LATE && n3 > 4 * NTHR) stage_head_weights(w3s, W3, OUT, H, ldw3, tid, 4 * NTHR);  // heads wider than 8 outputs
        if (FWD_BWD && tid < TMR) {
            const int b = m0 + tid;
            const bool ok = b < g.n_rows;
            if (!UNSCALED) {
                rowin[tid] = ok ? td_of_row(g, b, e) : 0.0f;
                rowin[TMR + tid] = (ok && g.weight) ? g.weight[b] : 1.0f;
            }
            rowin[2 * TMR + tid] = (ok && OUT > 1) ? g.act[b * g.ld_a] : 0.0f;
        }
        if constexpr (!CO) stage_first(st1, Ws, IN, tid);
        // (every barrier of this body hands data over through LDS only, so it does not drain vmcnt: the activation
        // tiles written out for the weight-gradient launch, the gathered rows and the next phase's prefetched weight
        // chunk stay in flight across the phase boundaries.  The one global write -> read inside a workgroup, a' of
        // the target chains, keeps a full barrier in fused_chain_kernel.)
        lds_barrier();

        BSTAMP(1);
        // ---- fc1 (fc2's first weight chunk is requested during its last K chunk)
        T::zero(acc);
        if constexpr (CO) {
            gemm_direct16<false>(acc, d1, bq1, xs, ldx_s, IN, lane, d2, bq2);
            if (FWD_BWD) d3.init(P + g.off[2], H, H, col0, lane);
        } else {
            gemm_tile<TMR, false, DBUF>(acc, st1, xs, ldx_s, IN, Ws, Ws1, tid, col0, st2, H);
        }
        BSTAMP(2);
        if (NSPL == 1 && !CO) stage_first(st2, Ws, H, tid);
        if (FWD_BWD && !CO) {
            if (DIRECT_BWD) dnn.init(P + g.off[2], H, H, col0, lane);
            else st3.init(P + g.off[2], H, H, tid);
        }
        float *wa = CO ? wa_co : Ws1, *as_ = ys;   // (consumer) W1[:, S:S+A] as [H][A]; a' of the tile as [TMR][32] (ys | dqs: free until the head)
        if (CONS) {
            const int A_ = g.ho.A, na = H * A_;
#pragma unroll
            for (int u = 0; u < HANDOFF_MAX_WA; ++u) {
                const int ei = tid + u * NTHR;
                if (ei < na) wa[ei] = war[u];
            }
            const unsigned tag = g.ho.base + (g.ho.tick ? (unsigned)*g.ho.tick : 0u);
            for (int t = tid; t < TMR * A_; t += NTHR) {
                const int r = t / A_, i = t - r * A_, b = m0 + r;
                float v = 0.0f;
                if (b < g.n_rows) v = handoff_poll(g.ho.pub + (int64_t)b * A_ + i, tag);
                as_[r * 32 + i] = v;
            }
            lds_barrier();
        }
        int m1q = 0;
        T::foreach4(acc, lane, [&](int row, int cw, f4 val) {
            const int col = col0 + cw;  // 4 consecutive columns (H % 32 == 0: all four in range or none)
            if (col < H) {
                if (CONS) {   // + a' W1[:, S:S+A]^T: the action part of fc1
                    const int A_ = g.ho.A;
                    for (int i = 0; i < A_; ++i) {
                        const float av = as_[row * 32 + i];
#pragma unroll
                        for (int q = 0; q < 4; ++q) val[q] += av * wa[(col + q) * A_ + i];
                    }
                }
                const f4 bq = *reinterpret_cast<const f4 *>(b1s + col);
                f4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(val[i] + bq[i], 0.0f);
                *reinterpret_cast<f4 *>(h1s + row * ldh + col) = v;
                if (g.H1 && (m0 + row) < g.n_rows)
                    *reinterpret_cast<f4 *>(g.H1 + ((int64_t)e * g.n_rows + m0 + row) * H + col) = v;
                if (CO && FWD_BWD) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) m1 |= (v[i] > 0.0f ? 1u : 0u) << (4 * m1q + i);
                }
            }
            ++m1q;
        });
        lds_barrier();
        BSTAMP(3);
        if (TMR == 16 && NSPL > 1) {
            // ---- column-split consumer: fc2 on this workgroup's columns from the register-resident fragments
            typename Tile<16>::Acc a2;
            Tile<16>::zero(a2);
            const int li_ = lane & 15, lg_ = lane >> 4;
#pragma unroll
            for (int c = 0; c < MAXCH; ++c) {
                if (c < NCH_) {
                    const f4 *ap = reinterpret_cast<const f4 *>(h1s + li_ * ldh + skb + 32 * c + lg_ * 8);
                    const f4 a0 = ap[0], a1 = ap[1];
#pragma unroll
                    for (int t = 0; t < 8; ++t)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            a2.v[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[c][u][t >> 2][t & 3], t < 4 ? a0[t & 3] : a1[t & 3],
                                                                           a2.v[u], 0, 0, 0);
                }
            }
            // the K-groups' partial tiles meet in LDS (staging buffer 0 is free: no fc2 staging in this form)
            float *redk = Ws;   // [8 waves][16 rows][32 columns]
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) redk[(wave * 16 + li_) * 32 + 16 * u + 4 * lg_ + r] = a2.v[u][r];
            lds_barrier();
            // thread t: row t / 32, columns (t % 32) + 32 i of the workgroup's CW: bias, ReLU, times W3, summed per row
            const int CW = H / NSPL, row = tid >> 5, c0_ = tid & 31;
            float qp_ = 0.0f;
            for (int i = 0; i < CW / 32; ++i) {
                float v = 0.0f;
                for (int kg = 0; kg < KG_; ++kg) v += redk[((kg * CG_ + i) * 16 + row) * 32 + c0_];
                const int col = split * CW + 32 * i + c0_;
                v = fmaxf(v + b2s[col], 0.0f);
                qp_ += v * w3s[col];
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) qp_ += __shfl_xor(qp_, o, 64);   // (the row's 32 threads are half a wave)
            if (c0_ == 0 && g.Y && (m0 + row) < g.n_rows)
                y_store(g.Y + ((int64_t)e * NSPL + split) * g.n_rows + m0 + row, split == 0 ? qp_ + b3s[0] : qp_);
            consumer_arrive();
            return;
        }
        // ---- fc2 (the backward-data phase re-reads W2 as a row-contiguous image: its first chunk is
        //      requested during fc2's last K chunk)
        T::zero(acc);
        // (W3LATE: the head's rows go in flight now, W3L_MAX per thread, and land in staging buffer 1 behind the K loop)
        constexpr int W3L_MAX = (HEAD_MAX * 256 + NTHR - 1) / NTHR;   // 32: heads up to 64 outputs over 256 hidden units
        float w3r[W3LATE ? W3L_MAX : 1];
        if (W3LATE) {
#pragma unroll
            for (int u = 0; u < W3L_MAX; ++u) {
                const int i3 = tid + u * NTHR;
                w3r[u] = (u * NTHR < n3) ? W3[i3 < n3 ? i3 : 0] : 0.0f;
            }
        }
        if constexpr (CO) {
            if (FWD_BWD) gemm_direct16<false>(acc, d2, bq2, h1s, ldh, H, lane, d3, bq3);
            else gemm_direct16<false>(acc, d2, bq2, h1s, ldh, H, lane, nodirect, bq3);
        } else {
            if (FWD_BWD && DIRECT_BWD) gemm_tile<TMR, false, DBUF>(acc, st2, h1s, ldh, H, Ws, Ws1, tid, col0, dnn, H);
            else if (FWD_BWD) gemm_tile<TMR, false, DBUF>(acc, st2, h1s, ldh, H, Ws, Ws1, tid, col0, st3, H);
            else gemm_tile<TMR, false, DBUF>(acc, st2, h1s, ldh, H, Ws, Ws1, tid, col0, none, 0);
        }
        if (W3LATE) {   // (gemm_tile ended with a barrier: nobody reads the staging buffers any more; visible to the head
                        //  behind the barrier that follows the fc2 epilogue)
#pragma unroll
            for (int u = 0; u < W3L_MAX; ++u) {
                const int i3 = tid + u * NTHR;
                if (i3 < n3) {
                    const int o = i3 / H;
                    w3s[o * ldw3 + (i3 - o * H)] = w3r[u];
                }
            }
        }
        BSTAMP(4);
        // Single-output heads (every continuous critic): the head's dot product q = h2 . W3 is taken from the fc2
        // accumulators right here -- each lane holds 8 or 16 columns of ONE row, so the row's partial is a lane sum, one
        // or two shuffles and an 8-way sum over the waves through LDS -- instead of a separate MFMA pass over h2s.
        // In the TD-independent backward mode (MODE_CRITIC_U) the head's backward, dz2u = W3 (.) [h2 > 0], needs nothing
        // but h2's sign either, so it is written (to LDS in place of h2, and out for the weight-gradient launch) in the
        // same pass: no head phase, no selector phase, no head-backward phase, two barriers fewer.
        const bool dot_head = OUT == 1;
        const bool fuse_dz2 = MODE == MODE_CRITIC_U && OUT == 1;
        float qp = 0.0f;
        T::foreach4(acc, lane, [&](int row, int cw, f4 val) {
            const int col = col0 + cw;
            if (col < H) {
                const f4 bq = *reinterpret_cast<const f4 *>(b2s + col);
                const bool rok = (m0 + row) < g.n_rows;
                f4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(val[i] + bq[i], 0.0f);
                if (g.H2 && rok) *reinterpret_cast<f4 *>(g.H2 + ((int64_t)e * g.n_rows + m0 + row) * H + col) = v;
                if (dot_head) {
                    const f4 w = *reinterpret_cast<const f4 *>(w3s + col);
#pragma unroll
                    for (int i = 0; i < 4; ++i) qp += v[i] * w[i];
                    if (fuse_dz2) {
                        f4 dz;
#pragma unroll
                        for (int i = 0; i < 4; ++i) dz[i] = (rok && v[i] > 0.0f) ? w[i] : 0.0f;
                        *reinterpret_cast<f4 *>(h2s + row * ldh + col) = dz;
                        if (g.DZ2 && rok)
                            *reinterpret_cast<f4 *>(g.DZ2 + ((int64_t)e * g.n_rows + m0 + row) * H + col) = dz;
                        if (g.W3S && bx == 0 && row == 0) *reinterpret_cast<f4 *>(g.W3S + (int64_t)e * H + col) = w;
                    } else {
                        *reinterpret_cast<f4 *>(h2s + row * ldh + col) = v;
                    }
                } else {
                    *reinterpret_cast<f4 *>(h2s + row * ldh + col) = v;
                }
            }
        });
        if (dot_head) {
            if (TMR == 16) qp += __shfl_xor(qp, 16, 64);   // 16-row tiles: a row's 32 columns sit in 4 lanes, else in 2
            qp += __shfl_xor(qp, 32, 64);
            if (lane < TMR) hpart[wave * TMR + lane] = qp;   // (waves beyond the layer width hold zeros)
        }
        lds_barrier();

        BSTAMP(5);
        BSTAMP(6);
        if (dot_head) {
            if (tid < TMR) {
                float v = b3s[0];
#pragma unroll
                for (int w = 0; w < 8; ++w) v += hpart[w * TMR + tid];
                ys[tid * ldo] = v;
                if (g.Y && (m0 + tid) < g.n_rows) y_store(g.Y + (int64_t)e * g.n_rows + m0 + tid, v);
                if (HO && MODE == MODE_CRITIC_U && g.ho.qpub && (m0 + tid) < g.n_rows)   // the actor workgroup of the tile is polling
                    handoff_publish(g.ho.qpub + (int64_t)e * g.n_rows + m0 + tid,
                                    g.ho.base + (g.ho.tick ? (unsigned)*g.ho.tick : 0u), v);
            }
        } else
        // ---- head on the matrix cores: wave w multiplies the k-slice [32w, 32w+32) of h2 with W3^T
        //      (a 16-wide B tile, rows >= OUT zero); the 8 partial tiles are summed through LDS.
        {
            // 16 head outputs at a time (heads wider than one MFMA tile, e.g. a 34-output actor, take several passes)
            const int li = lane & 15, lg = lane >> 4;
            for (int ob = 0; ob < ldo; ob += 16) {
                f32x4 hacc[TMR / 16];
#pragma unroll
                for (int q = 0; q < TMR / 16; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) hacc[q][i] = 0.0f;
                if (col0 < H) {  // col0 = 32*wave doubles as this wave's k-slice start
                    const int lo = ob + li;
                    const f4 *bp = reinterpret_cast<const f4 *>(w3s + (lo < OUT ? lo : 0) * ldw3 + col0 + lg * 8);
                    const float keep = lo < OUT ? 1.0f : 0.0f;  // rows >= OUT of the 16-wide B tile are zero
                    const f4 b0 = bp[0] * keep, b1 = bp[1] * keep;
#pragma unroll
                    for (int q = 0; q < TMR / 16; ++q) {
                        const f4 *ap = reinterpret_cast<const f4 *>(h2s + (16 * q + li) * ldh + col0 + lg * 8);
                        const f4 a0 = ap[0], a1 = ap[1];
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            hacc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(t < 4 ? a0[t & 3] : a1[t & 3],
                                                                           t < 4 ? b0[t & 3] : b1[t & 3], hacc[q], 0, 0, 0);
                    }
                }
                if (ob > 0) lds_barrier();  // the previous block's partials have been summed
#pragma unroll
                for (int q = 0; q < TMR / 16; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        hpart[(wave * TMR + 16 * q + 4 * lg + r) * MAX_OUT + li] = hacc[q][r];
                lds_barrier();
                const int row = tid >> 4, o = ob + (tid & 15);
                if (row < TMR && o < OUT) {
                    float v = b3s[o];
#pragma unroll
                    for (int w = 0; w < 8; ++w) v += hpart[(w * TMR + row) * MAX_OUT + (tid & 15)];
                    ys[row * ldo + o] = v;
                    if (g.Y && (m0 + row) < g.n_rows) g.Y[((int64_t)e * g.n_rows + m0 + row) * OUT + o] = v;
                }
            }
        }
    }
    BSTAMP(7);
    if (MODE == MODE_PLAIN) { consumer_arrive(); return; }
    lds_barrier();  // hpart (= staging buffer 0) has been consumed
    if (IS_CRITIC && !CO && !DIRECT_BWD) stage_first(st3, Ws, H, tid);

    if (MODE == MODE_SAMPLE) {
        // tanh-normal head: one thread per row
        // one thread per (row, action dimension) for the transcendental work, then one per row sums the
        // dimensions' log-probability terms in index order (the order a serial loop would use)
        const int A = OUT >> 1;
        float *lpt = dqs;  // [TMR][MAX_OUT] scratch, unused in this mode
        const unsigned long long ho_tag = g.ho.pub ? (unsigned long long)(g.ho.base + (g.ho.tick ? (unsigned)*g.ho.tick : 0u)) << 32 : 0ull;
        for (int t = tid; t < TMR * A; t += NTHR) {
            const int r = t / A, i = t - r * A, b = m0 + r;
            if (b < g.n_rows) {
                const float mu = ys[r * ldo + i], raw = ys[r * ldo + A + i];
                const float log_std = g.lo + 0.5f * (g.hi - g.lo) * (tanhf(raw) + 1.0f);
                const float sd = expf(log_std);
                // (gather folded in: the draw number is read from the input slot itself -- the device copy of the
                // slot is being written by this very launch)
                // (in-kernel noise: drawn in the prologue by THIS thread -- same (row, dimension) mapping -- into the row's upper half)
                const float ep = g.eps ? g.eps[(int64_t)b * A + i] : lpt[r * ldo + A + i];
                const float u = mu + sd * ep;
                const float dlt = u - mu;
                lpt[r * ldo + i] = (-(dlt * dlt) / (2.0f * sd * sd) - logf(sd) - LOG_SQRT_2PI) -
                                       2.0f * (LOG_2 - u - softplus_f(-2.0f * u));
                const float a_new = tanhf(u);
                g.act_dst[b * g.ld_act + g.act_col0 + i] = a_new;
                if (g.ho.pub)   // the tile's target-critic workgroups are polling for it
                    __hip_atomic_store(g.ho.pub + (int64_t)b * A + i, ho_tag | (unsigned long long)__float_as_uint(a_new),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        BSTAMP(9);
        lds_barrier();
        if (g.logp && tid < TMR && (m0 + tid) < g.n_rows) {
            float lp = 0.0f;
            for (int i = 0; i < A; ++i) lp += lpt[tid * ldo + i];
            g.logp[m0 + tid] = lp;
        }
        BSTAMP(8);
        return;
    }

    if (IS_CRITIC) {
        // ---- loss gradient per row (learning.py:90-98, 112)
        const float pw = (g.popart && g.pop) ? g.popart->w : 1.0f;
        const float pb = (g.popart && g.pop) ? g.popart->b : 0.0f;
        const float gscale = -2.0f * pw / (g.denom * (float)g.n_rows);
        float lossv = 0.0f, errv = 0.0f;  // this row's loss terms (threads < TMR), summed over wave 0 below
        if (ACTOR) {
            // dL/d(actor output) of this tile's rows (learning.py:392-408): the policy gradient passes through the
            // arg-min critic of every row (dQ/da, left UNSCALED by MODE_CRITIC_U), then through the tanh-normal head
            const int A = OUT >> 1;
            const ActorBwdArgs &ab = g.ab;
            const float alpha = ab.use_entropy ? expf(ab.log_alpha[0]) : 0.0f;
            const float gq = -pw * ab.inv_members / (float)g.n_rows;
            const float cen = alpha * ab.inv_members / (float)g.n_rows;
            float part = 0.0f;
            for (int t = tid; t < TMR * A; t += NTHR) {
                const int r = t / A, i = t - r * A, b = m0 + r;
                float dmu = 0.0f, dls = 0.0f;
                if (b < g.n_rows) {
                    // (chained actor update: the critics' workgroups of this tile publish Q and dQ/da as tagged granules)
                    const bool polled = HO && g.ho.qpub != nullptr;
                    const unsigned ptag = polled ? g.ho.base + (g.ho.tick ? (unsigned)*g.ho.tick : 0u) : 0u;
                    float mq = polled ? handoff_poll(g.ho.qpub + b, ptag) : ab.qc[b];
                    int am = 0;
                    for (int j = 1; j < ab.n_critics; ++j) {
                        const float v = polled ? handoff_poll(g.ho.qpub + (int64_t)j * g.n_rows + b, ptag)
                                               : ab.qc[(int64_t)j * g.n_rows + b];
                        if (v < mq) { mq = v; am = j; }
                    }
                    const float dxv = polled ? handoff_poll(g.ho.dxpub + ((int64_t)am * g.n_rows + b) * A + i, ptag)
                                             : ab.dxu[((int64_t)am * g.n_rows + b) * A + i];
                    const float gsum = gq * dxv;
                    const float mu = ab.aout[(int64_t)b * OUT + i], raw = ab.aout[(int64_t)b * OUT + A + i];
                    const float th = tanhf(raw);
                    const float sd = expf(ab.lo + 0.5f * (ab.hi - ab.lo) * (th + 1.0f));
                    const float ep = ab.eps ? ab.eps[(int64_t)b * A + i] : philox_normal(g.rng.seed, rng_draw(g.rng), b, i);
                    const float a = tanhf(mu + sd * ep);
                    const float gu = gsum * (1.0f - a * a);
                    dmu = gu + cen * 2.0f * a;
                    dls = (gu * sd * ep + cen * (-1.0f + 2.0f * a * sd * ep)) * 0.5f * (ab.hi - ab.lo) * (1.0f - th * th);
                    g.DQ[(int64_t)b * OUT + i] = dmu;
                    g.DQ[(int64_t)b * OUT + A + i] = dls;
                    if (i == 0) part += (pw * mq + pb) - (ab.use_entropy ? alpha * ab.logp[b] : 0.0f);
                }
                dqs[r * ldo + i] = dmu;
                dqs[r * ldo + A + i] = dls;
            }
            // the tile's loss term: fixed-order sum over the workgroup (partials are summed by ssac_actor_logs)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
            if (lane == 0) rowred[wave] = part;
            lds_barrier();
            if (tid == 0) {
                float tot = 0.0f;
                for (int w = 0; w < NTHR / 64; ++w) tot += rowred[w];
                ab.partials[bx] = tot;
            }
        } else if (UNSCALED && MODE == MODE_CRITIC_U && OUT == 1) {
            // (dz2u was written by the fc2 epilogue: nothing to select, no head backward)
        } else if (UNSCALED) {
            // selector of the head output the loss looks at (the taken action; the only output when OUT == 1)
            if (tid < TMR) {
                const int ai = OUT > 1 ? (int)rowin[2 * TMR + tid] : 0;
                for (int o = 0; o < OUT; ++o) dqs[tid * ldo + o] = (o == ai && (m0 + tid) < g.n_rows) ? 1.0f : 0.0f;
            }
        } else if (tid < TMR) {
            const int b = m0 + tid;
            int ai = 0;
            float dsel = 0.0f;
            if (b < g.n_rows) {
                if (OUT > 1) ai = (int)rowin[2 * TMR + tid];
                const float w = rowin[TMR + tid];
                const float err = rowin[tid] - (pw * ys[tid * ldo + ai] + pb);
                lossv = w * err * err;
                errv = err;
                dsel = gscale * w * err;
            }
            for (int o = 0; o < OUT; ++o) {
                const float d = (o == ai) ? dsel : 0.0f;
                dqs[tid * ldo + o] = d;
                if (b < g.n_rows) g.DQ[((int64_t)e * g.n_rows + b) * OUT + o] = d;
            }
        }
        if (!UNSCALED && !ACTOR && wave == 0) {  // fixed shuffle tree over the tile's rows: no LDS round trip, no serial loop
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lossv += __shfl_xor(lossv, o, 64); errv += __shfl_xor(errv, o, 64); }
            if (lane == 0) {
                const int64_t pi = ((int64_t)e * grid_x + bx) * 2;
                g.partials[pi] = lossv;
                g.partials[pi + 1] = errv;
            }
        }
        const bool dz2_done = MODE == MODE_CRITIC_U && OUT == 1;
        if (!dz2_done) lds_barrier();  // dqs visible to every wave
        BSTAMP(8);
        // ---- head backward: dz2 = (dq W3) (.) [h2 > 0], in place over h2s
        if (!dz2_done) {
            // thread -> 4 consecutive columns k = 4 (tid % 64), rows r = (tid >> 6), +8, ...  (H <= 256): 16-byte LDS
            // reads / writes and global stores.  gs accumulates over o in index order per element, as before.
            const int k = (tid & 63) * 4;
            if (k < H) {
                // all of this thread's h2 values are loaded before the first in-place store (the compiler cannot
                // prove the stores do not alias the later loads, and would serialise the LDS round trips)
                constexpr int NR = TMR / 8;
                const int r0 = tid >> 6;
                f4 hv[NR], gs[NR];
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    hv[j] = *reinterpret_cast<const f4 *>(h2s + (r0 + 8 * j) * ldh + k);
                    gs[j] = (f4){0.f, 0.f, 0.f, 0.f};
                }
                for (int o = 0; o < OUT; ++o) {
                    const f4 w = *reinterpret_cast<const f4 *>(w3s + o * ldw3 + k);
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        const float d = dqs[(r0 + 8 * j) * ldo + o];
#pragma unroll
                        for (int i = 0; i < 4; ++i) gs[j][i] += d * w[i];
                    }
                }
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int r = r0 + 8 * j;
                    f4 dz;
#pragma unroll
                    for (int i = 0; i < 4; ++i) dz[i] = hv[j][i] > 0.0f ? gs[j][i] : 0.0f;
                    *reinterpret_cast<f4 *>(h2s + r * ldh + k) = dz;
                    if (g.DZ2 && (m0 + r) < g.n_rows)
                        *reinterpret_cast<f4 *>(g.DZ2 + ((int64_t)e * g.n_rows + m0 + r) * H + k) = dz;
                }
            }
        }
        BSTAMP(9);
        // ---- backward-data of fc2: dz1 = (dz2 W2) (.) [h1 > 0]
        lds_barrier();  // dz2 (in h2s) and the staged first chunk of W2 are visible
        T::zero(acc);
        if constexpr (CO) gemm_direct16<true>(acc, d3, bq3, h2s, ldh, H, lane, nodirect, bq1);
        else if constexpr (DIRECT_BWD) gemm_direct_nn<TMR>(acc, dnn, h2s, ldh, H, lane);
        else gemm_tile<TMR, true, DBUF>(acc, st3, h2s, ldh, H, Ws, Ws1, tid, col0, none, 0);
        BSTAMP(10);
        const bool want_dx = MODE == MODE_CRITIC_U && g.DXU != nullptr;
        int m1r = 0;
        T::foreach4(acc, lane, [&](int row, int cw, f4 val) {
            const int col = col0 + cw;
            if (col < H) {
                f4 v;
                if (CO) {   // (h1 is gone from LDS: its mask was kept by the fc1 epilogue)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = ((m1 >> (4 * m1r + i)) & 1u) ? val[i] : 0.0f;
                } else {
                    const f4 hq = *reinterpret_cast<const f4 *>(h1s + row * ldh + col);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = hq[i] > 0.0f ? val[i] : 0.0f;
                }
                if (g.DZ1 && (m0 + row) < g.n_rows)
                    *reinterpret_cast<f4 *>(g.DZ1 + ((int64_t)e * g.n_rows + m0 + row) * H + col) = v;
                if (want_dx) *reinterpret_cast<f4 *>(h2s + row * ldh + col) = v;   // (dz2 is dead: gemm_tile ended with a barrier)
            }
            ++m1r;
        });
        if (want_dx) {
            // dX[b][c] = sum_k dz1u[b][k] W1[k][dx_col0 + c]: the (unscaled) gradient w.r.t. the ACTION columns of the
            // critic input -- all the online actor update needs from the critics' backward pass (learning.py:402-411).
            // W1's action columns (H x DC floats) go through LDS first (staging buffer 0 is free behind the last K loop):
            // read from global memory inside the k loop they were 256 strided loads per thread, ~5 us of a 35 us launch.
            const float *W1 = P + g.off[0];
            const int DC = g.dx_cols;
            float *wdx = Ws;   // [H][DC]
            for (int i = tid; i < H * DC; i += NTHR) {
                const int k = i / DC, cix = i - k * DC;
                wdx[i] = W1[(int64_t)k * IN + g.dx_col0 + cix];
            }
            lds_barrier();
            const unsigned dtag = (HO && g.ho.dxpub) ? g.ho.base + (g.ho.tick ? (unsigned)*g.ho.tick : 0u) : 0u;
            for (int t = tid; t < TMR * DC; t += NTHR) {
                const int r = t / DC, cix = t - r * DC;
                if ((m0 + r) < g.n_rows) {
                    float sx = 0.0f;
                    for (int k = 0; k < H; ++k) sx += h2s[r * ldh + k] * wdx[k * DC + cix];
                    g.DXU[((int64_t)e * g.n_rows + m0 + r) * DC + cix] = sx;
                    if (HO && g.ho.dxpub) handoff_publish(g.ho.dxpub + ((int64_t)e * g.n_rows + m0 + r) * DC + cix, dtag, sx);
                }
            }
        }
        BSTAMP(11);
    }
}


template <int MODE, int TMR, bool DBUF>
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(DBUF ? 2 : 4, DBUF ? 2 : 4)))
void fused_mlp_kernel(FusedArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = ssac_xcd_contiguous(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, g.xcd);
    fused_mlp_body<MODE, TMR, DBUF>(g, smem, L % gridDim.x, L / gridDim.x, gridDim.x);
}

// Two independent fused launches in ONE: workgroups [0, tiles_a) run the actor (+ tanh-normal sample) on 16-row
// tiles, the rest run the online critics' FORWARD (h1 / h2 / q saved) on 32-row tiles.  The critic forward does not
// depend on the sampled action, so it fills the ~220 CUs the 32 actor workgroups leave idle; the critic kernel
// that follows the target critics is then only its backward half (MODE_CRITIC_BWD).  Same-launch workgroups need no
// cross-stream signalling (which costs ~10 us on this platform, see SPLIT_FORWARD in learning.py).
// TC = row-tile size of the critic half (the same automatic choice a stand-alone forward would make, so the
// results are bit-identical to it).
// ADBUF = the actor half double-buffers its weight staging (false when a wide input + wide head leave no room).
template <int TC, bool ADBUF>
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fused_dual_kernel(FusedArgs ga, FusedArgs gc, int tiles_a, int critic_grid_x) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = ssac_xcd_contiguous(blockIdx.x, gridDim.x, gc.xcd);
    if (bid < tiles_a) {
        fused_mlp_body<MODE_SAMPLE, 16, ADBUF>(ga, smem, bid, 0, tiles_a);
    } else {
        const int L = bid - tiles_a;
        fused_mlp_body<MODE_PLAIN, TC, true>(gc, smem, L % critic_grid_x, L / critic_grid_x, critic_grid_x);
    }
}

// Second merged launch of an update: workgroups [0, tiles_t) run the target critics' forward on the REDQ subset
// (net_ids), the rest run the TD-independent half of the online critics' backward pass (MODE_CRITIC_BWDU) on the
// forward the first merged launch saved.
template <int TT, int TC>
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fused_dual2_kernel(FusedArgs gt, FusedArgs gc, int tiles_t, int target_grid_x, int critic_grid_x) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = ssac_xcd_contiguous(blockIdx.x, gridDim.x, gc.xcd);
    if (bid < tiles_t) {
        fused_mlp_body<MODE_PLAIN, TT, true>(gt, smem, bid % target_grid_x, bid / target_grid_x, target_grid_x);
    } else {
        const int L = bid - tiles_t;
        fused_mlp_body<MODE_CRITIC_BWDU, TC, true>(gc, smem, L % critic_grid_x, L / critic_grid_x, critic_grid_x);
    }
}

// ONE launch for everything of a critic update that does not need the TD target (continuous actions):
//   workgroups [0, tiles_t): the TARGET CHAIN of subset slot j on a 16-row tile -- actor forward on s' + tanh-normal
//     sample (a' and log pi; every slot recomputes the actor for its rows: 75 MFLOP per slot, and no launch boundary
//     between the actor and the target critics), then the target critic `ids[j]` on [s'|a'];
//   the rest: the online critics' forward AND the TD-independent half of their backward in the same workgroup
//     (MODE_CRITIC_U): h1 / h2 stay in LDS between the two, instead of being written out and read back by a
//     second launch.
// The replay gather and the start-of-update duties ride along as in fused_dual_kernel (actor part of slot 0).
// ADBUF = the actor pass double-buffers its weight staging (false: wide input + wide head, e.g. Humanoid's 376 -> 34).
template <int TC, bool ADBUF>
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fused_chain_kernel(FusedArgs ga, FusedArgs ga_rest, FusedArgs gt, FusedArgs gc, int tiles_t, int target_grid_x,
                        int critic_grid_x, DeferredLogsArgs dl, int dl_on) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    SSAC_LAB_ONLY(if (gc.tl && threadIdx.x == 0 && bid < 512) gc.tl[2 * bid] = __builtin_amdgcn_s_memrealtime();)
    if (dl_on && bid == (int)gridDim.x - 1) {
        // one extra workgroup: the PREVIOUS recorded update's log block -> its slot of the log ring (ssac_critic_logs.h)
        deferred_logs_body(dl, -1);
        return;
    }
    // The longer chain takes the FIRST workgroup ids (a launch's workgroups start in id order, ~1 us from first to last,
    // and when the launch is more than one round of workgroups the ids behind the first round wait for a whole
    // predecessor): 32-row critic tiles (forward + backward, ~58 k clocks) before the target chains (~57 k), but the
    // target chains before 16-row critic tiles (~45 k).
    constexpr bool CRIT_FIRST = TC == 32;
    const int n_main = (int)gridDim.x - (dl_on ? 1 : 0), n_crit = n_main - tiles_t;
    const int t_lo = CRIT_FIRST ? n_crit : 0, t_hi = CRIT_FIRST ? n_main : tiles_t;   // ids of the target chains
    if (bid >= t_lo && bid < t_hi) {
        // (XCD-contiguous order within each half, ssac_internal.h: the row tiles of one net -- which stream the same
        // weights -- sit on one or two XCDs instead of all eight)
        const int lb = ssac_xcd_contiguous_range(bid, t_lo, t_hi, gc.xcd);
        const int j = lb / target_grid_x, bx = lb - j * target_grid_x;
        if (j > 0) {
            // critic-sharded ranks ("only subset owners need to send"): a slot whose REDQ member lives on
            // another rank has no target critic to feed here, and its a' / log pi rows are slot 0's work -- no actor
            // pass for it; the target-critic pass below just stores +inf (the neutral element of the exchange's MIN)
            const int32_t *idsp = gt.ids;
            if (gt.gth_role == 4 && gt.gth.feed && gt.gth.ids_word >= 0) {
                if (gt.slot_now) {
                    idsp = reinterpret_cast<const int32_t *>(gt.slot_now + gt.gth.ids_word);
                } else {
                    const ssac_feed f = *gt.gth.feed;
                    idsp = reinterpret_cast<const int32_t *>(feed_slot(f) + gt.gth.ids_word);
                }
            }
            if (!(idsp && idsp[j] < 0)) fused_mlp_body<MODE_SAMPLE, 16, ADBUF>(ga_rest, smem, bx, 0, target_grid_x, -1);
        } else {
            fused_mlp_body<MODE_SAMPLE, 16, ADBUF>(ga, smem, bx, 0, target_grid_x, 0);
        }
        __threadfence_block();  // this workgroup's a' rows (global) are read back by its own target-critic pass
        __syncthreads();
        // (phase stamps: the actor pass in slots 0.., the target-critic pass of subset slot 0 in 16.., the critics in 32..)
        fused_mlp_body<MODE_PLAIN, 16, true>(gt, smem, bx, j, target_grid_x, j == 0 ? 16 : -1);
    } else {
        const int L = CRIT_FIRST ? ssac_xcd_contiguous_range(bid, 0, n_crit, gc.xcd)
                                 : ssac_xcd_contiguous_range(bid, tiles_t, n_main, gc.xcd);
        fused_mlp_body<MODE_CRITIC_U, TC, true>(gc, smem, L % critic_grid_x, L / critic_grid_x, critic_grid_x, 32);
    }
#ifdef SSAC_LAB
    if (gc.tl && bid < 512) {   // (the stamp buffer holds 512 workgroups per launch: tools/wg_timeline.py)
        __syncthreads();   // (drains this workgroup's stores too)
        if (threadIdx.x == 0) gc.tl[2 * bid + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// The chained launch with the target chains cut into PRODUCER and CONSUMER workgroups (round 4).
// In fused_chain_kernel a target chain is one workgroup running two dependent MLP passes -- actor on s', then the target
// critic on [s'|a'] -- and every subset slot repeats the actor pass: 26 + 23 k clocks + a global round trip between
// them, the longest workgroup of every under-filled launch.  Here:
//   workgroups [0, tiles_a): the ACTOR of a 16-row tile, ONCE (not per slot): forward, tanh-normal sample, log pi, the
//     [s'|a'] rows -- and a' published as tagged 8-byte granules (Handoff);
//   tiles_t consumers: the TARGET CRITIC ids[j] of (slot j, tile): everything that does not need a' FIRST -- prologue, its
//     own gather of the s' rows, fc1 on the state columns (the action columns of its x tile are zero) -- then it polls
//     the tile's granules, adds a' W1[:, S:S+A]^T to the fc1 accumulators and runs on.  Its critical path behind the
//     actor is fc2 + head (~14 k clocks) instead of a whole pass;
//   the rest: the online critics' tiles, unchanged.
// Producers take the first workgroup ids: they are dispatched before any consumer and never wait, so the polling cannot
// deadlock whatever part of the grid is resident.  The rank-A update adds a' W1^T after the state columns' sum (the
// one-pass kernel sums all columns of a K chunk in MFMA order): same value up to fp32 association.
template <int TC, bool ADBUF, bool AW3LATE = false>
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fused_chain_pc_kernel(FusedArgs ga, FusedArgs gt, FusedArgs gc, int tiles_a, int tiles_t, int target_grid_x,
                           int critic_grid_x, DeferredLogsArgs dl, int dl_on, XchgArgs xa, int xchg_on) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    SSAC_LAB_ONLY(if (gc.tl && threadIdx.x == 0 && bid < 512) gc.tl[2 * bid] = __builtin_amdgcn_s_memrealtime();)
    if (dl_on && bid == (int)gridDim.x - 1) {
        deferred_logs_body(dl, -1);   // the PREVIOUS recorded update's log block -> its slot of the log ring
        return;
    }
    if (xchg_on && bid == (int)gridDim.x - 1 - (dl_on ? 1 : 0)) {
        // Critic-sharded rank (round 5): the exchange of the subset's target Q runs HERE, in a tail workgroup of the launch
        // that produced it, instead of as a launch of its own between this one and the weight-gradient launch (measured on an
        // installed rank: that third launch -- two boundaries, a one-workgroup kernel -- cost 8.4 - 9.9 us per update,
        // profiles/r5_sharding_budget.md).  It waits until every target-critic workgroup of the launch has counted itself in
        // (their Q left as agent-scope stores), then does exactly what xchg_kernel does -- same protocol, same slots, flags and
        // acknowledgements -- and overlaps the critic tiles' tail.  It holds the highest workgroup id but the log workgroup's:
        // everything it waits for was dispatched before it.
        const int32_t *own = nullptr;
        if (gt.gth.feed && gt.gth.ids_word >= 0) {   // this update's id block, from the input slot itself (uncached)
            const uint32_t *gslot = gt.slot_now;
            if (!gslot) { const ssac_feed f = *gt.gth.feed; gslot = feed_slot(f); }
            own = reinterpret_cast<const int32_t *>(gslot + gt.gth.ids_word);
        }
        int *s_late = reinterpret_cast<int *>(smem) + 4;   // (behind the 16 bytes xchg_body keeps at the front of the block)
        if (threadIdx.x == 0) {
            const long long t0 = __builtin_amdgcn_s_memtime();
            const long long limit = *xa.dead ? 0 : xa.spin_limit;
            int late = 0;
            while (__hip_atomic_load(xa.arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)tiles_t) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memtime() - t0 > limit) { late = 1; break; }
            }
            // (LAB build, ssac_xchg_test_mode bit 2: behave as if the wait had given up -- the failure path's test)
            if (X_TEST_MODE(xa) & 4) late = 1;
            // a wait that gave up: the Qt in `data` is incomplete.  The exchange is told to FAIL (nothing sent, result poisoned,
            // error word raised, `dead` set -- xchg_body's !s_ok branch), and the counter is left alone: stragglers may still
            // bump it, and with `dead` set no later launch of this engine trusts it again
            if (!late) __hip_atomic_store(xa.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_late = late;
        }
        __syncthreads();
        const bool late = *s_late != 0;
        __syncthreads();   // (everyone has read the word before xchg_body's thread 0 reuses the front of the block)
        xchg_body<NTHR, true>(xa, smem, own, late);   // (its 16 bytes of LDS: the front of this workgroup's unused dynamic block)
        return;
    }
    // 32-row critic tiles (~58 k clocks) go before the consumers (~41 k from the launch's start, most of it waiting),
    // 16-row critic tiles (~40 k) behind them
    constexpr bool CRIT_FIRST = TC == 32;
    const int n_main = (int)gridDim.x - (dl_on ? 1 : 0) - (xchg_on ? 1 : 0), n_crit = n_main - tiles_a - tiles_t;
    const int t_lo = CRIT_FIRST ? tiles_a + n_crit : tiles_a, t_hi = t_lo + tiles_t;   // ids of the consumers
    if (bid < tiles_a) {
        fused_mlp_body<MODE_SAMPLE, 16, ADBUF, false, AW3LATE>(ga, smem, ssac_xcd_contiguous_range(bid, 0, tiles_a, gc.xcd), 0, tiles_a, 0);
    } else if (bid >= t_lo && bid < t_hi) {
        const int lb = ssac_xcd_contiguous_range(bid, t_lo, t_hi, gc.xcd);
        const int per_slot = target_grid_x * (gt.ho.nsplit > 1 ? gt.ho.nsplit : 1);   // consumers of a subset slot
        const int j = lb / per_slot, rem = lb - j * per_slot;
        const int sp = rem / target_grid_x, bx = rem - sp * target_grid_x;
        fused_mlp_body<MODE_PLAIN, 16, true, true>(gt, smem, bx, j, target_grid_x, (j == 0 && sp == 0) ? 16 : -1, sp);
    } else {
        const int c_lo = CRIT_FIRST ? tiles_a : t_hi;
        const int L = ssac_xcd_contiguous_range(bid, c_lo, c_lo + n_crit, gc.xcd);
        fused_mlp_body<MODE_CRITIC_U, TC, true>(gc, smem, L % critic_grid_x, L / critic_grid_x, critic_grid_x, 32);
    }
#ifdef SSAC_LAB
    if (gc.tl && bid < 512) {
        __syncthreads();
        if (threadIdx.x == 0) gc.tl[2 * bid + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// The producer / consumer chained launch with TWO workgroups resident per CU (round 5).  At the
// headline shape fused_chain_pc_kernel is exactly as long as one of its 160 32-row critic tiles, ~58 k clocks of which
// ~18 k (prologue, fc1, epilogues, dz1 store) overlap nothing: ~130 KB of LDS per workgroup means one workgroup per CU.
// Here every role runs 16-row tiles on the co-resident carve (fused_mlp_body<..., CO>: <= 80 KB of LDS, <= 128 VGPRs), so
// the 32 producers + 64 consumers + 320 critic tiles of the headline are ONE resident round at two per CU (4 waves per
// SIMD) -- the idea being that one tile's non-matrix phases hide under its neighbour's K loop.  Same arithmetic, operation
// by operation, as the 16-row tiles of fused_chain_pc_kernel<16, ...> (bit-identical outputs).
// MEASURED (profiles/r5_chain_coresident.md): NOT faster -- 35.0 - 38.6 us per launch against 29.8 (56 - 63 us per update
// against 51.4): a pair of 16-row tiles on a CU takes as long as the two one after the other (twice the prologues,
// epilogues, weight passes and barriers per 32 rows), and the actor -> target-critic chain slows down when it shares CUs.
// Kept as a selectable, tested form (ssac_chain_form(1)); the library's default is one workgroup per CU.
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(4, 4)))
void fused_chain_co_kernel(FusedArgs ga, FusedArgs gt, FusedArgs gc, int tiles_a, int tiles_t, int target_grid_x,
                           int critic_grid_x, DeferredLogsArgs dl, int dl_on) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    SSAC_LAB_ONLY(if (gc.tl && threadIdx.x == 0 && bid < 512) gc.tl[2 * bid] = __builtin_amdgcn_s_memrealtime();)
    if (dl_on && bid == (int)gridDim.x - 1) {
        deferred_logs_body(dl, -1);   // the PREVIOUS recorded update's log block -> its slot of the log ring
        return;
    }
    const int n_main = (int)gridDim.x - (dl_on ? 1 : 0), n_crit = n_main - tiles_a - tiles_t;
    // Workgroup ids: [critic tiles, first part][producers][consumers][critic tiles, rest].  The dispatcher hands out first
    // slots CU by CU and then second slots in the same CU order: with 256 - (producers + consumers) critic tiles in front, the
    // actor -> target-critic chain (ONE dependent chain that must end before the launch does) gets CUs of its own and the
    // tiles of the second round double up on critic CUs.  (Every workgroup of the launch is resident at once -- <= 512
    // at two per CU, checked by the launcher -- so the producers need not hold the lowest ids here.)
    int c_first = 256 - tiles_a - tiles_t;
    c_first = c_first < 0 ? 0 : (c_first > n_crit ? n_crit : c_first);
    const int p_lo = c_first, t_lo = p_lo + tiles_a, t_hi = t_lo + tiles_t;
#if !(defined(SSAC_LAB) && defined(SSAC_EXP_NO_PRIO))
    if (bid >= p_lo && bid < t_hi) __builtin_amdgcn_s_setprio(3);
#endif
    if (bid >= p_lo && bid < t_lo) {
        fused_mlp_body<MODE_SAMPLE, 16, false, false, false, true>(ga, smem, ssac_xcd_contiguous_range(bid, p_lo, t_lo, gc.xcd), 0, tiles_a, 0);
    } else if (bid >= t_lo && bid < t_hi) {
        const int lb = ssac_xcd_contiguous_range(bid, t_lo, t_hi, gc.xcd);
        const int j = lb / target_grid_x, bx = lb - j * target_grid_x;
        fused_mlp_body<MODE_PLAIN, 16, false, true, false, true>(gt, smem, bx, j, target_grid_x, j == 0 ? 16 : -1, 0);
    } else {
        // critic tiles: logical ids [0, c_first) in front of the chain's workgroups, [c_first, n_crit) behind
        const int L = bid < p_lo ? ssac_xcd_contiguous_range(bid, 0, c_first, gc.xcd)
                                 : c_first + ssac_xcd_contiguous_range(bid, t_hi, t_hi + (n_crit - c_first), gc.xcd);
        fused_mlp_body<MODE_CRITIC_U, 16, false, false, false, true>(gc, smem, L % critic_grid_x, L / critic_grid_x, critic_grid_x, 32);
    }
#ifdef SSAC_LAB
    if (gc.tl && bid < 512) {
        __syncthreads();
        if (threadIdx.x == 0) gc.tl[2 * bid + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

#ifdef SSAC_VGPR_PROBE   // (register budget of each role of fused_chain_co_kernel on its own: hipcc -DSSAC_VGPR_PROBE -S)
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(4, 4))) void probe_co_producer(FusedArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    fused_mlp_body<MODE_SAMPLE, 16, false, false, false, true>(g, smem, blockIdx.x, 0, gridDim.x, 0);
}
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(4, 4))) void probe_co_consumer(FusedArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    fused_mlp_body<MODE_PLAIN, 16, false, true, false, true>(g, smem, blockIdx.x, 0, gridDim.x, -1, 0);
}
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(4, 4))) void probe_co_critic(FusedArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    fused_mlp_body<MODE_CRITIC_U, 16, false, false, false, true>(g, smem, blockIdx.x, 0, gridDim.x, 32);
}
#endif

// The online actor update's three dependent passes as ONE launch (round 4).  Stand-alone they are
// actor forward + rsample (32 workgroups, ~15 us), every critic's forward + dQ/da (~35 us), arg-min routing + tanh-normal
// backward + actor backward-data (32 workgroups, ~15 us): three launch boundaries in front of small launches.  Here
//   workgroups [0, tiles_a): the ACTOR of a 16-row tile: forward, rsample, log pi, the [s | a] rows, a published as tagged
//     granules (Handoff::pub) -- then it WAITS for its rows' Q_j and dQ_j/da from every critic (Handoff::qpub / dxpub) and
//     runs its backward half (MODE_ACTOR_BWD) on the h1 / h2 / head outputs it has just written;
//   the rest: the critics' tiles (MODE_CRITIC_U as a hand-off CONSUMER: gather of the state columns, fc1 on them, then a
//     arrives, rank-A update, fc2, head, the unscaled backward, dQ/da) -- Q and dQ/da go out as granules.
// Actor workgroups hold the lowest ids: they are resident before any critic tile can wait for them, and while they wait
// for the critics they block nobody (a launch of more critic tiles than CUs drains behind them).
// ADBUF = false (round 6): the ACTOR's two passes with a single weight-staging buffer -- a wide input together with a wide
// head (Humanoid's 376 -> 256 -> 34: 35 KB of W3 resident beside a 384-column x tile) leaves no room for the second one;
// the critics' tiles keep theirs.  Same arithmetic, same order: the carve only decides how far ahead the weights stream.
template <int TC, bool ADBUF = true>
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fused_actor_chain_kernel(FusedArgs ga, FusedArgs gb, FusedArgs gc, int tiles_a, int critic_grid_x) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    if (bid < tiles_a) {
        fused_mlp_body<MODE_SAMPLE, 16, ADBUF>(ga, smem, bid, 0, tiles_a, -1);
        __threadfence_block();  // this workgroup's h1 / h2 / head-output rows (global) are read back by its backward half
        __syncthreads();
        fused_mlp_body<MODE_ACTOR_BWD, 16, ADBUF, true>(gb, smem, bid, 0, tiles_a, -1);
    } else {
        const int L = ssac_xcd_contiguous_range(bid, tiles_a, (int)gridDim.x, gc.xcd);
        fused_mlp_body<MODE_CRITIC_U, TC, true, true>(gc, smem, L % critic_grid_x, L / critic_grid_x, critic_grid_x, -1);
    }
}

long long *g_fused_dbg = nullptr;

int g_tile_rows = 0;  // 0 = automatic, else 16 or 32 (ssac_fused_tile_rows)
constexpr int CHAIN_FORM_DEFAULT = 0;
int g_chain_form = CHAIN_FORM_DEFAULT;  // ssac_chain_form: 0 = one workgroup per CU (fused_chain_pc_kernel), 1 = the co-resident form where it applies

// co: the co-resident carve (one activation tile instead of two); cons_wa: floats of W1's action columns a CO consumer parks
size_t fused_lds_bytes(int in_dim, int hidden, int out_dim, int tm = TM, bool dbuf = true, bool w3_late = false,
                       bool co = false, int cons_wa = 0) {
    const int KP = (in_dim + 31) & ~31;
    return sizeof(float) * ((size_t)tm * (KP + APAD) + (co ? 1 : 2) * (size_t)tm * (hidden + APAD) +
                            (co ? CO_SCRATCH : (dbuf ? 2 : 1) * WS_FLOATS) + 2 * tm * ((out_dim + 15) & ~15) + 64 +
                            2 * hidden + HEAD_MAX + (w3_late ? 0 : (size_t)out_dim * (hidden + APAD)) + 3 * tm + cons_wa);
}

// Kernel variant for one launch.  g_tile_rows (ssac_fused_tile_rows) forces one: 16 / 32 = double-buffered weight
// staging; 17 = 16 rows, single staging buffer (wide input + wide head shapes need it).
struct TileChoice { int tm; int variant; };  // variant 0 double-buffered staging, 1 single buffer

TileChoice choose_tile(const FusedArgs &g, int n_sel) {
    const bool fits32 = fused_lds_bytes(g.in_dim, g.hidden, g.out_dim, 32) <= 160 * 1024;
    // a wide input together with a wide head (e.g. 376 -> 34): only the single-buffer carve fits
    if (fused_lds_bytes(g.in_dim, g.hidden, g.out_dim, 16) > 160 * 1024) return {16, 1};
    switch (g_tile_rows) {
        case 17: return {16, 1};
        case 16: return {16, 0};
        case 32: return {fits32 ? 32 : 16, 0};
        default: break;
    }
    // automatic: 16-row tiles while the launch still fits the chip in one round (one workgroup per CU), and
    // whenever wide inputs / heads leave no room for 32 rows in the 160 KB of LDS
    const int wg16 = ((g.n_rows + 15) / 16) * n_sel;
    return {(wg16 <= 256 || !fits32) ? 16 : 32, 0};
}

bool fused_ok(const ssac_mlp *n) {
    return n && n->hidden % 32 == 0 && n->hidden <= 256 && n->out_dim >= 1 && n->out_dim <= HEAD_MAX &&
           n->in_dim >= 1 && fused_lds_bytes(n->in_dim, n->hidden, n->out_dim, 16, false) <= 160 * 1024;
}

// the actor pass with a double-buffered K loop and the head's rows parked in staging buffer 1 (W3LATE): for actors whose
// resident head image leaves no room for the second buffer
bool fused_dbuf_late_ok(const ssac_mlp *n) {
    return fused_ok(n) && (size_t)n->out_dim * (n->hidden + APAD) <= (size_t)WS_FLOATS &&
           fused_lds_bytes(n->in_dim, n->hidden, n->out_dim, 16, true, true) <= 160 * 1024;
}

// the merged launches run their critic halves with double-buffered staging only
bool fused_dbuf_ok(const ssac_mlp *n) {
    return fused_ok(n) && fused_lds_bytes(n->in_dim, n->hidden, n->out_dim, 16, true) <= 160 * 1024;
}

void fill_common(FusedArgs &g, const ssac_mlp *nets, const int32_t *ids, const float *X, int64_t ldx,
                 int64_t sX, int n_rows) {
    g.params = nets->params; g.net_stride = nets->net_stride;
    g.in_dim = nets->in_dim; g.hidden = nets->hidden; g.out_dim = nets->out_dim;
    ssac_mlp_layout(nets->in_dim, nets->hidden, nets->out_dim, g.off);
    g.ids = ids; g.X = X; g.ldx = ldx; g.sX = sX; g.n_rows = n_rows;
    g.dbg = g_fused_dbg; g.tl = g_ssac_timeline; g.xcd = g_ssac_xcd & 1;
}

template <int MODE, int TMR, bool DBUF>
int launch_fused_t(const FusedArgs &g, int n_sel, hipStream_t st) {
    static bool attr_set = false;
    const size_t lds = fused_lds_bytes(g.in_dim, g.hidden, g.out_dim, TMR, DBUF);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)fused_mlp_kernel<MODE, TMR, DBUF>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return ssac_fail("fused_mlp: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    dim3 grid((g.n_rows + TMR - 1) / TMR, n_sel);
    SSAC_LAUNCH((fused_mlp_kernel<MODE, TMR, DBUF>), grid, dim3(NTHR), lds, st, g);
    return ssac_check_launch("fused_mlp");
}

template <int MODE>
int launch_fused(const FusedArgs &g, int n_sel, hipStream_t st) {
    const TileChoice c = choose_tile(g, n_sel);
    if (c.variant == 1) return launch_fused_t<MODE, 16, false>(g, n_sel, st);
    return c.tm == 16 ? launch_fused_t<MODE, 16, true>(g, n_sel, st) : launch_fused_t<MODE, 32, true>(g, n_sel, st);
}

// ------------------------------------------------------------------ head weight gradient + Adam
// (body in ssac_head_wgrad.h; the update path normally runs it inside the merged weight-gradient launch)
constexpr int HW_GROUPS = 16;  // row groups per workgroup (1024 threads = 64 columns x 16 groups)
__global__ __launch_bounds__(64 * HW_GROUPS) void head_wgrad_kernel(HeadWgradArgs a) {
    __shared__ float lds[HW_GROUPS * 64 + HW_GROUPS];
    head_wgrad_body<HW_GROUPS>(a, lds, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------ log finalisation (1 WG; body in ssac_critic_logs.h)
__global__ __launch_bounds__(256) void critic_logs_kernel(CriticLogsArgs a) {
    __shared__ float red[12];
    critic_logs_body(a, red);
}

}  // namespace

#ifdef SSAC_LAB   // (ssac_hip_test.h, lab hooks: the product library does not define the symbol)
extern "C" int ssac_fused_debug_stamps(long long *dev_buf) {
    g_fused_dbg = dev_buf;
    return 0;
}
#endif
extern "C" int ssac_fused_supported(const ssac_mlp *nets) {
    return fused_dbuf_ok(nets) ? 1 : (fused_ok(nets) ? 2 : 0);
}

extern "C" int ssac_mlp3_fwd_fused(const ssac_mlp *nets, const int32_t *net_ids, int n_sel, const float *X,
                                   int64_t ldx, int64_t x_net_stride, int n_rows, float *H1, float *H2,
                                   float *Y, void *stream) {
    if (!fused_ok(nets)) return ssac_fail("ssac_mlp3_fwd_fused: shape not supported by the fused path");
    if (n_sel < 0 || n_sel > SSAC_MAX_NETS) return ssac_fail("ssac_mlp3_fwd_fused: n_sel out of range");
    if (n_sel == 0 || n_rows <= 0) return 0;
    FusedArgs g{};
    fill_common(g, nets, net_ids, X, ldx, x_net_stride, n_rows);
    g.H1 = H1; g.H2 = H2; g.Y = Y;
    return launch_fused<MODE_PLAIN>(g, n_sel, (hipStream_t)stream);
}

extern "C" int ssac_actor_sample_fused(const ssac_mlp *actor, const float *X, int64_t ldx, int n_rows,
                                       const float *eps, float log_std_lo, float log_std_hi,
                                       float *act_dst, int64_t ld_act, int64_t act_col0, float *logp,
                                       float *H1, float *H2, float *out, const ssac_rng *rng, void *stream) {
    if (!eps && !rng) return ssac_fail("ssac_actor_sample_fused: neither eps nor an rng stream given");
    if (!fused_ok(actor) || (actor->out_dim & 1))
        return ssac_fail("ssac_actor_sample_fused: shape not supported by the fused path");
    if (n_rows <= 0) return 0;
    FusedArgs g{};
    fill_common(g, actor, nullptr, X, ldx, 0, n_rows);
    g.H1 = H1; g.H2 = H2; g.Y = out;
    g.eps = eps; g.lo = log_std_lo; g.hi = log_std_hi;
    if (rng) g.rng = RngArgs{rng->seed, rng->counter, rng->offset};
    g.act_dst = act_dst; g.ld_act = ld_act; g.act_col0 = act_col0; g.logp = logp;
    return launch_fused<MODE_SAMPLE>(g, 1, (hipStream_t)stream);
}

// ssac_actor_sample_fused that also copies the state columns next to the sampled action: act_dst rows become the
// critics' input [s | a_theta] (mlps.py:124) without a separate copy launch
extern "C" int ssac_actor_sample_concat_fused(const ssac_mlp *actor, const float *X, int64_t ldx, int n_rows,
                                              const float *eps, float log_std_lo, float log_std_hi, float *xsa,
                                              int64_t ld_xsa, float *logp, float *H1, float *H2, float *out,
                                              const ssac_rng *rng, void *stream) {
    if (!eps && !rng) return ssac_fail("ssac_actor_sample_concat_fused: neither eps nor an rng stream given");
    if (!fused_ok(actor) || (actor->out_dim & 1)) return ssac_fail("ssac_actor_sample_concat_fused: shape not supported");
    if (!xsa || ld_xsa < actor->in_dim + actor->out_dim / 2) return ssac_fail("ssac_actor_sample_concat_fused: bad output");
    if (n_rows <= 0) return 0;
    FusedArgs g{};
    fill_common(g, actor, nullptr, X, ldx, 0, n_rows);
    g.H1 = H1; g.H2 = H2; g.Y = out;
    g.eps = eps; g.lo = log_std_lo; g.hi = log_std_hi;
    if (rng) g.rng = RngArgs{rng->seed, rng->counter, rng->offset};
    g.act_dst = xsa; g.ld_act = ld_xsa; g.act_col0 = actor->in_dim; g.logp = logp; g.copy_x = 1;
    return launch_fused<MODE_SAMPLE>(g, 1, (hipStream_t)stream);
}

extern "C" int ssac_target_fwd_critic_bwdu(const ssac_mlp *targets, const int32_t *net_ids, int n_sel, const float *X1,
                                           int64_t ldx1, int n_rows, float *Qt, const ssac_mlp *critics,
                                           const float *H1, const float *H2, const float *act, int64_t ld_act,
                                           float *DZ2u, float *DZ1u, void *stream) {
    if (!fused_dbuf_ok(targets) || !fused_dbuf_ok(critics))
        return ssac_fail("ssac_target_fwd_critic_bwdu: shape not supported by the merged launch");
    if (n_sel <= 0 || n_sel > SSAC_MAX_NETS) return ssac_fail("ssac_target_fwd_critic_bwdu: n_sel out of range");
    if (!H1 || !H2 || !DZ2u || !DZ1u || !Qt) return ssac_fail("ssac_target_fwd_critic_bwdu: missing buffer");
    if (critics->out_dim > 1 && !act) return ssac_fail("ssac_target_fwd_critic_bwdu: discrete needs actions");
    if (n_rows <= 0) return 0;
    FusedArgs gt{}, gc{};
    fill_common(gt, targets, net_ids, X1, ldx1, 0, n_rows);
    gt.Y = Qt;
    fill_common(gc, critics, nullptr, nullptr, 0, 0, n_rows);
    gc.H1 = const_cast<float *>(H1); gc.H2 = const_cast<float *>(H2);
    gc.act = act; gc.ld_a = ld_act; gc.DZ2 = DZ2u; gc.DZ1 = DZ1u;
    const int tt = choose_tile(gt, n_sel).tm, tc = choose_tile(gc, critics->n_nets).tm;
    const int tgx = (n_rows + tt - 1) / tt, cgx = (n_rows + tc - 1) / tc;
    size_t lds = fused_lds_bytes(targets->in_dim, targets->hidden, targets->out_dim, tt, true);
    const size_t lc = fused_lds_bytes(critics->in_dim, critics->hidden, critics->out_dim, tc, true);
    if (lc > lds) lds = lc;
    static bool attr_set = false;
    if (!attr_set) {
        const void *ks[4] = {(const void *)fused_dual2_kernel<16, 16>, (const void *)fused_dual2_kernel<16, 32>,
                             (const void *)fused_dual2_kernel<32, 16>, (const void *)fused_dual2_kernel<32, 32>};
        for (const void *k : ks)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return ssac_fail("fused_dual2: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    const int tiles_t = tgx * n_sel;
    const dim3 grid(tiles_t + cgx * critics->n_nets);
    hipStream_t st = (hipStream_t)stream;
    if (tt == 16 && tc == 16) SSAC_LAUNCH((fused_dual2_kernel<16, 16>), grid, dim3(NTHR), lds, st, gt, gc, tiles_t, tgx, cgx);
    else if (tt == 16) SSAC_LAUNCH((fused_dual2_kernel<16, 32>), grid, dim3(NTHR), lds, st, gt, gc, tiles_t, tgx, cgx);
    else if (tc == 16) SSAC_LAUNCH((fused_dual2_kernel<32, 16>), grid, dim3(NTHR), lds, st, gt, gc, tiles_t, tgx, cgx);
    else SSAC_LAUNCH((fused_dual2_kernel<32, 32>), grid, dim3(NTHR), lds, st, gt, gc, tiles_t, tgx, cgx);
    return ssac_check_launch("fused_dual2");
}

extern "C" int ssac_actor_sample_critic_fwd(const ssac_mlp *actor, const float *Xa, int64_t ldxa, int n_rows,
                                            const float *eps, float log_std_lo, float log_std_hi, float *act_dst,
                                            int64_t ld_act, int64_t act_col0, float *logp, const ssac_rng *rng,
                                            const ssac_mlp *critics, const float *Xc, int64_t ldxc, float *H1,
                                            float *H2, float *Q, const ssac_gather *gather, void *stream) {
    if (!eps && !rng) return ssac_fail("ssac_actor_sample_critic_fwd: neither eps nor an rng stream given");
    if (!fused_ok(actor) || (actor->out_dim & 1) || !fused_dbuf_ok(critics))
        return ssac_fail("ssac_actor_sample_critic_fwd: shape not supported by the merged launch");
    if (!H1 || !H2 || !Q) return ssac_fail("ssac_actor_sample_critic_fwd: H1 / H2 / Q missing");
    if (n_rows <= 0) return 0;
    FusedArgs ga{}, gc{};
    fill_common(ga, actor, nullptr, Xa, ldxa, 0, n_rows);
    ga.eps = eps; ga.lo = log_std_lo; ga.hi = log_std_hi;
    if (rng) ga.rng = RngArgs{rng->seed, rng->counter, rng->offset};
    ga.act_dst = act_dst; ga.ld_act = ld_act; ga.act_col0 = act_col0; ga.logp = logp;
    fill_common(gc, critics, nullptr, Xc, ldxc, 0, n_rows);
    gc.H1 = H1; gc.H2 = H2; gc.Y = Q;
    if (gather) {
        if (gather->s_elems != actor->in_dim || gather->s_elems + gather->a_elems != critics->in_dim)
            return ssac_fail("ssac_actor_sample_critic_fwd: gather sizes do not match the networks");
        if (!gather->s || !gather->s1 || !gather->act || !gather->rew || !gather->done || !gather->xsa ||
            !gather->x1sa || !gather->rew_out || !gather->done_out || (!gather->idx && !gather->feed))
            return ssac_fail("ssac_actor_sample_critic_fwd: incomplete ssac_gather");
        if (gather->feed && gather->n_logs > NTHR) return ssac_fail("ssac_actor_sample_critic_fwd: log block too large");
        ga.gth = *gather; ga.gth_role = 1;
        gc.gth = *gather; gc.gth_role = 2;
    } else if (!Xa || !Xc) {
        return ssac_fail("ssac_actor_sample_critic_fwd: Xa / Xc missing");
    }
    const int tc = choose_tile(gc, critics->n_nets).tm;  // what a stand-alone forward of the critics would use
    const int tiles_a = (n_rows + 15) / 16, cgx = (n_rows + tc - 1) / tc;
    const bool adbuf = fused_dbuf_ok(actor);
    size_t lds = fused_lds_bytes(actor->in_dim, actor->hidden, actor->out_dim, 16, adbuf);
    const size_t lc = fused_lds_bytes(critics->in_dim, critics->hidden, critics->out_dim, tc, true);
    if (lc > lds) lds = lc;
    static bool attr_set = false;
    if (!attr_set) {
        const void *ks[4] = {(const void *)fused_dual_kernel<16, true>, (const void *)fused_dual_kernel<32, true>,
                             (const void *)fused_dual_kernel<16, false>, (const void *)fused_dual_kernel<32, false>};
        for (const void *k : ks)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return ssac_fail("fused_dual: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    const dim3 grid(tiles_a + cgx * critics->n_nets);
    hipStream_t st = (hipStream_t)stream;
    if (tc == 16 && adbuf) SSAC_LAUNCH((fused_dual_kernel<16, true>), grid, dim3(NTHR), lds, st, ga, gc, tiles_a, cgx);
    else if (adbuf) SSAC_LAUNCH((fused_dual_kernel<32, true>), grid, dim3(NTHR), lds, st, ga, gc, tiles_a, cgx);
    else if (tc == 16) SSAC_LAUNCH((fused_dual_kernel<16, false>), grid, dim3(NTHR), lds, st, ga, gc, tiles_a, cgx);
    else SSAC_LAUNCH((fused_dual_kernel<32, false>), grid, dim3(NTHR), lds, st, ga, gc, tiles_a, cgx);
    if (gather && gather->feed)
        for (int a = 0; a < 2; ++a) ssac_record_slot_patch(a, offsetof(FusedArgs, slot_now));   // ga, gc
    return ssac_check_launch("fused_dual");
}

extern "C" int ssac_chain_update(const ssac_mlp *actor, const float *Xa, int64_t ldxa, int n_rows, const float *eps,
                                 float log_std_lo, float log_std_hi, float *x1sa, int64_t ld_x1, int64_t act_col0,
                                 float *logp, const ssac_rng *rng, const ssac_mlp *targets, const int32_t *net_ids,
                                 int n_sel, float *Qt, const ssac_mlp *critics, const float *Xc, int64_t ldxc,
                                 float *H1, float *H2, float *Q, float *DZ2u, float *DZ1u, float *W3_snapshot,
                                 const ssac_gather *gather, const ssac_deferred_logs *deferred,
                                 unsigned long long *handoff, int target_splits, ssac_xchg *xchg, void *stream) {
    if (!eps && !rng) return ssac_fail("ssac_chain_update: neither eps nor an rng stream given");
    if (target_splits != 1 && target_splits != 2 && target_splits != 4) return ssac_fail("ssac_chain_update: target_splits is 1, 2 or 4");
    if (target_splits > 1 && (!handoff || targets->hidden != 256))
        return ssac_fail("ssac_chain_update: column-split target critics need the hand-off form and hidden 256");
    if (!fused_ok(actor) || (actor->out_dim & 1) || !fused_dbuf_ok(targets) || !fused_dbuf_ok(critics))
        return ssac_fail("ssac_chain_update: shape not supported by the merged launch");
    if (critics->out_dim != 1 || targets->out_dim != 1) return ssac_fail("ssac_chain_update: single-output critics only");
    if (n_sel <= 0 || n_sel > SSAC_MAX_NETS) return ssac_fail("ssac_chain_update: n_sel out of range");
    // (DZ2u may be NULL: dz2u = W3 (.) [h2 > 0] is then not written out -- a weight-gradient launch that is given H2 and
    //  W3 rebuilds it in its operand staging, ssac_mlp_wgrad_all_lossfold with DZ2u == NULL)
    if (!x1sa || !logp || !Qt || !H1 || !H2 || !Q || !DZ1u) return ssac_fail("ssac_chain_update: missing buffer");
    if (!DZ2u && !W3_snapshot) return ssac_fail("ssac_chain_update: DZ2u == NULL needs the W3 snapshot buffer");
    if (act_col0 != actor->in_dim || targets->in_dim != actor->in_dim + actor->out_dim / 2)
        return ssac_fail("ssac_chain_update: [s'|a'] layout does not match the networks");
    if (n_rows <= 0) return 0;
    FusedArgs ga{}, gr{}, gt{}, gc{};
    fill_common(ga, actor, nullptr, Xa, ldxa, 0, n_rows);
    ga.eps = eps; ga.lo = log_std_lo; ga.hi = log_std_hi;
    if (rng) ga.rng = RngArgs{rng->seed, rng->counter, rng->offset};
    ga.act_dst = x1sa; ga.ld_act = ld_x1; ga.act_col0 = act_col0; ga.logp = logp;
    fill_common(gt, targets, net_ids, x1sa, ld_x1, 0, n_rows);
    gt.Y = Qt;
    fill_common(gc, critics, nullptr, Xc, ldxc, 0, n_rows);
    gc.H1 = H1; gc.H2 = H2; gc.Y = Q; gc.DZ2 = DZ2u; gc.DZ1 = DZ1u; gc.W3S = DZ2u ? nullptr : W3_snapshot;
    gc.xcd = ((g_ssac_xcd & 1) || !(g_ssac_xcd & 8)) ? 1 : 0;   // the chained launch: XCD-contiguous halves by default
    if (gather) {
        if (gather->s_elems != actor->in_dim || gather->s_elems + gather->a_elems != critics->in_dim)
            return ssac_fail("ssac_chain_update: gather sizes do not match the networks");
        if (!gather->s || !gather->s1 || !gather->act || !gather->rew || !gather->done || !gather->xsa ||
            gather->x1sa != x1sa || !gather->rew_out || !gather->done_out || (!gather->idx && !gather->feed))
            return ssac_fail("ssac_chain_update: incomplete ssac_gather");
        if (gather->feed && gather->n_logs > NTHR) return ssac_fail("ssac_chain_update: log block too large");
        ga.gth = *gather; ga.gth_role = 1;
        gc.gth = *gather; gc.gth_role = 2;
    } else if (!Xa || !Xc) {
        return ssac_fail("ssac_chain_update: Xa / Xc missing");
    }
    gr = ga;
    if (gather) gr.gth_role = 3;
    if (gather && gather->feed && gather->ids_word >= 0) { gt.gth = *gather; gt.gth_role = 4; }
    const int tc = choose_tile(gc, critics->n_nets).tm;
    const int tgx = (n_rows + 15) / 16, cgx = (n_rows + tc - 1) / tc;
    const bool adbuf = fused_dbuf_ok(actor);
    // (producer / consumer form only: an actor whose resident head image leaves no room for the second staging buffer runs
    // double-buffered with the head's rows parked behind fc2 -- W3LATE in fused_mlp_body)
    const int A_ = actor->out_dim / 2;
    const bool pc_form = handoff && A_ <= 32 && targets->hidden * A_ <= NTHR * HANDOFF_MAX_WA;
    const bool alate = pc_form && !adbuf && fused_dbuf_late_ok(actor);
    size_t lds = fused_lds_bytes(actor->in_dim, actor->hidden, actor->out_dim, 16, adbuf || alate, alate);
    const size_t lt = fused_lds_bytes(targets->in_dim, targets->hidden, targets->out_dim, 16, true);
    const size_t lc = fused_lds_bytes(critics->in_dim, critics->hidden, critics->out_dim, tc, true);
    if (lt > lds) lds = lt;
    if (lc > lds) lds = lc;
    if (lds > 160 * 1024) return ssac_fail("ssac_chain_update: LDS carve does not fit");
    static bool attr_set = false;
    if (!attr_set) {
        const void *ks[4] = {(const void *)fused_chain_kernel<16, true>, (const void *)fused_chain_kernel<32, true>,
                             (const void *)fused_chain_kernel<16, false>, (const void *)fused_chain_kernel<32, false>};
        for (const void *k : ks)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return ssac_fail("fused_chain: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    const int tiles_t = tgx * n_sel;
    DeferredLogsArgs dl{};
    const int dl_on = (deferred && deferred->feed) ? 1 : 0;
    if (dl_on)
        dl = DeferredLogsArgs{deferred->partials, deferred->n_nets, deferred->sumsq, deferred->n_ss, deferred->td_stats,
                              deferred->td_off, deferred->n_rows, deferred->denom, deferred->feed};
    hipStream_t st = (hipStream_t)stream;
    // Co-resident form (fused_chain_co_kernel): when 16-row tiles of the three roles make MORE than one workgroup per CU but
    // at most two, and each role's co-resident carve fits half a CU's LDS.  (Launches of <= 256 16-row tiles are one round
    // at one workgroup per CU already and keep the double-buffered K loop.)
    const int n16 = tgx + tgx * n_sel + tgx * critics->n_nets + ((deferred && deferred->feed) ? 1 : 0);
    const size_t co_lds_a = fused_lds_bytes(actor->in_dim, actor->hidden, actor->out_dim, 16, false, false, true);
    const size_t co_lds_t = fused_lds_bytes(targets->in_dim, targets->hidden, targets->out_dim, 16, false, false, true, targets->hidden * A_);
    const size_t co_lds_c = fused_lds_bytes(critics->in_dim, critics->hidden, critics->out_dim, 16, false, false, true);
    const size_t co_lds = co_lds_a > co_lds_t ? (co_lds_a > co_lds_c ? co_lds_a : co_lds_c) : (co_lds_t > co_lds_c ? co_lds_t : co_lds_c);
    const bool co_form = pc_form && g_chain_form == 1 && g_tile_rows == 0 && target_splits == 1 && n16 > 256 && n16 <= 512 &&
                         co_lds <= 80 * 1024;
    // RESIDENCY INVARIANT of the hand-off launches (any B x N, any number of rounds of workgroups): a consumer spins on granules
    // that only a PRODUCER writes; producers never wait for anybody; every producer holds a LOWER workgroup id than every
    // consumer (fused_chain_pc_kernel: producers [0, tiles_a), then critic tiles / consumers), and the dispatcher starts
    // workgroups in id order.  So when a consumer is resident and spinning, every producer has already been started: it is
    // running or done, never queued behind the spinner -- no occupancy, however small, can deadlock the launch (and a
    // producer that dies leaves a bounded spin + NaN, handoff_poll).  The co-resident form orders its ids differently and
    // therefore insists that the WHOLE launch is resident at once (n16 <= 512 at two workgroups per CU, checked below).
    // tests/test_hip_kernels.py::test_chain_launch_equals_the_separate_launches runs B 4096 / N 16 (eleven rounds).
    if (pc_form) {
        // producer / consumer form (fused_chain_pc_kernel): the actor ONCE per tile, a' handed to the tile's target critics
        static unsigned launch_no = 0;   // tags of eager launches: bit 31 set, so they never meet a recorded launch's
        Handoff ho{handoff, nullptr, 0u, actor->in_dim, A_, target_splits};
        // (a RECORDED launch re-issues these very argument bytes: its tag must come from device memory -- the update
        // counter of the input ring, reached through the gather or the deferred-log struct; the caller passes no hand-off
        // buffer for a recording that has neither)
        const ssac_feed *fd = (gather && gather->feed) ? gather->feed : ((deferred && deferred->feed) ? deferred->feed : nullptr);
        if (fd) { ho.tick = reinterpret_cast<const long long *>(&fd->tick); ho.base = 1u; }
        else ho.base = 0x80000000u | (++launch_no & 0x7fffffffu);
        ga.ho = ho;
        gt.ho = ho;
        if (gather) { gt.gth = *gather; gt.gth_role = 5; }
        static bool pc_attr = false;
        if (!pc_attr) {
            const void *ks[6] = {(const void *)fused_chain_pc_kernel<16, true>, (const void *)fused_chain_pc_kernel<32, true>,
                                 (const void *)fused_chain_pc_kernel<16, false>, (const void *)fused_chain_pc_kernel<32, false>,
                                 (const void *)fused_chain_pc_kernel<16, true, true>, (const void *)fused_chain_pc_kernel<32, true, true>};
            for (const void *k : ks)
                if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                    return ssac_fail("fused_chain_pc: cannot raise the dynamic LDS limit");
            pc_attr = true;
        }
        const int tiles_c = tiles_t * target_splits;   // consumers: one per (slot, column split, tile)
        // critic-sharded rank: the exchange of the subset's target Q as a tail workgroup of THIS launch (xchg != NULL)
        XchgArgs xa{};
        int xchg_on = 0;
        if (xchg) {
            if (co_form) return ssac_fail("ssac_chain_update: the in-launch exchange rides in the one-workgroup-per-CU form");
            if (ssac_xchg_fill_args(xchg, Qt, n_sel * n_rows, 0, net_ids, n_sel, target_splits, &xa)) return 1;
            if (!net_ids && !(gather && gather->feed && gather->ids_word >= 0))
                return ssac_fail("ssac_chain_update: the in-launch exchange needs the update's id block");
            gt.xarrive = xa.arrive;
            xchg_on = 1;
        }
        if (co_form) {
            static bool co_attr = false;
            if (!co_attr) {
                if (hipFuncSetAttribute((const void *)fused_chain_co_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
                    return ssac_fail("fused_chain_co: cannot raise the dynamic LDS limit");
                co_attr = true;
            }
            const dim3 grid_co(tgx + tiles_t + tgx * critics->n_nets + dl_on);
#if defined(SSAC_LAB) && defined(SSAC_EXP_NO_STORE)
            gc.H1 = gc.H2 = gc.DZ1 = gc.DZ2 = gc.Y = gc.W3S = nullptr;   // (experiment build: the critic tiles store nothing)
#endif
            SSAC_LAUNCH(fused_chain_co_kernel, grid_co, dim3(NTHR), co_lds, st, ga, gt, gc, tgx, tiles_t, tgx, tgx, dl, dl_on);
            if (gather && gather->feed)
                for (int a = 0; a < 3; ++a) ssac_record_slot_patch(a, offsetof(FusedArgs, slot_now));   // ga, gt, gc
            return ssac_check_launch("fused_chain_co");
        }
        const dim3 grid_pc(tgx + tiles_c + cgx * critics->n_nets + dl_on + xchg_on);
        if (alate && tc == 16) SSAC_LAUNCH((fused_chain_pc_kernel<16, true, true>), grid_pc, dim3(NTHR), lds, st, ga, gt, gc, tgx, tiles_c, tgx, cgx, dl, dl_on, xa, xchg_on);
        else if (alate) SSAC_LAUNCH((fused_chain_pc_kernel<32, true, true>), grid_pc, dim3(NTHR), lds, st, ga, gt, gc, tgx, tiles_c, tgx, cgx, dl, dl_on, xa, xchg_on);
        else if (tc == 16 && adbuf) SSAC_LAUNCH((fused_chain_pc_kernel<16, true>), grid_pc, dim3(NTHR), lds, st, ga, gt, gc, tgx, tiles_c, tgx, cgx, dl, dl_on, xa, xchg_on);
        else if (adbuf) SSAC_LAUNCH((fused_chain_pc_kernel<32, true>), grid_pc, dim3(NTHR), lds, st, ga, gt, gc, tgx, tiles_c, tgx, cgx, dl, dl_on, xa, xchg_on);
        else if (tc == 16) SSAC_LAUNCH((fused_chain_pc_kernel<16, false>), grid_pc, dim3(NTHR), lds, st, ga, gt, gc, tgx, tiles_c, tgx, cgx, dl, dl_on, xa, xchg_on);
        else SSAC_LAUNCH((fused_chain_pc_kernel<32, false>), grid_pc, dim3(NTHR), lds, st, ga, gt, gc, tgx, tiles_c, tgx, cgx, dl, dl_on, xa, xchg_on);
        if (gather && gather->feed)
            for (int a = 0; a < 3; ++a) ssac_record_slot_patch(a, offsetof(FusedArgs, slot_now));   // ga, gt, gc
        return ssac_check_launch("fused_chain_pc");
    }
    if (target_splits != 1) return ssac_fail("ssac_chain_update: column-split target critics need the hand-off form");
    if (xchg) return ssac_fail("ssac_chain_update: the in-launch exchange needs the hand-off (producer / consumer) form");
    const dim3 grid(tiles_t + cgx * critics->n_nets + dl_on);
    if (tc == 16 && adbuf) SSAC_LAUNCH((fused_chain_kernel<16, true>), grid, dim3(NTHR), lds, st, ga, gr, gt, gc, tiles_t, tgx, cgx, dl, dl_on);
    else if (adbuf) SSAC_LAUNCH((fused_chain_kernel<32, true>), grid, dim3(NTHR), lds, st, ga, gr, gt, gc, tiles_t, tgx, cgx, dl, dl_on);
    else if (tc == 16) SSAC_LAUNCH((fused_chain_kernel<16, false>), grid, dim3(NTHR), lds, st, ga, gr, gt, gc, tiles_t, tgx, cgx, dl, dl_on);
    else SSAC_LAUNCH((fused_chain_kernel<32, false>), grid, dim3(NTHR), lds, st, ga, gr, gt, gc, tiles_t, tgx, cgx, dl, dl_on);
    if (gather && gather->feed)
        for (int a = 0; a < 4; ++a) ssac_record_slot_patch(a, offsetof(FusedArgs, slot_now));   // ga, gr, gt, gc
    return ssac_check_launch("fused_chain");
}

// How many column splits of the target critics ssac_chain_update should be given for this shape: the largest of 4, 2, 1
// for which the whole launch -- one actor workgroup per tile, n_sel x splits consumers per tile, the online critics' tiles,
// the log workgroup -- is still ONE resident round of workgroups (256 CUs, one 512-thread workgroup each).
extern "C" int ssac_chain_target_splits(const ssac_mlp *actor, const ssac_mlp *targets, const ssac_mlp *critics, int n_rows,
                                        int n_sel) {
    if (!actor || !targets || !critics || n_rows <= 0 || n_sel <= 0) return 1;
    if (targets->hidden != 256 || targets->out_dim != 1 || actor->out_dim / 2 > 32) return 1;
    FusedArgs gc{};
    gc.n_rows = n_rows; gc.in_dim = critics->in_dim; gc.hidden = critics->hidden; gc.out_dim = critics->out_dim;
    const int tc = choose_tile(gc, critics->n_nets).tm;
    const int tgx = (n_rows + 15) / 16, cgx = (n_rows + tc - 1) / tc;
    for (int ns = 4; ns > 1; ns >>= 1)
        if (tgx + n_sel * ns * tgx + cgx * critics->n_nets + 1 <= 256) return ns;
    return 1;
}

extern "C" int ssac_critic_fwd_bwd_fused(const ssac_mlp *nets, const float *X, int64_t ldx, int n_rows,
                                         const float *td, const float *weight, const float *act,
                                         int64_t ld_act, const ssac_popart *popart, int pop, float denom,
                                         float *H1, float *H2, float *Q, float *DQ, float *DZ2, float *DZ1,
                                         float *partials, const ssac_td_spec *lazy_td, void *stream) {
    if (!fused_ok(nets)) return ssac_fail("ssac_critic_fwd_bwd_fused: shape not supported by the fused path");
    if (!H1 || !H2 || !DQ || !DZ2 || !DZ1 || !partials)
        return ssac_fail("ssac_critic_fwd_bwd_fused: missing output buffer");
    if (nets->out_dim > 1 && !act) return ssac_fail("ssac_critic_fwd_bwd_fused: discrete needs actions");
    if (n_rows <= 0) return 0;
    FusedArgs g{};
    fill_common(g, nets, nullptr, X, ldx, 0, n_rows);
    g.H1 = H1; g.H2 = H2; g.Y = Q;
    g.td = td; g.weight = weight; g.act = act; g.ld_a = ld_act; g.popart = popart; g.pop = pop;
    g.denom = denom; g.DQ = DQ; g.DZ2 = DZ2; g.DZ1 = DZ1; g.partials = partials;
    if (lazy_td) g.tds = *lazy_td;
    if (!td && !lazy_td) return ssac_fail("ssac_critic_fwd_bwd_fused: no TD target given");
    return launch_fused<MODE_CRITIC>(g, nets->n_nets, (hipStream_t)stream);
}

extern "C" int ssac_critic_bwd_fused(const ssac_mlp *nets, int n_rows, const float *td, const float *weight,
                                     const float *act, int64_t ld_act, const ssac_popart *popart, int pop,
                                     float denom, const float *H1, const float *H2, const float *Q, float *DQ,
                                     float *DZ2, float *DZ1, float *partials, const ssac_td_spec *lazy_td,
                                     void *stream) {
    if (!fused_ok(nets)) return ssac_fail("ssac_critic_bwd_fused: shape not supported by the fused path");
    if (nets->out_dim > 1 && !act) return ssac_fail("ssac_critic_bwd_fused: discrete needs actions");
    if (!H1 || !H2 || !Q) return ssac_fail("ssac_critic_bwd_fused: needs the saved forward (H1, H2, Q)");
    if (n_rows <= 0) return 0;
    FusedArgs g{};
    fill_common(g, nets, nullptr, nullptr, 0, 0, n_rows);
    g.H1 = const_cast<float *>(H1); g.H2 = const_cast<float *>(H2); g.Y = const_cast<float *>(Q);
    g.td = td; g.weight = weight; g.act = act; g.ld_a = ld_act; g.popart = popart; g.pop = pop;
    g.denom = denom; g.DQ = DQ; g.DZ2 = DZ2; g.DZ1 = DZ1; g.partials = partials;
    if (lazy_td) g.tds = *lazy_td;
    if (!td && !lazy_td) return ssac_fail("ssac_critic_bwd_fused: no TD target given");
    return launch_fused<MODE_CRITIC_BWD>(g, nets->n_nets, (hipStream_t)stream);
}

// ---- the online actor update's two fused launches (learning.py:344-421; csrc notes at MODE_ACTOR_BWD / DXU)
extern "C" int ssac_critic_fwd_dx_fused(const ssac_mlp *nets, const float *X, int64_t ldx, int n_rows, int dx_col0,
                                        int dx_cols, float *Q, float *DXu, void *stream) {
    if (!fused_dbuf_ok(nets) || nets->out_dim != 1)
        return ssac_fail("ssac_critic_fwd_dx_fused: single-output critics on the fused path only");
    if (!X || !Q || !DXu || dx_col0 < 0 || dx_cols <= 0 || dx_col0 + dx_cols > nets->in_dim)
        return ssac_fail("ssac_critic_fwd_dx_fused: bad arguments");
    if (n_rows <= 0) return 0;
    FusedArgs g{};
    fill_common(g, nets, nullptr, X, ldx, 0, n_rows);
    g.Y = Q; g.DXU = DXu; g.dx_col0 = dx_col0; g.dx_cols = dx_cols;
    return launch_fused<MODE_CRITIC_U>(g, nets->n_nets, (hipStream_t)stream);
}

extern "C" int ssac_actor_bwd_fused(const ssac_mlp *actor, const float *H1, const float *H2, int n_rows, const float *Qc,
                                    int n_critics, const float *DXu, const float *aout, const float *eps,
                                    const float *logp, const float *log_alpha, int use_entropy, float log_std_lo,
                                    float log_std_hi, float inv_members, const ssac_popart *popart, int pop,
                                    float *d_out, float *DZ2, float *DZ1, float *partials, void *stream) {
    if (!fused_ok(actor) || (actor->out_dim & 1)) return ssac_fail("ssac_actor_bwd_fused: shape not supported by the fused path");
    if (!H1 || !H2 || !Qc || !DXu || !aout || !eps || !log_alpha || !d_out || !DZ2 || !DZ1 || !partials || n_critics < 1 ||
        (use_entropy && !logp))
        return ssac_fail("ssac_actor_bwd_fused: missing argument");
    if (n_rows <= 0) return 0;
    FusedArgs g{};
    fill_common(g, actor, nullptr, nullptr, 0, 0, n_rows);
    g.H1 = const_cast<float *>(H1); g.H2 = const_cast<float *>(H2);
    g.popart = popart; g.pop = pop; g.DQ = d_out; g.DZ2 = DZ2; g.DZ1 = DZ1;
    g.ab = ActorBwdArgs{Qc, n_critics, DXu, aout, eps, logp, log_alpha, use_entropy, log_std_lo, log_std_hi, inv_members,
                        partials};
    return launch_fused<MODE_ACTOR_BWD>(g, 1, (hipStream_t)stream);
}

// ---- the chained form of the three launches above (fused_actor_chain_kernel)
extern "C" int ssac_actor_chain_fused(const ssac_mlp *actor, const float *X, int64_t ldx, int n_rows, const float *eps,
                                      const ssac_rng *rng, float log_std_lo, float log_std_hi, float *xsa, int64_t ld_xsa,
                                      float *logp, float *H1, float *H2, float *out, const ssac_mlp *critics, float *Q,
                                      float *DXu, const float *log_alpha, int use_entropy, float inv_members,
                                      const ssac_popart *popart, int pop, float *d_out, float *DZ2, float *DZ1,
                                      float *partials, unsigned long long *handoff, long long update_no,
                                      float *begin_logs, int n_logs, ssac_adam_ctl *begin_ctl, void *stream) {
    if (!eps && !rng) return ssac_fail("ssac_actor_chain_fused: neither eps nor an rng stream given");
    if (!fused_ok(actor) || (actor->out_dim & 1) || !fused_dbuf_ok(critics) || critics->out_dim != 1)
        return ssac_fail("ssac_actor_chain_fused: shape not supported by the chained launch");
    const bool adbuf = fused_dbuf_ok(actor);   // (false: the actor's passes stream their weights through ONE staging buffer)
    const int A = actor->out_dim / 2, S = actor->in_dim;
    if (critics->in_dim != S + A || A > 32 || critics->hidden * A > NTHR * HANDOFF_MAX_WA)
        return ssac_fail("ssac_actor_chain_fused: critic input is not [s | a] / too many action columns");
    if (!X || !xsa || ld_xsa < S + A || !H1 || !H2 || !out || !Q || !DXu || !log_alpha || !d_out || !DZ2 || !DZ1 ||
        !partials || !handoff || (use_entropy && !logp) || (begin_logs && n_logs > NTHR))
        return ssac_fail("ssac_actor_chain_fused: missing argument");
    if (n_rows <= 0) return 0;
    const int N = critics->n_nets;
    FusedArgs ga{}, gb{}, gc{};
    fill_common(ga, actor, nullptr, X, ldx, 0, n_rows);
    ga.H1 = H1; ga.H2 = H2; ga.Y = out;
    ga.eps = eps; ga.lo = log_std_lo; ga.hi = log_std_hi;
    if (rng) ga.rng = RngArgs{rng->seed, rng->counter, rng->offset};
    ga.act_dst = xsa; ga.ld_act = ld_xsa; ga.act_col0 = S; ga.logp = logp; ga.copy_x = 1;
    ga.begin_logs = begin_logs; ga.begin_n = begin_logs ? n_logs : 0; ga.begin_ctl = begin_ctl;
    // the launch's tag: 1 + update_no (a caller-numbered update: every RECORDED launch is -- its replays are renumbered
    // through ssac_replay_value, which rewrites the tag and the noise draw in the recorded argument bytes), or a host counter
    // with bit 31 set (update_no < 0: an eager launch nobody numbers)
    static unsigned launch_no = 0;
    if (update_no < 0 && g_ssac_recording) return ssac_fail("ssac_actor_chain_fused: a recorded launch needs an update number");
    Handoff ho{handoff, nullptr, update_no >= 0 ? (unsigned)((update_no + 1) & 0x7fffffff) : (0x80000000u | (++launch_no & 0x7fffffffu)),
               S, A, 1, handoff + (int64_t)n_rows * A, handoff + (int64_t)n_rows * A + (int64_t)N * n_rows};
    ga.ho = ho;
    fill_common(gb, actor, nullptr, nullptr, 0, 0, n_rows);
    gb.H1 = H1; gb.H2 = H2;
    gb.popart = popart; gb.pop = pop; gb.DQ = d_out; gb.DZ2 = DZ2; gb.DZ1 = DZ1;
    gb.ab = ActorBwdArgs{Q, N, DXu, out, eps, logp, log_alpha, use_entropy, log_std_lo, log_std_hi, inv_members, partials};
    gb.rng = ga.rng;
    gb.ho = ho;
    fill_common(gc, critics, nullptr, X, ldx, 0, n_rows);   // (state columns from the actor's input; a arrives by hand-off)
    gc.Y = Q; gc.DXU = DXu; gc.dx_col0 = S; gc.dx_cols = A;
    gc.ho = ho;
    gc.xcd = 1;
    const int tc = choose_tile(gc, N).tm;
    const int tiles_a = (n_rows + 15) / 16, cgx = (n_rows + tc - 1) / tc;
    // The actor workgroups WAIT for critic tiles that are dispatched behind them: they must never hold every slot of the
    // chip (one workgroup per CU at this LDS carve), or the critics they wait for could not start.  Half the CUs at most.
    if (tiles_a > SSAC_ACTOR_CHAIN_MAX_ROWS / 16)
        return ssac_fail("ssac_actor_chain_fused: more than SSAC_ACTOR_CHAIN_MAX_ROWS batch rows (use the three launches)");
    size_t lds = fused_lds_bytes(actor->in_dim, actor->hidden, actor->out_dim, 16, adbuf);
    const size_t lc = fused_lds_bytes(critics->in_dim, critics->hidden, critics->out_dim, tc, true);
    if (lc > lds) lds = lc;
    if (lds > 160 * 1024) return ssac_fail("ssac_actor_chain_fused: LDS carve does not fit");
    static bool attr_set = false;
    if (!attr_set) {
        const void *ks[4] = {(const void *)fused_actor_chain_kernel<16>, (const void *)fused_actor_chain_kernel<32>,
                             (const void *)fused_actor_chain_kernel<16, false>, (const void *)fused_actor_chain_kernel<32, false>};
        for (const void *k : ks)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return ssac_fail("fused_actor_chain: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    const dim3 grid(tiles_a + cgx * N);
    hipStream_t st = (hipStream_t)stream;
    if (tc == 16 && adbuf) SSAC_LAUNCH((fused_actor_chain_kernel<16>), grid, dim3(NTHR), lds, st, ga, gb, gc, tiles_a, cgx);
    else if (adbuf) SSAC_LAUNCH((fused_actor_chain_kernel<32>), grid, dim3(NTHR), lds, st, ga, gb, gc, tiles_a, cgx);
    else if (tc == 16) SSAC_LAUNCH((fused_actor_chain_kernel<16, false>), grid, dim3(NTHR), lds, st, ga, gb, gc, tiles_a, cgx);
    else SSAC_LAUNCH((fused_actor_chain_kernel<32, false>), grid, dim3(NTHR), lds, st, ga, gb, gc, tiles_a, cgx);
    if (update_no >= 0) {   // (recorded: a replay's number replaces update_no in the tag and in the noise draw)
        for (int a = 0; a < 3; ++a) ssac_record_value_patch(a, offsetof(FusedArgs, ho) + offsetof(Handoff, base), 0, 1);
        if (rng && !rng->counter)
            for (int a = 0; a < 2; ++a)
                ssac_record_value_patch(a, offsetof(FusedArgs, rng) + offsetof(RngArgs, offset), 1, rng->offset - update_no);
    }
    return ssac_check_launch("fused_actor_chain");
}

// words of the `handoff` buffer of ssac_actor_chain_fused: a (n_rows x A), Q (n_critics x n_rows), dQ/da (n_critics x n_rows x A)
extern "C" int64_t ssac_actor_chain_handoff_words(int n_rows, int n_critics, int action_dim) {
    return (int64_t)n_rows * action_dim + (int64_t)n_critics * n_rows * (1 + action_dim);
}

// row tiles the fused critic launch will use for (n_rows, n_nets): sizes the `partials` buffer
extern "C" int ssac_fused_row_tiles(const ssac_mlp *nets, int n_rows, int n_nets) {
    FusedArgs g{};
    g.n_rows = n_rows; g.in_dim = nets->in_dim; g.hidden = nets->hidden; g.out_dim = nets->out_dim;
    const int tm = choose_tile(g, n_nets).tm;
    return (n_rows + tm - 1) / tm;
}

extern "C" int ssac_chain_form(int form) {
    if (form != 0 && form != 1 && form != -1)
        return ssac_fail("ssac_chain_form: 0 (one workgroup per CU), 1 (co-resident 16-row tiles where they apply), -1 (the library's default)");
    g_chain_form = form < 0 ? CHAIN_FORM_DEFAULT : form;
    return 0;
}

extern "C" int ssac_fused_tile_rows(int rows) {
    if (rows != 0 && rows != 16 && rows != 17 && rows != 32)
        return ssac_fail("ssac_fused_tile_rows: 0 (auto), 16, 32, 17 (16 rows, single staging buffer)");
    g_tile_rows = rows;
    return 0;
}

extern "C" int ssac_head_wgrad(const ssac_mlp *nets, const int32_t *net_ids, int n_sel, const float *H2,
                               const float *DQ, int n_rows, float *adam_m, float *adam_v,
                               const ssac_adam_ctl *ctl, float *grads, float *sumsq,
                               int64_t sumsq_net_stride, float *target, float tau, void *stream) {
    if (!nets || nets->out_dim > MAX_OUT) return ssac_fail("ssac_head_wgrad: out_dim too large");
    if (!grads && (!adam_m || !adam_v || !ctl)) return ssac_fail("ssac_head_wgrad: Adam state missing");
    if (n_sel <= 0 || n_rows <= 0) return 0;
    int64_t off[6];
    ssac_mlp_layout(nets->in_dim, nets->hidden, nets->out_dim, off);
    dim3 grid((nets->hidden + 63) / 64, n_sel);
    HeadWgradArgs a{nets->params, nets->net_stride, nets->hidden, nets->out_dim, off[4], off[5], net_ids, H2, DQ,
                    n_rows, adam_m, adam_v, ctl, grads, sumsq, sumsq_net_stride, target, tau};
    SSAC_LAUNCH(head_wgrad_kernel, grid, dim3(64 * HW_GROUPS), 0, (hipStream_t)stream, a);
    return ssac_check_launch("head_wgrad");
}

// gradient-norm slots of the head layer per net (ssac_head_wgrad.h: one per 16 columns)
extern "C" int ssac_head_wgrad_tiles(const ssac_mlp *nets) {
    return nets ? (nets->hidden + SSAC_HEAD_SLOT_COLS - 1) / SSAC_HEAD_SLOT_COLS : -1;
}

namespace {
__global__ __launch_bounds__(64) void deferred_logs_kernel(DeferredLogsArgs d, int ring_slot) { deferred_logs_body(d, ring_slot); }
}  // namespace

extern "C" int ssac_deferred_logs_flush(const ssac_deferred_logs *d, int ring_slot, void *stream) {
    if (!d || !d->feed || !d->partials || ring_slot < 0) return ssac_fail("ssac_deferred_logs_flush: bad arguments");
    DeferredLogsArgs dl{d->partials, d->n_nets, d->sumsq, d->n_ss, d->td_stats, d->td_off, d->n_rows, d->denom, d->feed};
    SSAC_LAUNCH(deferred_logs_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dl, ring_slot);
    return ssac_check_launch("deferred_logs");
}

extern "C" int ssac_critic_logs(const float *partials, int n_nets, int tiles, int n_rows, float denom,
                                const float *sumsq, int n_sumsq, const ssac_adam_ctl *scale_by_clip,
                                float *logs, const ssac_td_spec *lazy_td, float *td_logs, ssac_feed *feed,
                                void *stream) {
    CriticLogsArgs a{partials, n_nets, tiles, n_rows, denom, sumsq, n_sumsq, scale_by_clip, logs, {}, td_logs, feed};
    if (lazy_td) a.tds = *lazy_td;
    SSAC_LAUNCH(critic_logs_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    return ssac_check_launch("critic_logs");
}
