// bf16-operand variant of the chained critic update for gfx950 (BASELINE.json config 2: "REDQ N=10 UTD=20 batch 256
// bf16").  The reference computes in fp32 only (super_sac/__init__.py:3), so this mode has no reference counterpart:
// fp32 MASTER weights, Adam moments and Polyak targets stay exactly as in the fp32 path; what changes is the operand
// type of the matrix products -- bf16 in, fp32 accumulate, v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate) -- fed from
// a bf16 SHADOW of every arena that the Adam epilogue keeps current.  Parity is tested against the same reference
// fixtures at a separately stated (bf16) tolerance; the headline benchmark stays fp32.
//
// Data layout (all new buffers bf16):
//   shadow arena per net   [ W1 (H x K1P, K zero-padded to a multiple of 16) | W2 (H x H) | W2^T (H x H) | W3 (out x H) ]
//                          every product is evaluated as D^T = W . act^T: the weight rows are the MFMA's A operand, and
//                          a lane's fragment (8 consecutive k of one row) is ONE 16-byte global load straight from the
//                          shadow -- no LDS staging, no K-loop barrier.  W2^T serves the backward-data product.
//   saved activations      H1T, H2T, DZ2uT, DZ1uT (n_nets x H x Bp) and XT (K1P x Bp): TRANSPOSED, batch contiguous, so
//                          the weight-gradient products (K = batch) also read both operands with 16-byte loads per lane.
//   Bp = batch rounded up to 16; the pad columns are zero (buffers are allocated zeroed and never written there).
#include <hip/hip_runtime.h>
#include <math.h>
#include <algorithm>
#include <type_traits>
#include <stdint.h>

#include "ssac_internal.h"
#include "ssac_philox.h"
#include "ssac_critic_logs.h"
#include "ssac_begin.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TM = 32;      // batch rows per workgroup
constexpr int NTHR = 512;   // 8 waves: wave w owns output features [32w, 32w + 32)
constexpr int LPAD = 8;     // LDS row padding (elements): 16-byte aligned rows, conflict-free ds_read_b128
constexpr int XA_LD = 32 + LPAD;   // row stride of the consumer's a' tile (two K-steps of 16)
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;
constexpr float LOG_2 = 0.69314718055994530942f;


__device__ __forceinline__ unsigned short f2bf(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
// v_exp_f32 / v_log_f32 / v_rcp_f32 forms (bf16 mode's sample epilogue)
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
__device__ __forceinline__ float fast_tanh(float x) {   // 1 - 2 / (e^{2x} + 1); saturates cleanly at +-1 (e^{2x} -> inf / 0)
    const float xc = fminf(fmaxf(x, -15.0f), 15.0f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(fast_exp(2.0f * xc) + 1.0f);
}

struct ShadowGeom { int64_t stride, o1, o2, o2t, o3; int k1p; };

// FRAGMENT-MAJOR shadow matrices (W1, W2, W2^T; W3 stays row-major).  The MFMA B operand of lane (n = lane & 31,
// half = lane >> 5) at K-step t is 8 consecutive k of weight row n; with row-major rows of 512 bytes one wave-level load
// touches 32 cache lines and uses 32 bytes of each -- the other 96 are fetched again by the next three steps unless
// the 16 KB L1 keeps 8 waves x 32 lines around, which it does not (18 loads per lane took ~8 k clocks at kernel start).
// So element (n, k) of a matrix with K = 16 * nsteps columns lives at
//     (((n >> 5) * nsteps + (k >> 4)) * 64 + (n & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7):
// step t of a 32-row block is ONE contiguous KiB, lane l's 16 bytes at l * 16 -- every line fetched once, fully used.
__host__ __device__ __forceinline__ int64_t frag_off(int nsteps, int n, int k) {
    return ((((int64_t)(n >> 5) * nsteps + (k >> 4)) * 64 + (n & 31) + 32 * ((k >> 3) & 1)) << 3) + (k & 7);
}
constexpr int FRAG_STEP = 512;   // elements between consecutive K-steps of a lane's fragment pointer

inline ShadowGeom shadow_geom(int in_dim, int hidden, int out_dim) {
    ShadowGeom g;
    g.k1p = (in_dim + 15) & ~15;
    g.o1 = 0;
    g.o2 = (int64_t)hidden * g.k1p;
    g.o2t = g.o2 + (int64_t)hidden * hidden;
    g.o3 = g.o2t + (int64_t)hidden * hidden;
    g.stride = (g.o3 + (int64_t)out_dim * hidden + 7) & ~(int64_t)7;
    return g;
}

enum { MODE_PLAIN = 0, MODE_SAMPLE = 1, MODE_CRITIC_U = 2 };

struct BfArgs {
    const float *params; int64_t net_stride; int in_dim, hidden, out_dim; int64_t off[6];   // fp32 master (biases)
    const unsigned short *shadow; ShadowGeom sg;
    const int32_t *ids;
    const float *X; int64_t ldx; int n_rows, bp;
    float *Y;                     // (n_sel, n_rows, out) fp32 head outputs, or null
    // MODE_SAMPLE
    const float *eps; float lo, hi; float *act_dst; int64_t ld_act, act_col0; float *logp; RngArgs rng;
    // MODE_CRITIC_U: transposed bf16 saves
    unsigned short *H1T, *H2T, *DZ2T, *DZ1T, *XT;
    ssac_gather gth; int gth_role;  // as in ssac_fused.hip: 1 actor half (+ start-of-update duties), 3 actor half,
                                    // 2 critic half, 4 rows from X with the subset ids read from the input slot
                                    // 5: hand-off consumer (s' rows like the actor half, nothing written, ids from the slot)
    long long *dbg;                 // optional s_memtime phase stamps of tile 0 (ssac_bf16_debug_stamps)
    long long *tl;                  // optional per-workgroup (start, end) stamps of the chained launch (ssac_debug_timeline)
    int xcd;                        // XCD-contiguous workgroup order (ssac_internal.h)
    Handoff ho;                     // bf_chain_pc_kernel: MODE_SAMPLE publishes a', MODE_PLAIN takes its action columns from it
};
constexpr int BF_HO_AMAX = 8;       // action dimensions a hand-off consumer keeps W1's action columns in registers for
#ifdef SSAC_LAB
#define BSTAMP(i) do { if (g.dbg && dbg_off >= 0 && bx == 0 && e == 0 && threadIdx.x == 0) g.dbg[dbg_off + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BSTAMP(i) do { } while (0)
#endif

// ---- one layer as D[b][n] = sum_k act[b][k] W[n][k] on v_mfma_f32_32x32x16_bf16.  A = the tile's 32 activation rows in
// LDS (lane (b = lane & 31, half = lane >> 5) reads 8 consecutive k of row b); B = 32 weight rows (row stride ldw, K
// contiguous): lane (n = lane & 31, half) reads 8 consecutive k of row n -- one 16-byte global load per MFMA straight
// from the shadow.  acc[r] of lane (n, half) = D[(r&3) + 8(r>>2) + 4 half][n]: a lane holds 4 CONSECUTIVE BATCH ROWS of
// one feature per register quad, so the transposed saves (feature-major, batch contiguous) leave as 8-byte stores.
// The weight fragments are PREFETCHED (Frags::load) one phase ahead, so no layer starts with an exposed global round
// trip: the first 16 K-steps (K <= 256: all of them) wait in registers, longer K continue with in-loop loads.
struct Frags {
    u16x8 w[16];
    // wp: this lane's fragment pointer (row n of the wave's 32 rows, + 8 half); steps beyond nsteps re-read the last
    // one (unconditional loads: a load inside a branch would drain the whole vmcnt queue at the join)
    __device__ __forceinline__ void load(const unsigned short *__restrict__ wp, int nsteps, int step = FRAG_STEP) {
#pragma unroll
        for (int t = 0; t < 16; ++t) w[t] = *reinterpret_cast<const u16x8 *>(wp + step * (t < nsteps ? t : nsteps - 1));
    }
};

// TR = false: D[b][n] (activation rows are the MFMA's A operand): acc[r] of lane (n, half) = D[(r&3) + 8(r>>2) + 4 half][n]
// TR = true : D^T[n][b] (weight rows are the A operand):            acc[r] of lane (b, half) = D^T[(r&3) + 8(r>>2) + 4 half][b]
//             -- a lane then holds 4 CONSECUTIVE FEATURES of one batch row per register quad: bias / ReLU / pack and ONE
//             8-byte LDS store per quad into the row-major activation tile (instead of four scattered 2-byte ones), and a
//             single-output head can be taken straight from the accumulators.  Same products, same K order.
template <bool TR = false>
__device__ __forceinline__ void bf_mma(f32x16 &acc, const Frags &f, const unsigned short *__restrict__ wp, int nsteps,
                                       const unsigned short *__restrict__ ap, int step = FRAG_STEP) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        if (t < nsteps) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8 *>(ap + 16 * t));
            const bf16x8 w = __builtin_bit_cast(bf16x8, f.w[t]);
            acc = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, acc, 0, 0, 0)
                     : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w, acc, 0, 0, 0);
        }
    }
    for (int t0 = 16; t0 < nsteps; t0 += 8) {   // K > 256 (wide observations): 8 steps in flight at a time
        u16x8 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = *reinterpret_cast<const u16x8 *>(wp + step * (t0 + u < nsteps ? t0 + u : nsteps - 1));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (t0 + u < nsteps) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8 *>(ap + 16 * (t0 + u)));
                const bf16x8 ww = __builtin_bit_cast(bf16x8, w[u]);
                acc = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(ww, a, acc, 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ww, acc, 0, 0, 0);
            }
        }
    }
}

__device__ __forceinline__ void zero_acc(f32x16 &a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 0.0f;
}

__device__ __forceinline__ bool bf_pos(unsigned short h) { return (h & 0x7fff) != 0 && !(h & 0x8000); }

// 4 consecutive batch rows [b0, b0+4) of feature row `dst` (batch contiguous): one 8-byte store when the whole quad is
// inside the batch, element-wise at a ragged end
__device__ __forceinline__ void store_quad(unsigned short *__restrict__ dst, int b0, int n_rows, u16x4 v) {
    if (b0 + 3 < n_rows) {
        *reinterpret_cast<u16x4 *>(dst + b0) = v;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (b0 + i < n_rows) dst[b0 + i] = v[i];
    }
}

// LDS carve (bytes) of one workgroup
__host__ __device__ inline size_t bf_lds_bytes(int in_dim, int hidden, int out_dim) {
    const int k1p = (in_dim + 15) & ~15;
    const int ldo = (out_dim + 31) & ~31;
    size_t b = 2 * ((size_t)TM * (k1p + LPAD) + 3 * (size_t)TM * (hidden + LPAD));             // xs, h1s, h2s, dz2s (bf16)
    b += 4 * (2 * (size_t)TM * ldo + 3 * (size_t)hidden + ldo + 64);                           // ys, lpt, b1s, b2s, w3f, b3s
    b = (b + 15) & ~(size_t)15;
    b += 2 * (size_t)(NTHR / 64) * TM * XA_LD;   // hand-off consumer: a wave-private [TM][32] tile of a' (bf16) per wave
    return b;
}

// HO: this instantiation may be the CONSUMER of a hand-off (bf_chain_pc_kernel's target-critic workgroups only: the other
// kernels' register budgets -- two sets of 16 weight fragments are live across the prologue -- must not carry its code)
template <int MODE, bool HO = false>
__device__ __forceinline__ void bf_mlp_body(const BfArgs &g, unsigned char *smem, const int bx, const int e,
                                            const int dbg_off = 0) {
    const int H = g.hidden, IN = g.in_dim, OUT = g.out_dim, K1P = g.sg.k1p;
    const int ldo = (OUT + 31) & ~31;
    const int ldx_s = K1P + LPAD, ldh = H + LPAD;
    unsigned short *xs = reinterpret_cast<unsigned short *>(smem);
    unsigned short *h1s = xs + TM * ldx_s;
    unsigned short *h2s = h1s + TM * ldh;
    unsigned short *dz2s = h2s + TM * ldh;                   // MODE_CRITIC_U: dz2u, written by the fc2 epilogue
    float *ys = reinterpret_cast<float *>(dz2s + TM * ldh);  // [TM][ldo]
    float *lpt = ys + TM * ldo;                              // [TM][ldo] scratch of the sample epilogue
    float *b1s = lpt + TM * ldo;                             // [H]
    float *b2s = b1s + H;                                    // [H]
    float *w3f = b2s + H;                                    // [H]   W3 row 0 as fp32 (single-output heads)
    float *b3s = w3f + H;                                    // [ldo]
    // hand-off consumer: wave-private [TM][XA_LD] tiles of a' (the carve's last block)
    unsigned short *xa = reinterpret_cast<unsigned short *>(smem + bf_lds_bytes(IN, H, OUT)) - (NTHR / 64) * TM * XA_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = bx * TM;

    BSTAMP(0);
    // ---- where this tile's rows come from (replay gather folded in; see ssac_gather in include/ssac_hip.h)
    const int64_t *gidx = nullptr;
    const uint32_t *gslot = nullptr;
    const int32_t *idsp = g.ids;
    if (g.gth_role) {
        gidx = g.gth.idx;
        if (g.gth.feed) {
            const ssac_feed f = *g.gth.feed;
            gslot = feed_slot(f);
            gidx = reinterpret_cast<const int64_t *>(gslot);
            if (g.gth_role == 1 && bx == 0) {  // start-of-update duties (ssac_begin_update)
                feed_pull(f);
                if (tid < g.gth.n_logs) g.gth.logs[tid] = 0.0f;
                if (tid == 0 && g.gth.ctl) adam_refresh(g.gth.ctl, g.gth.ctl->step + 1);
            }
            if ((g.gth_role == 4 || g.gth_role == 5) && g.gth.ids_word >= 0) idsp = reinterpret_cast<const int32_t *>(gslot + g.gth.ids_word);
        }
        if (g.gth_role == 4) gidx = nullptr;
    }
    // ---- the replay index of this thread's row is the OLDEST load in flight (16 lanes per row), so waiting for it
    //      does not wait for the weight fragments issued behind it
    const int xr = tid >> 4, xl = tid & 15;   // NTHR / 16 = TM rows, one pass
    const bool rok = (m0 + xr) < g.n_rows;
    const int64_t src = (gidx && rok) ? gidx[m0 + xr] : 0;
    const int net = idsp ? idsp[e] : e;
#ifdef SSAC_LAB
#define DSTAMP(i) do { if (g.dbg && dbg_off >= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); BSTAMP(i); } } while (0)
#else
#define DSTAMP(i) do { } while (0)
#endif
    DSTAMP(12);   // (debug runs only: serialises the prologue to time its dependent loads one by one)
    if (net < 0) {  // empty subset slot of a sharded rank: +inf, the neutral element of the min that follows
        if (MODE == MODE_PLAIN && g.Y)
            for (int i = tid; i < TM * OUT; i += NTHR) {
                const int r = i / OUT, o = i - r * OUT;
                if ((m0 + r) < g.n_rows) g.Y[((int64_t)e * g.n_rows + m0 + r) * OUT + o] = __builtin_inff();
            }
        return;
    }
    const float *P = g.params + (int64_t)net * g.net_stride;
    const unsigned short *S = g.shadow + (int64_t)net * g.sg.stride;
    const int col0 = wave * 32;
    const bool wave_on = col0 < H;
    const int c0 = wave_on ? col0 : 0;  // (idle waves prefetch wave 0's rows: unconditional loads, results unused)
    const int ns1 = K1P >> 4, nsh = H >> 4;
    const int n_me = col0 + li;         // the feature this lane owns in the hidden layers

    // ---- x tile: the row loads go out as soon as the index is back; every weight fragment of fc1 and fc2 and the
    //      small operands follow them into flight (they depend on nothing but the net id)
    const bool actor_half = (g.gth_role & 1) != 0;   // roles 1, 3 and 5
    // CONSUMER of a hand-off (bf_chain_pc_kernel's target-critic workgroups; ssac_fused.hip has the fp32 original): the
    // state columns of [s'|a'] are fetched here and fc1 runs on them while the tile's actor workgroup is still sampling; the
    // action columns of the x tile stay ZERO, a' W1[:, S:]^T is added to the fc1 accumulators when the granules arrive
    const bool CONS = HO && MODE == MODE_PLAIN && g.ho.pub != nullptr;
    const int XC = CONS ? g.ho.S : IN;   // columns of the x tile that are loaded
    const int Sg = gidx ? (int)g.gth.s_elems : 0;
    const float *srow = gidx ? (actor_half ? g.gth.s1 : g.gth.s) + src * Sg : g.X + (int64_t)(rok ? m0 + xr : 0) * g.ldx;
    const float *arow = gidx ? g.gth.act + src * g.gth.a_elems - Sg : srow;
    float xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {   // first 128 columns of the row (all of them for in_dim <= 128)
        const int k = xl + 16 * u;
        const bool ok = rok && k < XC;
        const bool in_act = gidx && k >= Sg;   // (arow is biased by -Sg: its first valid index is Sg)
        xv[u] = (in_act ? arow : srow)[ok ? k : (in_act ? Sg : 0)];
        if (!ok) xv[u] = 0.0f;
    }
    DSTAMP(13);
    // (vmcnt retires in order: everything the prologue's LDS stores wait for -- x, rewards, biases, W3 -- is requested
    //  BEFORE the 18 fragment loads per lane, so that wait does not also sit out the whole fc1 + fc2 weight fetch;
    //  with the small loads behind the fragments the prologue was 14-17 k clocks in every role)
    float *outp = (gidx && !CONS) ? (actor_half ? g.gth.x1sa : (e == 0 ? g.gth.xsa : nullptr)) : nullptr;
    const int64_t ldo_g = actor_half ? g.gth.ld_x1 : g.gth.ld_x;
    float rew_v = 0.0f, done_v = 0.0f;
    const bool rd_lane = gidx && actor_half && !CONS && xl == 0 && rok;
    if (rd_lane) { rew_v = g.gth.rew[src]; done_v = (float)g.gth.done[src]; }
    float bv1 = 0.0f, bv2 = 0.0f, wv3 = 0.0f, bv3 = 0.0f;
    if (tid < H) { bv1 = P[g.off[1] + tid]; bv2 = P[g.off[3] + tid]; wv3 = bf2f(S[g.sg.o3 + tid]); }
    if (tid < ldo) bv3 = tid < OUT ? P[g.off[5] + tid] : 0.0f;
    const unsigned short *wp1 = S + g.sg.o1 + frag_off(ns1, c0 + li, 8 * lh);   // (+ FRAG_STEP per K-step)
    const unsigned short *wp2 = S + g.sg.o2 + frag_off(nsh, c0 + li, 8 * lh);
    Frags f1, f2;
    DSTAMP(14);
    f1.load(wp1, ns1);
    f2.load(wp2, nsh);
    DSTAMP(15);
    if (MODE == MODE_SAMPLE && !g.eps) {
        // in-kernel noise: this tile's standard normals are computed while the prologue's loads are in flight (ssac_fused.hip)
        const int A = OUT >> 1;
        for (int t = tid; t < TM * A; t += NTHR) {
            const int r = t / A, i = t - r * A, b = m0 + r;
            const int64_t draw = (gslot && g.gth.rng_word >= 0)
                                     ? g.rng.offset + *reinterpret_cast<const int64_t *>(gslot + g.gth.rng_word)
                                     : rng_draw(g.rng);
            lpt[r * ldo + A + i] = b < g.n_rows ? philox_normal(g.rng.seed, draw, b, i) : 0.0f;
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = xl + 16 * u;
        if (k < K1P) {
            xs[xr * ldx_s + k] = f2bf(xv[u]);
            if (rok && k < IN && outp) outp[(int64_t)(m0 + xr) * ldo_g + k] = xv[u];
        }
    }
    for (int k0 = xl + 128; k0 < K1P; k0 += 128) {   // wide observations: 8 more columns per lane per round
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 16 * u;
            const bool ok = rok && k < XC;
            const bool in_act = gidx && k >= Sg;
            v[u] = (in_act ? arow : srow)[ok ? k : (in_act ? Sg : 0)];
            if (!ok) v[u] = 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 16 * u;
            if (k < K1P) {
                xs[xr * ldx_s + k] = f2bf(v[u]);
                if (rok && k < IN && outp) outp[(int64_t)(m0 + xr) * ldo_g + k] = v[u];
            }
        }
    }
    if (rd_lane) { g.gth.rew_out[m0 + xr] = rew_v; g.gth.done_out[m0 + xr] = done_v; }
    if (tid < H) { b1s[tid] = bv1; b2s[tid] = bv2; w3f[tid] = wv3; }
    if (tid < ldo) b3s[tid] = bv3;
    if (CONS) {   // the wave's a' tile: zero but for the action columns, which arrive later
        const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = lane; i < TM * XA_LD / 8; i += 64) *reinterpret_cast<u16x8 *>(xa + wave * (TM * XA_LD) + 8 * i) = z;
    }
    lds_barrier();
    BSTAMP(1);
    if (MODE == MODE_CRITIC_U && e == 0 && g.XT) {
        // the [s|a] tile transposed (K1P x Bp, batch contiguous) for the fc1 weight gradient: 4 batch rows per store
        for (int i = tid; i < K1P * (TM / 4); i += NTHR) {
            const int k = i % K1P, b0 = (i / K1P) * 4;
            u16x4 v;
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = xs[(b0 + u) * ldx_s + k];
            store_quad(g.XT + frag_off(g.bp >> 4, k, m0 + b0) - b0, b0, g.n_rows - m0, v);
        }
    }

    f32x16 acc;
    // The passes that save nothing transposed (actor, target critics, the plain forward) run D^T = W . act^T (bf_mma<true>);
    // the critic workgroup keeps D = act . W^T, whose accumulators are the feature-major saves of the weight-gradient launch.
    constexpr bool TR = MODE != MODE_CRITIC_U;
    const unsigned short *wp3;   // the matrix after fc2: W2^T (backward-data) or the head's rows
    constexpr int step3 = MODE == MODE_CRITIC_U ? FRAG_STEP : 16;
    // ---- fc1
    zero_acc(acc);
    bf_mma<TR>(acc, f1, wp1, ns1, xs + li * ldx_s + 8 * lh);
    BSTAMP(2);
    if constexpr (TR) {
        if (CONS) {
            // a' W1[:, S:S+A]^T as ONE or TWO more MFMAs (round 5; was 128 fused multiply-adds per lane behind an LDS tile and
            // a workgroup barrier): the W1 fragments of the K-steps that hold the action columns are requested now; every
            // wave polls the tile's granules itself -- lane (row, half) takes 4 of the <= 8 action dimensions -- into its
            // OWN zeroed [TM][32] tile (columns k - 16 t0; no workgroup barrier: a wave's LDS accesses are in order), reads
            // the two B fragments back and multiplies.  a' is rounded to bf16 as the one-workgroup chain's x tile rounds it.
            const int A_ = g.ho.A, S_ = g.ho.S;
            const int t0 = S_ >> 4, t1 = (S_ + A_ - 1) >> 4;   // (A <= 8: at most two K-steps)
            const bf16x8 wact0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8 *>(wp1 + FRAG_STEP * t0));
            const bf16x8 wact1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8 *>(wp1 + FRAG_STEP * t1));
            unsigned short *xaw = xa + wave * (TM * XA_LD);
            const unsigned tag = g.ho.base + (g.ho.tick ? (unsigned)*g.ho.tick : 0u);
            const int b = m0 + li;
#pragma unroll
            for (int dd = 0; dd < BF_HO_AMAX / 2; ++dd) {
                const int d = (BF_HO_AMAX / 2) * lh + dd;
                if (d < A_) {
                    const float v = b < g.n_rows ? handoff_poll(g.ho.pub + (int64_t)b * A_ + d, tag) : 0.0f;
                    xaw[li * XA_LD + (S_ + d - 16 * t0)] = f2bf(v);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const bf16x8 xa0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8 *>(xaw + li * XA_LD + 8 * lh));
            const bf16x8 xa1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8 *>(xaw + li * XA_LD + 16 + 8 * lh));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wact0, xa0, acc, 0, 0, 0);
            if (t1 != t0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wact1, xa1, acc, 0, 0, 0);
        }
        // heads wider than one output (the actor, discrete critics): W3's rows go in flight now, into fc1's registers
        const int hrow = (col0 + li) < OUT ? col0 + li : 0;
        wp3 = S + g.sg.o3 + (int64_t)hrow * H + 8 * lh;   // (W3: row-major, 16 elements per K-step)
        if (OUT > 1) f1.load(wp3, nsh, 16);
        if (wave_on) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = col0 + 8 * q + 4 * lh;   // 4 consecutive features of row li
                const f4 bq = *reinterpret_cast<const f4 *>(b1s + n0);
                u16x4 hv;
#pragma unroll
                for (int i = 0; i < 4; ++i) hv[i] = f2bf(fmaxf(acc[4 * q + i] + bq[i], 0.0f));
                *reinterpret_cast<u16x4 *>(h1s + li * ldh + n0) = hv;
            }
        }
        lds_barrier();
        BSTAMP(3);
        // ---- fc2
        zero_acc(acc);
        bf_mma<true>(acc, f2, wp2, nsh, h1s + li * ldh + 8 * lh);
        BSTAMP(4);
        if (OUT == 1) {
            // single output (critics): the head's dot product straight from the accumulators -- h2 is rounded to bf16 where
            // it used to be stored -- summed over the lane's 16 features, the two halves, then the 8 waves through LDS
            float qp = 0.0f;
            if (wave_on) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n0 = col0 + 8 * q + 4 * lh;
                    const f4 bq = *reinterpret_cast<const f4 *>(b2s + n0), wq = *reinterpret_cast<const f4 *>(w3f + n0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) qp += bf2f(f2bf(fmaxf(acc[4 * q + i] + bq[i], 0.0f))) * wq[i];
                }
            }
            qp += __shfl_xor(qp, 32, 64);
            if (lh == 0) lpt[wave * TM + li] = qp;   // (lpt: the sample epilogue's scratch, unused by a single-output pass)
            lds_barrier();
            BSTAMP(5);
            if (tid < TM) {
                float v = b3s[0];
#pragma unroll
                for (int w = 0; w < NTHR / 64; ++w) v += lpt[w * TM + tid];
                ys[tid * ldo] = v;
                if (g.Y && (m0 + tid) < g.n_rows) g.Y[(int64_t)e * g.n_rows + m0 + tid] = v;
            }
        } else {
            if (wave_on) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n0 = col0 + 8 * q + 4 * lh;
                    const f4 bq = *reinterpret_cast<const f4 *>(b2s + n0);
                    u16x4 hv;
#pragma unroll
                    for (int i = 0; i < 4; ++i) hv[i] = f2bf(fmaxf(acc[4 * q + i] + bq[i], 0.0f));
                    *reinterpret_cast<u16x4 *>(h2s + li * ldh + n0) = hv;
                }
            }
            lds_barrier();
            BSTAMP(5);
            if (col0 < OUT) {
                // wave w computes outputs [32w, 32w + 32) over the whole K (rows past OUT read row 0 and are dropped)
                zero_acc(acc);
                bf_mma<true>(acc, f1, wp3, nsh, h2s + li * ldh + 8 * lh, 16);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = col0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (o < OUT) {
                        const float v = acc[r] + b3s[o];
                        ys[li * ldo + o] = v;
                        if (g.Y && (m0 + li) < g.n_rows) g.Y[((int64_t)e * g.n_rows + m0 + li) * OUT + o] = v;
                    }
                }
            }
        }
    } else {
    if (CONS) {
        // W1's action columns of this lane's feature, from the bf16 shadow (what the MFMA would have multiplied): requested
        // here -- fc1's fragments are dead, the next prefetch has not been issued: the body's register budget is full
        // without them -- and back long before a' is
        float wa[BF_HO_AMAX];
#pragma unroll
        for (int d = 0; d < BF_HO_AMAX; ++d)
            wa[d] = d < g.ho.A ? bf2f(S[g.sg.o1 + frag_off(ns1, c0 + li, g.ho.S + d)]) : 0.0f;
        // a' of the tile's rows: poll the granules (tag: this launch's), rounded to bf16 as the one-workgroup chain's x tile
        // rounds them, parked in the sample epilogue's scratch [TM][BF_HO_AMAX]
        const int A_ = g.ho.A;
        const unsigned tag = g.ho.base + (g.ho.tick ? (unsigned)*g.ho.tick : 0u);
        for (int t = tid; t < TM * A_; t += NTHR) {
            const int r = t / A_, d = t - r * A_, b = m0 + r;
            lpt[r * BF_HO_AMAX + d] = b < g.n_rows ? bf2f(f2bf(handoff_poll(g.ho.pub + (int64_t)b * A_ + d, tag))) : 0.0f;
        }
        lds_barrier();
        if (wave_on) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float *ar = lpt + (8 * q + 4 * lh + i) * BF_HO_AMAX;
                    float s_ = 0.0f;
#pragma unroll
                    for (int d = 0; d < BF_HO_AMAX; ++d) s_ += ar[d] * wa[d];   // (d >= A: wa = 0)
                    acc[4 * q + i] += s_;
                }
        }
    }
    // the NEXT matrix after fc2 goes in flight now, into fc1's registers: W2^T (backward-data) or the head's rows
    if (MODE == MODE_CRITIC_U) {
        wp3 = S + g.sg.o2t + frag_off(nsh, c0 + li, 8 * lh);
    } else {
        const int hrow = (col0 + li) < OUT ? col0 + li : 0;
        wp3 = S + g.sg.o3 + (int64_t)hrow * H + 8 * lh;   // (W3: row-major, 16 elements per K-step)
    }
    f1.load(wp3, nsh, step3);
    if (wave_on) {
        const float bias = b1s[n_me];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b0 = 8 * q + 4 * lh;
            u16x4 hv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                hv[i] = f2bf(fmaxf(acc[4 * q + i] + bias, 0.0f));
                h1s[(b0 + i) * ldh + n_me] = hv[i];
            }
            if (MODE == MODE_CRITIC_U)
                store_quad(g.H1T + (int64_t)e * H * g.bp + frag_off(g.bp >> 4, n_me, m0 + b0) - b0, b0, g.n_rows - m0, hv);
        }
    }
    lds_barrier();
    BSTAMP(3);
    // ---- fc2
    zero_acc(acc);
    bf_mma(acc, f2, wp2, nsh, h1s + li * ldh + 8 * lh);
    BSTAMP(4);
    if (wave_on) {
        const float bias = b2s[n_me];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b0 = 8 * q + 4 * lh;
            u16x4 hv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                hv[i] = f2bf(fmaxf(acc[4 * q + i] + bias, 0.0f));
                h2s[(b0 + i) * ldh + n_me] = hv[i];
            }
            if (MODE == MODE_CRITIC_U) {
                store_quad(g.H2T + (int64_t)e * H * g.bp + frag_off(g.bp >> 4, n_me, m0 + b0) - b0, b0, g.n_rows - m0, hv);
                // single-output critics: the head's backward needs h2's sign only -- dz2u[b][j] = W3[j] [h2[b][j] > 0] is
                // written right here (its own LDS tile: the head below still reads h2), not in a pass of its own behind
                // the head (4.5 k clocks of a 28 k-clock workgroup: 2-byte LDS read-modify-writes)
                const unsigned short wj = f2bf(w3f[n_me]);
                u16x4 dz;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dz[i] = bf_pos(hv[i]) ? wj : (unsigned short)0;
                    dz2s[(b0 + i) * ldh + n_me] = dz[i];
                }
                store_quad(g.DZ2T + (int64_t)e * H * g.bp + frag_off(g.bp >> 4, n_me, m0 + b0) - b0, b0, g.n_rows - m0, dz);
            }
        }
    }
    lds_barrier();
    BSTAMP(5);
    // ---- head
    if (OUT == 1) {
        // single output (critics): 16 lanes per row, fp32 dot product of the bf16 operands, shuffle tree
        float s = 0.0f;
        for (int k = xl * 8; k < H; k += 128) {
            const u16x8 hv = *reinterpret_cast<const u16x8 *>(h2s + xr * ldh + k);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += bf2f(hv[u]) * w3f[k + u];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (xl == 0) {
            const float v = s + b3s[0];
            ys[xr * ldo] = v;
            if (g.Y && rok) g.Y[(int64_t)e * g.n_rows + m0 + xr] = v;
        }
    } else if (col0 < OUT) {
        // wave w computes outputs [32w, 32w + 32) over the whole K (rows past OUT read row 0 and are dropped)
        zero_acc(acc);
        bf_mma(acc, f1, wp3, nsh, h2s + li * ldh + 8 * lh, step3);
        const int o = col0 + li;
        if (o < OUT) {
            const float bias = b3s[o];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float v = acc[r] + bias;
                ys[b * ldo + o] = v;
                if (g.Y && (m0 + b) < g.n_rows) g.Y[((int64_t)e * g.n_rows + m0 + b) * OUT + o] = v;
            }
        }
    }
    }
    BSTAMP(6);
    if (MODE == MODE_PLAIN) return;
    lds_barrier();

    if (MODE == MODE_SAMPLE) {
        // tanh-normal head in fp32 (distributions.py:9-15, 64-104), as in ssac_fused.hip
        const int A = OUT >> 1;
        for (int t = tid; t < TM * A; t += NTHR) {
            const int r = t / A, i = t - r * A, b = m0 + r;
            if (b < g.n_rows) {
                // (bf16 mode: hardware exp2 / log2 / rcp, ~1 ulp each, instead of the libm-grade sequences the fp32 path
                //  keeps for parity with the reference -- this epilogue is a dependent chain on the launch's critical path,
                //  actor -> a' -> target critics, and its consumers round a' to bf16 anyway.  dlt / sd = eps exactly, log sd =
                //  log_std by construction)
                const float mu = ys[r * ldo + i], raw = ys[r * ldo + A + i];
                const float log_std = g.lo + 0.5f * (g.hi - g.lo) * (fast_tanh(raw) + 1.0f);
                const float sd = fast_exp(log_std);
                const float ep = g.eps ? g.eps[(int64_t)b * A + i] : lpt[r * ldo + A + i];   // (drawn in the prologue by this thread)
                const float u = mu + sd * ep;
                const float z = -2.0f * u;
                const float sp = z > 20.0f ? z : fast_log(1.0f + fast_exp(z));   // softplus(-2u)
                lpt[r * ldo + i] = (-0.5f * ep * ep - log_std - LOG_SQRT_2PI) - 2.0f * (LOG_2 - u - sp);
                const float a_new = fast_tanh(u);
                g.act_dst[b * g.ld_act + g.act_col0 + i] = a_new;
                if (g.ho.pub)   // the tile's target-critic workgroups are polling for it
                    handoff_publish(g.ho.pub + (int64_t)b * A + i, g.ho.base + (g.ho.tick ? (unsigned)*g.ho.tick : 0u), a_new);
            }
        }
        lds_barrier();
        if (g.logp && tid < TM && (m0 + tid) < g.n_rows) {
            float lp = 0.0f;
            for (int i = 0; i < A; ++i) lp += lpt[tid * ldo + i];
            g.logp[m0 + tid] = lp;
        }
        BSTAMP(7);
        return;
    }

    // ---- MODE_CRITIC_U (single-output critics): the TD-independent half of the backward pass: dz2u was written by the fc2
    //      epilogue (dz2s; visible since the barrier behind it), dz1u = (dz2u . W2) (.) [h1 > 0]
    BSTAMP(7);
    zero_acc(acc);
    bf_mma(acc, f1, wp3, nsh, dz2s + li * ldh + 8 * lh);
    BSTAMP(8);
    if (wave_on) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b0 = 8 * q + 4 * lh;
            u16x4 dv;
#pragma unroll
            for (int i = 0; i < 4; ++i) dv[i] = bf_pos(h1s[(b0 + i) * ldh + n_me]) ? f2bf(acc[4 * q + i]) : (unsigned short)0;
            store_quad(g.DZ1T + (int64_t)e * H * g.bp + frag_off(g.bp >> 4, n_me, m0 + b0) - b0, b0, g.n_rows - m0, dv);
        }
    }
    BSTAMP(9);
}

// ---------------------------------------------------------------------------------------------------------------
// Large-batch ensemble-Q forward (single-output critics, SURVEY 8(d)'s B = 4 096 ... 65 536 rows of the kernel).
// bf_mlp_body gives every 32-row tile its own workgroup, and every workgroup fetches its net's whole weight slab
// (W1 + W2: ~140 KB of bf16) from L2 for 32 rows of work: ~128 B / clk / CU of operand demand against a 64 B / clk L2
// port -- 0.07 of the bf16 matrix peak however large the batch.  Here a workgroup is PERSISTENT over row tiles: each
// wave loads the fragments of its 32 output features ONCE (fc1: NS1 K-steps, fc2: 16 -- they stay in registers) and
// then streams 64-row tiles through them, two MFMAs per fragment (rows 0-31 / 32-63 of the tile share it), the next
// tile's rows in flight while the current one is multiplied.  Per tile the chip-side traffic is the tile's own rows
// in and its Q values out.
template <int NS1>
__global__ __launch_bounds__(NTHR) void bf_stream_kernel(BfArgs g, int tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TS = 64;
    const int H = g.hidden, IN = g.in_dim, K1P = g.sg.k1p;
    const int ldx_s = K1P + LPAD, ldh = H + LPAD;
    unsigned short *xs = reinterpret_cast<unsigned short *>(smem);
    unsigned short *h1s = xs + TS * ldx_s;
    unsigned short *h2s = h1s + TS * ldh;
    float *b1s = reinterpret_cast<float *>(h2s + TS * ldh);
    float *b2s = b1s + H, *w3f = b2s + H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int e = blockIdx.y;
    const int net = g.ids ? g.ids[e] : e;
    if (net < 0) {   // empty subset slot of a sharded rank
        for (int64_t i = blockIdx.x * (int64_t)NTHR + tid; i < g.n_rows; i += (int64_t)gridDim.x * NTHR)
            g.Y[(int64_t)e * g.n_rows + i] = __builtin_inff();
        return;
    }
    const float *P = g.params + (int64_t)net * g.net_stride;
    const unsigned short *S = g.shadow + (int64_t)net * g.sg.stride;
    const int col0 = wave * 32;
    const bool wave_on = col0 < H;
    const int c0 = wave_on ? col0 : 0, n_me = col0 + li;
    const int ns1 = K1P >> 4, nsh = H >> 4;
    // ---- the wave's weight fragments, once
    u16x8 w1[NS1], w2[16];
    {
        const unsigned short *wp1 = S + g.sg.o1 + frag_off(ns1, c0 + li, 8 * lh);
        const unsigned short *wp2 = S + g.sg.o2 + frag_off(nsh, c0 + li, 8 * lh);
#pragma unroll
        for (int t = 0; t < NS1; ++t) w1[t] = *reinterpret_cast<const u16x8 *>(wp1 + FRAG_STEP * (t < ns1 ? t : ns1 - 1));
#pragma unroll
        for (int t = 0; t < 16; ++t) w2[t] = *reinterpret_cast<const u16x8 *>(wp2 + FRAG_STEP * (t < nsh ? t : nsh - 1));
    }
    if (tid < H) { b1s[tid] = P[g.off[1] + tid]; b2s[tid] = P[g.off[3] + tid]; w3f[tid] = bf2f(S[g.sg.o3 + tid]); }
    const float b3 = P[g.off[5]];
    // ---- x staging: thread -> row tid >> 3 (64 rows), columns (tid & 7) + 8 u: XV values per thread per tile
    constexpr int XV = NS1 * 2;            // K1P / 8
    const int xr = tid >> 3, xl = tid & 7;
    float xv[XV];
    auto load_x = [&](int tile) {
        const int row = tile * TS + xr;
        const bool rok = tile < tiles && row < g.n_rows;
        const float *src = g.X + (int64_t)(rok ? row : 0) * g.ldx;
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int k = xl + 8 * u;
            const bool ok = rok && k < IN;
            xv[u] = src[ok ? k : 0];
            if (!ok) xv[u] = 0.0f;
        }
    };
    int tile = blockIdx.x;
    load_x(tile);
    for (; tile < tiles; tile += gridDim.x) {
        const int m0 = tile * TS;
        // (the previous tile's head has read h2s / its fc1 has read xs: the barrier at the end of the loop body)
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int k = xl + 8 * u;
            if (k < K1P) xs[xr * ldx_s + k] = f2bf(xv[u]);
        }
        load_x(tile + gridDim.x);          // next tile's rows travel under this tile's matrix work
        lds_barrier();
        f32x16 a0, a1;
        // ---- fc1
        zero_acc(a0); zero_acc(a1);
#pragma unroll
        for (int t = 0; t < NS1; ++t) {
            if (t < ns1) {
                const bf16x8 w = __builtin_bit_cast(bf16x8, w1[t]);
                const u16x8 x0 = *reinterpret_cast<const u16x8 *>(xs + li * ldx_s + 8 * lh + 16 * t);
                const u16x8 x1 = *reinterpret_cast<const u16x8 *>(xs + (32 + li) * ldx_s + 8 * lh + 16 * t);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x0), w, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x1), w, a1, 0, 0, 0);
            }
        }
        if (wave_on) {
            // (2-byte scattered stores: swapping one value per row pair with the neighbouring lane through a DPP quad
            // permute to store 4 bytes measured slower, 178 vs 156 us at B 65 536 -- the selects cost more than the
            // 16 store instructions saved)
            const float bias = b1s[n_me];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = (r & 3) + 8 * (r >> 2) + 4 * lh;
                h1s[b * ldh + n_me] = f2bf(fmaxf(a0[r] + bias, 0.0f));
                h1s[(32 + b) * ldh + n_me] = f2bf(fmaxf(a1[r] + bias, 0.0f));
            }
        }
        lds_barrier();
        // ---- fc2
        zero_acc(a0); zero_acc(a1);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t < nsh) {
                const bf16x8 w = __builtin_bit_cast(bf16x8, w2[t]);
                const u16x8 x0 = *reinterpret_cast<const u16x8 *>(h1s + li * ldh + 8 * lh + 16 * t);
                const u16x8 x1 = *reinterpret_cast<const u16x8 *>(h1s + (32 + li) * ldh + 8 * lh + 16 * t);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x0), w, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x1), w, a1, 0, 0, 0);
            }
        }
        if (wave_on) {
            const float bias = b2s[n_me];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = (r & 3) + 8 * (r >> 2) + 4 * lh;
                h2s[b * ldh + n_me] = f2bf(fmaxf(a0[r] + bias, 0.0f));
                h2s[(32 + b) * ldh + n_me] = f2bf(fmaxf(a1[r] + bias, 0.0f));
            }
        }
        lds_barrier();
        // ---- head: 8 lanes per row, fp32 dot product of the bf16 operands (the same products, k-order and shuffle
        //      tree aside, as bf_mlp_body's single-output head)
        {
            float sacc = 0.0f;
            for (int k = xl * 8; k < H; k += 64) {
                const u16x8 hv = *reinterpret_cast<const u16x8 *>(h2s + xr * ldh + k);
#pragma unroll
                for (int u = 0; u < 8; ++u) sacc += bf2f(hv[u]) * w3f[k + u];
            }
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
            if (xl == 0 && (m0 + xr) < g.n_rows) g.Y[(int64_t)e * g.n_rows + m0 + xr] = sacc + b3;
        }
        // (no barrier here: the next tile's xs stores follow this tile's fc1 reads by two barriers, and its h2s stores
        //  come behind two MORE barriers that no wave passes before it has left this head)
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Large-batch ensemble-Q forward, REGISTER-CHAINED (round 5; hidden 256, 17 <= in_dim <= 32, single-output critics).
// bf_stream_kernel above keeps the WEIGHTS in registers and sends every activation through LDS: x -> LDS -> fc1 -> h1 as
// 32 two-byte scattered stores per lane -> barrier -> fc2 (one ds_read_b128 per MFMA) -> h2 -> LDS -> barrier -> head:
// three barriers per 64 rows, one LDS read per MFMA (the LDS port's whole bandwidth at the matrix peak), 0.24 of the bf16
// peak at B 65 536.  Here it is the other way round: the net's W1 / W2 live in LDS for the lifetime of the (persistent)
// workgroup -- 144 KB, as MFMA A-operand fragments -- and a WAVE owns 64 batch rows end to end:
//   * every product is D^T = W . act^T, so a lane (b = lane & 31, half) holds, for ITS batch row, 16 features of each
//     32-feature block: acc[r] = D^T[32 blk + (r & 3) + 8 (r >> 2) + 4 half][b];
//   * bias + ReLU + bf16 rounding happen in registers and the 8 packed values of accumulator registers 8 s .. 8 s + 7 ARE
//     the next layer's B-operand fragment of K-step (blk, s) -- for the k values {16 s + 4 half + 0..3, 16 s + 4 half +
//     8..11} of the block, not 8 consecutive ones: the MFMA does not care which k a slot carries as long as A and B agree,
//     so the LDS image of W2 is written with that permutation once per workgroup;
//   * x comes straight from memory in the B-fragment layout (a lane reads 8 consecutive floats of its row per K-step), h2
//     never exists outside the accumulators (the head's dot product is taken from them), Q leaves as one coalesced store.
// No activation ever touches LDS, no barrier after the weight image is built, every weight fragment read feeds two MFMAs
// (the wave's two 32-row blocks): half the LDS traffic per FLOP, waves fully independent of each other.
// Same bf16 operands, same fp32 products and the same rounding points as bf_mlp_body / bf_stream_kernel; the summation
// order of the head differs (tests/test_hip_bf16.py compares the three).
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ReLU as ONE integer max on the bit pattern (negative floats are negative integers; fmaxf costs a second instruction that
// quiets a possible signalling NaN first): -0 and negative NaNs become +0, positive values are untouched
__device__ __forceinline__ float relu_bits(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

__device__ __forceinline__ bf16x8 pack_bf8(const float (&v)[8]) {
    const f32x8 t = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    return __builtin_convertvector(t, bf16x8);
}

constexpr int RC_H = 256, RC_NB = RC_H / 32, RC_NS1 = 2;
constexpr size_t RC_LDS = 2 * ((size_t)RC_NB * RC_NB * 2 + RC_NB * RC_NS1) * 512 + 4 * 3 * RC_H;

__global__ __launch_bounds__(NTHR) void bf_regchain_kernel(BfArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int H = RC_H, NB = RC_NB, NS1 = RC_NS1;
    unsigned short *w2s = reinterpret_cast<unsigned short *>(smem);   // chunk ((j NB + i) 2 + s): lane l's 8 elements at l * 8
    unsigned short *w1s = w2s + NB * NB * 2 * 512;                    // chunk (i NS1 + s)
    float *b1s = reinterpret_cast<float *>(w1s + NB * NS1 * 512);
    float *b2s = b1s + H;
    unsigned short *w3b = reinterpret_cast<unsigned short *>(b2s + H);   // W3's row as it is in the shadow (bf16)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int e = blockIdx.y, IN = g.in_dim;
    const int net = g.ids ? g.ids[e] : e;
    if (net < 0) {   // empty subset slot of a sharded rank
        for (int64_t i = blockIdx.x * (int64_t)NTHR + tid; i < g.n_rows; i += (int64_t)gridDim.x * NTHR)
            g.Y[(int64_t)e * g.n_rows + i] = __builtin_inff();
        return;
    }
#ifdef SSAC_LAB
    if (g.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g.dbg[5] = __builtin_amdgcn_s_memtime();
#endif
    const float *P = g.params + (int64_t)net * g.net_stride;
    const unsigned short *S = g.shadow + (int64_t)net * g.sg.stride;
    // ---- the net's weights -> LDS, once.  W1: the shadow's fragment-major chunks as they are.  W2: chunk (j, t) of the
    //      shadow holds k = 16 t + 8 h' + 0..7 in lane (n, h'); lane (n, half) of the image takes elements 4 half .. + 3
    //      of BOTH of them (k = 16 t + 4 half + 0..3 and + 8)
    {
        const unsigned short *s1 = S + g.sg.o1, *s2 = S + g.sg.o2;
        for (int i = tid; i < NB * NS1 * 64; i += NTHR)
            *reinterpret_cast<u16x8 *>(w1s + i * 8) = *reinterpret_cast<const u16x8 *>(s1 + i * 8);
        // (a thread takes row n of a chunk: both halves in, both halves out -- 16-byte loads and stores, all 16 loads of a
        //  thread in flight before the first store)
        constexpr int PER = NB * NB * 2 * 32 / NTHR;   // 8
        u16x8 n0[PER], n1[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = tid + k * NTHR, c = idx >> 5, n = idx & 31;
            n0[k] = *reinterpret_cast<const u16x8 *>(s2 + ((int64_t)c * 64 + n) * 8);
            n1[k] = *reinterpret_cast<const u16x8 *>(s2 + ((int64_t)c * 64 + 32 + n) * 8);
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = tid + k * NTHR, c = idx >> 5, n = idx & 31;
            const u16x8 lo = {n0[k][0], n0[k][1], n0[k][2], n0[k][3], n1[k][0], n1[k][1], n1[k][2], n1[k][3]};
            const u16x8 hi = {n0[k][4], n0[k][5], n0[k][6], n0[k][7], n1[k][4], n1[k][5], n1[k][6], n1[k][7]};
            *reinterpret_cast<u16x8 *>(w2s + (c * 64 + n) * 8) = lo;
            *reinterpret_cast<u16x8 *>(w2s + (c * 64 + 32 + n) * 8) = hi;
        }
        if (tid < H) { b1s[tid] = P[g.off[1] + tid]; b2s[tid] = P[g.off[3] + tid]; w3b[tid] = S[g.sg.o3 + tid]; }
    }
    const float b3 = P[g.off[5]];
    __syncthreads();

    // ---- work of this wave: a contiguous range of 32-row UNITS, taken two at a time (the two row blocks of an iteration
    //      share every weight fragment read); an odd last unit runs the same body with one row block (half the MFMAs).
    //      32-row granularity matters: 65 536 rows over 25 workgroups per net are 10.24 units per wave -- whole 64-row tiles
    //      made that 5 or 6 iterations with every workgroup waiting for its 6-iteration waves
    const int units = (g.n_rows + 31) >> 5, nwv = gridDim.x * (NTHR / 64), wv = blockIdx.x * (NTHR / 64) + wave;
    const int ubase = units / nwv, urem = units - ubase * nwv;
    const int u_lo = wv * ubase + (wv < urem ? wv : urem), u_hi = u_lo + ubase + (wv < urem ? 1 : 0);

    // x of one iteration in the B-fragment layout: lane (b, half) wants x[row b of unit u + rb][16 s + 8 half + 0..7]: two
    // 16-byte loads per lane and K-step (load_x: issue only -- nothing here waits for the data).  The ragged K-step (in_dim
    // not a multiple of 16) has one half whose 8-float window crosses the end of the row: those lanes read the LAST 8 floats
    // of the row instead (always inside it) and pack_x, an iteration later, shifts them down in registers -- the shift is
    // the same for every lane that shifts, so it is a 3-stage barrel shifter on scalar conditions, 32 selects; the other
    // half's window lies fully inside the row (kept) or fully outside (zero)
    auto load_x = [&](float (&xf)[2][NS1][8], int u, int nrb) {
        // (the lane's row / half through an asm the compiler cannot see through: everything derived from them here is the
        //  same in every iteration, and hoisted out of the loop it is live across fc2 -- i.e. spilled, and reloaded behind an
        //  s_waitcnt vmcnt(0) that also waits for the rows just requested)
        int lnx = lane;
        asm volatile("" : "+v"(lnx));
        const int lix = lnx & 31, lhx = lnx >> 5;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            if (rb >= nrb) break;   // (uniform)
            int row = (u + rb) * 32 + lix;
            row = row < g.n_rows ? row : g.n_rows - 1;   // (rows past the batch: a valid row, results dropped)
            // (32-bit byte offsets from the uniform base -- the launcher checks that X spans < 4 GiB: no 64-bit address
            //  pairs per lane, which this kernel has no registers for)
            const uint32_t rowoff = (uint32_t)row * (uint32_t)g.ldx;
#pragma unroll
            for (int s = 0; s < NS1; ++s) {
                const int rem = IN - 16 * s;   // (uniform) k values of this K-step inside the row
                const bool hi_part = rem >= 8;
                // first float of the lane's window: half 0 / half 1 (both uniform)
                const int st0 = (rem >= 16 || hi_part) ? 16 * s : IN - 8;
                const int st1 = rem >= 16 ? 16 * s + 8 : IN - 8;
                const int start = st0 + lhx * (st1 - st0);
                const char *p = reinterpret_cast<const char *>(g.X) + (size_t)((rowoff + (uint32_t)start) * 4u);
                const f4 a = *reinterpret_cast<const f4u *>(p), b = *reinterpret_cast<const f4u *>(p + 16);
#pragma unroll
                for (int c = 0; c < 4; ++c) { xf[rb][s][c] = a[c]; xf[rb][s][4 + c] = b[c]; }
            }
        }
    };
    auto pack_x = [&](const float (&t0)[8], int s) -> bf16x8 {
        const int rem = IN - 16 * s;
        float v[8];
        if (rem >= 16) {   // (uniform)
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = t0[c];
        } else {
            const bool hi_part = rem >= 8;                    // (uniform) half 0 whole, half 1 shifts; else half 0 shifts, half 1 empty
            const bool shifts = hi_part ? lh == 1 : lh == 0;
            const int sh = hi_part ? 16 - rem : 8 - rem;      // (uniform) 1 .. 8
            float t1[8], t2[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) t1[c] = (sh & 1) ? (c + 1 < 8 ? t0[c + 1] : 0.0f) : t0[c];
#pragma unroll
            for (int c = 0; c < 8; ++c) t2[c] = (sh & 2) ? (c + 2 < 8 ? t1[c + 2] : 0.0f) : t1[c];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float w = (sh & 4) ? (c + 4 < 8 ? t2[c + 4] : 0.0f) : t2[c];
                w = (sh & 8) ? 0.0f : w;
                v[c] = shifts ? w : (hi_part ? t0[c] : 0.0f);
            }
        }
        return pack_bf8(v);
    };
    float xf[2][NS1][8] = {};
    if (u_lo < u_hi) load_x(xf, u_lo, u_hi - u_lo >= 2 ? 2 : 1);
#ifdef SSAC_LAB
    if (g.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { g.dbg[6] = __builtin_amdgcn_s_memtime(); g.dbg[8] = u_hi - u_lo; }
#endif

#ifdef SSAC_LAB
#define RSTAMP(i) do { if (g.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && u == u_lo + 2) g.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RSTAMP(i) do { } while (0)
#endif
    // one iteration: TWO row blocks (units u, u + 1) or one
    auto body = [&](auto two_c, int u, int u_next, int n_next) {
        constexpr bool TWO = decltype(two_c)::value;
        constexpr int NR = TWO ? 2 : 1;
        RSTAMP(0);
        bf16x8 xb[NR][NS1];
#pragma unroll
        for (int rb = 0; rb < NR; ++rb)
#pragma unroll
            for (int s = 0; s < NS1; ++s) xb[rb][s] = pack_x(xf[rb][s], s);
        // ---- fc1: feature block i -> the B fragments of fc2's K-steps (i, 0) and (i, 1).  The bias is the MFMA's C operand
        //      (the accumulators start from it), so the epilogue is max + pack: 1.5 VALU instructions per element.
        //      Software-pipelined one block deep BY HAND: block i + 1's bias vectors and weight fragments are requested
        //      before block i's MFMAs.  (Left alone, the compiler hoists W1's fragments and the biases -- the same for every
        //      tile -- out of the tile loop, 192 registers that live in scratch, or multiplies all eight blocks first: the
        //      loads hang on an offset it cannot see through, z, re-made once per block by a volatile asm statement, and
        //      each block's results are "used" by another one: volatile asm statements keep their order.)
        bf16x8 h1[NR][NB][2];
        f4 bqn[4];
        bf16x8 wn[NS1];
        auto fc1_fetch = [&](int i, int z) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bqn[q] = *reinterpret_cast<const f4 *>(b1s + 32 * i + 8 * q + 4 * lh + z);
#pragma unroll
            for (int s = 0; s < NS1; ++s) wn[s] = *reinterpret_cast<const bf16x8 *>(w1s + ((i * NS1 + s) * 64 + lane) * 8 + z);
        };
        {
            int z = 0;
            asm volatile("" : "+v"(z));
            fc1_fetch(0, z);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            f32x16 a[NR];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int rb = 0; rb < NR; ++rb) a[rb][4 * q + c] = bqn[q][c];
            bf16x8 w[NS1];
#pragma unroll
            for (int s = 0; s < NS1; ++s) w[s] = wn[s];
            if (i + 1 < NB) {
                int z = 0;
                asm volatile("" : "+v"(z));
                fc1_fetch(i + 1, z);
            }
#pragma unroll
            for (int s = 0; s < NS1; ++s)
#pragma unroll
                for (int rb = 0; rb < NR; ++rb) a[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s], xb[rb][s], a[rb], 0, 0, 0);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int rb = 0; rb < NR; ++rb) {
                    float v[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = relu_bits(a[rb][8 * s2 + c]);
                    h1[rb][i][s2] = pack_bf8(v);
                }
            if (TWO) asm volatile("" :: "v"(h1[0][i][0]), "v"(h1[0][i][1]), "v"(h1[NR - 1][i][0]), "v"(h1[NR - 1][i][1]));
            else asm volatile("" :: "v"(h1[0][i][0]), "v"(h1[0][i][1]));
        }
        RSTAMP(1);
        if (n_next > 0) load_x(xf, u_next, n_next);   // the next iteration's rows travel under fc2
        RSTAMP(2);
        __builtin_amdgcn_sched_barrier(0);
        // ---- fc2 + head: output block j; the accumulators start from b2, and the head's dot product is taken straight
        //      from them: max, pack two features to bf16 (the rounding point of h2), v_dot2c_f32_bf16 with the packed pair
        //      of W3 -- 2 VALU instructions per element, h2 never exists outside the registers.  The weight fragments run
        //      three K-steps ahead of their MFMAs ACROSS the j loop (the chunks of block j + 1 follow block j's in LDS), and
        //      block j + 1's bias vectors are requested in the middle of block j's epilogue.
        float qv[NR];
#pragma unroll
        for (int rb = 0; rb < NR; ++rb) qv[rb] = 0.0f;
        constexpr int PF = 2;   // weight fragments in flight ahead of their MFMAs (a third costs 4 registers the kernel does not have: spills)
        int lnf = lane;   // (as in load_x: the lane's LDS offsets are re-derived here, not kept live -- i.e. spilled -- across fc1)
        asm volatile("" : "+v"(lnf));
        bf16x8 wf[PF];
#pragma unroll
        for (int t = 0; t < PF; ++t) wf[t] = *reinterpret_cast<const bf16x8 *>(w2s + (t * 64 + lnf) * 8);
        f4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f4 *>(b2s + 8 * q + 4 * lh);
#pragma unroll 1
        for (int j = 0; j < NB; ++j) {
            f32x16 a[NR];
#if defined(SSAC_LAB) && defined(SSAC_EXP_RC_NOEPI)
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) zero_acc(a[rb]);
#else
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int rb = 0; rb < NR; ++rb) a[rb][4 * q + c] = bq[q][c];
#endif
            const unsigned short *wj = w2s + (j * NB * 2 * 64 + lnf) * 8;
            bf16x8 w[2 * NB + PF];
#pragma unroll
            for (int t = 0; t < PF; ++t) w[t] = wf[t];
#pragma unroll
            for (int t = 0; t < 2 * NB; ++t) {
                // (t + PF >= 16: the first chunks of block j + 1; behind the last block they are W1's image -- read, unused)
                w[t + PF] = *reinterpret_cast<const bf16x8 *>(wj + (t + PF) * 512);
#pragma unroll
                for (int rb = 0; rb < NR; ++rb)
                    a[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[t], h1[rb][t >> 1][t & 1], a[rb], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < PF; ++t) wf[t] = w[2 * NB + t];
#pragma unroll
            for (int t = 0; t < 2 * NB; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NR, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#if defined(SSAC_LAB) && defined(SSAC_EXP_RC_NOEPI)
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) qv[rb] += a[rb][0] + a[rb][15];   // (measurement build, wrong results: no fc2 epilogue)
#else
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // W3 of features 32 j + 8 q + 4 half + 0..3: two packed bf16 pairs (8 bytes; read again per row block)
                    const bf16x4 wq = *reinterpret_cast<const bf16x4 *>(w3b + 32 * j + 8 * q + 4 * lh);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const bf16x2 wp = {wq[2 * c], wq[2 * c + 1]};
                        const f32x2 m = {relu_bits(a[rb][4 * q + 2 * c]), relu_bits(a[rb][4 * q + 2 * c + 1])};
                        qv[rb] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_convertvector(m, bf16x2), wp, qv[rb], false);
                    }
                }
                if (rb == 0) {   // block j + 1's biases (behind the last block: block 0's, unused)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f4 *>(b2s + 32 * ((j + 1) & (NB - 1)) + 8 * q + 4 * lh);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        RSTAMP(3);
#pragma unroll
        for (int rb = 0; rb < NR; ++rb) qv[rb] += __shfl_xor(qv[rb], 32, 64);
        // lanes 0..31: unit u, lanes 32..63: unit u + 1 -- one coalesced store
        const int row = u * 32 + lane;
        const float val = (TWO && lh) ? qv[NR - 1] : qv[0];
        if ((TWO || lh == 0) && row < g.n_rows) g.Y[(int64_t)e * g.n_rows + row] = val + b3;
        RSTAMP(4);
    };
    int u = u_lo;
    for (; u + 2 <= u_hi; u += 2) {
        const int left = u_hi - (u + 2);
        body(std::true_type{}, u, u + 2, left >= 2 ? 2 : left);
    }
    if (u < u_hi) body(std::false_type{}, u, 0, 0);
#ifdef SSAC_LAB
    if (g.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g.dbg[7] = __builtin_amdgcn_s_memtime();
#endif
}

template <int MODE>
__global__ __launch_bounds__(NTHR) void bf_mlp_kernel(BfArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf_mlp_body<MODE>(g, smem, blockIdx.x, blockIdx.y);
}

// everything of a critic update that does not need the TD target, ONE launch (see fused_chain_kernel in ssac_fused.hip):
// workgroups [0, tiles_t): target chain of subset slot j (actor forward + sample, then target critic ids[j] on [s'|a']);
// the rest: online critics' forward + TD-independent backward.
__global__ __launch_bounds__(NTHR) void bf_chain_kernel(BfArgs ga, BfArgs ga_rest, BfArgs gt, BfArgs gc, int tiles_t,
                                                       int grid_x, DeferredLogsArgs dl, int dl_on) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int bid = blockIdx.x;
    if (dl_on && bid == (int)gridDim.x - 1) {   // the previous recorded update's log block -> its ring slot
        deferred_logs_body(dl, -1);
        return;
    }
    // (each half in XCD-contiguous order, as in fused_chain_kernel: the row tiles of one net share its weight shadows)
    const int n_main = (int)gridDim.x - (dl_on ? 1 : 0);
    if (bid < tiles_t) {
        const int lb = ssac_xcd_contiguous_range(bid, 0, tiles_t, gc.xcd);
        const int j = lb / grid_x, bx = lb - j * grid_x;
        // (kernel parameters are never written to -- that would copy them to scratch; the phase stamps of the actor
        //  pass go to slots 0.., of subset slot 0's target-critic pass to 16.., of the critic half to 32..)
        if (j == 0) bf_mlp_body<MODE_SAMPLE>(ga, smem, bx, 0, 0);
        else bf_mlp_body<MODE_SAMPLE>(ga_rest, smem, bx, 0, -1);
        __threadfence_block();  // this workgroup's a' rows (global) are read back by its own target-critic pass
        __syncthreads();
        bf_mlp_body<MODE_PLAIN>(gt, smem, bx, j, j == 0 ? 16 : -1);
    } else {
        const int L = ssac_xcd_contiguous_range(bid, tiles_t, n_main, gc.xcd);
        bf_mlp_body<MODE_CRITIC_U>(gc, smem, L % grid_x, L / grid_x, 32);
    }
}

// The same launch with the target chains cut into PRODUCER and CONSUMER workgroups (round 4; fused_chain_pc_kernel in
// ssac_fused.hip is the fp32 original): [0, tiles_a) the ACTOR of a 32-row tile, once (not per subset slot), a' published
// as tagged granules; tiles_t consumers, one per (slot, tile): the target critic's gather and fc1 on the state columns, then
// the granules, the rank-A term, fc2 and the head; the online critics' tiles between them.  In bf16 the matrix work of a
// pass is ~2 k clocks and everything else latency: the one-workgroup chain was 25.7 k (actor) + 17.5 k (target critic) =
// 43.5 k clocks against 28.4 k for a critic tile; here the target critic's 10.6 k of prologue + fc1 hide behind the actor.
__global__ __launch_bounds__(NTHR) void bf_chain_pc_kernel(BfArgs ga, BfArgs gt, BfArgs gc, int tiles_a, int tiles_t,
                                                          int grid_x, DeferredLogsArgs dl, int dl_on) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int bid = blockIdx.x;
    // (LAB build: per-workgroup (start, end) stamps, ssac_debug_timeline -- slots of the chained launch, as in ssac_fused.hip)
    SSAC_LAB_ONLY(long long *tl = gc.tl; if (tl && threadIdx.x == 0 && bid < 512) tl[2 * bid] = __builtin_amdgcn_s_memrealtime();)
    if (dl_on && bid == (int)gridDim.x - 1) {
        deferred_logs_body(dl, -1);
        return;
    }
    const int n_main = (int)gridDim.x - (dl_on ? 1 : 0), n_crit = n_main - tiles_a - tiles_t;
    if (bid < tiles_a) {
        bf_mlp_body<MODE_SAMPLE>(ga, smem, ssac_xcd_contiguous_range(bid, 0, tiles_a, gc.xcd), 0, 0);
    } else if (bid < tiles_a + n_crit) {   // (the critic tiles before the consumers: a launch of more than one round)
        const int L = ssac_xcd_contiguous_range(bid, tiles_a, tiles_a + n_crit, gc.xcd);
        bf_mlp_body<MODE_CRITIC_U>(gc, smem, L % grid_x, L / grid_x, 32);
    } else {
        const int lb = ssac_xcd_contiguous_range(bid, tiles_a + n_crit, n_main, gc.xcd);
        const int j = lb / grid_x, bx = lb - j * grid_x;
        bf_mlp_body<MODE_PLAIN, true>(gt, smem, bx, j, j == 0 ? 16 : -1);
    }
#ifdef SSAC_LAB
    if (tl && bid < 512) {
        __syncthreads();
        if (threadIdx.x == 0) tl[2 * bid + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------
// shadow maintenance
// ------------------------------------------------------------------------------------------------------------
__global__ void bf_sync_kernel(const float *__restrict__ params, int64_t net_stride, int in_dim, int hidden, int out_dim,
                               int64_t off_w1, int64_t off_w2, int64_t off_w3, unsigned short *__restrict__ shadow,
                               ShadowGeom sg) {
    const int e = blockIdx.y;
    const float *P = params + (int64_t)e * net_stride;
    unsigned short *S = shadow + (int64_t)e * sg.stride;
    const int64_t n1 = (int64_t)hidden * sg.k1p, n2 = (int64_t)hidden * hidden, n3 = (int64_t)out_dim * hidden;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n1 + n2 + n3; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n1) {
            const int r = (int)(i / sg.k1p), k = (int)(i - (int64_t)r * sg.k1p);
            S[sg.o1 + frag_off(sg.k1p >> 4, r, k)] = k < in_dim ? f2bf(P[off_w1 + (int64_t)r * in_dim + k]) : (unsigned short)0;
        } else if (i < n1 + n2) {
            const int64_t t = i - n1;
            const int r = (int)(t / hidden), c = (int)(t - (int64_t)r * hidden);
            const unsigned short v = f2bf(P[off_w2 + t]);
            S[sg.o2 + frag_off(hidden >> 4, r, c)] = v;
            S[sg.o2t + frag_off(hidden >> 4, c, r)] = v;
        } else {
            S[sg.o3 + (i - n1 - n2)] = f2bf(P[off_w3 + (i - n1 - n2)]);
        }
    }
}

// Polyak on the fp32 masters of the weights' segments + the target's shadow (biases: fp32 only)
__global__ void bf_polyak_kernel(float *__restrict__ target, const float *__restrict__ source, int64_t net_stride,
                                 int in_dim, int hidden, int out_dim, int64_t off_w1, int64_t off_w2, int64_t off_w3,
                                 int64_t n_per_net, float tau, unsigned short *__restrict__ shadow, ShadowGeom sg) {
    const int e = blockIdx.y;
    float *T = target + (int64_t)e * net_stride;
    const float *Sp = source + (int64_t)e * net_stride;
    unsigned short *S = shadow + (int64_t)e * sg.stride;
    const int64_t end1 = off_w1 + (int64_t)hidden * in_dim, end2 = off_w2 + (int64_t)hidden * hidden,
                  end3 = off_w3 + (int64_t)out_dim * hidden;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_per_net; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = T[i] * (1.0f - tau) + Sp[i] * tau;   // learning_utils.py:160-162
        T[i] = v;
        if (i >= off_w1 && i < end1) {
            const int64_t t = i - off_w1;
            const int r = (int)(t / in_dim), k = (int)(t - (int64_t)r * in_dim);
            S[sg.o1 + frag_off(sg.k1p >> 4, r, k)] = f2bf(v);
        } else if (i >= off_w2 && i < end2) {
            const int64_t t = i - off_w2;
            const int r = (int)(t / hidden), c = (int)(t - (int64_t)r * hidden);
            const unsigned short h = f2bf(v);
            S[sg.o2 + frag_off(hidden >> 4, r, c)] = h;
            S[sg.o2t + frag_off(hidden >> 4, c, r)] = h;
        } else if (i >= off_w3 && i < end3) {
            S[sg.o3 + (i - off_w3)] = f2bf(v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// weight gradients (K = batch) + loss gradient folded in + Adam (+Polyak) + shadow refresh.  Single-output critics.
//   dW2[j][i] = sum_b c_b dz2u[b][j] h1[b][i]      A = DZ2uT rows j, B = H1T rows i
//   dW1[j][i] = sum_b c_b dz1u[b][j] x[b][i]       A = DZ1uT rows j, B = XT rows i
//   dW3[i]    = sum_b c_b h2[b][i], db3 = sum_b c_b   (VALU workgroup per net)
// c_b = dL/dq of (net, row b), evaluated per workgroup into LDS (loss_fold_table).  A 256-thread workgroup owns a 64x64
// tile as 2x2 waves of one 32x32 accumulator; both operands are read straight from global memory, 16 bytes per lane
// per MFMA, one 4-step group ahead.  No split-K: every gradient element is produced by ONE wave in a fixed order.
// ------------------------------------------------------------------------------------------------------------
struct BfWgradArgs {
    float *params; int64_t net_stride; int in_dim, hidden; int64_t off[6];
    unsigned short *shadow; ShadowGeom sg;
    const unsigned short *H1T, *H2T, *DZ2T, *DZ1T, *XT;
    int n_rows, bp, n_nets;
    float *am, *av; const ssac_adam_ctl *ctl;
    float *grads;         // != null: GRADIENT mode (clip_grad_norm_ path): the gradients are stored here, fp32, in the
                          // arena's layout, and nothing is updated -- clip + Adam + shadow refresh follow as launches
    float *target; unsigned short *tshadow; float tau;
    const uint32_t *late_word;   // != null: the target update waits for the decision in this word (late-bound Polyak)
    float *sumsq; int64_t sumsq_stride;
    LossFoldArgs lf;
    LogFoldArgs fold;     // fold.done != null: the update's logs are finalised by the last workgroup (ssac_critic_logs.h)
    int tiles2, tiles1;   // 64x64 tiles of fc2 / fc1 per net; then 1 head workgroup per net
    int xcd;                     // XCD-contiguous workgroup order (ssac_internal.h)
    long long *dbg;
    long long *tl;               // optional per-workgroup (start, end) stamps, slots from 1024 (ssac_debug_timeline)
};
#ifdef SSAC_LAB
#define WSTAMP(i) do { if (g.dbg && blockIdx.x == 0 && threadIdx.x == 0) g.dbg[48 + i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSTAMP(i) do { } while (0)
#endif

// Adam on one element.  bf16 mode has no reference to be bit-compatible with (the fp32 path keeps torch's exact sequence:
// ssac_gemm.hip), and this epilogue is 16 elements per lane with ONE wave of the workgroup per SIMD: the IEEE square root and
// the two IEEE divisions were ~30 of its ~40 instructions per element, a dependent chain nothing overlaps -- 11.5 k of the
// workgroup's 29 k clocks (LAB stamps; re-arranging its ~230 stores per lane into 16-byte ones through LDS changed nothing).
// Here: v_sqrt_f32 / v_rcp_f32 (1 ulp each) and the bias correction as a multiplication by its reciprocal -- the update differs
// from the exact sequence by a few ulp of the STEP, i.e. ~1e-7 of lr.
__device__ __forceinline__ float adam_elem(float p, float g, float &m, float &v, const ssac_adam_ctl &c, float inv_bc2_sqrt) {
    if (c.weight_decay != 0.0f) g = g + c.weight_decay * p;
    m = m + (1.0f - c.beta1) * (g - m);
    v = v * c.beta2 + (1.0f - c.beta2) * g * g;
    const float denom = __builtin_amdgcn_sqrtf(v) * inv_bc2_sqrt + c.eps;
    return p - c.step_size * (m * __builtin_amdgcn_rcpf(denom));
}

__device__ __forceinline__ u16x8 scale_frag(const u16x8 a, const float *__restrict__ c, float &colsum) {
    const f4 c0 = *reinterpret_cast<const f4 *>(c), c1 = *reinterpret_cast<const f4 *>(c + 4);
    u16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v = bf2f(a[i]) * (i < 4 ? c0[i] : c1[i - 4]);
        colsum += v;
        o[i] = f2bf(v);
    }
    return o;
}

__global__ __launch_bounds__(256) void bf_wgrad_kernel(BfWgradArgs g) {
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    SSAC_LAB_ONLY(if (g.tl && threadIdx.x == 0 && blockIdx.x < 512) g.tl[1024 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();)
    float *tab = wlds;                 // [bp] row scales (zero beyond n_rows)
    float *red = wlds + g.bp;          // [16] scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_net = g.tiles2 + g.tiles1 + 1;
    // (XCD-contiguous: the tiles of one net -- which share its transposed activation saves -- on one or two XCDs)
    const int lbid = ssac_xcd_contiguous(blockIdx.x, gridDim.x, g.xcd);
    const int e = lbid / per_net, t = lbid - e * per_net;
    const int H = g.hidden, IN = g.in_dim, K1P = g.sg.k1p;
    float *P = g.params + (int64_t)e * g.net_stride;
    float *M = g.am + (int64_t)e * g.net_stride, *V = g.av + (int64_t)e * g.net_stride;
    float *Gr = g.grads ? g.grads + (int64_t)e * g.net_stride : nullptr;
    // late-bound Polyak (include/ssac_hip.h): tau bits left by the update's first launch, 0 = no soft_update followed
    const uint32_t late_bits = g.late_word ? *g.late_word : 0u;
    const bool pol = g.target != nullptr && (g.late_word == nullptr || late_bits != 0u);
    const float tau = g.late_word ? __uint_as_float(late_bits) : g.tau;
    float *T = pol ? g.target + (int64_t)e * g.net_stride : nullptr;
    unsigned short *S = g.shadow + (int64_t)e * g.sg.stride;
    unsigned short *TS = (pol && g.tshadow) ? g.tshadow + (int64_t)e * g.sg.stride : nullptr;
    const bool head = t == per_net - 1;
    WSTAMP(0);
    // ---- tile geometry (GEMM tiles)
    const bool fc2 = t < g.tiles2;
    const int tt = fc2 ? t : t - g.tiles2;
    const int Ncols = fc2 ? H : IN;                 // valid gradient columns
    const int gx = fc2 ? (H + 63) / 64 : (IN + 63) / 64;
    const int by = tt / gx, bxn = tt - by * gx;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int row = by * 64 + wm * 32 + li;         // gradient row j (0 .. H)
    const int col = bxn * 64 + wn * 32 + li;        // gradient column i
    const int Brows = fc2 ? H : K1P;                // rows of the B operand's transposed buffer
    // (the transposed saves are fragment-major like the weight shadows: K-step s of a 32-row block is one KiB)
    const unsigned short *AT = (fc2 ? g.DZ2T : g.DZ1T) + (int64_t)e * H * g.bp + frag_off(g.bp >> 4, row < H ? row : 0, 8 * lh);
    const unsigned short *BT = fc2 ? g.H1T + (int64_t)e * H * g.bp + frag_off(g.bp >> 4, col < Brows ? col : 0, 8 * lh)
                                   : g.XT + frag_off(g.bp >> 4, col < Brows ? col : 0, 8 * lh);
    const int nsteps = g.bp >> 4;
    constexpr int G = 8;
    u16x8 a0[G], b0[G], a1[G], b1[G];
    auto load = [&](u16x8 (&a)[G], u16x8 (&b)[G], int s0) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int s = s0 + u < nsteps ? s0 + u : nsteps - 1;   // clamped: redundant loads, never used
            a[u] = *reinterpret_cast<const u16x8 *>(AT + FRAG_STEP * s);
            b[u] = *reinterpret_cast<const u16x8 *>(BT + FRAG_STEP * s);
        }
    };
    // ---- everything that does not depend on the loss gradient goes in flight first: the first operand group and
    //      the optimizer state of this wave's 16 gradient elements (read again only in the epilogue)
    const int64_t off_w = fc2 ? g.off[2] : g.off[0];
    const int ldc = fc2 ? H : IN;
    float pv[16], mv[16], vv[16], tv[16];
    if (!head) {
        load(a0, b0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = by * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int64_t a = off_w + (int64_t)(j < H ? j : 0) * ldc + (col < Ncols ? col : 0);
            pv[r] = P[a]; mv[r] = Gr ? 0.0f : M[a]; vv[r] = Gr ? 0.0f : V[a];
            tv[r] = T ? T[a] : 0.0f;
        }
    }
    // ---- loss gradient of this net's rows -> LDS (the first fc2 tile of each net also reduces the loss terms)
    for (int b = g.n_rows + tid; b < g.bp; b += 256) tab[b] = 0.0f;
    // (the two-step table of the fp32 tiles -- inputs requested ahead of the prefetch, LDS-only barrier -- was tried here
    //  and measured slower: 33.6 vs 33.0 us per update, profiles/r5_bf16_forward.md)
    loss_fold_table(g.lf, e, tab, t == 0, red, t == 0 && e == 0);
    if (((g.fold.done && g.fold.td_logs) || g.fold.deferred_stats) && t == 0 && e == 0)
        log_fold_td_stats(g.fold, g.lf.tds, red);
    __syncthreads();
    WSTAMP(1);
    const ssac_adam_ctl ctl = *g.ctl;
    const float inv_bc2 = 1.0f / ctl.bc2_sqrt;
    float ss = 0.0f;
    bool early = false;
    if (head) {
        // ---- head layer (VALU): thread i owns W3[i]
        const int i = tid;
        float gw = 0.0f, gb = 0.0f;
        if (i < H) {
            const unsigned short *hbase = g.H2T + (int64_t)e * H * g.bp;
            auto hp_at = [&](int k) { return hbase + frag_off(g.bp >> 4, i, k); };   // 8 consecutive k are contiguous
            // (round 5: FOUR partial sums -- one dependent chain of bp fused multiply-adds per thread made the head workgroup
            //  the LAST one of the launch, 12 - 14 us against 10 - 11 for the GEMM tiles (tools/r5/bf16_timeline.py) -- and the
            //  row scales as 16-byte LDS reads)
            float gq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int b0_ = 0; b0_ < g.bp; b0_ += 32) {
                u16x8 hv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) hv[q] = *reinterpret_cast<const u16x8 *>(hp_at(b0_ + 8 * q < g.bp ? b0_ + 8 * q : 0));
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (b0_ + 8 * q < g.bp) {
                        const f4 c0 = *reinterpret_cast<const f4 *>(tab + b0_ + 8 * q), c1 = *reinterpret_cast<const f4 *>(tab + b0_ + 8 * q + 4);
#pragma unroll
                        for (int u = 0; u < 8; ++u) gq[q] += (u < 4 ? c0[u] : c1[u - 4]) * bf2f(hv[q][u]);
                    }
            }
            gw = (gq[0] + gq[1]) + (gq[2] + gq[3]);
        }
        if (tid < 64) {
            for (int b = tid; b < g.bp; b += 64) gb += tab[b];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) gb += __shfl_xor(gb, o, 64);
        }
        if (i < H) {
            const int64_t a = g.off[4] + i;
            if (Gr) {
                Gr[a] = gw;
            } else {
                float m = M[a], v = V[a];
                const float pn = adam_elem(P[a], gw, m, v, ctl, inv_bc2);
                M[a] = m; V[a] = v; P[a] = pn;
                S[g.sg.o3 + i] = f2bf(pn);
                if (T) { const float tn = T[a] * (1.0f - tau) + pn * tau; T[a] = tn; TS[g.sg.o3 + i] = f2bf(tn); }
            }
            ss += gw * gw;
        }
        if (tid == 0) {
            const int64_t a = g.off[5];
            if (Gr) {
                Gr[a] = gb;
            } else {
                float m = M[a], v = V[a];
                const float pn = adam_elem(P[a], gb, m, v, ctl, inv_bc2);
                M[a] = m; V[a] = v; P[a] = pn;
                if (T) T[a] = T[a] * (1.0f - tau) + pn * tau;
            }
            ss += gb * gb;
        }
    } else {
        const float *cs = tab + 8 * lh;
        f32x16 acc;
        zero_acc(acc);
        float colsum = 0.0f;   // bias gradient: sum_b c_b dz[b][row] from the scaled A fragments
        auto mm = [&](const u16x8 (&a)[G], const u16x8 (&b)[G], int s0) {
#pragma unroll
            for (int u = 0; u < G; ++u) {
                if (s0 + u < nsteps) {
                    const u16x8 as = scale_frag(a[u], cs + 16 * (s0 + u), colsum);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, as),
                                                                  __builtin_bit_cast(bf16x8, b[u]), acc, 0, 0, 0);
                }
            }
        };
        for (int s0 = 0; s0 < nsteps; s0 += 2 * G) {
            load(a1, b1, s0 + G);
            mm(a0, b0, s0);
            load(a0, b0, s0 + 2 * G);
            mm(a1, b1, s0 + G);
        }
        WSTAMP(2);
        colsum += __shfl_xor(colsum, 32, 64);   // the two k halves of row `row`
        // the bias element of this lane's row (the n-tile-0 waves with wn == 0 own it): loaded before the stores below
        const bool bias_lane = bxn == 0 && wn == 0 && lh == 0 && row < H;
        const int64_t ab = (fc2 ? g.off[3] : g.off[1]) + (row < H ? row : 0);
        const float pb = P[ab], mb0 = Gr ? 0.0f : M[ab], vb0 = Gr ? 0.0f : V[ab], tb0 = T ? T[ab] : 0.0f;
        // ---- gradient-norm partial first (it needs the gradients only): with the logs folded in, the arrival ticket is
        //      drawn BEFORE the optimizer stores, whose drain it must not wait for
        if (col < Ncols) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = by * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (j < H) ss += acc[r] * acc[r];
            }
        }
        if (bias_lane) ss += colsum * colsum;
        early = true;
        if (g.sumsq) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            __syncthreads();
            if (lane == 0) red[wave] = ss;
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_store(g.sumsq + (int64_t)e * g.sumsq_stride + t, red[0] + red[1] + red[2] + red[3],
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (g.fold.done) red[8] = log_fold_arrive(g.fold, gridDim.x) ? 1.0f : 0.0f;
            }
        }
        // ---- Adam epilogue straight from the accumulator: acc[r] = dW[by*64 + wm*32 + (r&3) + 8(r>>2) + 4 lh][col];
        //      the optimizer state was loaded at kernel start, so this is arithmetic + stores only
        if (Gr) {
            if (col < Ncols) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = by * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (j < H) Gr[off_w + (int64_t)j * ldc + col] = acc[r];
                }
            }
            if (bias_lane) Gr[ab] = colsum;
        } else if (col < Ncols) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u16x4 hq, tq;
                const int j0 = by * 64 + wm * 32 + 8 * q + 4 * lh;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u, r = 4 * q + u;
                    hq[u] = 0; tq[u] = 0;
                    if (j < H) {
                        const int64_t a = off_w + (int64_t)j * ldc + col;
                        const float gr = acc[r];
                        float m = mv[r], v = vv[r];
                        const float pn = adam_elem(pv[r], gr, m, v, ctl, inv_bc2);
                        M[a] = m; V[a] = v; P[a] = pn;
                        hq[u] = f2bf(pn);
                        if (fc2) S[g.sg.o2 + frag_off(H >> 4, j, col)] = hq[u];
                        else S[g.sg.o1 + frag_off(K1P >> 4, j, col)] = hq[u];
                        if (T) {
                            const float tn = tv[r] * (1.0f - tau) + pn * tau;
                            T[a] = tn;
                            tq[u] = f2bf(tn);
                            if (fc2) TS[g.sg.o2 + frag_off(H >> 4, j, col)] = tq[u];
                            else TS[g.sg.o1 + frag_off(K1P >> 4, j, col)] = tq[u];
                        }
                    }
                }
                if (fc2 && j0 + 3 < H) {   // W2^T shadow: 4 consecutive j (= k of W2^T, j0 % 4 == 0) of its row `col`
                    *reinterpret_cast<u16x4 *>(S + g.sg.o2t + frag_off(H >> 4, col, j0)) = hq;
                    if (T) *reinterpret_cast<u16x4 *>(TS + g.sg.o2t + frag_off(H >> 4, col, j0)) = tq;
                }
            }
        }
        if (bias_lane && !Gr) {
            float m = mb0, v = vb0;
            const float pn = adam_elem(pb, colsum, m, v, ctl, inv_bc2);
            M[ab] = m; V[ab] = v; P[ab] = pn;
            if (T) T[ab] = tb0 * (1.0f - tau) + pn * tau;
        }
    }
    WSTAMP(3);
    if (g.sumsq && !early) {   // (the head workgroup: its partial and ticket come after its few stores)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        __syncthreads();
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(g.sumsq + (int64_t)e * g.sumsq_stride + t, red[0] + red[1] + red[2] + red[3],
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g.fold.done) red[8] = log_fold_arrive(g.fold, gridDim.x) ? 1.0f : 0.0f;
        }
    }
    if (g.fold.done) {
        __syncthreads();
        if (red[8] != 0.0f && tid < 64) log_fold_finish(g.fold);
    } else if (g.fold.deferred_stats && g.fold.feed && blockIdx.x == 0 && tid == 0) {
        g.fold.feed->tick += 1;   // deferred finalisation: the update is over for the input ring
    }
#ifdef SSAC_LAB
    if (g.tl && blockIdx.x < 512) {
        __syncthreads();
        if (tid == 0) g.tl[1024 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

long long *g_bf_dbg = nullptr;

bool g_bf_regchain = true;
bool bf_ok(const ssac_mlp *n) {
    return n && n->hidden % 32 == 0 && n->hidden >= 32 && n->hidden <= 256 && n->out_dim >= 1 && n->out_dim <= 64 &&
           n->in_dim >= 1 && bf_lds_bytes(n->in_dim, n->hidden, n->out_dim) <= 160 * 1024;
}

void bf_fill(BfArgs &g, const ssac_mlp *nets, const uint16_t *shadow, const int32_t *ids, const float *X, int64_t ldx,
             int n_rows) {
    g.params = nets->params; g.net_stride = nets->net_stride;
    g.in_dim = nets->in_dim; g.hidden = nets->hidden; g.out_dim = nets->out_dim;
    ssac_mlp_layout(nets->in_dim, nets->hidden, nets->out_dim, g.off);
    g.shadow = shadow; g.sg = shadow_geom(nets->in_dim, nets->hidden, nets->out_dim);
    g.ids = ids; g.X = X; g.ldx = ldx; g.n_rows = n_rows; g.bp = (n_rows + 15) & ~15;
    g.dbg = g_bf_dbg; g.tl = g_ssac_timeline;
    g.xcd = (g_ssac_xcd & 8) ? 0 : 1;   // (ssac_xcd_order bit 3: the chained launches in hardware order)
}

template <typename K>
int raise_lds(K kernel, bool &done) {
    if (done) return 0;
    if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return ssac_fail("ssac_bf16: cannot raise the dynamic LDS limit");
    done = true;
    return 0;
}

}  // namespace

extern "C" int64_t ssac_bf16_layout(int in_dim, int hidden, int out_dim, int64_t offsets[4]) {
    const ShadowGeom g = shadow_geom(in_dim, hidden, out_dim);
    if (offsets) { offsets[0] = g.o1; offsets[1] = g.o2; offsets[2] = g.o2t; offsets[3] = g.o3; }
    return g.stride;
}

extern "C" int ssac_bf16_supported(const ssac_mlp *nets) { return bf_ok(nets) ? 1 : 0; }
// A/B switch of the large-batch forward (tools / tests): 1 (default) = the register-chained kernel where it applies,
// 0 = bf_stream_kernel everywhere
extern "C" int ssac_bf16_fwd_form(int form) { g_bf_regchain = form != 0; return 0; }
#ifdef SSAC_LAB   // (ssac_hip_test.h, lab hooks: the product library does not define the symbol)
extern "C" int ssac_bf16_debug_stamps(long long *dev_buf) {
    g_bf_dbg = dev_buf;
    return 0;
}
#endif

extern "C" int ssac_bf16_sync(const ssac_mlp *nets, uint16_t *shadow, void *stream) {
    if (!nets || !shadow) return ssac_fail("ssac_bf16_sync: missing argument");
    int64_t off[6];
    ssac_mlp_layout(nets->in_dim, nets->hidden, nets->out_dim, off);
    const ShadowGeom sg = shadow_geom(nets->in_dim, nets->hidden, nets->out_dim);
    const int64_t n = (int64_t)nets->hidden * (sg.k1p + nets->hidden + nets->out_dim);
    dim3 grid((unsigned)((n + 255) / 256 > 512 ? 512 : (n + 255) / 256), nets->n_nets);
    SSAC_LAUNCH(bf_sync_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float *)nets->params, nets->net_stride,
                nets->in_dim, nets->hidden, nets->out_dim, off[0], off[2], off[4], (unsigned short *)shadow, sg);
    return ssac_check_launch("bf16_sync");
}

extern "C" int ssac_bf16_polyak(const ssac_mlp *target, const ssac_mlp *source, float tau, uint16_t *target_shadow,
                                void *stream) {
    if (!target || !source || !target_shadow) return ssac_fail("ssac_bf16_polyak: missing argument");
    if (target->n_nets != source->n_nets || target->net_stride != source->net_stride || target->in_dim != source->in_dim ||
        target->hidden != source->hidden || target->out_dim != source->out_dim)
        return ssac_fail("ssac_bf16_polyak: arenas differ");
    int64_t off[6];
    const int64_t n = ssac_mlp_layout(target->in_dim, target->hidden, target->out_dim, off);
    const ShadowGeom sg = shadow_geom(target->in_dim, target->hidden, target->out_dim);
    dim3 grid((unsigned)((n + 255) / 256 > 256 ? 256 : (n + 255) / 256), target->n_nets);
    SSAC_LAUNCH(bf_polyak_kernel, grid, dim3(256), 0, (hipStream_t)stream, target->params, (const float *)source->params,
                target->net_stride, target->in_dim, target->hidden, target->out_dim, off[0], off[2], off[4], n, tau,
                (unsigned short *)target_shadow, sg);
    return ssac_check_launch("bf16_polyak");
}

extern "C" int ssac_bf16_mlp3_fwd(const ssac_mlp *nets, const uint16_t *shadow, const int32_t *net_ids, int n_sel,
                                  const float *X, int64_t ldx, int n_rows, float *Y, void *stream) {
    if (!bf_ok(nets) || !shadow) return ssac_fail("ssac_bf16_mlp3_fwd: shape not supported by the bf16 path");
    if (n_sel < 0 || n_sel > SSAC_MAX_NETS) return ssac_fail("ssac_bf16_mlp3_fwd: n_sel out of range");
    if (n_sel == 0 || n_rows <= 0) return 0;
    BfArgs g{};
    bf_fill(g, nets, shadow, net_ids, X, ldx, n_rows);
    g.Y = Y;
    // large batches of single-output critics: persistent workgroups that keep their weight fragments (bf_stream_kernel)
    const int tiles64 = (n_rows + 63) / 64;
    // (measured, N 10: the register-chained kernel's fixed cost -- 144 KB of weights into LDS per workgroup -- is paid back from
    //  ~6 000 rows on: 18.8 vs 18.7 us at B 6 144, 19.2 vs 25.8 at 8 192, 89 vs 149 at 65 536; profiles/r5_bf16_forward.md)
    if (nets->out_dim == 1 && Y && nets->hidden == RC_H && g.sg.k1p == 16 * RC_NS1 && tiles64 * n_sel >= 1024 && g_bf_regchain &&
        (int64_t)n_rows * ldx * 4 < (1LL << 32)) {
        // register-chained form (bf_regchain_kernel): one persistent workgroup per CU, its 8 waves take 32-row units in pairs
        const int per_net = std::max(1, std::min((tiles64 + 7) / 8, 256 / n_sel));
        static bool arc = false;
        if (raise_lds(bf_regchain_kernel, arc)) return 1;
        SSAC_LAUNCH(bf_regchain_kernel, dim3(per_net, n_sel), dim3(NTHR), RC_LDS, (hipStream_t)stream, g);
        return ssac_check_launch("bf16_mlp3_fwd (register-chained)");
    }
    if (nets->out_dim == 1 && Y && g.sg.k1p <= 64 && tiles64 * n_sel >= 512) {
        const size_t lds_s = 2 * (64 * (size_t)(g.sg.k1p + LPAD) + 2 * 64 * (size_t)(nets->hidden + LPAD)) + 4 * 3 * (size_t)nets->hidden + 64;
        // one resident workgroup per CU (the fragments + two accumulators + the fragment reads in flight take ~200
        // registers; held to 128 for two workgroups per CU the kernel spills and runs 3x slower)
        const int per_net = std::max(1, std::min(tiles64, 256 / n_sel));
        static bool a2 = false, a4 = false;
        if (g.sg.k1p <= 32) {
            if (raise_lds(bf_stream_kernel<2>, a2)) return 1;
            SSAC_LAUNCH(bf_stream_kernel<2>, dim3(per_net, n_sel), dim3(NTHR), lds_s, (hipStream_t)stream, g, tiles64);
        } else {
            if (raise_lds(bf_stream_kernel<4>, a4)) return 1;
            SSAC_LAUNCH(bf_stream_kernel<4>, dim3(per_net, n_sel), dim3(NTHR), lds_s, (hipStream_t)stream, g, tiles64);
        }
        return ssac_check_launch("bf16_mlp3_fwd (stream)");
    }
    static bool attr = false;
    if (raise_lds(bf_mlp_kernel<MODE_PLAIN>, attr)) return 1;
    const size_t lds = bf_lds_bytes(nets->in_dim, nets->hidden, nets->out_dim);
    SSAC_LAUNCH(bf_mlp_kernel<MODE_PLAIN>, dim3((n_rows + TM - 1) / TM, n_sel), dim3(NTHR), lds, (hipStream_t)stream, g);
    return ssac_check_launch("bf16_mlp3_fwd");
}

extern "C" int ssac_bf16_chain_update(const ssac_mlp *actor, const uint16_t *actor_shadow, const float *Xa, int64_t ldxa,
                                      int n_rows, const float *eps, float log_std_lo, float log_std_hi, float *x1sa,
                                      int64_t ld_x1, int64_t act_col0, float *logp, const ssac_rng *rng,
                                      const ssac_mlp *targets, const uint16_t *target_shadow, const int32_t *net_ids,
                                      int n_sel, float *Qt, const ssac_mlp *critics, const uint16_t *critic_shadow,
                                      const float *Xc, int64_t ldxc, float *Q, uint16_t *H1T, uint16_t *H2T,
                                      uint16_t *DZ2uT, uint16_t *DZ1uT, uint16_t *XT, const ssac_gather *gather,
                                      const ssac_deferred_logs *deferred, unsigned long long *handoff, void *stream) {
    if (!eps && !rng) return ssac_fail("ssac_bf16_chain_update: neither eps nor an rng stream given");
    if (!bf_ok(actor) || (actor->out_dim & 1) || !bf_ok(targets) || !bf_ok(critics))
        return ssac_fail("ssac_bf16_chain_update: shape not supported by the bf16 path");
    if (critics->out_dim != 1 || targets->out_dim != 1) return ssac_fail("ssac_bf16_chain_update: single-output critics only");
    if (n_sel <= 0 || n_sel > SSAC_MAX_NETS) return ssac_fail("ssac_bf16_chain_update: n_sel out of range");
    if (!actor_shadow || !target_shadow || !critic_shadow || !x1sa || !logp || !Qt || !Q || !H1T || !H2T || !DZ2uT ||
        !DZ1uT || !XT)
        return ssac_fail("ssac_bf16_chain_update: missing buffer");
    if (act_col0 != actor->in_dim || targets->in_dim != actor->in_dim + actor->out_dim / 2)
        return ssac_fail("ssac_bf16_chain_update: [s'|a'] layout does not match the networks");
    if (n_rows <= 0) return 0;
    BfArgs ga{}, gr{}, gt{}, gc{};
    bf_fill(ga, actor, actor_shadow, nullptr, Xa, ldxa, n_rows);
    ga.eps = eps; ga.lo = log_std_lo; ga.hi = log_std_hi;
    if (rng) ga.rng = RngArgs{rng->seed, rng->counter, rng->offset};
    ga.act_dst = x1sa; ga.ld_act = ld_x1; ga.act_col0 = act_col0; ga.logp = logp;
    bf_fill(gt, targets, target_shadow, net_ids, x1sa, ld_x1, n_rows);
    gt.Y = Qt;
    bf_fill(gc, critics, critic_shadow, nullptr, Xc, ldxc, n_rows);
    gc.Y = Q; gc.H1T = H1T; gc.H2T = H2T; gc.DZ2T = DZ2uT; gc.DZ1T = DZ1uT; gc.XT = XT;
    if (gather) {
        if (gather->s_elems != actor->in_dim || gather->s_elems + gather->a_elems != critics->in_dim)
            return ssac_fail("ssac_bf16_chain_update: gather sizes do not match the networks");
        if (!gather->s || !gather->s1 || !gather->act || !gather->rew || !gather->done || !gather->xsa ||
            gather->x1sa != x1sa || !gather->rew_out || !gather->done_out || (!gather->idx && !gather->feed))
            return ssac_fail("ssac_bf16_chain_update: incomplete ssac_gather");
        if (gather->feed && gather->n_logs > NTHR) return ssac_fail("ssac_bf16_chain_update: log block too large");
        ga.gth = *gather; ga.gth_role = 1;
        gc.gth = *gather; gc.gth_role = 2;
    } else if (!Xa || !Xc) {
        return ssac_fail("ssac_bf16_chain_update: Xa / Xc missing");
    }
    gr = ga;
    if (gather) gr.gth_role = 3;
    if (gather && gather->feed && gather->ids_word >= 0) { gt.gth = *gather; gt.gth_role = 4; }
    size_t lds = bf_lds_bytes(actor->in_dim, actor->hidden, actor->out_dim);
    const size_t lt = bf_lds_bytes(targets->in_dim, targets->hidden, targets->out_dim);
    const size_t lc = bf_lds_bytes(critics->in_dim, critics->hidden, critics->out_dim);
    if (lt > lds) lds = lt;
    if (lc > lds) lds = lc;
    static bool attr = false;
    if (raise_lds(bf_chain_kernel, attr)) return 1;
    const int gx = (n_rows + TM - 1) / TM;
    const int tiles_t = gx * n_sel;
    DeferredLogsArgs dl{};
    const int dl_on = (deferred && deferred->feed) ? 1 : 0;
    if (dl_on)
        dl = DeferredLogsArgs{deferred->partials, deferred->n_nets, deferred->sumsq, deferred->n_ss, deferred->td_stats,
                              deferred->td_off, deferred->n_rows, deferred->denom, deferred->feed};
    const int A_ = actor->out_dim / 2;
    if (handoff && A_ <= BF_HO_AMAX) {
        // producer / consumer form (bf_chain_pc_kernel); tags as in ssac_chain_update: the input ring's update counter for a
        // recorded launch (the caller passes no buffer for a recording without one), a host counter with bit 31 set else
        static unsigned launch_no = 0;
        Handoff ho{handoff, nullptr, 0u, actor->in_dim, A_, 1, nullptr, nullptr};
        const ssac_feed *fd = (gather && gather->feed) ? gather->feed : ((deferred && deferred->feed) ? deferred->feed : nullptr);
        if (fd) { ho.tick = reinterpret_cast<const long long *>(&fd->tick); ho.base = 1u; }
        else ho.base = 0x80000000u | (++launch_no & 0x7fffffffu);
        ga.ho = ho;
        gt.ho = ho;
        if (gather) { gt.gth = *gather; gt.gth_role = 5; }
        static bool attr_pc = false;
        if (raise_lds(bf_chain_pc_kernel, attr_pc)) return 1;
        SSAC_LAUNCH(bf_chain_pc_kernel, dim3(gx + tiles_t + gx * critics->n_nets + dl_on), dim3(NTHR), lds, (hipStream_t)stream,
                    ga, gt, gc, gx, tiles_t, gx, dl, dl_on);
        return ssac_check_launch("bf16_chain_pc");
    }
    SSAC_LAUNCH(bf_chain_kernel, dim3(tiles_t + gx * critics->n_nets + dl_on), dim3(NTHR), lds, (hipStream_t)stream, ga, gr,
                gt, gc, tiles_t, gx, dl, dl_on);
    return ssac_check_launch("bf16_chain");
}

extern "C" int ssac_bf16_wgrad_tiles(const ssac_mlp *nets) {
    if (!nets) return -1;
    const int t = (nets->hidden + 63) / 64;
    return t * t + t * ((nets->in_dim + 63) / 64) + 1;
}

extern "C" int ssac_bf16_wgrad_lossfold(const ssac_mlp *nets, uint16_t *shadow, const uint16_t *XT, const uint16_t *H1T,
                                        const uint16_t *H2T, const uint16_t *DZ2uT, const uint16_t *DZ1uT, const float *Q,
                                        const float *td, const ssac_td_spec *lazy_td, const float *weight,
                                        const ssac_popart *popart, int pop, float denom,
                                        float *partials, int n_rows, float *adam_m, float *adam_v,
                                        const ssac_adam_ctl *ctl, float *grads, float *sumsq, int64_t sumsq_net_stride,
                                        float *target, uint16_t *target_shadow, float tau, const ssac_logfold *logfold,
                                        void *stream) {
    if (!bf_ok(nets) || nets->out_dim != 1) return ssac_fail("ssac_bf16_wgrad_lossfold: single-output critics only");
    if (!shadow || !XT || !H1T || !H2T || !DZ2uT || !DZ1uT || !Q || !partials || (!td && !lazy_td) || !adam_m || !adam_v || !ctl)
        return ssac_fail("ssac_bf16_wgrad_lossfold: missing argument");
    if (target && !target_shadow) return ssac_fail("ssac_bf16_wgrad_lossfold: Polyak needs the target's shadow");
    if (n_rows <= 0) return 0;
    if (n_rows > 8192) return ssac_fail("ssac_bf16_wgrad_lossfold: more than 8192 rows");
    BfWgradArgs g{};
    g.params = nets->params; g.net_stride = nets->net_stride; g.in_dim = nets->in_dim; g.hidden = nets->hidden;
    ssac_mlp_layout(nets->in_dim, nets->hidden, nets->out_dim, g.off);
    g.shadow = shadow; g.sg = shadow_geom(nets->in_dim, nets->hidden, nets->out_dim);
    g.H1T = H1T; g.H2T = H2T; g.DZ2T = DZ2uT; g.DZ1T = DZ1uT; g.XT = XT;
    g.n_rows = n_rows; g.bp = (n_rows + 15) & ~15; g.n_nets = nets->n_nets;
    g.am = adam_m; g.av = adam_v; g.ctl = ctl; g.target = target; g.tshadow = target_shadow; g.tau = tau;
    g.grads = grads;
    if (grads && (target || (logfold && logfold->late_word)))
        return ssac_fail("ssac_bf16_wgrad_lossfold: gradient mode updates nothing (no Polyak step in the same launch)");
    g.sumsq = sumsq; g.sumsq_stride = sumsq_net_stride;
    g.lf.q = Q; g.lf.td = td; if (lazy_td) g.lf.tds = *lazy_td;
    g.lf.weight = weight; g.lf.popart = popart; g.lf.pop = pop; g.lf.denom = denom; g.lf.partials = partials;
    g.lf.n_rows = n_rows;
    const int t = (nets->hidden + 63) / 64;
    g.tiles2 = t * t; g.tiles1 = t * ((nets->in_dim + 63) / 64);
    g.dbg = g_bf_dbg; g.tl = g_ssac_timeline;
    g.xcd = (g_ssac_xcd >> 1) & 1;
    if (logfold && logfold->done_counter) {
        if (!sumsq) return ssac_fail("ssac_bf16_wgrad_lossfold: the folded logs need the sumsq slots");
        g.fold = LogFoldArgs{logfold->done_counter, logfold->logs, lazy_td ? logfold->td_logs : nullptr, logfold->feed,
                             nullptr, partials, nets->n_nets, sumsq, (int)(nets->n_nets * sumsq_net_stride), n_rows, denom};
    } else if (logfold && logfold->deferred_stats) {
        if (!logfold->feed || !lazy_td) return ssac_fail("ssac_bf16_wgrad_lossfold: deferred logs need a feed and the in-launch TD target");
        g.fold = LogFoldArgs{};
        g.fold.feed = logfold->feed;
        g.fold.deferred_stats = logfold->deferred_stats;
        g.fold.n_rows = n_rows;
    }
    if (logfold && logfold->late_word) {
        if (!target) return ssac_fail("ssac_bf16_wgrad_lossfold: the late-bound Polyak needs the target arena");
        g.late_word = logfold->late_word;
    }
    const size_t lds = sizeof(float) * (g.bp + 16);
    static bool attr = false;
    if (raise_lds(bf_wgrad_kernel, attr)) return 1;
    SSAC_LAUNCH(bf_wgrad_kernel, dim3((g.tiles2 + g.tiles1 + 1) * nets->n_nets), dim3(256), lds, (hipStream_t)stream, g);
    return ssac_check_launch("bf16_wgrad");
}
