// Ensemble GEMM for the actor/critic MLP layers on gfx950 (MI355X), exact fp32.
//
// One launch computes, for every selected net e of a packed ensemble,
//     C_e (M x N) = op(A_e) * op(B_e)      in fp32 on the matrix cores
// with v_mfma_f32_32x32x2_f32 (exact fp32: bitwise a k-ordered fmaf chain, so the result
// is comparable with the reference's fp32 torch path at fp32 round-off), and finishes
// with a fused epilogue: bias+ReLU (forward), ReLU-mask (backward-data), or Adam (+Polyak)
// directly on the weight tile (backward-weights) so gradients never round-trip HBM.
//
// Tiling: a 256-thread workgroup (4 waves, one per SIMD) owns a 64x64 tile of C as 2x2
// waves of one 32x32 MFMA accumulator each (16 VGPRs).  K is consumed in chunks of 32
// staged through LDS; the next chunk's global loads are issued before the MFMAs of the
// current one so HBM/L2 latency hides under the 64-cycle MFMAs.  Operand fragments are
// single ds_read_b32 per MFMA operand; both LDS layouts below are bank-conflict free for
// the 32-lane groups ds_read_b32/ds_write_b32 are serviced in:
//   K-contiguous source  -> Xs[row][k], row stride 33 floats   (bank = (row + k) % 32)
//   row-contiguous source-> Xt[k][row], row stride 64 floats   (consecutive lanes, same k)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "ssac_internal.h"
#include "ssac_head_wgrad.h"
#include "ssac_critic_logs.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 64, BN = 64, BK = 32, NTHREADS = 256;
constexpr int LDS_KC = 33;            // stride of a K-contiguous tile row
constexpr int LDS_RC = 64;            // stride of a row-contiguous tile row
constexpr int TILE_FLOATS = 64 * LDS_KC;  // 2112 >= 32*64

enum { EPI_STORE = 0, EPI_BIAS = 1, EPI_BIAS_RELU = 2, EPI_MASK = 3, EPI_ADAM = 4, EPI_GRAD = 5 };

struct GemmArgs {
    const float *A; int64_t lda, sA; int idsA;
    const float *B; int64_t ldb, sB; int idsB;
    float *C; int64_t ldc, sC; int idsC;
    int M, N, K;
    const int32_t *ids;
    // EPI_BIAS*: bias[n], per-net stride sBias, indexed like B (params side)
    const float *bias; int64_t sBias;
    // EPI_MASK: C *= (mask > 0)
    const float *mask; int64_t ldmask, sMask;
    // EPI_ADAM / EPI_GRAD: C is the weight (M x N, ldc); same indexing for m, v, grads, target.
    float *am, *av;          // Adam moments of the weight
    float *pb, *bm, *bv;     // bias param + moments (M entries), per-net stride sC
    float *gw, *gb;          // gradient outputs (EPI_GRAD)
    float *tw, *tb;          // Polyak target weight / bias (nullable)
    float tau;
    const ssac_adam_ctl *ctl;
    float *sumsq;            // partial sums of g^2: sumsq[e*sumsq_stride + tile] (nullable)
    int64_t sumsq_stride;
    int vec;                 // bit0: A rows are 16-byte aligned, bit1: B rows (row-contiguous operands)
    long long *dbg;          // optional s_memtime stamps of workgroup (0,0,0), thread 0
    int grid_x, grid_y;      // tiles along N and M (filled by the launcher)
    const float *rowscale;   // TN mode: A[k][m] is multiplied by rowscale[e*sRow + k] while it is loaded (null: 1)
    int64_t sRow;
    int Ktot;                // > 0: batch entry e covers k in [e*K, min((e+1)*K, Ktot)) (split-K over blocks)
    int64_t sGb;             // EPI_GRAD: bias-gradient stride per batch entry (0 = same as sC)
    int xcd;                 // XCD-contiguous tile order (ssac_internal.h)
    // TN mode, vector loads: A is not read as stored but REBUILT from it -- A[k][m] = a_sign_w[e][m] where the stored
    // value is positive, else 0 (dz2u = W3 (.) [h2 > 0] from the saved h2: the chained launch then does not write dz2u,
    // 5 MB per update at the metric shape).  a_sign_w: (n_nets x M) SNAPSHOT of the head rows the chained launch took --
    // not the arena's W3, which the head workgroups of this very launch are updating
    const float *a_sign_w;
    int no_bias;             // the bias gradient of this layer is somebody else's job (the head workgroups: HeadWgradArgs::b2_w3s)
};

__device__ __forceinline__ int64_t batch_off(const int32_t *ids, int use_ids, int e, int64_t stride) {
    return (int64_t)((use_ids && ids) ? ids[e] : e) * stride;
}

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load of a dword-aligned address (gfx950: no wider requirement)

// ---- global -> register staging (8 floats per thread per operand per chunk, held in the SAME two f4 registers the
//      16-byte loader uses -- one staging set per operand, whichever load path a launch takes), branch-free:
//      out-of-range lanes read element 0 (always valid) and select 0.
template <bool KCONTIG>
__device__ __forceinline__ void load_chunk(f4 (&r)[2], const float *__restrict__ S, int64_t ld,
                                           int R0, int R, int k0, int K, int tid) {
    if (KCONTIG) {
        // S is (R x K) row-major: thread -> k = tid&31, rows (tid>>5) + 8p
        const int kk = tid & 31, rr = tid >> 5;
        const bool kok = (k0 + kk) < K;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = R0 + rr + 8 * p;
            const bool ok = kok && row < R;
            const float v = S[ok ? (int64_t)row * ld + k0 + kk : 0];
            r[p >> 2][p & 3] = ok ? v : 0.0f;
        }
    } else {
        // S is (K x R) row-major: thread -> row = tid&63, k = (tid>>6) + 4p
        const int rr = tid & 63, kk = tid >> 6;
        const bool rok = (R0 + rr) < R;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int k = k0 + kk + 4 * p;
            const bool ok = rok && k < K;
            const float v = S[ok ? (int64_t)k * ld + R0 + rr : 0];
            r[p >> 2][p & 3] = ok ? v : 0.0f;
        }
    }
}

template <bool KCONTIG>
__device__ __forceinline__ void store_chunk(const f4 (&r)[2], float *__restrict__ Xs, int tid) {
    if (KCONTIG) {
        const int kk = tid & 31, rr = tid >> 5;
#pragma unroll
        for (int p = 0; p < 8; ++p) Xs[(rr + 8 * p) * LDS_KC + kk] = r[p >> 2][p & 3];
    } else {
        const int rr = tid & 63, kk = tid >> 6;
#pragma unroll
        for (int p = 0; p < 8; ++p) Xs[(kk + 4 * p) * LDS_RC + rr] = r[p >> 2][p & 3];
    }
}

template <bool KCONTIG>
__device__ __forceinline__ float frag(const float *__restrict__ Xs, int row, int k) {
    return KCONTIG ? Xs[row * LDS_KC + k] : Xs[k * LDS_RC + row];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Adam on one element, torch.optim.Adam arithmetic (lerp / addcmul / addcdiv forms).
__device__ __forceinline__ float adam_elem(float p, float g, float &m, float &v,
                                           const ssac_adam_ctl &c) {
    if (c.weight_decay != 0.0f) g = g + c.weight_decay * p;
    m = m + (1.0f - c.beta1) * (g - m);
    v = v * c.beta2 + (1.0f - c.beta2) * g * g;
#if defined(SSAC_LAB) && defined(SSAC_EXP_FAST_ADAM)
    // (measurement build only: the bound of an epilogue on v_sqrt / v_rcp instead of the IEEE sequences -- NOT torch's bits)
    const float denom = __builtin_amdgcn_sqrtf(v) * __builtin_amdgcn_rcpf(c.bc2_sqrt) + c.eps;
    return p - c.step_size * (m * __builtin_amdgcn_rcpf(denom));
#else
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    return p - c.step_size * (m / denom);
#endif
}

// Gradient-norm partial of one weight-gradient tile.  A layer owns ceil(M/32) x ceil(N/32) slots per net
// (ssac_wgrad_tiles): the 32 x 32 tiles of the latency variant write one each (wgrad_small_body), a 64 x 64 tile writes
// the first of the (up to) four it covers and zeroes the others -- the slots of a layer sum to the same value
// whichever variant ran last.  (agent scope: with the logs folded into the launch the reader sits on another XCD)
// (the stores themselves: `sumsq_finish` in ens_gemm_body -- four lanes of one store instruction)

// KS = number of K-split groups of 4 waves inside the workgroup.  Group kg consumes chunks
// kg, kg+KS, ... with its own LDS staging; the partial tiles are summed through LDS before the
// epilogue.  It buys latency hiding (KS waves per SIMD) for launches with few tiles and a long K
// (weight gradients: K = batch), where most CUs would otherwise run one wave per SIMD.
#ifdef SSAC_LAB
#define GSTAMP(i) do { if (g.dbg && bx == 0 && by == 0 && bz == 0 && threadIdx.x == 0) g.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif

// Row-contiguous operand (K x R row-major, e.g. dY and X of the weight-gradient GEMM), 16-byte
// loads: thread -> 4 consecutive rows r4 = 4*(tid&15), k = (tid>>4) + 16q.  The source pointers
// are built once and advanced by a uniform stride; the LDS image is Xt[k*64 + r] (b128 stores).
struct RcVecLoader {
    const float *base;  // workgroup-uniform operand base (an SGPR pair: the loads use saddr + 32-bit lane offsets)
    uint32_t off;       // float offset of row kfirst + kk, columns R0 + r4 .. + 3 of this thread (second row: + ld16)
    uint32_t ld16;      // 16 rows, in floats (uniform)
    f4 v[2];
    f4 sw;              // sign mode (GemmArgs::a_sign_w): the weights of this thread's 4 columns
    bool sign;
    int kk, r4;
    int klast;          // k0 of the rows held in v (late row scale: applied when the rows are stored)
    bool rok;
    bool any_ragged;    // R % 4 != 0: some thread's 4 columns straddle the matrix edge
    int nv;             // how many of this thread's 4 columns exist (4: all; 1..3: the straddling vector)
    uint32_t back;      // 4 - nv for the straddling vector (its load starts that many elements earlier), else 0
    __device__ __forceinline__ void init(const float *S, int64_t ld, int R0, int R, int kfirst, int tid) {
        sign = false;
        r4 = (tid & 15) * 4;
        kk = tid >> 4;
        base = S;
        rok = (R0 + r4) < R;
        nv = rok ? min(4, R - (R0 + r4)) : 4;
        any_ragged = __builtin_amdgcn_readfirstlane((R & 3) != 0 ? 1 : 0) != 0;
        back = (uint32_t)(4 - nv);
        ld16 = (uint32_t)(16 * ld);
        off = rok ? (uint32_t)((kfirst + kk) * ld + R0 + r4) : 0u;
        klast = 0;
    }
    // predicated-off lanes read the (always valid, 16-byte aligned) first element of the operand
    // KFULL: K is a multiple of the chunk depth (the lean kernel's launch condition): no row of a chunk lies beyond K
    // ... and the lanes whose columns do not exist are NOT zeroed there: a dead column of A or B only ever reaches output
    // rows / columns beyond the matrix, which every epilogue guards (stores, bias sums, gradient-norm partials) -- so the
    // lean loop carries no per-lane predicate at all: load, [rotate a ragged vector], [scale], store.
    template <bool KFULL = false>
    __device__ __forceinline__ void load(int k0, int K, uint32_t adv, const float *scale = nullptr) {
        if constexpr (KFULL) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = k0 + kk + 16 * q;
                f4 x = *reinterpret_cast<const f4u *>(base + (size_t)(off + q * ld16) - back);
                if (any_ragged) {   // (uniform) rotate the straddling lane's vector left by `back`: x[c] = y[c + back]
                    const bool b1 = (back & 1u) != 0, b2 = (back & 2u) != 0;
                    const f4 t = {b1 ? x[1] : x[0], b1 ? x[2] : x[1], b1 ? x[3] : x[2], b1 ? x[0] : x[3]};
                    x = (f4){b2 ? t[2] : t[0], b2 ? t[3] : t[1], b2 ? t[0] : t[2], b2 ? t[1] : t[3]};
                }
                v[q] = scale ? x * scale[k] : x;
            }
            klast = k0;
            off += adv;
            return;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = k0 + kk + 16 * q;
            const bool ok = rok && k < K;
            const uint32_t o = ok ? off + q * ld16 : 0u;
            // (the general kernels take this loader for 16-byte aligned rows of a multiple of 4 floats only: GemmArgs::vec)
            const f4 x = *reinterpret_cast<const f4 *>(base + (size_t)o);
            const float sc = scale ? scale[ok ? k : 0] : 1.0f;
            v[q] = ok ? x * sc : (f4){0.f, 0.f, 0.f, 0.f};
        }
        klast = k0;
        off += adv;
    }
    // late: row scales in LDS (loss_fold_table), applied here so the loads need not wait for the table
    template <bool KFULL = false>
    __device__ __forceinline__ void store(float *Xt, const float *late = nullptr, int K = 0) const {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = klast + kk + 16 * q;
            f4 x = v[q];
            if (sign) {
#pragma unroll
                for (int c = 0; c < 4; ++c) x[c] = x[c] > 0.0f ? sw[c] : 0.0f;
            }
            *reinterpret_cast<f4 *>(Xt + (kk + 16 * q) * LDS_RC + r4) =
                late ? x * late[KFULL ? k : ((rok && k < K) ? k : 0)] : x;
        }
    }
};

// late_rs / lf / lf_mode: the merged weight-gradient launch with the loss gradient folded in.  lf_mode 1: the fold's
// inputs are requested BEFORE the first operand chunk (loss_fold_issue; in-order returns: they arrive first) and the
// LDS table `late_rs` of per-row scales is finished with the operand chunk in flight (loss_fold_finish + an LDS-only
// barrier); lf_mode 2: the one-step table with the net's loss statistics (loss_fold_table; the first fc2 tile of a net
// in multi-round launches); 0: no fold.  (lf is passed down as a plain reference to the kernel argument: a struct
// holding that reference made the compiler copy the whole argument block to scratch.)
// fold / last: the update's logs are folded into this launch (LogFoldArgs): thread 0 draws the arrival ticket right after
// the workgroup's gradient-norm partial is stored -- BEFORE the optimizer stores, whose drain it must not wait for --
// and reports through *last whether this workgroup arrived last.
// Late-bound Polyak (include/ssac_hip.h): on = the target update of this launch waits for the decision the update's
// first launch left in feed->late_word (tau bits, 0 = no soft_update followed this update)
struct LateTau { bool on; uint32_t bits; };

// VECONLY: both operands take the 16-byte loader (decided per launch on the host): the element-wise loaders are not
// compiled into the K loop at all.  They used to sit in it behind uniform branches -- never taken at the metric shape,
// but the loop body was ~700 instructions long and its registers were allocated for the worst path.  One instantiation
// of this body per KERNEL (a second one inside the same kernel makes the compiler copy the argument block to scratch),
// hence the kernel carries the flag.
template <bool A_KC, bool B_KC, int EPI, int KS, bool VECONLY = false>
__device__ __forceinline__ void ens_gemm_body(const GemmArgs &g, float *lds, int bx, int by, int bz,
                                              float *late_rs, const LossFoldArgs &lf, int lf_mode,
                                              const LogFoldArgs &fold, int &last,
                                              const LateTau &lt = LateTau{false, 0u}) {
    const int tid_all = threadIdx.x;
    // (K-group and wave index are the same in every lane of a wave: said so explicitly, everything derived from them --
    // the tile quadrant, "this wave sums the bias", "this wave's columns do not exist" -- is a scalar branch instead of
    // an exec-masked region inside the K loop)
    // (lean kernel only: in the general kernels the same hint made the NT forward GEMM of the hidden-1024 critics twice
    // as slow -- 34 -> 62 us, found in the pixel configurations' kernel trace)
    const int kg = VECONLY ? __builtin_amdgcn_readfirstlane(tid_all >> 8) : (tid_all >> 8), tid = tid_all & 255;
    // per K-group: two staging buffers (double buffering), each [A tile | B tile]
    float *buf0 = lds + kg * (4 * TILE_FLOATS);
    float *buf1 = buf0 + 2 * TILE_FLOATS;
    float *red = lds + KS * (4 * TILE_FLOATS);  // 64*KS floats: bias-grad / sumsq scratch

    const int lane = tid & 63, wave = VECONLY ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    // Quadrant of the 64 x 64 tile this wave multiplies.  A wave's SIMD is its index within the K-group, so the mapping
    // is ROTATED by the K-group: the two waves that own the left column block sit on SIMDs {0, 2}, {3, 1}, {2, 0}, {1, 3}
    // for K-groups 0..3.  When the right block does not exist (a weight gradient with <= 32 columns: fc1 of a
    // low-dimensional observation, 23 of 64 tile columns at the metric shape) its waves skip their fragment reads and
    // MFMAs, and the remaining half of the matrix work is spread over all four SIMDs instead of loading two and idling
    // two: the fc1 tiles -- the slowest workgroups of the merged launch until then -- finish ~3 us before the fc2 tiles
    // (tools/wg_timeline.py).
    // (weight gradients only -- the TN products; the forward / backward-data GEMMs keep the plain mapping)
    constexpr bool TN_ROT = !A_KC && !B_KC;
    const int wq = TN_ROT ? ((wave + kg) & 3) : wave;
    const int wm = wq >> 1, wn = wq & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int e = bz;
    const int m0 = by * BM, n0 = bx * BN;
    const bool dead_cols = TN_ROT && (n0 + wn * 32) >= g.N;   // (wave-uniform) this wave's 32 columns lie beyond the matrix

    const float *A = g.A + batch_off(g.ids, g.idsA, e, g.sA);
    const float *B = g.B + batch_off(g.ids, g.idsB, e, g.sB);
    const int64_t coff = batch_off(g.ids, g.idsC, e, g.sC);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;

    float bias_acc = 0.0f;  // TN mode, column sums of A (bias gradient), threads < 64 of n-tile 0
    const bool want_bias_grad = (EPI == EPI_ADAM || EPI == EPI_GRAD) && bx == 0 && !g.no_bias;
    const int Kloc = g.Ktot > 0 ? max(0, min(g.K, g.Ktot - e * g.K)) : g.K;
    const int nchunks = (Kloc + BK - 1) / BK;
#if defined(SSAC_LAB) && defined(SSAC_EXP_WGRAD_SKIP_ITER)
    const int iters = (nchunks + KS - 1) / KS - ((TN_ROT && g.N > 32) ? 1 : 0);   // (measurement build, WRONG results: an fc2 tile without its last K iteration)
#else
    const int iters = (nchunks + KS - 1) / KS;
#endif
    constexpr bool TN = !A_KC && !B_KC;
    // 16-byte loads need 16-byte aligned rows on both operands (checked per launch on the host)
    // ... and a batch entry's operand must span < 2^31 floats (32-bit lane offsets in RcVecLoader)
    static_assert(!VECONLY || TN, "the 16-byte loader reads row-contiguous operands");
    const bool vecA = VECONLY ? true : (TN && (g.vec & 1) && (int64_t)(Kloc + 1) * g.lda < (1LL << 31));
    const bool vecB = VECONLY ? true : (TN && (g.vec & 2) && (int64_t)(Kloc + 1) * g.ldb < (1LL << 31));

    RcVecLoader va, vb;
    if (vecA) {
        va.init(A, g.lda, m0, g.M, kg * BK, tid);
        if (g.a_sign_w) {
            va.sign = true;
            va.sw = va.rok ? *reinterpret_cast<const f4 *>(g.a_sign_w + (int64_t)e * g.M + m0 + va.r4) : (f4){0.f, 0.f, 0.f, 0.f};
        }
    }
    if (vecB) vb.init(B, g.ldb, n0, g.N, kg * BK, tid);
    // (lane offsets are 32-bit: one batch entry's operand spans < 2^32 floats -- the launchers check it)
    const uint32_t advA = (uint32_t)(KS * BK * g.lda), advB = (uint32_t)(KS * BK * g.ldb);

    const float *rscale = (TN && g.rowscale && !late_rs) ? g.rowscale + (int64_t)e * g.sRow : nullptr;
    int ra_k0 = 0;
    auto loadA = [&](int k0) {
        if (vecA) { va.template load<VECONLY>(k0, Kloc, advA, rscale); return; }
        ra_k0 = k0;
        load_chunk<A_KC>(va.v, A, g.lda, m0, g.M, k0, Kloc, tid);
        if (TN && rscale) {  // (K x R) layout: this thread's 8 values sit at k = k0 + (tid >> 6) + 4 p
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int k = k0 + (tid >> 6) + 4 * p;
                va.v[p >> 2][p & 3] *= rscale[k < Kloc ? k : 0];
            }
        }
    };
    auto loadB = [&](int k0) { if (vecB) vb.template load<VECONLY>(k0, Kloc, advB); else load_chunk<B_KC>(vb.v, B, g.ldb, n0, g.N, k0, Kloc, tid); };
    auto storeA = [&](float *d) {
        if (vecA) { va.template store<VECONLY>(d, TN ? late_rs : nullptr, Kloc); return; }
        if (TN && late_rs) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int k = ra_k0 + (tid >> 6) + 4 * p;
                va.v[p >> 2][p & 3] *= late_rs[k < Kloc ? k : 0];
            }
        }
        store_chunk<A_KC>(va.v, d, tid);
    };
    auto storeB = [&](float *d) { if (vecB) vb.template store<VECONLY>(d); else store_chunk<B_KC>(vb.v, d, tid); };

    // ---- optimizer state of this thread's 4 tile elements, requested during the LAST K chunk (the operand staging
    //      registers are free by then) so that the epilogue finds p / m / v [/ target] in registers instead of opening
    //      with a round trip to HBM / the Infinity Cache: 16-byte form, one float4 per thread (KS = 4) only.
    constexpr int NT_ALL_ = NTHREADS * KS;
    constexpr bool PREFETCH_OPT = EPI == EPI_ADAM && (BM * BN / 4) / NT_ALL_ == 1;
    const bool vecC = (EPI == EPI_ADAM || EPI == EPI_GRAD) && (g.N & 3) == 0 && (g.ldc & 3) == 0 && (coff & 3) == 0 &&
                      ((((uintptr_t)g.C | (uintptr_t)g.am | (uintptr_t)g.av | (uintptr_t)g.tw | (uintptr_t)g.gw) & 15) == 0);
    const bool pol_ = g.tw != nullptr && (!lt.on || lt.bits != 0u);
    f4 pf_p, pf_m, pf_v, pf_t;
    bool pf_done = false;
    auto opt_prefetch = [&]() {
        if (!vecC) return;
        const int row = tid_all >> 4, col = (tid_all & 15) * 4;
        const bool ok = (m0 + row) < g.M && (n0 + col) < g.N;
        const int64_t a = ok ? coff + (int64_t)(m0 + row) * g.ldc + n0 + col : coff;
        pf_p = *reinterpret_cast<const f4 *>(g.C + a);
        pf_m = *reinterpret_cast<const f4 *>(g.am + a);
        pf_v = *reinterpret_cast<const f4 *>(g.av + a);
        pf_t = pol_ ? *reinterpret_cast<const f4 *>(g.tw + a) : (f4){0.f, 0.f, 0.f, 0.f};
        pf_done = true;
    };

    // ... and its cache lines are pulled towards this XCD's L2 at the very start of the workgroup (one dword per
    // element group, result unused): the epilogues of all ~200 workgroups of the launch fall into the same few
    // microseconds, and their 12.8 MB of optimizer-state reads + 12.8 MB of writes ran at the HBM rate there (8.7 k
    // clocks per workgroup) -- requested here, the reads travel during the K loop, which is nowhere near the memory
    // bandwidth.  Ordinary loads whose four results stay live until the epilogue consumes them with an empty asm
    // statement (a load nobody uses would be deleted).  (Until late in round 3 these were inline-asm loads into ONE
    // register the compiler did not know to be in flight: correct only as long as the allocator never spilled or
    // re-used that register during the K loop -- the general kernel, at 128 VGPRs with spills, eventually did, and the
    // late-arriving dword overwrote an address: a memory access fault at the metric shape.)
    float l2_sink[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (PREFETCH_OPT && vecC) {
        const int row = tid_all >> 4, col = (tid_all & 15) * 4;
        if ((m0 + row) < g.M && (n0 + col) < g.N) {
            const int64_t a = coff + (int64_t)(m0 + row) * g.ldc + n0 + col;
            l2_sink[0] = g.C[a];
            l2_sink[1] = g.am[a];
            l2_sink[2] = g.av[a];
            if (pol_) l2_sink[3] = g.tw[a];
        }
    }

    GSTAMP(0);
    // Software-pipelined K loop in half chunks: the fragments of the second half of chunk it are read while its
    // first half multiplies, the first half of chunk it+1 is read (right after the barrier that publishes it) while
    // the second half multiplies, and the staging stores / next global loads are issued between the two, so LDS
    // and global latency sit under matrix work instead of in front of it
    // (with read -> 16 MFMAs -> barrier the 4*KS waves of the workgroup all waited at the same points and the
    // matrix pipe idled through every staging phase: 34k clocks for 16k clocks of MFMA work at K = 512).
    // Buffer reuse: chunk it+1 overwrites the buffer of chunk it-1, whose fragments every wave finished
    // reading before the barrier of iteration it-1.
    const bool bias_wave = want_bias_grad && TN && wn == 0;  // bias gradient = column sums of A, from the fragments
    constexpr int HT = BK / 4;  // MFMAs per half chunk
    auto rd = [&](float (&fa)[HT], float (&fb)[HT], const float *buf, int half) {
        if (dead_cols) return;
        const float *As = buf, *Bs = buf + TILE_FLOATS;
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            fa[t] = frag<A_KC>(As, wm * 32 + li, 2 * (half * HT + t) + lh);
            fb[t] = frag<B_KC>(Bs, wn * 32 + li, 2 * (half * HT + t) + lh);
        }
    };
    auto mm = [&](const float (&fa)[HT], const float (&fb)[HT]) {
        if (bias_wave) {
#pragma unroll
            for (int t = 0; t < HT; ++t) bias_acc += fa[t];
        }
        if (dead_cols) return;
#pragma unroll
        for (int t = 0; t < HT; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc, 0, 0, 0);
    };
#ifdef SSAC_LAB
#define GDSTAMP(i) do { if (g.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); GSTAMP(i); } } while (0)
#else
#define GDSTAMP(i) do { } while (0)
#endif
    LossFoldRegs lfr;
    if (lf_mode == 1) loss_fold_issue(lf, e, lfr);
    loadA(kg * BK);
    loadB(kg * BK);
    GDSTAMP(5);   // (debug runs only: serialises the prologue to time its pieces)
    if (lf_mode == 1) { loss_fold_finish(lf, e, lfr, late_rs); lds_barrier(); }
    else if (lf_mode == 2) { loss_fold_table(lf, e, late_rs, true, late_rs + lf.n_rows, false); __syncthreads(); }
    GDSTAMP(6);
    storeA(buf0);
    storeB(buf0 + TILE_FLOATS);
    if (iters > 1) { loadA((KS + kg) * BK); loadB((KS + kg) * BK); }
    GDSTAMP(7);
    lds_barrier();
    float f0a[HT], f0b[HT], f1a[HT], f1b[HT];  // first / second half of the current chunk
    if (iters > 0) rd(f0a, f0b, buf0, 0);
    GSTAMP(1);
    for (int it = 0; it + 1 < iters; ++it) {
        float *cur = (it & 1) ? buf1 : buf0;
        float *nxt = (it & 1) ? buf0 : buf1;
        rd(f1a, f1b, cur, 1);
        // (tools/lab/kloop_lab.hip: the staging work in front of these MFMAs, or staggered between the K-groups so that
        // two waves of a SIMD stage while two multiply, measured no faster -- 21.2-22.4 k clocks per K = 512 loop either
        // way against 17.0 k for the bare MFMAs + barriers; the fragment reads cost ~2.2 k and the staging ~2.3 k of
        // it wherever they stand.)
        mm(f0a, f0b);
        storeA(nxt); storeB(nxt + TILE_FLOATS);
        if (it + 2 < iters) { const int k0 = ((it + 2) * KS + kg) * BK; loadA(k0); loadB(k0); }
        lds_barrier();  // (LDS hand-off only: the loads just issued stay in flight over the next half chunk)
        rd(f0a, f0b, nxt, 0);
        mm(f1a, f1b);
    }
    if (iters > 0) {
        // last chunk, peeled: nothing is staged any more, so the operand staging registers are dead here and the
        // optimizer state takes their place (inside the loop the two live ranges would have overlapped)
        rd(f1a, f1b, ((iters - 1) & 1) ? buf1 : buf0, 1);
        if (PREFETCH_OPT) opt_prefetch();
        mm(f0a, f0b);
        mm(f1a, f1b);
    }
    if (bias_wave) bias_acc += __shfl_xor(bias_acc, 32, 64);  // the two k parities of column li

    GSTAMP(2);
    asm volatile("" :: "v"(l2_sink[0]), "v"(l2_sink[1]), "v"(l2_sink[2]), "v"(l2_sink[3]));   // (the L2 touches end here)
    if (EPI == EPI_ADAM || EPI == EPI_GRAD) {
        // ---- weight-gradient epilogue on ALL threads of the workgroup.  Every K-group parks its
        // partial 64x64 tile in LDS; element idx = row*64 + col is then finished (partials summed in
        // a fixed order -> deterministic) by thread idx % (256*KS), so the optimizer's three loads
        // and three stores per element are spread over 4*KS waves and issued as independent batches
        // instead of one latency-bound chain per accumulator register.
        float *mine = lds + kg * (4 * TILE_FLOATS);
        lds_barrier();   // (LDS hand-offs only: the prefetched optimizer state stays in flight)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            mine[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + wn * 32 + li] = acc[r];
        if (bias_wave && lh == 0) red[kg * 64 + wm * 32 + li] = bias_acc;
        // The arrival ticket of the folded logs is drawn by the LAST wave (sumsq_finish), the net's loss partials were stored
        // by wave 0 at the front (loss_fold_table): wave 0 drains its stores in front of THIS barrier, which the last wave
        // passes before it draws -- the order no longer rests on the K loop's waits (by now only the prefetched optimizer
        // state is in flight in this wave, and it is consumed right behind the barrier)
        if (fold.done && (tid_all >> 6) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        GSTAMP(3);
        constexpr int NT_ALL = NTHREADS * KS;
        constexpr int PER = (BM * BN) / NT_ALL;  // 16 / KS elements per thread
        ssac_adam_ctl ctl;
        if (EPI == EPI_ADAM) ctl = *g.ctl;
        // target update in this epilogue: always (static tau) for eager callers, or only when this update's Polyak
        // request is in the ring tail (late-bound)
        const bool pol = g.tw != nullptr && (!lt.on || lt.bits != 0u);
        const float tau = lt.on ? __uint_as_float(lt.bits) : g.tau;
        float ss = 0.0f;
        // Gradient-norm partial of the tile (round 5).  The sum over the tile's 4 KS wave partials and the slot stores used to
        // be ONE thread's serial work in front of its optimizer stores -- 2.2 k clocks that only the first wave paid, and
        // with it the workgroup.  Now the LAST wave takes them BEHIND its own optimizer stores: a fixed xor tree over the
        // partials (deterministic), the slot of the tile's first 32 x 32 block takes the sum and the siblings are zeroed by
        // lanes 1..3 of the same store instruction (one slot per 32 x 32 block, above); lane 0 then draws the arrival ticket of the
        // folded logs (its wave's stores drained first: log_fold_arrive).
        auto sumsq_finish = [&]() {
            if (!(g.sumsq && (tid_all >> 6) == 4 * KS - 1)) return;
            float tot = lane < 4 * KS ? red[lane] : 0.0f;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);   // (4 KS <= 16 partials in lanes 0..15)
            const int gx = (g.N + 31) >> 5, gy = (g.M + 31) >> 5;
            const int r_ = 2 * by + (lane >> 1), c_ = 2 * bx + (lane & 1);
            if (lane < 4 && r_ < gy && c_ < gx)
                __hip_atomic_store(g.sumsq + (int64_t)e * g.sumsq_stride + r_ * gx + c_, lane == 0 ? tot : 0.0f, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            if (fold.done && lane == 0) last = log_fold_arrive(fold, gridDim.x) ? 1 : 0;
        };
        // 16-byte form: a thread owns 4 consecutive columns of one row (weight rows of a multiple of 4 floats on 16-byte
        // aligned arenas: fc2 and every hidden-to-hidden layer).  The optimizer's traffic is 7-8 streams per element
        // (p, m, v [, target] in; the same out) that come from HBM / the Infinity Cache once per update; as dword
        // accesses that was 16 + 16 memory instructions per thread and the slowest phase of the workgroup after the K
        // loop (11 k clocks at the metric shape), as dwordx4 it is 4 + 4.  Same sums, same Adam arithmetic per element.
        if (vecC) {
            constexpr int PER4 = (BM * BN / 4) / NT_ALL;  // 4 / KS float4 per thread
            f4 g4[PER4], p4[PER4], m4[PER4], v4[PER4], t4[PER4];
            int64_t c4[PER4];
            bool ok4[PER4];
#pragma unroll
            for (int j = 0; j < PER4; ++j) {
                const int id = tid_all + j * NT_ALL;
                const int row = id >> 4, col = (id & 15) * 4;
                ok4[j] = (m0 + row) < g.M && (n0 + col) < g.N;
                c4[j] = coff + (int64_t)(m0 + row) * g.ldc + n0 + col;
                if (EPI == EPI_ADAM) {
                    if (PREFETCH_OPT && pf_done) {
                        p4[j] = pf_p; m4[j] = pf_m; v4[j] = pf_v; t4[j] = pf_t;
                    } else {
                        const int64_t a = ok4[j] ? c4[j] : coff;
                        p4[j] = *reinterpret_cast<const f4 *>(g.C + a);
                        m4[j] = *reinterpret_cast<const f4 *>(g.am + a);
                        v4[j] = *reinterpret_cast<const f4 *>(g.av + a);
                        t4[j] = pol ? *reinterpret_cast<const f4 *>(g.tw + a) : (f4){0.f, 0.f, 0.f, 0.f};
                    }
                }
                f4 sum = *reinterpret_cast<const f4 *>(lds + row * 64 + col);
#pragma unroll
                for (int gq = 1; gq < KS; ++gq) sum += *reinterpret_cast<const f4 *>(lds + gq * (4 * TILE_FLOATS) + row * 64 + col);
                g4[j] = sum;
            }
            GSTAMP(8);
            const int gm_ = m0 + tid_all;
            const bool bias_thr_ = want_bias_grad && tid_all < 64 && gm_ < g.M;
            const int64_t bi_ = (EPI == EPI_GRAD && g.sGb) ? (int64_t)e * g.sGb + gm_ : coff + gm_;
            float bsum_ = 0.0f, bpv_ = 0.0f, bmv_ = 0.0f, bvv_ = 0.0f, btv_ = 0.0f;
            if (bias_thr_) {
                bsum_ = red[tid_all];
#pragma unroll
                for (int gq = 1; gq < KS; ++gq) bsum_ += red[gq * 64 + tid_all];
                if (EPI == EPI_ADAM) { bpv_ = g.pb[bi_]; bmv_ = g.bm[bi_]; bvv_ = g.bv[bi_]; btv_ = (pol && g.tb) ? g.tb[bi_] : 0.0f; }
            }
#pragma unroll
            for (int j = 0; j < PER4; ++j)
                if (ok4[j]) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) ss += g4[j][c] * g4[j][c];
                }
            if (bias_thr_) ss += bsum_ * bsum_;
            if (g.sumsq) {
                ss = wave_sum(ss);
                GSTAMP(10);
                lds_barrier();  // red[] (bias partials) and the partial tiles have been consumed
                GSTAMP(11);
                if (lane == 0) red[tid_all >> 6] = ss;
                lds_barrier();
                GSTAMP(12);
                // (the sum of the wave partials and the slot stores: sumsq_finish, behind the optimizer stores)
            }
            GSTAMP(9);
#pragma unroll
            for (int j = 0; j < PER4; ++j) {
                if (!ok4[j]) continue;
                if (EPI == EPI_GRAD) {
                    *reinterpret_cast<f4 *>(g.gw + c4[j]) = g4[j];
                } else {
                    f4 pn, tn;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float m = m4[j][c], v = v4[j][c];
                        pn[c] = adam_elem(p4[j][c], g4[j][c], m, v, ctl);
                        m4[j][c] = m; v4[j][c] = v;
                        tn[c] = t4[j][c] * (1.0f - tau) + pn[c] * tau;
                    }
                    *reinterpret_cast<f4 *>(g.am + c4[j]) = m4[j];
                    *reinterpret_cast<f4 *>(g.av + c4[j]) = v4[j];
                    *reinterpret_cast<f4 *>(g.C + c4[j]) = pn;
                    if (pol) *reinterpret_cast<f4 *>(g.tw + c4[j]) = tn;
                }
            }
            if (bias_thr_) {
                if (EPI == EPI_GRAD) {
                    g.gb[bi_] = bsum_;
                } else {
                    float m = bmv_, v = bvv_;
                    const float pn = adam_elem(bpv_, bsum_, m, v, ctl);
                    g.bm[bi_] = m;
                    g.bv[bi_] = v;
                    g.pb[bi_] = pn;
                    if (pol && g.tb) g.tb[bi_] = btv_ * (1.0f - tau) + pn * tau;
                }
            }
            sumsq_finish();
            GSTAMP(4);
            return;
        }
        float gval[PER], pv[PER], mv[PER], vv[PER], tv[PER];
        int64_t ci[PER];
        bool ok[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int idx = tid_all + j * NT_ALL;
            const int row = idx >> 6, col = idx & 63;
            float sum = lds[idx];
#pragma unroll
            for (int gq = 1; gq < KS; ++gq) sum += lds[gq * (4 * TILE_FLOATS) + idx];
            gval[j] = sum;
            ok[j] = (m0 + row) < g.M && (n0 + col) < g.N;
            ci[j] = coff + (int64_t)(m0 + row) * g.ldc + n0 + col;
        }
        if (EPI == EPI_ADAM) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int64_t a = ok[j] ? ci[j] : coff;
                pv[j] = g.C[a]; mv[j] = g.am[a]; vv[j] = g.av[a];
                tv[j] = pol ? g.tw[a] : 0.0f;
            }
        }
        // the bias gradient of this tile's rows and its optimizer state (threads < 64 of n-tile 0)
        const int gm = m0 + tid_all;
        const bool bias_thr = want_bias_grad && tid_all < 64 && gm < g.M;
        const int64_t bi = (EPI == EPI_GRAD && g.sGb) ? (int64_t)e * g.sGb + gm : coff + gm;
        float bsum = 0.0f, bpv = 0.0f, bmv = 0.0f, bvv = 0.0f, btv = 0.0f;
        if (bias_thr) {
            bsum = red[tid_all];
#pragma unroll
            for (int gq = 1; gq < KS; ++gq) bsum += red[gq * 64 + tid_all];
            if (EPI == EPI_ADAM) { bpv = g.pb[bi]; bmv = g.bm[bi]; bvv = g.bv[bi]; btv = (pol && g.tb) ? g.tb[bi] : 0.0f; }
        }
        // ---- gradient-norm partial of the tile: per-wave sums here, finished by the LAST wave behind its optimizer
        //      stores (sumsq_finish above: xor tree over the wave partials, then the arrival ticket of the folded logs)
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (ok[j]) ss += gval[j] * gval[j];
        if (bias_thr) ss += bsum * bsum;
        if (g.sumsq) {
            ss = wave_sum(ss);
            __syncthreads();  // red[] (bias partials) has been consumed
            if (lane == 0) red[tid_all >> 6] = ss;
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (!ok[j]) continue;
            if (EPI == EPI_GRAD) {
                g.gw[ci[j]] = gval[j];
            } else {
                float m = mv[j], v = vv[j];
                const float pn = adam_elem(pv[j], gval[j], m, v, ctl);
                g.am[ci[j]] = m;
                g.av[ci[j]] = v;
                g.C[ci[j]] = pn;
                if (pol) g.tw[ci[j]] = tv[j] * (1.0f - tau) + pn * tau;
            }
        }
        if (bias_thr) {
            if (EPI == EPI_GRAD) {
                g.gb[bi] = bsum;
            } else {
                float m = bmv, v = bvv;
                const float pn = adam_elem(bpv, bsum, m, v, ctl);
                g.bm[bi] = m;
                g.bv[bi] = v;
                g.pb[bi] = pn;
                if (pol && g.tb) g.tb[bi] = btv * (1.0f - tau) + pn * tau;
            }
        }
        sumsq_finish();
        GSTAMP(4);
        return;
    }

    if (KS > 1) {
        // sum the K-split partial tiles into group 0 (fixed order -> deterministic)
        __syncthreads();
        float *mine = lds + kg * (4 * TILE_FLOATS);  // 64x64 floats fit in one group's staging area
        if (kg > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mine[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + wn * 32 + li] = acc[r];
        }
        __syncthreads();
        for (int gq = 1; gq < KS && kg == 0; ++gq) {
            const float *other = lds + gq * (4 * TILE_FLOATS);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[r] += other[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + wn * 32 + li];
        }
    }

    // ---- activation epilogues (group 0 only).  C/D layout of 32x32 MFMA: col = lane&31,
    //      row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool leader = kg == 0;
    const int gn = n0 + wn * 32 + li;
    const bool nok = leader && gn < g.N;
    float bias_n = 0.0f;
    if ((EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) && nok)
        bias_n = g.bias[batch_off(g.ids, g.idsB, e, g.sBias) + gn];
    float maskv[16];
    if (EPI == EPI_MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const bool in = nok && gm < g.M;
            maskv[r] = g.mask[in ? (int64_t)e * g.sMask + (int64_t)gm * g.ldmask + gn : 0];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (!(nok && gm < g.M)) continue;
        const float val = acc[r];
        const int64_t ci = coff + (int64_t)gm * g.ldc + gn;
        if (EPI == EPI_STORE) g.C[ci] = val;
        else if (EPI == EPI_BIAS) g.C[ci] = val + bias_n;
        else if (EPI == EPI_BIAS_RELU) g.C[ci] = fmaxf(val + bias_n, 0.0f);
        else if (EPI == EPI_MASK) g.C[ci] = maskv[r] > 0.0f ? val : 0.0f;
    }
}

template <bool A_KC, bool B_KC, int EPI, int KS, bool VECONLY = false>
__global__ __launch_bounds__(NTHREADS *KS) void ens_gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int per = gridDim.x * gridDim.y;
    const int L = ssac_xcd_contiguous(blockIdx.z * per + blockIdx.y * gridDim.x + blockIdx.x, per * gridDim.z, g.xcd);
    const int bz = L / per, rem = L - bz * per;
    int last = 0;
    ens_gemm_body<A_KC, B_KC, EPI, KS, VECONLY>(g, lds, rem % gridDim.x, rem / gridDim.x, bz, nullptr, LossFoldArgs{}, 0,
                                                LogFoldArgs{}, last);
}

// Two problems in ONE launch (the fc2 and fc1 weight gradients of an update): workgroups
// [0, tiles0) work on g0, the rest on g1, so the small problem fills CUs the big one leaves idle.
// Optionally a third piece: the head layer's (VALU) weight gradient as `head_tiles` extra workgroups.
// And optionally the update's log finalisation, run by whichever workgroup finishes LAST (device counter).
extern int g_gemm_lean;   // ssac_gemm_lean: 0 = always the general kernel

struct GemmPair {
    GemmArgs g0, g1; int tiles0; int tiles01; int head_grid_x; HeadWgradArgs head;
    int head_total;                       // number of head workgroups (head_grid_x per net)
    int td_wg;                            // 1: one more workgroup behind them stores the TD targets + their statistics
    int lf_nets;                          // ... and reduces the loss terms of these many nets (0 with stats_in_head)
    int stats_in_head;                    // the first head workgroup of every net reduces that net's loss terms
    int xcd;                              // XCD-contiguous tile order (ssac_internal.h)
    int xcd_mix;                          // ... per workgroup class (every class count a multiple of 8)
    LossFoldArgs lf;                      // lf.q != null: dL/dq evaluated per workgroup (ssac_critic_logs.h)
    LogFoldArgs fold;                     // fold.done != null: the update's logs are finalised by the last workgroup
    const uint32_t *late_word;            // != null: the target update waits for the decision in this word
    long long *tl;                        // optional per-workgroup (start, end) stamps (ssac_debug_timeline), slots from 1024
};

template <bool A_KC, bool B_KC, int EPI, int KS, bool VECONLY = false>
__global__ __launch_bounds__(NTHREADS *KS) void ens_gemm_pair_kernel(GemmPair p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (the XCD-contiguous order covers the GEMM + head workgroups only: the TD workgroup behind them keeps its own id,
    // so the tiles land on the same XCDs with or without it)
    SSAC_LAB_ONLY(if (p.tl && threadIdx.x == 0 && blockIdx.x < 512) p.tl[1024 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();)
    const int n_main = p.tiles01 + p.head_total;
    int bid = (int)blockIdx.x < n_main ? ssac_xcd_contiguous(blockIdx.x, n_main, p.xcd) : (int)blockIdx.x;
    if (p.xcd_mix && (int)blockIdx.x < n_main) {
        // XCD-contiguous PER CLASS: XCD x takes a contiguous eighth of the fc2 tiles, of the fc1 tiles and of the head
        // workgroups (heavy first), instead of a contiguous eighth of the concatenated list -- which gave five XCDs 30
        // fc2 tiles each and two XCDs nothing but the short head workgroups
        const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int a8 = p.tiles0 >> 3, f8 = (p.tiles01 - p.tiles0) >> 3, h8 = p.head_total >> 3;
        bid = slot < a8 ? x * a8 + slot
                        : (slot < a8 + f8 ? p.tiles0 + x * f8 + (slot - a8) : p.tiles01 + x * h8 + (slot - a8 - f8));
    }
    float *tab = lds + KS * (4 * TILE_FLOATS) + 64 * KS;  // folded loss gradient: [n_rows] row scales, then scratch
    const bool fold = p.lf.q != nullptr;
    const LateTau lt{p.late_word != nullptr, p.late_word ? *p.late_word : 0u};
    int last = 0;
    bool drawn = false;
    if (p.td_wg && bid == p.tiles01 + p.head_total) {
        // The TD targets themselves (the caller's td_target tensor) and their statistics for the logs: a workgroup of
        // its own.  As a side job of net 0's first GEMM tile (round 1) it made that tile the slowest workgroup of the
        // launch -- a global write -> read round trip and five barriers in front of its K loop -- and the launch is as
        // long as its slowest workgroup.
        // It also reduces every net's loss / TD-error terms (round 1: the first fc2 tile of each net, in front of its
        // K loop): this workgroup has nothing else to do and is done long before the GEMM tiles.
        for (int e = 0; e < p.lf_nets; ++e) {
            loss_fold_table(p.lf, e, tab, true, tab + p.lf.n_rows, e == 0);
            __syncthreads();
        }
        if (p.lf_nets == 0) {   // the nets' loss terms are reduced elsewhere (head workgroups / first tiles): only the TD part
            loss_fold_table(p.lf, 0, tab, false, tab + p.lf.n_rows, true);
            __syncthreads();
        }
        if ((p.fold.done && p.fold.td_logs) || p.fold.deferred_stats)
            log_fold_td_stats(p.fold, p.lf.tds, tab + p.lf.n_rows);
    } else if (bid >= p.tiles01) {  // head-layer weight gradient + Adam beside the GEMM tiles
#if defined(SSAC_LAB) && defined(SSAC_EXPERIMENT_SKIP_HEAD)
        return;
#endif
        const int L = bid - p.tiles01;
        const int e = L / p.head_grid_x;
        if (fold) {
            // (the first head workgroup of a net also reduces the net's loss terms: it builds the very table they are
            // sums over anyway.  Round 2 gave all ten nets' terms to the TD workgroup -- a serial pass of ~2 us per net,
            // 21 us on its own: profiles/r3_kernel_stats.md)
            loss_fold_table(p.lf, e, tab, p.stats_in_head && (L % p.head_grid_x) == 0, tab + p.lf.n_rows, false);
            __syncthreads();
        }
        bool hpol = p.head.target != nullptr;
        float htau = p.head.tau;
        if (lt.on) {
            hpol = hpol && lt.bits != 0u;
            htau = __uint_as_float(lt.bits);
        }
        head_wgrad_body<4 * KS>(p.head, lds, L % p.head_grid_x, e, fold ? tab : nullptr, hpol, htau,
                                KS * 4 * TILE_FLOATS + 64 * KS);
    } else {
        const bool first = bid < p.tiles0;
#if defined(SSAC_LAB) && defined(SSAC_EXPERIMENT_SKIP_FC1)
        if (!first) return;   // (measurement builds only: how long is the launch without the fc1 problem's tiles?)
#endif
#if defined(SSAC_LAB) && defined(SSAC_EXPERIMENT_SKIP_FC2)
        if (first) return;
#endif
        const GemmArgs &g = first ? p.g0 : p.g1;
        const int L = first ? bid : bid - p.tiles0;
        const int per = g.grid_x * g.grid_y;
        const int bz = L / per, rem = L - bz * per;
        // multi-round launches (no TD workgroup pass over the nets): the first fc2 tile of each net also reduces that
        // net's loss terms -- the one-step table with statistics; every other tile takes the two-step form
        const bool stats = p.lf_nets == 0 && !p.stats_in_head && first && rem == 0;
        ens_gemm_body<A_KC, B_KC, EPI, KS, VECONLY>(g, lds, rem % g.grid_x, rem / g.grid_x, bz, fold ? tab : nullptr, p.lf,
                                                    fold ? (stats ? 2 : 1) : 0, p.fold, last, lt);
        drawn = true;
    }
    if (p.fold.done) {
        // a GEMM tile's ticket was drawn inside its epilogue by lane 0 of its last wave, behind that wave's optimizer stores
        // and behind the barrier in front of which wave 0 drained the loss partials; the wave of the last arriver finalises the logs
        float *flag = lds + KS * (4 * TILE_FLOATS);   // (the bias / sumsq scratch: consumed by now)
        __syncthreads();
        // (a GEMM tile's ticket was drawn by lane 0 of its LAST wave -- sumsq_finish; the other workgroup classes draw it here)
        if (threadIdx.x == (drawn ? (4 * KS - 1) * 64 : 0))
            flag[0] = (drawn ? last != 0 : log_fold_arrive(p.fold, gridDim.x)) ? 1.0f : 0.0f;
        __syncthreads();
        if (flag[0] != 0.0f && threadIdx.x < 64) log_fold_finish(p.fold);
    } else if (p.fold.deferred_stats && p.fold.feed && blockIdx.x == 0 && threadIdx.x == 0) {
        p.fold.feed->tick += 1;   // deferred finalisation: the update is over for the input ring (no reader in this launch)
    }
#ifdef SSAC_LAB
    if (p.tl && blockIdx.x < 512) {   // (the stamp buffer holds 512 workgroups per launch)
        __syncthreads();
        if (threadIdx.x == 0) p.tl[1024 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// LATENCY VARIANT of the merged weight-gradient launch (round 4; VERDICT round 3 "under-filled launches").
//
// A launch of 64 x 64 tiles is as long as ONE tile's serial chain -- first operand chunk + loss fold, a K loop over the
// whole batch, the partial-tile hand-off, the Adam epilogue: ~40 k clocks at the metric shape -- however few tiles it
// has: with 2 critics (SAC; a rank holding 2 of 16) 32 + 8 + 8 workgroups keep 48 of 256 CUs busy for 20 us.  Here a
// workgroup owns a 32 x 32 tile = ONE v_mfma_f32_32x32x2_f32 accumulator, and its 8 waves split K: wave w takes the
// 32-deep chunks w, w + 8, ... (B 512: two chunks = 32 MFMAs per wave).  Both operands are row-contiguous over the
// tile's 32 rows / columns (A[k][m], B[k][n]: the TN product of a weight gradient), so a lane's MFMA operand for step t
// is ONE dword at k = k0 + 2 t + (lane >> 5), row / column (lane & 31): 128-byte coalesced loads straight into the
// operand registers -- no LDS staging, no barrier in the K loop, every load of a chunk in flight at once.  The 8 partial
// tiles meet in LDS (32 KB) and are summed in wave order (deterministic); each thread then finishes 2 elements (Adam
// [+ Polyak] with the optimizer state requested at kernel start).  4x the workgroups of the 64 x 64 form, each ~5x
// shorter: 128 + 16 + 32 + 1 workgroups for 2 critics at the metric shape.  The summation order over k differs from the
// 64 x 64 kernel's (rounding only; the fixtures' fp32 tolerances hold for both: tests/test_hip_cases.py runs both).
// ---------------------------------------------------------------------------------------------------------------
constexpr int ST = 32;                  // tile edge
constexpr int SW = 8;                   // waves per workgroup = K-split ways
constexpr int STHREADS = 64 * SW;
constexpr int S_HEAD_COLS = 16, S_HEAD_GROUPS = STHREADS / S_HEAD_COLS;
constexpr int S_PART = SW * ST * ST;    // floats: the partial tiles
constexpr int S_RED = SW * ST + 2 * SW; // bias partials [SW][32] + gradient-norm partials [SW] + scratch
// head workgroups reuse the front of the LDS block: 2 * GROUPS * COLS + GROUPS floats (ssac_head_wgrad.h) <= S_PART
static_assert(2 * S_HEAD_GROUPS * S_HEAD_COLS + S_HEAD_GROUPS <= S_PART, "head scratch must fit the partial-tile area");
// (the 64 x 32 form runs 32-column head workgroups of half as many row groups: the same scratch)

struct SmallFrag { float a[16], b[16]; };

// MB = 32-row blocks of the tile: 1 -> a 32 x 32 tile whose 8 waves split K eight ways (the only form launched).  MB = 2 -- a
// 64 x 32 tile, two row blocks x four K-groups -- was measured slower than the 64 x 64 kernel wherever the 32 x 32 form does
// not fit (see wgrad_merged) and is not instantiated.
template <int EPI, int MB>
__device__ __forceinline__ void wgrad_small_body(const GemmArgs &g, float *lds, int bx, int by, int bz, float *late_rs,
                                                 const LossFoldArgs &lf, int lf_mode, const LogFoldArgs &fold, int &last,
                                                 const LateTau &lt) {
    constexpr int KW = SW / MB;   // K-split ways
    constexpr int NE = 2 * MB;    // tile elements finished per thread
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SSAC_LAB
#define SSTAMP(i) do { if (g.dbg && bx == 0 && by == 0 && bz == 0 && threadIdx.x == 0) g.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SSTAMP(i) do { } while (0)
#endif
    SSTAMP(0);
    const int li = lane & 31, lh = lane >> 5;
    const int mb = wave / KW, kw = wave - mb * KW;   // (uniform) this wave's row block and K-group
    const int e = bz, mt0 = by * (ST * MB), m0 = mt0 + mb * ST, n0 = bx * ST;
    float *part = lds, *red = lds + S_PART;
    const float *A = g.A + batch_off(g.ids, g.idsA, e, g.sA);
    const float *B = g.B + batch_off(g.ids, g.idsB, e, g.sB);
    const int64_t coff = batch_off(g.ids, g.idsC, e, g.sC);
    const int K = g.K;
    // rows / columns beyond the matrix read the block's first row / column (always valid -- a row block that lies
    // entirely beyond the matrix reads block 0's): they only reach accumulator rows / columns the epilogue guards
    const bool mok = (m0 + li) < g.M, nok = (n0 + li) < g.N;
    const int m0c = m0 < g.M ? m0 : mt0;
    // Operand loads are BUFFER loads: one descriptor per operand in SGPRs (built from wave-uniform values only), the lane's
    // byte offset in ONE VGPR, the row of MFMA step t as a scalar offset -- the 32 loads of a chunk need no address
    // register each (as flat loads with 64-bit lane addresses the two register sets + 32 address pairs did not fit 128
    // VGPRs, i.e. two workgroups per CU).
    auto uniform_ptr = [](const float *p_) {
        const uint64_t u = (uint64_t)(uintptr_t)p_;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return (void *)(uintptr_t)(((uint64_t)hi << 32) | lo);
    };
    const uint32_t lda = __builtin_amdgcn_readfirstlane((uint32_t)g.lda), ldb = __builtin_amdgcn_readfirstlane((uint32_t)g.ldb);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(A + m0c), 0, 0x7ffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(B + n0), 0, 0x7ffffffc, 0x00020000);
    const int alane = 4 * (int)((mok ? (uint32_t)li : 0u) + (uint32_t)lh * lda), blane = 4 * (int)((nok ? (uint32_t)li : 0u) + (uint32_t)lh * ldb);
    const bool sign = g.a_sign_w != nullptr;
    const float sw = sign ? g.a_sign_w[(int64_t)e * g.M + (mok ? m0 + li : m0c)] : 0.0f;
    const bool ragged = (K & 31) != 0;   // (uniform; K is even: small_ok)
    const float *rscale = (g.rowscale && !late_rs) ? g.rowscale + (int64_t)e * g.sRow : nullptr;
    const bool want_bias = bx == 0 && !g.no_bias;
    const int nchunks = (K + 31) >> 5;

    // ---- this thread finishes NE tile elements: idx = tid + 512 j -> row idx >> 5 (of the 32 MB), column idx & 31
    const bool pol = g.tw != nullptr && (!lt.on || lt.bits != 0u);
    const float tau = lt.on ? __uint_as_float(lt.bits) : g.tau;
    auto elem = [&](int j, int64_t &c) {
        const int idx = tid + j * STHREADS, row = idx >> 5, col = idx & 31;
        c = coff + (int64_t)(mt0 + row) * g.ldc + n0 + col;
        return (mt0 + row) < g.M && (n0 + col) < g.N;
    };
    LossFoldRegs lfr;
    if (lf_mode == 1) loss_fold_issue(lf, e, lfr);
    auto load = [&](SmallFrag &f, int c) {
        const int sa = (int)(4u * (uint32_t)(c * 32) * lda), sb = (int)(4u * (uint32_t)(c * 32) * ldb);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (!ragged || c * 32 + 2 * t < K) {   // (uniform: K is even, so both k parities of step t exist or neither)
                f.a[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, alane, sa + (int)(8u * t * lda), 0));
                f.b[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, blane, sb + (int)(8u * t * ldb), 0));
            } else {
                f.a[t] = 0.0f;
                f.b[t] = 0.0f;
            }
        }
    };
    SmallFrag f0, f1;
    if (kw < nchunks) load(f0, kw);
    if (lf_mode == 1) { loss_fold_finish(lf, e, lfr, late_rs); lds_barrier(); }
    else if (lf_mode == 2) { loss_fold_table(lf, e, late_rs, true, late_rs + lf.n_rows, false); __syncthreads(); }

    SSTAMP(1);   // (loss-fold table ready)
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    float bias_acc = 0.0f;
    // (the row scales are fetched per SOURCE -- LDS table, global vector, none -- in uniform branches of their own, the
    // table through an LDS-qualified pointer: one `late_rs ? late_rs[k] : rscale[k]` per element turns into 16 generic-
    // address loads, each followed by a wait for everything in flight, the operand loads included)
    typedef __attribute__((address_space(3))) const float lds_cfloat;
    lds_cfloat *tabl = (lds_cfloat *)late_rs;
    auto consume = [&](SmallFrag &f, int c) {
        const int k0 = c * 32 + lh;
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // in halves of 8 steps: 8 scale registers live instead of 16
            float sc[8];
            if (late_rs) {
                if (!ragged) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) sc[t] = tabl[k0 + 2 * (8 * h + t)];
                } else {
#pragma unroll
                    for (int t = 0; t < 8; ++t) sc[t] = tabl[k0 + 2 * (8 * h + t) < K ? k0 + 2 * (8 * h + t) : 0];
                }
            } else if (rscale) {
#pragma unroll
                for (int t = 0; t < 8; ++t) sc[t] = rscale[(!ragged || k0 + 2 * (8 * h + t) < K) ? k0 + 2 * (8 * h + t) : 0];
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) sc[t] = 1.0f;
            }
            if (sign) {
#pragma unroll
                for (int t = 0; t < 8; ++t) f.a[8 * h + t] = (f.a[8 * h + t] > 0.0f ? sw : 0.0f) * sc[t];
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) f.a[8 * h + t] *= sc[t];   // (steps beyond a ragged K hold zeros already)
            }
            if (want_bias) {
#pragma unroll
                for (int t = 0; t < 8; ++t) bias_acc += f.a[8 * h + t];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[8 * h + t], f.b[8 * h + t], acc, 0, 0, 0);
        }
    };
    int c = kw;
    bool have1 = c + KW < nchunks;
    if (have1) load(f1, c + KW);
    // The optimizer state of the thread's elements is requested HERE: it has not been touched since the last update and
    // comes from HBM (~2 us) -- behind the K loop that round trip was a phase of its own in every workgroup; in front of
    // the operand loads it would hold THEM back (vector memory returns in order).  Behind the first two chunks' loads it
    // travels while they multiply.
    // (p, m, v here; the Polyak target -- every other update -- behind the K loop, under the exchange through LDS: with it
    // the 64 x 32 form's four elements per thread no longer fit 128 registers beside the two operand sets)
    float pv[NE], mv[NE], vv[NE], tv[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) { pv[j] = 0.f; mv[j] = 0.f; vv[j] = 0.f; tv[j] = 0.f; }
    auto opt_load = [&]() {
        if (EPI != EPI_ADAM) return;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            int64_t cj;
            const int64_t a_ = elem(j, cj) ? cj : coff;
            pv[j] = g.C[a_]; mv[j] = g.am[a_]; vv[j] = g.av[a_];
        }
    };
    // (the 64 x 32 form holds four elements per thread: beside the two operand sets their state does not fit 128
    // registers, so it is requested behind the K loop and travels under the exchange through LDS)
    constexpr bool EARLY = MB == 1;
    if (EARLY) opt_load();
    for (; c + KW < nchunks; c += 2 * KW) {   // two register sets: chunk c + KW loads while chunk c multiplies
        if (!have1) load(f1, c + KW);
        have1 = false;
        consume(f0, c);
        if (c + 2 * KW < nchunks) load(f0, c + 2 * KW);
        consume(f1, c + KW);
    }
    if (c < nchunks) consume(f0, c);   // (an odd number of chunks for this wave)
    if (!EARLY) opt_load();

    SSTAMP(2);   // (K loop done)
    if (EPI == EPI_ADAM && pol) {
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            int64_t cj;
            tv[j] = g.tw[elem(j, cj) ? cj : coff];
        }
    }
    // ---- the 8 partial tiles meet in LDS; C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8 (r>>2) + 4 (lane>>5)
    float *mine = part + wave * (ST * ST);
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * lh) * ST + li] = acc[r];
    if (want_bias) {
        bias_acc += __shfl_xor(bias_acc, 32, 64);   // the two k parities of row li
        if (lh == 0) red[wave * ST + li] = bias_acc;
    }
    lds_barrier();
    SSTAMP(3);   // (partials exchanged)
    float gval[NE];
    int64_t ci[NE];
    bool ok[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        ok[j] = elem(j, ci[j]);
        // element idx = tid + 512 j lies in row block idx >> 10 (its K-groups' partials are summed in group order)
        const int idx = tid + j * STHREADS, blk = idx >> 10, off = idx & 1023;
        float sum = part[(blk * KW) * (ST * ST) + off];
#pragma unroll
        for (int w = 1; w < KW; ++w) sum += part[(blk * KW + w) * (ST * ST) + off];
        gval[j] = sum;
    }
    const int gm = mt0 + tid;
    const bool bias_thr = want_bias && tid < ST * MB && gm < g.M;
    const int64_t bi = (EPI == EPI_GRAD && g.sGb) ? (int64_t)e * g.sGb + gm : coff + gm;
    float bsum = 0.0f, bpv = 0.0f, bmv = 0.0f, bvv = 0.0f, btv = 0.0f;
    if (bias_thr) {
        const int blk = tid >> 5, r_ = tid & 31;
        bsum = red[(blk * KW) * ST + r_];
#pragma unroll
        for (int w = 1; w < KW; ++w) bsum += red[(blk * KW + w) * ST + r_];
        if (EPI == EPI_ADAM) { bpv = g.pb[bi]; bmv = g.bm[bi]; bvv = g.bv[bi]; btv = (pol && g.tb) ? g.tb[bi] : 0.0f; }
    }
    // ---- gradient-norm partial first (it needs the gradients only): with the logs folded in, the arrival ticket is
    //      drawn before -- not behind -- the optimizer stores
    if (g.sumsq) {
        float ss = 0.0f;
#pragma unroll
        for (int j = 0; j < NE; ++j) ss += ok[j] ? gval[j] * gval[j] : 0.0f;
        if (bias_thr) ss += bsum * bsum;
        ss = wave_sum(ss);
        float *red2 = red + SW * ST;
        if (lane == 0) red2[wave] = ss;
        lds_barrier();
        if (tid == 0) {
            float tot = 0.0f;
#pragma unroll
            for (int w = 0; w < SW; ++w) tot += red2[w];
            // one slot per 32 x 32 tile (ssac_wgrad_tiles): the first of this workgroup's row blocks takes the sum
            const int gx = (g.N + ST - 1) / ST, gy = (g.M + ST - 1) / ST;
            float *slot = g.sumsq + (int64_t)e * g.sumsq_stride + (by * MB) * gx + bx;
            __hip_atomic_store(slot, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MB == 2 && by * MB + 1 < gy) __hip_atomic_store(slot + gx, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fold.done) last = log_fold_arrive(fold, gridDim.x) ? 1 : 0;
        }
    }
    SSTAMP(4);   // (gradient-norm partial out)
    if (EPI == EPI_GRAD) {
#pragma unroll
        for (int j = 0; j < NE; ++j)
            if (ok[j]) g.gw[ci[j]] = gval[j];
        if (bias_thr) g.gb[bi] = bsum;
        return;
    }
    const ssac_adam_ctl ctl = *g.ctl;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        if (!ok[j]) continue;
        const float pn = adam_elem(pv[j], gval[j], mv[j], vv[j], ctl);
        g.am[ci[j]] = mv[j]; g.av[ci[j]] = vv[j]; g.C[ci[j]] = pn;
        if (pol) g.tw[ci[j]] = tv[j] * (1.0f - tau) + pn * tau;
    }
    if (bias_thr) {
        const float pn = adam_elem(bpv, bsum, bmv, bvv, ctl);
        g.bm[bi] = bmv; g.bv[bi] = bvv; g.pb[bi] = pn;
        if (pol && g.tb) g.tb[bi] = btv * (1.0f - tau) + pn * tau;
    }
    SSTAMP(5);
#undef SSTAMP
}

// the merged launch in its latency form: [fc2 tiles | fc1 tiles | head workgroups (16 MB columns each) | TD workgroup]
template <int EPI, int MB>
__global__ __launch_bounds__(STHREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void wgrad_small_pair_kernel(GemmPair p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    SSAC_LAB_ONLY(if (p.tl && threadIdx.x == 0 && blockIdx.x < 512) p.tl[1024 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();)
    const int n_main = p.tiles01 + p.head_total;
    int bid = (int)blockIdx.x < n_main ? ssac_xcd_contiguous(blockIdx.x, n_main, p.xcd) : (int)blockIdx.x;
    if (p.xcd_mix && (int)blockIdx.x < n_main) {   // XCD-contiguous PER CLASS (see ens_gemm_pair_kernel)
        const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int a8 = p.tiles0 >> 3, f8 = (p.tiles01 - p.tiles0) >> 3, h8 = p.head_total >> 3;
        bid = slot < a8 ? x * a8 + slot
                        : (slot < a8 + f8 ? p.tiles0 + x * f8 + (slot - a8) : p.tiles01 + x * h8 + (slot - a8 - f8));
    }
    float *tab = lds + S_PART + S_RED;  // folded loss gradient: [n_rows] row scales, then scratch
    const bool fold = p.lf.q != nullptr;
    const LateTau lt{p.late_word != nullptr, p.late_word ? *p.late_word : 0u};
    int last = 0;
    bool drawn = false;
    if (p.td_wg && bid == p.tiles01 + p.head_total) {
        // the TD targets themselves + their statistics (and the nets' loss terms when no head workgroup takes them)
        for (int e = 0; e < p.lf_nets; ++e) {
            loss_fold_table(p.lf, e, tab, true, tab + p.lf.n_rows, e == 0);
            __syncthreads();
        }
        if (p.lf_nets == 0) {
            loss_fold_table(p.lf, 0, tab, false, tab + p.lf.n_rows, true);
            __syncthreads();
        }
        if ((p.fold.done && p.fold.td_logs) || p.fold.deferred_stats)
            log_fold_td_stats(p.fold, p.lf.tds, tab + p.lf.n_rows);
    } else if (bid >= p.tiles01) {
        const int L = bid - p.tiles01;
        const int e = L / p.head_grid_x;
        if (fold) {
            loss_fold_table(p.lf, e, tab, p.stats_in_head && (L % p.head_grid_x) == 0, tab + p.lf.n_rows, false);
            __syncthreads();
        }
        bool hpol = p.head.target != nullptr;
        float htau = p.head.tau;
        if (lt.on) {
            hpol = hpol && lt.bits != 0u;
            htau = __uint_as_float(lt.bits);
        }
        head_wgrad_body<S_HEAD_GROUPS / MB, S_HEAD_COLS * MB>(p.head, lds, L % p.head_grid_x, e, fold ? tab : nullptr, hpol, htau,
                                                              S_PART + S_RED);
    } else {
        const bool first = bid < p.tiles0;
        const GemmArgs &g = first ? p.g0 : p.g1;
        const int L = first ? bid : bid - p.tiles0;
        const int per = g.grid_x * g.grid_y;
        const int bz = L / per, rem = L - bz * per;
        const bool stats = p.lf_nets == 0 && !p.stats_in_head && first && rem == 0;
        wgrad_small_body<EPI, MB>(g, lds, rem % g.grid_x, rem / g.grid_x, bz, fold ? tab : nullptr, p.lf,
                              fold ? (stats ? 2 : 1) : 0, p.fold, last, lt);
        drawn = true;
    }
    if (p.fold.done) {
        float *flag = lds + S_PART;   // (the bias / gradient-norm scratch: consumed by now)
        __syncthreads();
        if (threadIdx.x == 0) flag[0] = (drawn ? last != 0 : log_fold_arrive(p.fold, gridDim.x)) ? 1.0f : 0.0f;
        __syncthreads();
        if (flag[0] != 0.0f && threadIdx.x < 64) log_fold_finish(p.fold);
    } else if (p.fold.deferred_stats && p.fold.feed && blockIdx.x == 0 && threadIdx.x == 0) {
        p.fold.feed->tick += 1;   // deferred finalisation: the update is over for the input ring
    }
#ifdef SSAC_LAB
    if (p.tl && blockIdx.x < 512) {   // (the stamp buffer holds 512 workgroups per launch)
        __syncthreads();
        if (threadIdx.x == 0) p.tl[1024 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// 0 = automatic (the latency form when the whole launch fits one resident round of its small workgroups), 1 = always 64 x 64
// tiles, 2 = 32 x 32 tiles whenever the shapes allow (ssac_wgrad_variant)
int g_wgrad_variant = 0;
int g_wgrad_small_max = 512;   // automatic choice: the latency form up to this many workgroups (2 resident per CU)

bool small_ok(const GemmArgs &g) {
    return g.Ktot <= 0 && g.K >= 2 && (g.K & 1) == 0 && (int64_t)(g.K + 64) * g.lda < (1LL << 28) &&
           (int64_t)(g.K + 64) * g.ldb < (1LL << 28);   // (byte offsets of the buffer loads stay below 2^31)
}

template <int EPI, int MB>
int launch_pair_small(GemmPair &p, int batch, hipStream_t st) {
    static bool attr_set = false;
    const size_t lds = sizeof(float) * (S_PART + S_RED + (p.lf.q ? p.lf.n_rows + 2 * SW + 16 : 0));
    if (lds > 150 * 1024) return ssac_fail("wgrad_small_pair: the folded loss table does not fit LDS");
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)wgrad_small_pair_kernel<EPI, MB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                150 * 1024) != hipSuccess)
            return ssac_fail("wgrad_small_pair: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    for (GemmArgs *g : {&p.g0, &p.g1}) {
        g->grid_x = (g->N + ST - 1) / ST;
        g->grid_y = (g->M + ST * MB - 1) / (ST * MB);
    }
    if (p.head_grid_x > 0) p.head_grid_x = (p.head.hidden + S_HEAD_COLS * MB - 1) / (S_HEAD_COLS * MB);
    p.xcd = (g_ssac_xcd >> 1) & 1;
    p.tl = g_ssac_timeline;
    p.tiles0 = p.g0.grid_x * p.g0.grid_y * batch;
    p.tiles01 = p.tiles0 + p.g1.grid_x * p.g1.grid_y * batch;
    p.head_total = p.head_grid_x > 0 ? p.head_grid_x * batch : 0;
    p.xcd_mix = (p.xcd && (g_ssac_xcd & 4) == 0 && p.tiles0 % 8 == 0 && (p.tiles01 - p.tiles0) % 8 == 0 && p.head_total % 8 == 0) ? 1 : 0;
    p.td_wg = (p.lf.q && p.lf.tds.q_t) ? 1 : 0;
    p.stats_in_head = (p.lf.q && p.head_total > 0) ? 1 : 0;
    p.lf_nets = (!p.stats_in_head && p.td_wg) ? batch : 0;
    const int total = p.tiles01 + p.head_total + p.td_wg;
    SSAC_LAUNCH((wgrad_small_pair_kernel<EPI, MB>), dim3(total), dim3(STHREADS), lds, st, p);
    return ssac_check_launch("wgrad_small_pair");
}

template <bool A_KC, bool B_KC, int EPI, int KS>
int launch_pair_ks(GemmPair &p, int batch0, int batch1, hipStream_t st) {
    static bool attr_set = false;
    const size_t lds = sizeof(float) * (KS * 4 * TILE_FLOATS + 64 * KS + (p.lf.q ? p.lf.n_rows + 16 * KS : 0));
    constexpr int PAIR_LDS_MAX = 160 * 1024 - 256;  // the kernel also has a few bytes of static LDS
    if (lds > PAIR_LDS_MAX) return ssac_fail("ens_gemm_pair: the folded loss table does not fit LDS");
    // the lean kernel (16-byte loader on both operands of both problems, nothing else compiled into the K loop)
    constexpr bool TN_ = !A_KC && !B_KC;
    auto vec_ok = [](const GemmArgs &g) {
        return (g.vec & 1) && (g.vec & 6) && g.Ktot <= 0 && g.K % BK == 0 && (int64_t)(g.K + 1) * g.lda < (1LL << 31) &&
               (int64_t)(g.K + 1) * g.ldb < (1LL << 31);
    };
    const bool lean = TN_ && g_gemm_lean && vec_ok(p.g0) && vec_ok(p.g1);
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)ens_gemm_pair_kernel<A_KC, B_KC, EPI, KS>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS_MAX) != hipSuccess)
            return ssac_fail("ens_gemm_pair: cannot raise the dynamic LDS limit");
        if constexpr (TN_) {
            if (hipFuncSetAttribute((const void *)ens_gemm_pair_kernel<A_KC, B_KC, EPI, KS, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS_MAX) != hipSuccess)
                return ssac_fail("ens_gemm_pair: cannot raise the dynamic LDS limit");
        }
        attr_set = true;
    }
    p.xcd = (g_ssac_xcd >> 1) & 1;
    p.tl = g_ssac_timeline;
    p.xcd_mix = 0;
    p.tiles0 = p.g0.grid_x * p.g0.grid_y * batch0;
    p.tiles01 = p.tiles0 + p.g1.grid_x * p.g1.grid_y * batch1;
    p.head_total = p.head_grid_x > 0 ? p.head_grid_x * batch0 : 0;
    p.xcd_mix = (p.xcd && (g_ssac_xcd & 4) == 0 && p.tiles0 % 8 == 0 && (p.tiles01 - p.tiles0) % 8 == 0 && p.head_total % 8 == 0) ? 1 : 0;
    p.td_wg = (p.lf.q && p.lf.tds.q_t) ? 1 : 0;
    // the TD workgroup takes over the per-net loss terms only while the launch is ONE round of workgroups (it then has
    // ~25 us of slack); in a multi-round launch (N = 16) a serial pass over 16 nets would itself become the tail
    p.stats_in_head = (p.lf.q && p.head_total > 0) ? 1 : 0;
    p.lf_nets = (!p.stats_in_head && p.td_wg && p.tiles01 + p.head_total + 1 <= 256) ? batch0 : 0;
    const int total = p.tiles01 + p.head_total + p.td_wg;
    if constexpr (TN_) {
        if (lean) {
            SSAC_LAUNCH((ens_gemm_pair_kernel<A_KC, B_KC, EPI, KS, true>), dim3(total), dim3(NTHREADS * KS), lds, st, p);
            return ssac_check_launch("ens_gemm_pair");
        }
    }
    SSAC_LAUNCH((ens_gemm_pair_kernel<A_KC, B_KC, EPI, KS>), dim3(total), dim3(NTHREADS * KS), lds, st, p);
    return ssac_check_launch("ens_gemm_pair");
}

template <bool A_KC, bool B_KC, int EPI, int KS>
int launch_ks(const GemmArgs &g, dim3 grid, hipStream_t st) {
    static bool attr_set = false;
    const size_t lds = sizeof(float) * (KS * 4 * TILE_FLOATS + 64 * KS);
    constexpr bool TN_ = !A_KC && !B_KC;
    if (!attr_set && lds > 48 * 1024) {
        if (hipFuncSetAttribute((const void *)ens_gemm_kernel<A_KC, B_KC, EPI, KS>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ssac_fail("ens_gemm: cannot raise the dynamic LDS limit");
        if constexpr (TN_) {
            if (hipFuncSetAttribute((const void *)ens_gemm_kernel<A_KC, B_KC, EPI, KS, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return ssac_fail("ens_gemm: cannot raise the dynamic LDS limit");
        }
        attr_set = true;
    }
    if constexpr (TN_) {   // weight gradients: the lean kernel when the operands qualify (see launch_pair_ks)
        extern int g_gemm_lean;
        if (g_gemm_lean && (g.vec & 1) && (g.vec & 6) && g.Ktot <= 0 && g.K % BK == 0 &&
            (int64_t)(g.K + 1) * g.lda < (1LL << 31) && (int64_t)(g.K + 1) * g.ldb < (1LL << 31)) {
            SSAC_LAUNCH((ens_gemm_kernel<A_KC, B_KC, EPI, KS, true>), grid, dim3(NTHREADS * KS), lds, st, g);
            return ssac_check_launch("ens_gemm");
        }
    }
    SSAC_LAUNCH((ens_gemm_kernel<A_KC, B_KC, EPI, KS>), grid, dim3(NTHREADS * KS), lds, st, g);
    return ssac_check_launch("ens_gemm");
}

template <bool A_KC, bool B_KC, int EPI>
int launch(const GemmArgs &g_in, int batch, hipStream_t st) {
    GemmArgs g = g_in;
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, batch);
    if (grid.x == 0 || grid.y == 0 || batch == 0) return 0;
    g.grid_x = grid.x; g.grid_y = grid.y; g.xcd = (g_ssac_xcd >> 1) & 1;
    const int tiles = grid.x * grid.y * batch;
    const int nchunks = (g.K + BK - 1) / BK;
    // K-split inside the workgroup when the launch cannot fill the chip with tiles alone
    if (tiles <= 256 && nchunks >= 8) return launch_ks<A_KC, B_KC, EPI, 4>(g, grid, st);
    if (tiles <= 512 && nchunks >= 4) return launch_ks<A_KC, B_KC, EPI, 2>(g, grid, st);
    return launch_ks<A_KC, B_KC, EPI, 1>(g, grid, st);
}

// Heads (N <= 8 outputs) over a deep layer: the 64x64 MFMA tile is 1/64 .. 1/8 full and the launch is 8-16
// workgroups streaming their rows one 32-deep chunk at a time (30 us at hidden 1024 / B 512).  Here a WAVE owns a row:
// the lanes take 4 consecutive k each (16-byte loads of x and of every output's weight row, all in flight at once),
// a shuffle tree sums the 64 partials.  fp32 sums, k-order differs from the MFMA chain (rounding only).
constexpr int HEAD_MAX_N = 8;
template <bool RELU>
__global__ __launch_bounds__(256) void mlp_head_rows_kernel(GemmArgs g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave, e = blockIdx.y;
    if (row >= g.M) return;
    const float *x = g.A + batch_off(g.ids, g.idsA, e, g.sA) + (int64_t)row * g.lda;
    const float *W = g.B + batch_off(g.ids, g.idsB, e, g.sB);
    float acc[HEAD_MAX_N];
#pragma unroll
    for (int n = 0; n < HEAD_MAX_N; ++n) acc[n] = 0.0f;
    for (int k = lane * 4; k < g.K; k += 256) {
        const f4 xv = *reinterpret_cast<const f4 *>(x + k);
#pragma unroll
        for (int n = 0; n < HEAD_MAX_N; ++n) {
            if (n < g.N) {
                const f4 wv = *reinterpret_cast<const f4 *>(W + (int64_t)n * g.ldb + k);
                acc[n] += xv[0] * wv[0];
                acc[n] += xv[1] * wv[1];
                acc[n] += xv[2] * wv[2];
                acc[n] += xv[3] * wv[3];
            }
        }
    }
    const float *bias = g.bias + batch_off(g.ids, g.idsB, e, g.sBias);
    float *c = g.C + batch_off(g.ids, g.idsC, e, g.sC) + (int64_t)row * g.ldc;
#pragma unroll
    for (int n = 0; n < HEAD_MAX_N; ++n) {
        if (n < g.N) {
            const float v = wave_sum(acc[n]) + bias[n];
            if (lane == 0) c[n] = RELU ? fmaxf(v, 0.0f) : v;
        }
    }
}

// the head kernel takes a forward layer with at most 8 outputs and K >= 256, 16-byte aligned rows
bool head_rows_ok(const GemmArgs &g) {
    return g.N >= 1 && g.N <= HEAD_MAX_N && g.K >= 256 && (g.K & 3) == 0 && (g.lda & 3) == 0 && (g.ldb & 3) == 0 &&
           (g.sA & 3) == 0 && (g.sB & 3) == 0 && (((uintptr_t)g.A | (uintptr_t)g.B) & 15) == 0;
}

int launch_head_rows(const GemmArgs &g, int batch, bool relu, hipStream_t st) {
    if (g.M <= 0 || batch <= 0) return 0;
    const dim3 grid((g.M + 3) / 4, batch);
    if (relu) SSAC_LAUNCH(mlp_head_rows_kernel<true>, grid, dim3(256), 0, st, g);
    else SSAC_LAUNCH(mlp_head_rows_kernel<false>, grid, dim3(256), 0, st, g);
    return ssac_check_launch("mlp_head_rows");
}

long long *g_gemm_dbg = nullptr;
int g_gemm_lean = 1;
int g_gemm_head_bias = 1;   // debugging knob (ssac_gemm_lean bit 1 = off)
extern "C" int ssac_gemm_lean(int on) { g_gemm_lean = (on & 1) ? 1 : 0; g_gemm_head_bias = (on & 2) ? 0 : 1; return 0; }

struct LayerGeom { int64_t off_w, off_b; int rows, cols; };

bool layer_geom(const ssac_mlp *n, int layer, LayerGeom &L) {
    int64_t off[6];
    ssac_mlp_layout(n->in_dim, n->hidden, n->out_dim, off);
    if (layer == 0) { L = {off[0], off[1], n->hidden, n->in_dim}; return true; }
    if (layer == 1) { L = {off[2], off[3], n->hidden, n->hidden}; return true; }
    if (layer == 2) { L = {off[4], off[5], n->out_dim, n->hidden}; return true; }
    return false;
}

}  // namespace

#ifdef SSAC_LAB   // (ssac_hip_test.h, lab hooks: the product library does not define the symbol)
extern "C" int ssac_gemm_debug_stamps(long long *dev_buf) {
    g_gemm_dbg = dev_buf;
    return 0;
}
#endif

// Which form the merged weight-gradient launch takes: 0 = automatic, 1 = 64 x 64 tiles, 2 = 32 x 32 tiles (the latency
// form) whenever the shapes allow.  Both are parity-tested on every fixture (tests/test_hip_cases.py).
extern "C" int ssac_wgrad_variant(int variant) {
    if (variant < 0 || variant > 2) return ssac_fail("ssac_wgrad_variant: 0 (automatic), 1 (64 x 64 tiles), 2 (32 x 32 tiles)");
    g_wgrad_variant = variant;
    return 0;
}

extern "C" int64_t ssac_mlp_layout(int in_dim, int hidden, int out_dim, int64_t off[6]) {
    int64_t o = 0;
    off[0] = o; o += (int64_t)hidden * in_dim;
    off[1] = o; o += hidden;
    off[2] = o; o += (int64_t)hidden * hidden;
    off[3] = o; o += hidden;
    off[4] = o; o += (int64_t)out_dim * hidden;
    off[5] = o; o += out_dim;
    return (o + 3) & ~(int64_t)3;
}

extern "C" int ssac_mlp_layer_fwd(const ssac_mlp *nets, int layer, const int32_t *net_ids, int n_sel,
                                  const float *X, int64_t ldx, int64_t x_net_stride, int n_rows,
                                  float *Y, int64_t ldy, int64_t y_net_stride, int relu, void *stream) {
    LayerGeom L;
    if (!nets || !layer_geom(nets, layer, L)) return ssac_fail("ssac_mlp_layer_fwd: bad layer");
    if (n_sel < 0 || n_sel > SSAC_MAX_NETS) return ssac_fail("ssac_mlp_layer_fwd: n_sel out of range");
    GemmArgs g{};
    g.A = X; g.lda = ldx; g.sA = x_net_stride; g.idsA = 0;
    g.B = nets->params + L.off_w; g.ldb = L.cols; g.sB = nets->net_stride; g.idsB = 1;
    g.C = Y; g.ldc = ldy; g.sC = y_net_stride; g.idsC = 0;
    g.M = n_rows; g.N = L.rows; g.K = L.cols;
    g.ids = net_ids;
    g.bias = nets->params + L.off_b; g.sBias = nets->net_stride;
    hipStream_t st = (hipStream_t)stream;
    if (head_rows_ok(g)) return launch_head_rows(g, n_sel, relu != 0, st);
    return relu ? launch<true, true, EPI_BIAS_RELU>(g, n_sel, st) : launch<true, true, EPI_BIAS>(g, n_sel, st);
}

extern "C" int ssac_mlp_layer_dgrad(const ssac_mlp *nets, int layer, const int32_t *net_ids, int n_sel,
                                    const float *dY, int64_t ldy, int64_t y_net_stride,
                                    const float *mask, int64_t ldmask, int64_t mask_net_stride,
                                    int n_rows, float *dX, int64_t ldx, int64_t x_net_stride,
                                    void *stream) {
    LayerGeom L;
    if (!nets || !layer_geom(nets, layer, L)) return ssac_fail("ssac_mlp_layer_dgrad: bad layer");
    if (n_sel < 0 || n_sel > SSAC_MAX_NETS) return ssac_fail("ssac_mlp_layer_dgrad: n_sel out of range");
    GemmArgs g{};
    g.A = dY; g.lda = ldy; g.sA = y_net_stride; g.idsA = 0;       // (n_rows x rows), K-contiguous
    g.B = nets->params + L.off_w; g.ldb = L.cols; g.sB = nets->net_stride; g.idsB = 1;  // (rows x cols)
    g.C = dX; g.ldc = ldx; g.sC = x_net_stride; g.idsC = 0;
    g.M = n_rows; g.N = L.cols; g.K = L.rows;
    g.ids = net_ids;
    g.mask = mask; g.ldmask = ldmask; g.sMask = mask_net_stride;
    hipStream_t st = (hipStream_t)stream;
    return mask ? launch<true, false, EPI_MASK>(g, n_sel, st) : launch<true, false, EPI_STORE>(g, n_sel, st);
}

extern "C" int ssac_wgrad_tiles(const ssac_mlp *nets, int layer) {
    LayerGeom L;
    if (!nets || !layer_geom(nets, layer, L)) return -1;
    // gradient-norm slots of the layer per net: one per 32 x 32 tile (sumsq_finish); heads of <= 16 outputs go through
    // the VALU head workgroups, which own one slot per 16 columns (ssac_head_wgrad.h)
    if (layer == 2 && nets->out_dim <= 16) return (nets->hidden + SSAC_HEAD_SLOT_COLS - 1) / SSAC_HEAD_SLOT_COLS;
    return ((L.rows + 31) / 32) * ((L.cols + 31) / 32);
}

namespace {
bool build_wgrad_args(GemmArgs &g, const ssac_mlp *nets, int layer, const int32_t *net_ids,
                      const float *X, int64_t ldx, int64_t x_net_stride, const float *dY, int64_t ldy,
                      int64_t y_net_stride, int n_rows, float *adam_m, float *adam_v, const ssac_adam_ctl *ctl,
                      float *grads, float *sumsq, int64_t sumsq_net_stride, float *target, float tau) {
    LayerGeom L;
    if (!nets || !layer_geom(nets, layer, L)) return false;
    g = GemmArgs{};
    g.A = dY; g.lda = ldy; g.sA = y_net_stride; g.idsA = 0;
    g.B = X; g.ldb = ldx; g.sB = x_net_stride; g.idsB = 0;
    g.C = nets->params + L.off_w; g.ldc = L.cols; g.sC = nets->net_stride; g.idsC = 1;
    g.M = L.rows; g.N = L.cols; g.K = n_rows;
    g.grid_x = (g.N + BN - 1) / BN; g.grid_y = (g.M + BM - 1) / BM;
    g.ids = net_ids;
    // bit 0 / bit 1: the A / B rows are 16-byte aligned and a multiple of 4 floats long (the 16-byte loader of the
    // general kernel); bit 2: the B rows take the LEAN kernel's loader -- any row length and leading dimension
    // (dword-aligned 16-byte loads; a ragged last vector is read element-wise): the 23-float [s | a] rows of the
    // metric shape as well
    g.vec = (((uintptr_t)dY & 15) == 0 && (ldy & 3) == 0 && (y_net_stride & 3) == 0 && (L.rows & 3) == 0 ? 1 : 0) |
            (((uintptr_t)X & 15) == 0 && (ldx & 3) == 0 && (x_net_stride & 3) == 0 && (L.cols & 3) == 0 ? 2 : 0) |
            ((((uintptr_t)X & 3) == 0 && L.cols >= 4) ? 4 : 0);
    g.pb = nets->params + L.off_b;
    g.ctl = ctl; g.sumsq = sumsq; g.sumsq_stride = sumsq_net_stride; g.dbg = g_gemm_dbg; g.tau = tau;
    if (grads) { g.gw = grads + L.off_w; g.gb = grads + L.off_b; }
    else {
        g.am = adam_m + L.off_w; g.av = adam_v + L.off_w; g.bm = adam_m + L.off_b; g.bv = adam_v + L.off_b;
        if (target) { g.tw = target + L.off_w; g.tb = target + L.off_b; }
    }
    return true;
}
}  // namespace

// fc2 and fc1 weight gradients (+Adam/Polyak or gradient store) of every selected net in ONE launch.
// sumsq1 / sumsq0: the layers' slots inside the per-net sumsq row (see ssac_mlp_layer_wgrad).
static int wgrad_merged(const ssac_mlp *nets, const int32_t *net_ids, int n_sel, const float *X, int64_t ldx,
                        int64_t x_net_stride, const float *H1, const float *DZ2, const float *DZ1,
                        const float *H2, const float *DQ, int n_rows, float *adam_m, float *adam_v,
                        const ssac_adam_ctl *ctl, float *grads, float *sumsq1, float *sumsq0, float *sumsq2,
                        int64_t sumsq_net_stride, float *target, float tau, void *stream,
                        const float *rowscale = nullptr, const LossFoldArgs *lossfold = nullptr,
                        const ssac_logfold *logfold = nullptr, const float *w3_snapshot = nullptr,
                        const ssac_actor_logfold *afold = nullptr);

extern "C" int ssac_mlp_wgrad_all_scaled(const ssac_mlp *nets, const int32_t *net_ids, int n_sel, const float *X,
                                         int64_t ldx, int64_t x_net_stride, const float *H1, const float *H2,
                                         const float *DZ2u, const float *DZ1u, const float *row_scale, int n_rows,
                                         float *adam_m, float *adam_v, const ssac_adam_ctl *ctl, float *grads,
                                         float *sumsq2, float *sumsq1, float *sumsq0, int64_t sumsq_net_stride,
                                         float *target, float tau, void *stream) {
    if (!nets || nets->out_dim != 1) return ssac_fail("ssac_mlp_wgrad_all_scaled: single-output heads only");
    if (!H2 || !row_scale) return ssac_fail("ssac_mlp_wgrad_all_scaled: H2 / row_scale missing");
    // out_dim == 1: the per-row scale IS the head's output gradient dq (n_sel x n_rows x 1)
    return wgrad_merged(nets, net_ids, n_sel, X, ldx, x_net_stride, H1, DZ2u, DZ1u, H2, row_scale, n_rows, adam_m,
                        adam_v, ctl, grads, sumsq1, sumsq0, sumsq2, sumsq_net_stride, target, tau, stream, row_scale);
}

extern "C" int ssac_mlp_wgrad_all_lossfold(const ssac_mlp *nets, const float *X, int64_t ldx, int64_t x_net_stride,
                                           const float *H1, const float *H2, const float *DZ2u, const float *DZ1u,
                                           const float *W3_snapshot,
                                           const float *Q, const float *td, const ssac_td_spec *lazy_td,
                                           const float *weight, const ssac_popart *popart, int pop, float denom,
                                           float *partials, int n_rows, float *adam_m, float *adam_v,
                                           const ssac_adam_ctl *ctl, float *grads, float *sumsq2, float *sumsq1,
                                           float *sumsq0, int64_t sumsq_net_stride, float *target, float tau,
                                           const ssac_logfold *logfold, void *stream) {
    if (!nets || nets->out_dim != 1) return ssac_fail("ssac_mlp_wgrad_all_lossfold: single-output heads only");
    if (!H2 || !Q || !partials || (!td && !lazy_td)) return ssac_fail("ssac_mlp_wgrad_all_lossfold: missing argument");
    // DZ2u == NULL: the fc2 tiles rebuild dz2u = W3 (.) [h2 > 0] from H2 while staging it (16-byte operand rows needed)
    if (!DZ2u && (!W3_snapshot || (nets->hidden & 3) || (((uintptr_t)H2 | (uintptr_t)W3_snapshot) & 15)))
        return ssac_fail("ssac_mlp_wgrad_all_lossfold: DZ2u == NULL needs the W3 snapshot and 16-byte aligned H2 rows");
    if (n_rows > 4096) return ssac_fail("ssac_mlp_wgrad_all_lossfold: more than 4096 rows (use ssac_critic_loss_bwd)");
    LossFoldArgs lf{};
    lf.q = Q; lf.td = td; if (lazy_td) lf.tds = *lazy_td;
    lf.weight = weight; lf.popart = popart; lf.pop = pop; lf.denom = denom; lf.partials = partials; lf.n_rows = n_rows;
    return wgrad_merged(nets, nullptr, nets->n_nets, X, ldx, x_net_stride, H1, DZ2u, DZ1u, H2, Q, n_rows, adam_m,
                        adam_v, ctl, grads, sumsq1, sumsq0, sumsq2, sumsq_net_stride, target, tau, stream, nullptr, &lf,
                        logfold, W3_snapshot);
}

extern "C" int ssac_mlp_wgrad_fc12(const ssac_mlp *nets, const int32_t *net_ids, int n_sel, const float *X,
                                   int64_t ldx, int64_t x_net_stride, const float *H1, const float *DZ2,
                                   const float *DZ1, int n_rows, float *adam_m, float *adam_v,
                                   const ssac_adam_ctl *ctl, float *grads, float *sumsq1, float *sumsq0,
                                   int64_t sumsq_net_stride, float *target, float tau, void *stream) {
    return wgrad_merged(nets, net_ids, n_sel, X, ldx, x_net_stride, H1, DZ2, DZ1, nullptr, nullptr, n_rows, adam_m,
                        adam_v, ctl, grads, sumsq1, sumsq0, nullptr, sumsq_net_stride, target, tau, stream);
}

extern "C" int ssac_mlp_wgrad_all(const ssac_mlp *nets, const int32_t *net_ids, int n_sel, const float *X,
                                  int64_t ldx, int64_t x_net_stride, const float *H1, const float *H2,
                                  const float *DZ2, const float *DZ1, const float *DQ, int n_rows, float *adam_m,
                                  float *adam_v, const ssac_adam_ctl *ctl, float *grads, float *sumsq2,
                                  float *sumsq1, float *sumsq0, int64_t sumsq_net_stride, float *target, float tau,
                                  void *stream) {
    if (!nets || nets->out_dim > 16) return ssac_fail("ssac_mlp_wgrad_all: head wider than 16 outputs");
    if (!H2 || !DQ) return ssac_fail("ssac_mlp_wgrad_all: H2 / DQ missing");
    return wgrad_merged(nets, net_ids, n_sel, X, ldx, x_net_stride, H1, DZ2, DZ1, H2, DQ, n_rows, adam_m, adam_v, ctl,
                        grads, sumsq1, sumsq0, sumsq2, sumsq_net_stride, target, tau, stream);
}

static int wgrad_merged(const ssac_mlp *nets, const int32_t *net_ids, int n_sel, const float *X, int64_t ldx,
                        int64_t x_net_stride, const float *H1, const float *DZ2, const float *DZ1,
                        const float *H2, const float *DQ, int n_rows, float *adam_m, float *adam_v,
                        const ssac_adam_ctl *ctl, float *grads, float *sumsq1, float *sumsq0, float *sumsq2,
                        int64_t sumsq_net_stride, float *target, float tau, void *stream,
                        const float *rowscale, const LossFoldArgs *lossfold, const ssac_logfold *logfold,
                        const float *w3_snapshot, const ssac_actor_logfold *afold) {
    if (n_sel < 0 || n_sel > SSAC_MAX_NETS) return ssac_fail("ssac_mlp_wgrad_fc12: n_sel out of range");
    if (!grads && (!adam_m || !adam_v || !ctl)) return ssac_fail("ssac_mlp_wgrad_fc12: Adam state missing");
    if (n_sel == 0 || n_rows <= 0) return 0;
    const int H = nets->hidden;
    GemmPair p{};
    int64_t off[6];
    ssac_mlp_layout(nets->in_dim, nets->hidden, nets->out_dim, off);
    if (H2) {  // head-layer weight gradient as extra workgroups of the same launch
        p.head = HeadWgradArgs{nets->params, nets->net_stride, nets->hidden, nets->out_dim, off[4], off[5], net_ids, H2,
                               DQ, n_rows, adam_m, adam_v, ctl, grads, sumsq2, sumsq_net_stride, target, tau};
        p.head_grid_x = (nets->hidden + 63) / 64;
    }
    // DZ2 == NULL (loss-fold launches only): the fc2 tiles read H2 and rebuild dz2u = W3 (.) [h2 > 0] in their staging
    const float *A2 = DZ2 ? DZ2 : H2;
    if (!A2) return ssac_fail("ssac_mlp_wgrad_fc12: DZ2 missing");
    if (!build_wgrad_args(p.g0, nets, 1, net_ids, H1, H, (int64_t)n_rows * H, A2, H, (int64_t)n_rows * H, n_rows,
                          adam_m, adam_v, ctl, grads, sumsq1, sumsq_net_stride, target, tau) ||
        !build_wgrad_args(p.g1, nets, 0, net_ids, X, ldx, x_net_stride, DZ1, H, (int64_t)n_rows * H, n_rows,
                          adam_m, adam_v, ctl, grads, sumsq0, sumsq_net_stride, target, tau))
        return ssac_fail("ssac_mlp_wgrad_fc12: bad arena");
    if (!DZ2) {
        if (!(p.g0.vec & 1) || nets->out_dim != 1 || net_ids || !w3_snapshot)
            return ssac_fail("ssac_mlp_wgrad_fc12: the sign-rebuilt dz2u needs the W3 snapshot, 16-byte aligned H2 rows and the whole ensemble");
        p.g0.a_sign_w = w3_snapshot;
    }
    if (p.head_grid_x > 0 && nets->out_dim == 1 && g_gemm_head_bias && (lossfold || rowscale)) {
        // single-output heads, UNSCALED backward (dz2 = c_row * (W3 (.) [h2 > 0]) by construction: nothing but the head
        // reaches fc2's output -- not so with DR3, whose launches pass a scaled dz2): the head workgroups read h2 anyway
        // and take the fc2 bias gradient as well (HeadWgradArgs::with_b2) -- the same sums in the same order whether
        // dz2u was stored or is rebuilt from h2's sign
        p.g0.no_bias = 1;
        p.head.with_b2 = 1;
        p.head.off_b2 = off[3];
    }
    p.g1.dbg = nullptr;   // (debug stamps: the first fc2 tile only -- both problems have a workgroup (0, 0, 0))
    if (rowscale) { p.g0.rowscale = p.g1.rowscale = rowscale; p.g0.sRow = p.g1.sRow = n_rows; }
    if (lossfold) p.lf = *lossfold;
    if (logfold && logfold->done_counter) {
        if (!lossfold || grads || !sumsq0 || !sumsq1 || !sumsq2 || net_ids)
            return ssac_fail("ssac_mlp_wgrad_all_lossfold: the folded logs need the loss fold, Adam mode and sumsq slots");
        // the gradient-norm partials of the launch: n_sel rows of sumsq_net_stride slots starting at the lowest pointer
        const float *ss_base = sumsq0 < sumsq1 ? sumsq0 : sumsq1;
        if (sumsq2 < ss_base) ss_base = sumsq2;
        p.fold = LogFoldArgs{logfold->done_counter, logfold->logs, lossfold->tds.q_t ? logfold->td_logs : nullptr,
                             logfold->feed, nullptr, lossfold->partials, n_sel, ss_base,
                             (int)(n_sel * sumsq_net_stride), n_rows, lossfold->denom};
    } else if (logfold && logfold->deferred_stats) {
        if (!lossfold || grads || !logfold->feed)
            return ssac_fail("ssac_mlp_wgrad_all_lossfold: deferred logs need the loss fold, Adam mode and a feed");
        p.fold = LogFoldArgs{};
        p.fold.feed = logfold->feed;
        p.fold.deferred_stats = lossfold->tds.q_t ? logfold->deferred_stats : nullptr;
        p.fold.n_rows = n_rows;
        if (!p.fold.deferred_stats) return ssac_fail("ssac_mlp_wgrad_all_lossfold: deferred logs need the in-launch TD target");
    }
    if (logfold && logfold->late_word) {
        if (!target || grads) return ssac_fail("ssac_mlp_wgrad_all_lossfold: the late-bound Polyak needs a target arena and Adam mode");
        p.late_word = logfold->late_word;
    }
    bool actor_ring = false;
    if (afold && afold->done_counter) {
        // the online actor update's two logs ride in this launch (ssac_actor_logs' arithmetic, by the last workgroup to arrive)
        if (lossfold || logfold || grads || !sumsq0 || !sumsq1 || !sumsq2 || net_ids || n_sel != 1 || !afold->partials ||
            afold->n_tiles <= 0 || afold->n_rows <= 0 || !afold->logs_loss || (afold->ring && (!afold->block || afold->width <= 0 || afold->ring_slot < 0)))
            return ssac_fail("ssac_mlp_wgrad_all_actor: the folded actor logs need ONE net, Adam mode, every sumsq slot and the tiles' loss terms");
        const float *ss_base = sumsq0 < sumsq1 ? sumsq0 : sumsq1;
        if (sumsq2 < ss_base) ss_base = sumsq2;
        p.fold = LogFoldArgs{};
        p.fold.done = afold->done_counter;
        p.fold.logs = afold->logs_loss;
        p.fold.partials = afold->partials; p.fold.n_nets = afold->n_tiles;
        p.fold.sumsq = ss_base; p.fold.n_ss = (int)sumsq_net_stride;
        p.fold.n_rows = afold->n_rows;
        p.fold.actor = 1; p.fold.a_scale = -afold->inv_members / (float)afold->n_rows; p.fold.a_gn = afold->logs_gn;
        p.fold.a_block = afold->block; p.fold.a_width = afold->width;
        p.fold.a_pub = afold->ring ? afold->ring + afold->ring_slot * afold->width : nullptr;
        actor_ring = afold->ring != nullptr;
    }
    // (a recorded launch: every replay names its own ring slot -- ssac_replay_value2's second number)
    auto ring_patch = [&]() {
        if (actor_ring)
            ssac_record_value_patch(0, offsetof(GemmPair, fold) + offsetof(LogFoldArgs, a_pub), 2, (long long)(uintptr_t)afold->ring,
                                    (long long)afold->width * 4);
    };
    hipStream_t st = (hipStream_t)stream;
    const int tiles = (p.g0.grid_x * p.g0.grid_y + p.g1.grid_x * p.g1.grid_y) * n_sel;
    const int nchunks = (n_rows + BK - 1) / BK;
    {
        // latency form (wgrad_small_pair_kernel, 32 x 32 tiles): taken while ALL of its workgroups are resident at once
        // (two per CU) -- the launch then lasts one short workgroup instead of one long one.  (A 64 x 32 form of the same
        // body -- half the workgroups, for launches too big for this one -- was built and measured: 57.2 vs 53.0 us per
        // update at N 10, 63.7 vs 59.6 at N 8: its 256 dword loads per lane make the address unit a second bottleneck
        // beside the matrix pipe.  Dropped; profiles/r4_latency_forms.md.)
        auto t32 = [](const GemmArgs &g) { return ((g.M + ST - 1) / ST) * ((g.N + ST - 1) / ST); };
        const int small_wgs = (t32(p.g0) + t32(p.g1) + (p.head_grid_x > 0 ? (H + S_HEAD_COLS - 1) / S_HEAD_COLS : 0)) * n_sel + 1;
        const bool can = small_ok(p.g0) && small_ok(p.g1) && nets->out_dim <= 16;
        if (can && (g_wgrad_variant == 2 || (g_wgrad_variant == 0 && small_wgs <= g_wgrad_small_max))) {
            const int rc = grads ? launch_pair_small<EPI_GRAD, 1>(p, n_sel, st) : launch_pair_small<EPI_ADAM, 1>(p, n_sel, st);
            ring_patch();
            return rc;
        }
    }
    if (grads) {
        if (tiles <= 256 && nchunks >= 8) return launch_pair_ks<false, false, EPI_GRAD, 4>(p, n_sel, n_sel, st);
        if (tiles <= 512 && nchunks >= 4) return launch_pair_ks<false, false, EPI_GRAD, 2>(p, n_sel, n_sel, st);
        return launch_pair_ks<false, false, EPI_GRAD, 1>(p, n_sel, n_sel, st);
    }
    // (measured again in round 3 at the metric shape: 4 K-groups 58.3 us per update, 2 K-groups 61.4, none 73.3)
    const int rc = (tiles <= 256 && nchunks >= 8) ? launch_pair_ks<false, false, EPI_ADAM, 4>(p, n_sel, n_sel, st)
                   : (tiles <= 512 && nchunks >= 4) ? launch_pair_ks<false, false, EPI_ADAM, 2>(p, n_sel, n_sel, st)
                                                    : launch_pair_ks<false, false, EPI_ADAM, 1>(p, n_sel, n_sel, st);
    ring_patch();
    return rc;
}

// ssac_mlp_wgrad_all with the online actor update's two logs folded in (round 6; was a launch of its own, ssac_actor_logs)
extern "C" int ssac_mlp_wgrad_all_actor(const ssac_mlp *nets, const float *X, int64_t ldx, const float *H1, const float *H2,
                                        const float *DZ2, const float *DZ1, const float *DQ, int n_rows, float *adam_m,
                                        float *adam_v, const ssac_adam_ctl *ctl, float *sumsq2, float *sumsq1, float *sumsq0,
                                        int64_t sumsq_net_stride, const ssac_actor_logfold *fold, void *stream) {
    if (!nets || nets->out_dim > 16 || nets->n_nets != 1) return ssac_fail("ssac_mlp_wgrad_all_actor: one net, head of <= 16 outputs");
    if (!H2 || !DQ || !fold) return ssac_fail("ssac_mlp_wgrad_all_actor: H2 / DQ / fold missing");
    return wgrad_merged(nets, nullptr, 1, X, ldx, 0, H1, DZ2, DZ1, H2, DQ, n_rows, adam_m, adam_v, ctl, nullptr, sumsq1, sumsq0,
                        sumsq2, sumsq_net_stride, nullptr, 0.0f, stream, nullptr, nullptr, nullptr, nullptr, fold);
}

extern "C" int ssac_mlp_layer_wgrad(const ssac_mlp *nets, int layer, const int32_t *net_ids, int n_sel,
                                    const float *X, int64_t ldx, int64_t x_net_stride,
                                    const float *dY, int64_t ldy, int64_t y_net_stride, int n_rows,
                                    float *adam_m, float *adam_v, const ssac_adam_ctl *ctl,
                                    float *grads, float *sumsq, int64_t sumsq_net_stride,
                                    float *target, float tau, void *stream) {
    LayerGeom L;
    if (!nets || !layer_geom(nets, layer, L)) return ssac_fail("ssac_mlp_layer_wgrad: bad layer");
    if (n_sel < 0 || n_sel > SSAC_MAX_NETS) return ssac_fail("ssac_mlp_layer_wgrad: n_sel out of range");
    if (!grads && (!adam_m || !adam_v || !ctl)) return ssac_fail("ssac_mlp_layer_wgrad: Adam state missing");
    GemmArgs g{};
    g.A = dY; g.lda = ldy; g.sA = y_net_stride; g.idsA = 0;  // (n_rows x rows) read as K x M
    g.B = X; g.ldb = ldx; g.sB = x_net_stride; g.idsB = 0;   // (n_rows x cols) read as K x N
    g.C = nets->params + L.off_w; g.ldc = L.cols; g.sC = nets->net_stride; g.idsC = 1;
    g.M = L.rows; g.N = L.cols; g.K = n_rows;
    g.ids = net_ids;
    g.vec = (((uintptr_t)dY & 15) == 0 && (ldy & 3) == 0 && (y_net_stride & 3) == 0 && (L.rows & 3) == 0 ? 1 : 0) |
            (((uintptr_t)X & 15) == 0 && (ldx & 3) == 0 && (x_net_stride & 3) == 0 && (L.cols & 3) == 0 ? 2 : 0);
    g.pb = nets->params + L.off_b;
    g.ctl = ctl;
    g.sumsq = sumsq;
    g.sumsq_stride = sumsq_net_stride;
    g.dbg = g_gemm_dbg;
    g.tau = tau;
    hipStream_t st = (hipStream_t)stream;
    if (grads) {
        g.gw = grads + L.off_w; g.gb = grads + L.off_b;
        return launch<false, false, EPI_GRAD>(g, n_sel, st);
    }
    g.am = adam_m + L.off_w; g.av = adam_v + L.off_w;
    g.bm = adam_m + L.off_b; g.bv = adam_v + L.off_b;
    if (target) { g.tw = target + L.off_w; g.tb = target + L.off_b; }
    return launch<false, false, EPI_ADAM>(g, n_sel, st);
}

// ---------------------------------------------------------------------------------------------
// Single-problem GEMM entry points for the pixel encoders (convolutions run as im2col + GEMM).
// ---------------------------------------------------------------------------------------------
extern "C" int ssac_linear_fwd(const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias,
                               float *Y, int64_t ldy, int M, int N, int K, int relu, void *stream) {
    GemmArgs g{};
    g.A = X; g.lda = ldx; g.B = W; g.ldb = ldw; g.C = Y; g.ldc = ldy;
    g.M = M; g.N = N; g.K = K; g.bias = bias;
    hipStream_t st = (hipStream_t)stream;
    if (!bias) return ssac_fail("ssac_linear_fwd: bias required");
    return relu ? launch<true, true, EPI_BIAS_RELU>(g, 1, st) : launch<true, true, EPI_BIAS>(g, 1, st);
}

// Y = X W^T + b for a short, very deep problem (the encoder's fc: M = batch, N = embedding, K = C*H*W): the M x N
// tiles alone leave the chip idle, so K is cut into slices of k_per_slice (a multiple of 32) that run as
// separate workgroups into `partial` (slices x M x N), then ssac_reduce_slices_bias sums them in a fixed order.
extern "C" int ssac_linear_fwd_splitk(const float *X, int64_t ldx, const float *W, int64_t ldw, float *partial,
                                      int M, int N, int K, int k_per_slice, void *stream) {
    if (k_per_slice <= 0 || (k_per_slice & 31)) return ssac_fail("ssac_linear_fwd_splitk: slice must be a multiple of 32");
    const int slices = (K + k_per_slice - 1) / k_per_slice;
    GemmArgs g{};
    g.A = X; g.lda = ldx; g.sA = k_per_slice;
    g.B = W; g.ldb = ldw; g.sB = k_per_slice;
    g.C = partial; g.ldc = N; g.sC = (int64_t)M * N;
    g.M = M; g.N = N; g.K = k_per_slice; g.Ktot = K;
    return launch<true, true, EPI_STORE>(g, slices, (hipStream_t)stream);
}

// The same split-K forward for N <= 64 outputs as a pure operand STREAM (the pixel encoders' fc: M 512, N 50, K 39 200 --
// 80 MB of X against 2.6 GFLOP: both roofs near 20 us, the tiled kernel above takes 45).  A wave owns (32 rows, one K
// slice): both operands are K-contiguous, so a lane reads ITS row 16 bytes at a time (lane half h takes k + 4 h .. + 3 of
// every 8) straight into the MFMA operand registers -- no LDS, no barrier -- and keeps FS_DEPTH groups of 8 k in flight
// (one wave per SIMD: the loads must cover the round trip themselves).  The X fragment feeds both 32-column halves of the
// output.  The four waves of a workgroup take four row blocks of the SAME slice (their W reads hit the L1).  Partials
// [slice][M][N] as ssac_linear_fwd_splitk; the k order inside a slice differs from the tiled kernel's (rounding only).
constexpr int FS_DEPTH = 8;
__global__ __launch_bounds__(256) void linear_fwd_stream_kernel(const float *__restrict__ X, int64_t ldx,
                                                                const float *__restrict__ W, int64_t ldw,
                                                                float *__restrict__ partial, int M, int N, int K,
                                                                int k_per_slice) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const int slice = blockIdx.x, r0 = (blockIdx.y * 4 + wave) * 32;
    if (r0 >= M) return;
    const int k_lo = slice * k_per_slice, k_hi = min(K, k_lo + k_per_slice), ngroups = (k_hi - k_lo) >> 3;
    const float *xr = X + (int64_t)min(r0 + li, M - 1) * ldx + k_lo + 4 * lh;
    const float *w0 = W + (int64_t)min(li, N - 1) * ldw + k_lo + 4 * lh;
    const float *w1 = W + (int64_t)min(32 + li, N - 1) * ldw + k_lo + 4 * lh;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
    f4 xa[FS_DEPTH], wa[FS_DEPTH], wb[FS_DEPTH];
#pragma unroll
    for (int d = 0; d < FS_DEPTH; ++d) {
        if (d < ngroups) {
            xa[d] = *reinterpret_cast<const f4 *>(xr + 8 * d);
            wa[d] = *reinterpret_cast<const f4 *>(w0 + 8 * d);
            wb[d] = *reinterpret_cast<const f4 *>(w1 + 8 * d);
        }
    }
    for (int g0 = 0; g0 < ngroups; g0 += FS_DEPTH) {
#pragma unroll
        for (int d = 0; d < FS_DEPTH; ++d) {
            if (g0 + d < ngroups) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[d][j], wa[d][j], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[d][j], wb[d][j], acc1, 0, 0, 0);
                }
            }
            if (g0 + d + FS_DEPTH < ngroups) {
                const int o = 8 * (g0 + d + FS_DEPTH);
                xa[d] = *reinterpret_cast<const f4 *>(xr + o);
                wa[d] = *reinterpret_cast<const f4 *>(w0 + o);
                wb[d] = *reinterpret_cast<const f4 *>(w1 + o);
            }
        }
    }
    float *out = partial + (int64_t)slice * M * N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M) {
            if (li < N) out[(int64_t)row * N + li] = acc0[r];
            if (32 + li < N) out[(int64_t)row * N + 32 + li] = acc1[r];
        }
    }
}

// 1 when ssac_linear_fwd_stream covers the problem: N <= 64, K and the slice multiples of 8, 16-byte aligned rows
extern "C" int ssac_linear_fwd_stream_supported(int M, int N, int K, int k_per_slice, int64_t ldx, int64_t ldw) {
    return M > 0 && N >= 1 && N <= 64 && K >= 8 && (K & 7) == 0 && k_per_slice >= 8 && (k_per_slice & 7) == 0 &&
           (ldx & 3) == 0 && (ldw & 3) == 0;
}

extern "C" int ssac_linear_fwd_stream(const float *X, int64_t ldx, const float *W, int64_t ldw, float *partial, int M,
                                      int N, int K, int k_per_slice, void *stream) {
    if (!ssac_linear_fwd_stream_supported(M, N, K, k_per_slice, ldx, ldw) || (((uintptr_t)X | (uintptr_t)W) & 15))
        return ssac_fail("ssac_linear_fwd_stream: shape / alignment not covered (see ssac_linear_fwd_stream_supported)");
    const int slices = (K + k_per_slice - 1) / k_per_slice;
    SSAC_LAUNCH(linear_fwd_stream_kernel, dim3(slices, (M + 127) / 128), dim3(256), 0, (hipStream_t)stream, X, ldx, W, ldw,
                partial, M, N, K, k_per_slice);
    return ssac_check_launch("linear_fwd_stream");
}

extern "C" int ssac_linear_dgrad(const float *dY, int64_t ldy, const float *W, int64_t ldw, float *dX,
                                 int64_t ldx, int M, int N_in, int K_out, void *stream) {
    GemmArgs g{};
    g.A = dY; g.lda = ldy; g.B = W; g.ldb = ldw; g.C = dX; g.ldc = ldx;
    g.M = M; g.N = N_in; g.K = K_out;
    return launch<true, false, EPI_STORE>(g, 1, (hipStream_t)stream);
}

extern "C" int ssac_linear_dgrad_masked(const float *dY, int64_t ldy, const float *W, int64_t ldw, const float *mask,
                                        int64_t ldmask, float *dX, int64_t ldx, int M, int N_in, int K_out, void *stream) {
    if (!mask) return ssac_fail("ssac_linear_dgrad_masked: null mask");
    GemmArgs g{};
    g.A = dY; g.lda = ldy; g.B = W; g.ldb = ldw; g.C = dX; g.ldc = ldx;
    g.M = M; g.N = N_in; g.K = K_out;
    g.mask = mask; g.ldmask = ldmask;
    return launch<true, false, EPI_MASK>(g, 1, (hipStream_t)stream);
}

// dW partials: for slice z, partial_w[z] (M_out x N_in) = dY[rows of z]^T X[rows of z],
// partial_b[z] (M_out) = colsum(dY[rows of z]); rows_per_slice rows each (the last one shorter).
extern "C" int ssac_linear_wgrad_splitk(const float *dY, int64_t ldy, const float *X, int64_t ldx,
                                        float *partial_w, float *partial_b, int M_out, int N_in,
                                        int n_rows, int rows_per_slice, void *stream) {
    if (rows_per_slice <= 0) return ssac_fail("ssac_linear_wgrad_splitk: bad slice size");
    const int slices = (n_rows + rows_per_slice - 1) / rows_per_slice;
    GemmArgs g{};
    g.A = dY; g.lda = ldy; g.sA = (int64_t)rows_per_slice * ldy;
    g.B = X; g.ldb = ldx; g.sB = (int64_t)rows_per_slice * ldx;
    g.C = partial_w; g.ldc = N_in; g.sC = (int64_t)M_out * N_in;
    g.gw = partial_w; g.gb = partial_b; g.sGb = M_out;
    g.M = M_out; g.N = N_in; g.K = rows_per_slice; g.Ktot = n_rows;
    g.vec = (((uintptr_t)dY & 15) == 0 && (ldy & 3) == 0 && (M_out & 3) == 0 ? 1 : 0) |
            (((uintptr_t)X & 15) == 0 && (ldx & 3) == 0 && (N_in & 3) == 0 ? 2 : 0);
    return launch<false, false, EPI_GRAD>(g, slices, (hipStream_t)stream);
}
