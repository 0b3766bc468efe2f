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

#include "ssac_internal.h"

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
};

__device__ __forceinline__ int64_t batch_off(const int32_t *ids, int use_ids, int e, int64_t stride) {
    return (int64_t)((use_ids && ids) ? ids[e] : e) * stride;
}

// ---- global -> register staging (8 floats per thread per operand per chunk) ----
template <bool KCONTIG>
__device__ __forceinline__ void load_chunk(float (&r)[8], const float *__restrict__ S, int64_t ld,
                                           int R0, int R, int k0, int K, int tid) {
    if (KCONTIG) {
        // S is (R x K) row-major: thread -> k = tid&31, rows (tid>>5) + 8p
        const int kk = tid & 31, rr = tid >> 5;
        const bool kok = (k0 + kk) < K;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = R0 + rr + 8 * p;
            r[p] = (kok && row < R) ? S[(int64_t)row * ld + k0 + kk] : 0.0f;
        }
    } else {
        // S is (K x R) row-major: thread -> row = tid&63, k = (tid>>6) + 4p
        const int rr = tid & 63, kk = tid >> 6;
        const bool rok = (R0 + rr) < R;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int k = k0 + kk + 4 * p;
            r[p] = (rok && k < K) ? S[(int64_t)k * ld + R0 + rr] : 0.0f;
        }
    }
}

template <bool KCONTIG>
__device__ __forceinline__ void store_chunk(const float (&r)[8], float *__restrict__ Xs, int tid) {
    if (KCONTIG) {
        const int kk = tid & 31, rr = tid >> 5;
#pragma unroll
        for (int p = 0; p < 8; ++p) Xs[(rr + 8 * p) * LDS_KC + kk] = r[p];
    } else {
        const int rr = tid & 63, kk = tid >> 6;
#pragma unroll
        for (int p = 0; p < 8; ++p) Xs[(kk + 4 * p) * LDS_RC + rr] = r[p];
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
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    return p - c.step_size * (m / denom);
}

template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(NTHREADS) void ens_gemm_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE_FLOATS + 64];
    float *As = lds;
    float *Bs = lds + TILE_FLOATS;
    float *red = lds + 2 * TILE_FLOATS;  // 64 floats: bias-grad / sumsq scratch

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int e = blockIdx.z;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    const float *A = g.A + batch_off(g.ids, g.idsA, e, g.sA);
    const float *B = g.B + batch_off(g.ids, g.idsB, e, g.sB);
    const int64_t coff = batch_off(g.ids, g.idsC, e, g.sC);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;

    float ra[8], rb[8];
    float bias_acc = 0.0f;  // TN mode, column sums of A (bias gradient), threads < 64 of n-tile 0
    const bool want_bias_grad = (EPI == EPI_ADAM || EPI == EPI_GRAD) && blockIdx.x == 0;

    const int nchunks = (g.K + BK - 1) / BK;
    load_chunk<A_KC>(ra, A, g.lda, m0, g.M, 0, g.K, tid);
    load_chunk<B_KC>(rb, B, g.ldb, n0, g.N, 0, g.K, tid);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();  // everyone finished reading the previous chunk
        store_chunk<A_KC>(ra, As, tid);
        store_chunk<B_KC>(rb, Bs, tid);
        __syncthreads();
        if (c + 1 < nchunks) {
            load_chunk<A_KC>(ra, A, g.lda, m0, g.M, (c + 1) * BK, g.K, tid);
            load_chunk<B_KC>(rb, B, g.ldb, n0, g.N, (c + 1) * BK, g.K, tid);
        }
        if (want_bias_grad && tid < 64) {
            // A is staged row-contiguous in the weight-gradient mode: As[k][m]
#pragma unroll 8
            for (int k = 0; k < BK; ++k) bias_acc += As[k * LDS_RC + tid];
        }
#pragma unroll
        for (int t = 0; t < BK / 2; ++t) {
            const float a = frag<A_KC>(As, wm * 32 + li, 2 * t + lh);
            const float b = frag<B_KC>(Bs, wn * 32 + li, 2 * t + lh);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }

    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int gn = n0 + wn * 32 + li;
    const bool nok = gn < g.N;
    float ss = 0.0f;
    ssac_adam_ctl ctl;
    if (EPI == EPI_ADAM) ctl = *g.ctl;
    float bias_n = 0.0f;
    if ((EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) && nok)
        bias_n = g.bias[batch_off(g.ids, g.idsB, e, g.sBias) + gn];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (!(nok && gm < g.M)) continue;
        float val = acc[r];
        const int64_t ci = coff + (int64_t)gm * g.ldc + gn;
        if (EPI == EPI_STORE) {
            g.C[ci] = val;
        } else if (EPI == EPI_BIAS) {
            g.C[ci] = val + bias_n;
        } else if (EPI == EPI_BIAS_RELU) {
            g.C[ci] = fmaxf(val + bias_n, 0.0f);
        } else if (EPI == EPI_MASK) {
            const float hm = g.mask[(int64_t)e * g.sMask + (int64_t)gm * g.ldmask + gn];
            g.C[ci] = hm > 0.0f ? val : 0.0f;
        } else if (EPI == EPI_GRAD) {
            g.gw[ci] = val;
            ss += val * val;
        } else if (EPI == EPI_ADAM) {
            ss += val * val;
            float m = g.am[ci], v = g.av[ci];
            const float pn = adam_elem(g.C[ci], val, m, v, ctl);
            g.am[ci] = m;
            g.av[ci] = v;
            g.C[ci] = pn;
            if (g.tw) g.tw[ci] = g.tw[ci] * (1.0f - g.tau) + pn * g.tau;
        }
    }
    if (EPI == EPI_ADAM || EPI == EPI_GRAD) {
        if (want_bias_grad && tid < 64) {
            const int gm = m0 + tid;
            if (gm < g.M) {
                const int64_t bi = coff + gm;
                ss += bias_acc * bias_acc;
                if (EPI == EPI_GRAD) {
                    g.gb[bi] = bias_acc;
                } else {
                    float m = g.bm[bi], v = g.bv[bi];
                    const float pn = adam_elem(g.pb[bi], bias_acc, m, v, ctl);
                    g.bm[bi] = m;
                    g.bv[bi] = v;
                    g.pb[bi] = pn;
                    if (g.tb) g.tb[bi] = g.tb[bi] * (1.0f - g.tau) + pn * g.tau;
                }
            }
        }
        if (g.sumsq) {
            ss = wave_sum(ss);
            __syncthreads();
            if (lane == 0) red[wave] = ss;
            __syncthreads();
            if (tid == 0) {
                g.sumsq[(int64_t)e * g.sumsq_stride + blockIdx.y * gridDim.x + blockIdx.x] =
                    red[0] + red[1] + red[2] + red[3];
            }
        }
    }
}

template <bool A_KC, bool B_KC, int EPI>
int launch(const GemmArgs &g, int batch, hipStream_t st) {
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, batch);
    if (grid.x == 0 || grid.y == 0 || batch == 0) return 0;
    hipLaunchKernelGGL((ens_gemm_kernel<A_KC, B_KC, EPI>), grid, dim3(NTHREADS), 0, st, g);
    return ssac_check_launch("ens_gemm");
}

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
    return ((L.rows + BM - 1) / BM) * ((L.cols + BN - 1) / BN);
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
    g.pb = nets->params + L.off_b;
    g.ctl = ctl;
    g.sumsq = sumsq;
    g.sumsq_stride = sumsq_net_stride;
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
