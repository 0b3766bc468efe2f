// Pixel-encoder pieces (reference nets/cnns.py:37-103) that are not GEMMs: patch gather
// (im2col) with the input normalisation fused, its adjoint (col2im) with the ReLU mask fused,
// LayerNorm+tanh forward/backward, slice reduction for the split-K weight gradients.
//
// Convolutions run as  im2col -> ens_gemm (exact-fp32 MFMA, bias+ReLU epilogue)  with activations
// kept channels-last, i.e. the GEMM output (rows = (b, y, x), cols = channel) IS the next layer's
// input; column order of a patch is (c, ky, kx) = the flattened nn.Conv2d weight, so the weight
// tensor is used in place.  The final nn.Linear over the NCHW-flattened feature map is the same
// gather with kernel = the whole map.  These kernels are HBM-bound byte movers: one thread per
// element, consecutive threads along the contiguous output dimension.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ssac_internal.h"

namespace {

struct Im2colArgs {
    const void *src; int src_u8;
    int64_t sb, sc, sy, sx;   // element strides of the source for (batch, channel, y, x)
    int B, C, Hi, Wi, k, stride, Ho, Wo;
    float div, shift;         // value = float(src) / div + shift   (cnns.py:60: obs/255 - 0.5)
    float *col;               // (B*Ho*Wo, C*k*k)
};

__global__ void im2col_kernel(Im2colArgs a) {
    const int64_t ckk = (int64_t)a.C * a.k * a.k;
    const int64_t total = (int64_t)a.B * a.Ho * a.Wo * ckk;
    const uint8_t *s8 = (const uint8_t *)a.src;
    const float *sf = (const float *)a.src;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / ckk;
        const int col = (int)(i - row * ckk);
        const int c = col / (a.k * a.k), kr = col - c * a.k * a.k;
        const int ky = kr / a.k, kx = kr - ky * a.k;
        const int ox = (int)(row % a.Wo);
        const int64_t t = row / a.Wo;
        const int oy = (int)(t % a.Ho), b = (int)(t / a.Ho);
        const int64_t si = b * a.sb + c * a.sc + (int64_t)(oy * a.stride + ky) * a.sy +
                           (int64_t)(ox * a.stride + kx) * a.sx;
        const float v = a.src_u8 ? (float)s8[si] : sf[si];
        a.col[i] = a.div == 1.0f && a.shift == 0.0f ? v : v / a.div + a.shift;
    }
}

// The same column matrix, one thread per (output pixel, channel, ky): it copies the k consecutive kx taps -- consecutive
// source pixels, consecutive column entries -- with ONE index decode (32-bit) instead of one 64-bit div/mod chain per
// element; neighbouring threads write neighbouring k-float segments of a row.  Same values, bit for bit.
__global__ __launch_bounds__(256) void im2col_runs_kernel(Im2colArgs a, int n_items) {
    const int ck = a.C * a.k, ckk = ck * a.k;
    const uint8_t *s8 = (const uint8_t *)a.src;
    const float *sf = (const float *)a.src;
    const bool plain = a.div == 1.0f && a.shift == 0.0f;
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < n_items; it += gridDim.x * blockDim.x) {
        const int row = it / ck, r = it - row * ck;
        const int c = r / a.k, ky = r - c * a.k;
        const int ox = row % a.Wo, t = row / a.Wo;
        const int oy = t % a.Ho, b = t / a.Ho;
        const int64_t si = b * a.sb + c * a.sc + (int64_t)(oy * a.stride + ky) * a.sy + (int64_t)(ox * a.stride) * a.sx;
        float *dst = a.col + (int64_t)row * ckk + r * a.k;
        for (int kx = 0; kx < a.k; ++kx) {
            const float v = a.src_u8 ? (float)s8[si + kx * a.sx] : sf[si + kx * a.sx];
            dst[kx] = plain ? v : v / a.div + a.shift;
        }
    }
}

struct Col2imArgs {
    const float *dcol;        // (B*Ho*Wo, C*k*k)
    float *dx; int64_t sb, sc, sy, sx;   // destination strides
    const float *mask; int64_t mb, mc, my, mx;  // optional ReLU mask source (same logical shape as dx)
    int B, C, Hi, Wi, k, stride, Ho, Wo;
};

__global__ void col2im_kernel(Col2imArgs a) {
    const int64_t total = (int64_t)a.B * a.C * a.Hi * a.Wi;
    const int64_t ckk = (int64_t)a.C * a.k * a.k;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        // decode with channel fastest (destination is channels-last in every use)
        const int c = (int)(i % a.C);
        int64_t t = i / a.C;
        const int ix = (int)(t % a.Wi); t /= a.Wi;
        const int iy = (int)(t % a.Hi);
        const int b = (int)(t / a.Hi);
        float acc = 0.0f;
        for (int ky = 0; ky < a.k; ++ky) {
            const int ny = iy - ky;
            if (ny < 0 || ny % a.stride) continue;
            const int oy = ny / a.stride;
            if (oy >= a.Ho) continue;
            for (int kx = 0; kx < a.k; ++kx) {
                const int nx = ix - kx;
                if (nx < 0 || nx % a.stride) continue;
                const int ox = nx / a.stride;
                if (ox >= a.Wo) continue;
                acc += a.dcol[((int64_t)(b * a.Ho + oy) * a.Wo + ox) * ckk + (c * a.k + ky) * a.k + kx];
            }
        }
        if (a.mask) {
            const float m = a.mask[b * a.mb + c * a.mc + (int64_t)iy * a.my + (int64_t)ix * a.mx];
            acc = m > 0.0f ? acc : 0.0f;
        }
        a.dx[b * a.sb + c * a.sc + (int64_t)iy * a.sy + (int64_t)ix * a.sx] = acc;
    }
}

// The same adjoint for a column matrix whose columns run (ky, kx, c) -- the product of dY with the weight in
// channels-last column order (ssac_permute_cp) -- into a contiguous channels-last dx (B, Hi, Wi, C), C % 4 == 0:
// a thread owns 4 channels of one input pixel and every tap it adds is a 16-byte read next to its neighbours'
// (the (c, ky, kx) order above puts the channels of one tap k*k floats apart: 64-byte strides at k = 4).
// Same taps in the same order (ky, then kx), so the sums are the ones col2im_kernel forms.
__global__ __launch_bounds__(256) void col2im_cl_kernel(const float *__restrict__ dcol, float *__restrict__ dx,
                                                        const float *__restrict__ mask, int B, int C, int Hi, int Wi,
                                                        int k, int stride, int Ho, int Wo) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int c4n = C >> 2;
    const int64_t total = (int64_t)B * Hi * Wi * c4n;
    const int64_t ckk = (int64_t)C * k * k;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        int64_t t = i / c4n;
        const int ix = (int)(t % Wi); t /= Wi;
        const int iy = (int)(t % Hi);
        const int b = (int)(t / Hi);
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int ky = iy % stride; ky < k && ky <= iy; ky += stride) {
            const int oy = (iy - ky) / stride;
            if (oy >= Ho) continue;
            for (int kx = ix % stride; kx < k && kx <= ix; kx += stride) {
                const int ox = (ix - kx) / stride;
                if (ox >= Wo) continue;
                acc += *reinterpret_cast<const f4 *>(dcol + ((int64_t)(b * Ho + oy) * Wo + ox) * ckk +
                                                      (ky * k + kx) * C + 4 * c4);
            }
        }
        if (mask) {
            const f4 m = *reinterpret_cast<const f4 *>(mask + 4 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = m[j] > 0.0f ? acc[j] : 0.0f;
        }
        *reinterpret_cast<f4 *>(dx + 4 * i) = acc;
    }
}

// dY (.)= [Y > 0]  (ReLU backward on a contiguous buffer)
__global__ void relu_mask_kernel(float *__restrict__ dy, const float *__restrict__ y, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dy[i] = y[i] > 0.0f ? dy[i] : 0.0f;
}

// out[i] = sum_s partial[s*n + i], fixed order
// sum of partial[z * n + i] over z = z0, z0 + dz, ... < slices, added in that order; the loads of 8 terms are requested
// together (the additions are a dependent chain, the loads are not: one round trip per 8 terms instead of per term)
__device__ __forceinline__ float sum_slices(const float *__restrict__ partial, int64_t n, int64_t i, int z0, int dz, int slices) {
    float s = 0.0f;
    for (int z = z0; z < slices; z += 8 * dz) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)min(z + u * dz, slices - 1) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (z + u * dz < slices) s += v[u];
    }
    return s;
}

__global__ void reduce_slices_kernel(const float *__restrict__ partial, int slices, int64_t n,
                                     float *__restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        out[i] = sum_slices(partial, n, i, 0, 1, slices);
    }
}

// the same sum for MANY slices (implicit-conv weight gradients: ~700 slices): 16 waves per workgroup, wave w
// adds the slices z = w, w+16, ... of its 64 outputs, then the 16 partial sums are added in a fixed order --
// a fixed reduction tree, so the result is still run-to-run deterministic.
__global__ __launch_bounds__(1024) void reduce_slices_wide_kernel(const float *__restrict__ partial, int slices,
                                                                  int64_t n, float *__restrict__ out) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    red[w][lane] = i < n ? sum_slices(partial, n, i, w, 16, slices) : 0.0f;
    __syncthreads();
    if (w == 0 && i < n) {
        float t = red[0][lane];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][lane];
        out[i] = t;
    }
}

// two reductions in one launch (a convolution layer's weight AND bias slices): workgroups [0, ceil(n0 / 64)) take the first
// problem, the rest the second; per output the sums of reduce_slices_kernel / reduce_slices_wide_kernel, in their order
__global__ __launch_bounds__(1024) void reduce_slices_pair_kernel(const float *__restrict__ p0, int64_t n0, float *__restrict__ o0,
                                                                  const float *__restrict__ p1, int64_t n1, float *__restrict__ o1,
                                                                  int slices, int wide) {
    __shared__ float red[16][64];
    const int g0 = (int)((n0 + 63) / 64);
    const bool second = (int)blockIdx.x >= g0;
    const float *partial = second ? p1 : p0;
    float *out = second ? o1 : o0;
    const int64_t n = second ? n1 : n0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = (int64_t)(second ? blockIdx.x - g0 : blockIdx.x) * 64 + lane;
    if (!wide) {   // few slices: one serial sum per output (wave 0)
        if (w == 0 && i < n) out[i] = sum_slices(partial, n, i, 0, 1, slices);
        return;
    }
    red[w][lane] = i < n ? sum_slices(partial, n, i, w, 16, slices) : 0.0f;
    __syncthreads();
    if (w == 0 && i < n) {
        float t = red[0][lane];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][lane];
        out[i] = t;
    }
}

// out[m*ldo + n] = bias[n] + sum_s partial[(s*M + m)*N + n], fixed order (slice 0 first).  The additions of an element
// are a dependent chain, its loads are not: 16 slices are requested at once (one round trip per 16 instead of per slice).
__global__ void reduce_slices_bias_kernel(const float *__restrict__ partial, int slices, int M, int N,
                                          const float *__restrict__ bias, float *__restrict__ out, int64_t ldo) {
    const int64_t n_el = (int64_t)M * N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_el;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
        float s = 0.0f;
        for (int z0 = 0; z0 < slices; z0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = partial[(int64_t)min(z0 + u, slices - 1) * n_el + i];
#pragma unroll
            for (int u = 0; u < 16; ++u) if (z0 + u < slices) s += v[u];
        }
        out[m * ldo + n] = s + (bias ? bias[n] : 0.0f);
    }
}

// dst[n][p][c] = src[n][c][p]  (to_cl) or dst[n][c][p] = src[n][p][c]  (!to_cl): the fc weight (emb x C x P, the
// NCHW flatten order of cnns.py:63,98) <-> the channels-last order the feature maps are stored in.  A workgroup moves a
// 32 x 32 tile of one matrix through LDS: both the reads and the writes are 128-byte rows (the element-wise form read
// with a stride of a whole row: 32 cache lines per wave load).
__global__ __launch_bounds__(256) void permute_cp_kernel(const float *__restrict__ src, float *__restrict__ dst, int N, int Cc,
                                                         int P, int to_cl) {
    __shared__ float tile[32][33];
    // rows x cols of the SOURCE matrix of one n: (Cc x P) when to_cl, else (P x Cc)
    const int R = to_cl ? Cc : P, Q = to_cl ? P : Cc;
    const int tr = (R + 31) >> 5, tq = (Q + 31) >> 5, per_n = tr * tq;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int64_t t = blockIdx.x; t < (int64_t)N * per_n; t += gridDim.x) {
        const int n = (int)(t / per_n), rem = (int)(t - (int64_t)n * per_n), r0 = (rem / tq) << 5, q0 = (rem % tq) << 5;
        const float *sn = src + (int64_t)n * R * Q;
        float *dn = dst + (int64_t)n * R * Q;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + ty + 8 * j, q = q0 + tx;
            if (r < R && q < Q) tile[ty + 8 * j][tx] = sn[(int64_t)r * Q + q];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + ty + 8 * j, r = r0 + tx;
            if (r < R && q < Q) dn[(int64_t)q * R + r] = tile[tx][ty + 8 * j];
        }
        __syncthreads();
    }
}

// out = dy * [y > 0] (out may alias dy)
__global__ void relu_mask_to_kernel(const float *__restrict__ dy, const float *__restrict__ y, int64_t n,
                                    float *__restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = y[i] > 0.0f ? dy[i] : 0.0f;
}

// per-block partial sums of x^2 (for clip_grad_norm_ over an encoder's gradient arena)
__global__ void sumsq_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ out) {
    __shared__ float red[4];
    float s = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // (a thread's terms are added in index order; 8 of them are requested together: one round trip per 8 instead of per term)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += 8 * stride) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i + u * stride < n ? x[i + u * stride] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i + u * stride < n) s += v[u] * v[u];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// LayerNorm (eps 1e-5, biased variance) + tanh, one thread per row (cnns.py:66-68)
__global__ void ln_tanh_fwd_kernel(const float *__restrict__ x, int64_t ldx, const float *__restrict__ gamma,
                                   const float *__restrict__ beta, int n_rows, int D, float *__restrict__ out,
                                   int64_t ldo, float *__restrict__ xhat, float *__restrict__ rstd) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_rows) return;
    const float *xr = x + b * ldx;
    float mean = 0.0f;
    for (int j = 0; j < D; ++j) mean += xr[j];
    mean /= (float)D;
    float var = 0.0f;
    for (int j = 0; j < D; ++j) { const float d = xr[j] - mean; var += d * d; }
    var /= (float)D;
    const float rs = 1.0f / sqrtf(var + 1e-5f);
    if (rstd) rstd[b] = rs;
    for (int j = 0; j < D; ++j) {
        const float xh = (xr[j] - mean) * rs;
        if (xhat) xhat[(int64_t)b * D + j] = xh;
        out[b * ldo + j] = tanhf(xh * gamma[j] + beta[j]);
    }
}

// The same arithmetic (a row's two sums run serially over its features, in index order -> same bits) with everything
// else taken off the serial chain: LN_ROWS rows per workgroup are staged in LDS with coalesced loads (row stride D + 1:
// conflict-free row walks), 64 threads form the rows' mean / rstd, then all 256 threads normalise, apply tanh -- the
// expensive part: one thread per row spent 50 serial tanhf per row -- and store, one element each at a time.
constexpr int LN_ROWS = 16;   // rows per workgroup: the per-row sums are serial, everything around them scales with the rows (B 512: 32 workgroups)
__global__ __launch_bounds__(256) void ln_tanh_fwd_lds_kernel(const float *__restrict__ x, int64_t ldx,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, int n_rows, int D,
                                                              float *__restrict__ out, int64_t ldo,
                                                              float *__restrict__ xhat, float *__restrict__ rstd) {
    extern __shared__ float lds[];   // xs[LN_ROWS][D + 1] | mean[LN_ROWS] | rs[LN_ROWS]
    const int ld = D + 1, tid = threadIdx.x, r0 = blockIdx.x * LN_ROWS;
    const int rows = min(LN_ROWS, n_rows - r0);
    float *xs = lds, *mean_s = lds + LN_ROWS * ld, *rs_s = mean_s + LN_ROWS;
    for (int i = tid; i < rows * D; i += 256) {
        const int r = i / D, j = i - r * D;
        xs[r * ld + j] = x[(int64_t)(r0 + r) * ldx + j];
    }
    __syncthreads();
    if (tid < rows) {
        const float *xr = xs + tid * ld;
        float mean = 0.0f;
        for (int j = 0; j < D; ++j) mean += xr[j];
        mean /= (float)D;
        float var = 0.0f;
        for (int j = 0; j < D; ++j) { const float d = xr[j] - mean; var += d * d; }
        var /= (float)D;
        const float rs = 1.0f / sqrtf(var + 1e-5f);
        if (rstd) rstd[r0 + tid] = rs;
        mean_s[tid] = mean;
        rs_s[tid] = rs;
    }
    __syncthreads();
    for (int i = tid; i < rows * D; i += 256) {
        const int r = i / D, j = i - r * D;
        const float xh = (xs[r * ld + j] - mean_s[r]) * rs_s[r];
        if (xhat) xhat[(int64_t)(r0 + r) * D + j] = xh;
        out[(int64_t)(r0 + r) * ldo + j] = tanhf(xh * gamma[j] + beta[j]);
    }
}

// backward: d_out (n_rows x D) wrt tanh output -> dx (pre-LayerNorm), dgamma, dbeta.  Two launches:
//   rows:    one WAVE per row (lanes over the features, coalesced), the row's two means by a shuffle tree;
//   columns: one workgroup per feature, 256 threads over the rows, fixed-order tree -> dgamma[j], dbeta[j].
// (Round 1 ran this as ONE workgroup with a thread per row and serial feature loops: 170 us at B 512 / D 50.)
__global__ __launch_bounds__(256) void ln_tanh_bwd_rows_kernel(
    const float *__restrict__ d_out, int64_t ldd, const float *__restrict__ out, int64_t ldo,
    const float *__restrict__ xhat, const float *__restrict__ rstd, const float *__restrict__ gamma,
    int n_rows, int D, float *__restrict__ dx, int64_t ldx, float *__restrict__ dy_scratch) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_rows) return;
    float m1 = 0.0f, m2 = 0.0f;
    for (int j = lane; j < D; j += 64) {
        const float o = out[(int64_t)b * ldo + j];
        const float dy = d_out[(int64_t)b * ldd + j] * (1.0f - o * o);
        dy_scratch[(int64_t)b * D + j] = dy;
        const float dxh = dy * gamma[j];
        m1 += dxh;
        m2 += dxh * xhat[(int64_t)b * D + j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o, 64); m2 += __shfl_xor(m2, o, 64); }
    m1 /= (float)D;
    m2 /= (float)D;
    const float rs = rstd[b];
    for (int j = lane; j < D; j += 64) {
        const float o = out[(int64_t)b * ldo + j];
        const float dxh = d_out[(int64_t)b * ldd + j] * (1.0f - o * o) * gamma[j];
        dx[(int64_t)b * ldx + j] = rs * (dxh - m1 - xhat[(int64_t)b * D + j] * m2);
    }
}

__global__ __launch_bounds__(256) void ln_tanh_bwd_cols_kernel(const float *__restrict__ dy_scratch,
                                                               const float *__restrict__ xhat, int n_rows, int D,
                                                               float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ float red[2][4];
    const int j = blockIdx.x, tid = threadIdx.x;
    float g = 0.0f, bb = 0.0f;
    for (int b = tid; b < n_rows; b += 256) {
        const float dy = dy_scratch[(int64_t)b * D + j];
        g += dy * xhat[(int64_t)b * D + j];
        bb += dy;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { g += __shfl_xor(g, o, 64); bb += __shfl_xor(bb, o, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = g; red[1][tid >> 6] = bb; }
    __syncthreads();
    if (tid == 0) {
        dgamma[j] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        dbeta[j] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

inline int grid_for(int64_t n, int block = 256, int cap = 16384) {
    int64_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int ssac_im2col(const void *src, int src_u8, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                           int B, int C, int Hi, int Wi, int k, int stride, float div, float shift,
                           float *col, void *stream) {
    if (k < 1 || stride < 1 || Hi < k || Wi < k) return ssac_fail("ssac_im2col: bad geometry");
    Im2colArgs a{src, src_u8, sb, sc, sy, sx, B, C, Hi, Wi, k, stride, (Hi - k) / stride + 1,
                 (Wi - k) / stride + 1, div, shift, col};
    const int64_t total = (int64_t)B * a.Ho * a.Wo * C * k * k;
    if (total <= 0) return 0;
    const int64_t items = total / k;   // (pixel, channel, ky) runs of k taps
    if (items < (1ll << 31) - 65536) {
        SSAC_LAUNCH(im2col_runs_kernel, dim3(grid_for(items, 256, 65536)), dim3(256), 0, ST, a, (int)items);
        return ssac_check_launch("im2col");
    }
    SSAC_LAUNCH(im2col_kernel, dim3(grid_for(total)), dim3(256), 0, ST, a);
    return ssac_check_launch("im2col");
}

extern "C" int ssac_col2im(const float *dcol, float *dx, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                           const float *mask, int64_t mb, int64_t mc, int64_t my, int64_t mx, int B, int C,
                           int Hi, int Wi, int k, int stride, void *stream) {
    if (k < 1 || stride < 1 || Hi < k || Wi < k) return ssac_fail("ssac_col2im: bad geometry");
    Col2imArgs a{dcol, dx, sb, sc, sy, sx, mask, mb, mc, my, mx, B, C, Hi, Wi, k, stride,
                 (Hi - k) / stride + 1, (Wi - k) / stride + 1};
    const int64_t total = (int64_t)B * C * Hi * Wi;
    if (total <= 0) return 0;
    SSAC_LAUNCH(col2im_kernel, dim3(grid_for(total)), dim3(256), 0, ST, a);
    return ssac_check_launch("col2im");
}

extern "C" int ssac_col2im_cl(const float *dcol, float *dx, const float *mask, int B, int C, int Hi, int Wi, int k,
                              int stride, void *stream) {
    if (k < 1 || stride < 1 || Hi < k || Wi < k || (C & 3)) return ssac_fail("ssac_col2im_cl: bad geometry (C % 4 == 0)");
    const int64_t total = (int64_t)B * Hi * Wi * (C / 4);
    if (total <= 0) return 0;
    SSAC_LAUNCH(col2im_cl_kernel, dim3(grid_for(total)), dim3(256), 0, ST, dcol, dx, mask, B, C, Hi, Wi, k, stride,
                (Hi - k) / stride + 1, (Wi - k) / stride + 1);
    return ssac_check_launch("col2im_cl");
}

extern "C" int ssac_relu_mask(float *dy, const float *y, int64_t n, void *stream) {
    if (n <= 0) return 0;
    SSAC_LAUNCH(relu_mask_kernel, dim3(grid_for(n)), dim3(256), 0, ST, dy, y, n);
    return ssac_check_launch("relu_mask");
}

extern "C" int ssac_reduce_slices(const float *partial, int slices, int64_t n, float *out, void *stream) {
    if (n <= 0 || slices <= 0) return 0;
    if (slices >= 64) {
        SSAC_LAUNCH(reduce_slices_wide_kernel, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, ST, partial, slices, n,
                    out);
        return ssac_check_launch("reduce_slices");
    }
    SSAC_LAUNCH(reduce_slices_kernel, dim3(grid_for(n)), dim3(256), 0, ST, partial, slices, n, out);
    return ssac_check_launch("reduce_slices");
}

extern "C" int ssac_reduce_slices_pair(const float *partial0, int64_t n0, float *out0, const float *partial1, int64_t n1,
                                       float *out1, int slices, void *stream) {
    if (slices <= 0 || n0 <= 0 || n1 <= 0) return ssac_fail("ssac_reduce_slices_pair: bad sizes");
    const unsigned grid = (unsigned)((n0 + 63) / 64 + (n1 + 63) / 64);
    SSAC_LAUNCH(reduce_slices_pair_kernel, dim3(grid), dim3(1024), 0, ST, partial0, n0, out0, partial1, n1, out1, slices,
                slices >= 64 ? 1 : 0);
    return ssac_check_launch("reduce_slices_pair");
}

extern "C" int ssac_reduce_slices_bias(const float *partial, int slices, int M, int N, const float *bias,
                                       float *out, int64_t ld_out, void *stream) {
    if (M <= 0 || N <= 0 || slices <= 0) return 0;
    SSAC_LAUNCH(reduce_slices_bias_kernel, dim3(grid_for((int64_t)M * N)), dim3(256), 0, ST, partial, slices, M, N,
                bias, out, ld_out);
    return ssac_check_launch("reduce_slices_bias");
}

extern "C" int ssac_permute_cp(const float *src, float *dst, int n, int channels, int pixels, int to_channels_last,
                               void *stream) {
    const int64_t total = (int64_t)n * channels * pixels;
    if (total <= 0) return 0;
    const int64_t tiles = (int64_t)n * ((channels + 31) / 32) * ((pixels + 31) / 32);
    SSAC_LAUNCH(permute_cp_kernel, dim3((unsigned)(tiles < 4096 ? tiles : 4096)), dim3(256), 0, ST, src, dst, n, channels,
                pixels, to_channels_last);
    return ssac_check_launch("permute_cp");
}

extern "C" int ssac_relu_mask_to(const float *dy, const float *y, int64_t n, float *out, void *stream) {
    if (n <= 0) return 0;
    SSAC_LAUNCH(relu_mask_to_kernel, dim3(grid_for(n)), dim3(256), 0, ST, dy, y, n, out);
    return ssac_check_launch("relu_mask_to");
}

extern "C" int ssac_sumsq_blocks(void) { return 256; }

extern "C" int ssac_sumsq(const float *x, int64_t n, float *out_partials, void *stream) {
    SSAC_LAUNCH(sumsq_kernel, dim3(256), dim3(256), 0, ST, x, n, out_partials);
    return ssac_check_launch("sumsq");
}

extern "C" int ssac_ln_tanh_fwd(const float *x, int64_t ldx, const float *gamma, const float *beta,
                                int n_rows, int dim, float *out, int64_t ldo, float *xhat, float *rstd,
                                void *stream) {
    if (n_rows <= 0) return 0;
    const size_t lds = sizeof(float) * (LN_ROWS * (size_t)(dim + 1) + 2 * LN_ROWS);
    if (lds <= 64 * 1024)
        SSAC_LAUNCH(ln_tanh_fwd_lds_kernel, dim3((n_rows + LN_ROWS - 1) / LN_ROWS), dim3(256), lds, ST, x, ldx, gamma, beta, n_rows, dim,
                    out, ldo, xhat, rstd);
    else
        SSAC_LAUNCH(ln_tanh_fwd_kernel, dim3((n_rows + 63) / 64), dim3(64), 0, ST, x, ldx, gamma, beta,
                    n_rows, dim, out, ldo, xhat, rstd);
    return ssac_check_launch("ln_tanh_fwd");
}

extern "C" int ssac_ln_tanh_bwd(const float *d_out, int64_t ldd, const float *out, int64_t ldo,
                                const float *xhat, const float *rstd, const float *gamma, int n_rows,
                                int dim, float *dx, int64_t ldx, float *dy_scratch, float *dgamma,
                                float *dbeta, void *stream) {
    if (n_rows <= 0) return 0;
    SSAC_LAUNCH(ln_tanh_bwd_rows_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, ST, d_out, ldd, out, ldo, xhat, rstd, gamma,
                n_rows, dim, dx, ldx, dy_scratch);
    SSAC_LAUNCH(ln_tanh_bwd_cols_kernel, dim3(dim), dim3(256), 0, ST, (const float *)dy_scratch, xhat, n_rows, dim, dgamma,
                dbeta);
    return ssac_check_launch("ln_tanh_bwd");
}
