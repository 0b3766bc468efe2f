// Implicit-GEMM convolutions for the pixel encoders' inner layers (nets/cnns.py:41-44, 76-78) on gfx950.
//
// The im2col + GEMM path (ssac_conv.hip) writes and re-reads a (B*Ho*Wo) x (ci*k*k) column matrix per layer --
// 807 MB for DrQ's 32->32 3x3 layers at batch 512, nine times the feature map -- and is bound by that HBM
// traffic.  Here the patch gather happens in the operand loads: activations are channels-last
// (B, H, W, C) fp32, so the 32 channels of one input pixel are 128 contiguous bytes, and a K chunk of the GEMM is
// "one (ky, kx) tap x one block of 32 input channels".  Every lane of a wave loads its own pixel's 16 floats per
// chunk straight from global memory (the map is read ~k*k times, from L2), weights for the workgroup's 32 output
// channels sit in LDS for the whole launch (persistent workgroups), and the product runs on
// v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered).  Requirements: ci % 32 == 0, co % 32 == 0 (all layers
// except the first, which reads the uint8 NCHW image and stays on im2col).
//
//   forward        y[b,oy,ox,co]  = relu(bias[co] + sum_{ky,kx,c} x[b, oy*s+ky, ox*s+kx, c] * W[co][c][ky][kx])
//   backward-data  dx[b,iy,ix,c]  = [x[b,iy,ix,c] > 0] * sum_{ky,kx,co} dy[b,(iy-ky)/s,(ix-kx)/s,co] * W[co][c][ky][kx]
//                  (x is the ReLU output of the previous layer, so the mask is the ReLU derivative)
//   weight grad    dW[co][c][ky][kx] = sum_{b,oy,ox} dy[b,oy,ox,co] * x[b, oy*s+ky, ox*s+kx, c]   (split over
//                  workgroups into partial slices, reduced in a fixed order by ssac_reduce_slices)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssac_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int CV_THREADS = 256;   // 4 waves, each a 32-pixel x 32-channel output tile
constexpr int CV_PIX = 128;       // pixels per workgroup tile
constexpr int WL_LD = 32;         // LDS row stride of a weight row: 32 floats, the 16-byte columns XOR-swizzled by the row

struct ConvArgs {
    const float *x;      // (B, Hi, Wi, ci) channels-last          [fwd: input; dgrad: the layer input (mask)]
    const float *w;      // (co, ci, k, k) nn.Conv2d layout
    const float *bias;   // (co)
    const float *dy;     // (B, Ho, Wo, co)                         [dgrad / wgrad]
    float *out;          // fwd: y (B,Ho,Wo,co); dgrad: dx (B,Hi,Wi,ci)
    int B, Hi, Wi, ci, Ho, Wo, co, k, s;
};

// element (row, col) of a 32-float weight row in LDS: the 16-byte column index is XORed with (row >> 1) & 7.  A
// ds_read_b128 is served in groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) over 64 banks = 256 bytes;
// rows are 128 bytes, so a group's 8 even rows (and its 8 odd rows) must land on 8 different 16-byte columns: their
// row >> 1 values are distinct mod 8 in every group, hence conflict-free without a padding column
__device__ __forceinline__ int wl_off(int row, int col) {
    return row * WL_LD + ((((col >> 2) ^ ((row >> 1) & 7)) << 2) | (col & 3));
}

// chunk index -> (ky, kx, channel block)
__device__ __forceinline__ void chunk_decode(int ch, int k, int cblocks, int &ky, int &kx, int &cb) {
    cb = ch % cblocks;
    const int t = ch / cblocks;
    kx = t % k;
    ky = t / k;
}

// ---------------------------------------------------------------------------------------------
// forward: grid (persistent pixel tiles, co / 32)
// ---------------------------------------------------------------------------------------------
// chunk -> offset iterator in chunk order (cb fastest, then kx, then ky), uniform integer steps instead of a
// division chain per chunk; next() returns the current chunk's (ky, kx, cb) and moves on, staying on the last chunk
// (the prefetches past the end re-read it)
struct ChunkIter {
    int k, cblocks, left;
    int ky = 0, kx = 0, cb = 0;
    __device__ ChunkIter(int k_, int cblocks_, int nch) : k(k_), cblocks(cblocks_), left(nch - 1) {}
    __device__ __forceinline__ void next(int &oky, int &okx, int &ocb) {
        oky = ky; okx = kx; ocb = cb;
        if (left > 0) {
            --left;
            if (++cb == cblocks) { cb = 0; if (++kx == k) { kx = 0; ++ky; } }
        }
    }
};

__global__ __launch_bounds__(CV_THREADS) void conv_fwd_kernel(ConvArgs g, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [chunks][32 co][WL_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int cblocks = g.ci >> 5, nch = g.k * g.k * cblocks, kk = g.k * g.k;
    const int co0 = blockIdx.y * 32;
    // weights of this workgroup's 32 output channels -> LDS, chunk-major, k (= channel within block) contiguous
    for (int ch = 0; ch < nch; ++ch) {
        int ky, kx, cb;
        chunk_decode(ch, g.k, cblocks, ky, kx, cb);
        const float *src = g.w + ((int64_t)co0 * g.ci + cb * 32) * kk + ky * g.k + kx;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * CV_THREADS, c = i & 31, co = i >> 5;
            wl[wl_off(ch * 32 + co, c)] = src[((int64_t)co * g.ci + c) * kk];
        }
    }
    __syncthreads();
    const float bias = g.bias[co0 + li];
    const int64_t M = (int64_t)g.B * g.Ho * g.Wo;
    // this lane's input pixel of a tile (lanes beyond the last pixel read pixel 0: valid memory, results dropped)
    auto tile_base = [&](int tile) {
        const int64_t m = (int64_t)tile * CV_PIX + wave * 32 + li;
        const int64_t mm = m < M ? m : 0;
        const int ox = (int)(mm % g.Wo);
        const int64_t t = mm / g.Wo;
        const int oy = (int)(t % g.Ho), b = (int)(t / g.Ho);
        return g.x + (((int64_t)b * g.Hi + oy * g.s) * g.Wi + ox * g.s) * g.ci + lh * 16;
    };
    auto compute = [&](f32x16 &acc, const f4 (&a)[4], int ch) {
        if (ch < nch) {
            f4 bf[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                bf[q] = *reinterpret_cast<const f4 *>(wl + wl_off(ch * 32 + li, lh * 16 + 4 * q));
#pragma unroll
            for (int tt = 0; tt < 16; ++tt)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tt >> 2][tt & 3], bf[tt >> 2][tt & 3], acc, 0, 0, 0);
        }
    };
    auto store = [&](const f32x16 &acc, int tile) {
        // C layout: column = lane & 31 (output channel), rows (r & 3) + 8 (r >> 2) + 4 lh (pixel within the wave)
        const int64_t m_wave = (int64_t)tile * CV_PIX + wave * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t mr = m_wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (mr < M) g.out[mr * g.co + co0 + li] = fmaxf(acc[r] + bias, 0.0f);
        }
    };
    // ONE operand stream over all of this workgroup's tiles: the loads run two chunks ahead of the MFMAs ACROSS tile
    // boundaries -- with a per-tile pipeline every tile began with an exposed round trip for its first chunk and ended with
    // two loads nobody used.  The three operand buffers rotate over the stream; a tile ends wherever its last chunk falls.
    int ltile = blockIdx.x;
    if (ltile >= n_tiles) return;
    const float *lbase = tile_base(ltile);
    ChunkIter lit(g.k, cblocks, nch);
    int lleft = nch;
    auto load = [&](f4 (&a)[4]) {
        int ky, kx, cb;
        lit.next(ky, kx, cb);
        const float *p = lbase + (ky * g.Wi + kx) * g.ci + cb * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f4 *>(p + 4 * q);
        if (--lleft == 0) {   // that was the tile's last chunk: the stream moves to the workgroup's next tile
            ltile += gridDim.x;
            lbase = tile_base(ltile < n_tiles ? ltile : blockIdx.x);   // (past the end: redundant loads of a valid tile)
            lit = ChunkIter(g.k, cblocks, nch);
            lleft = nch;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    int ctile = blockIdx.x, cch = 0;   // the tile / chunk the MFMAs are at
    auto step = [&](const f4 (&a)[4]) {
        if (ctile >= n_tiles) return;
        compute(acc, a, cch);
        if (++cch == nch) {
            store(acc, ctile);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            cch = 0;
            ctile += gridDim.x;
        }
    };
    f4 a0[4], a1[4], a2[4];
    load(a0);
    load(a1);
    while (ctile < n_tiles) {
        load(a2); step(a0);
        load(a0); step(a1);
        load(a1); step(a2);
    }
}

// ---------------------------------------------------------------------------------------------
// backward-data: grid (persistent input-pixel tiles, ci / 32)
// ---------------------------------------------------------------------------------------------
template <bool S1>  // S1: stride 1 (every layer the engine sends here) -- no per-tap divisions
__global__ __launch_bounds__(CV_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3)))   // (three persistent workgroups per CU: <= 168 VGPRs)
void conv_dgrad_kernel(ConvArgs g, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [chunks][32 c][WL_LD] : W[co][c0+c][ky][kx], co contiguous
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int oblocks = g.co >> 5, nch = g.k * g.k * oblocks, kk = g.k * g.k;
    const int c0 = blockIdx.y * 32;
    for (int ch = 0; ch < nch; ++ch) {
        int ky, kx, ob;
        chunk_decode(ch, g.k, oblocks, ky, kx, ob);
        const float *src = g.w + ((int64_t)(ob * 32) * g.ci + c0) * kk + ky * g.k + kx;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * CV_THREADS, o = i & 31, c = i >> 5;
            wl[wl_off(ch * 32 + c, o)] = src[((int64_t)o * g.ci + c) * kk];
        }
    }
    __syncthreads();
    const int64_t M = (int64_t)g.B * g.Hi * g.Wi;
    // this lane's input pixel of a tile
    struct Pix { int iy, ix; bool ok; const float *dyb; };
    auto tile_pix = [&](int tile) {
        const int64_t m = (int64_t)tile * CV_PIX + wave * 32 + li;
        Pix p;
        p.ok = m < M;
        const int64_t mm = p.ok ? m : 0;
        p.ix = (int)(mm % g.Wi);
        const int64_t t = mm / g.Wi;
        p.iy = (int)(t % g.Hi);
        p.dyb = g.dy + (int64_t)(t / g.Hi) * g.Ho * g.Wo * g.co + lh * 16;
        return p;
    };
    // chunk (ky, kx, ob) reads dy[b, (iy-ky)/s, (ix-kx)/s, ob*32 + 16 lh ..]: valid taps only, else zeros
    auto load_at = [&](f4 (&a)[4], const Pix &px, int ky, int kx, int ob) {
        const int ny = px.iy - ky, nx = px.ix - kx;
        int oy, ox;
        bool v;
        if (S1) {
            oy = ny; ox = nx;
            v = px.ok && (unsigned)ny < (unsigned)g.Ho && (unsigned)nx < (unsigned)g.Wo;
        } else {
            oy = ny / g.s; ox = nx / g.s;
            v = px.ok && ny >= 0 && nx >= 0 && oy * g.s == ny && ox * g.s == nx && oy < g.Ho && ox < g.Wo;
        }
        const float *p = px.dyb + (v ? (oy * g.Wo + ox) * g.co : 0) + ob * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f4 x = *reinterpret_cast<const f4 *>(p + 4 * q);
            a[q] = v ? x : (f4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto compute = [&](f32x16 &acc, const f4 (&a)[4], int ch) {
        if (ch < nch) {
            f4 bf[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                bf[q] = *reinterpret_cast<const f4 *>(wl + wl_off(ch * 32 + li, lh * 16 + 4 * q));
#pragma unroll
            for (int tt = 0; tt < 16; ++tt)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tt >> 2][tt & 3], bf[tt >> 2][tt & 3], acc, 0, 0, 0);
        }
    };
    // the ReLU mask of a tile's outputs (the layer's input activation): 16 coalesced loads per lane, requested at the START
    // of the tile so that they come back under its K loop (read in the epilogue they were an exposed round trip per tile)
    auto mask_load = [&](float (&xm)[16], int tile) {
        const int64_t m_wave = (int64_t)tile * CV_PIX + wave * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t mr = m_wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
            xm[r] = g.x[(mr < M ? mr : 0) * g.ci + c0 + li];
        }
    };
    auto store = [&](const f32x16 &acc, const float (&xm)[16], int tile) {
        const int64_t m_wave = (int64_t)tile * CV_PIX + wave * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t mr = m_wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (mr < M) g.out[mr * g.ci + c0 + li] = xm[r] > 0.0f ? acc[r] : 0.0f;
        }
    };
    // one operand stream over all of this workgroup's tiles (conv_fwd_kernel): the loads run two chunks ahead of the MFMAs
    // across tile boundaries
    int ltile = blockIdx.x;
    if (ltile >= n_tiles) return;
    Pix lpx = tile_pix(ltile);
    ChunkIter lit(g.k, oblocks, nch);
    int lleft = nch;
    auto load = [&](f4 (&a)[4]) {
        int ky, kx, ob;
        lit.next(ky, kx, ob);
        load_at(a, lpx, ky, kx, ob);
        if (--lleft == 0) {
            ltile += gridDim.x;
            lpx = tile_pix(ltile < n_tiles ? ltile : blockIdx.x);   // (past the end: redundant loads of a valid tile)
            lit = ChunkIter(g.k, oblocks, nch);
            lleft = nch;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    int ctile = blockIdx.x, cch = 0;
    float xm[16];
    mask_load(xm, ctile);
    auto step = [&](const f4 (&a)[4]) {
        if (ctile >= n_tiles) return;
        compute(acc, a, cch);
        if (++cch == nch) {
            store(acc, xm, ctile);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            cch = 0;
            ctile += gridDim.x;
            if (ctile < n_tiles) mask_load(xm, ctile);
        }
    };
    f4 a0[4], a1[4], a2[4];
    load(a0);
    load(a1);
    while (ctile < n_tiles) {
        load(a2); step(a0);
        load(a0); step(a1);
        load(a1); step(a2);
    }
}

// ---------------------------------------------------------------------------------------------
// backward-data of a STRIDED layer (s > 1): the input pixels are processed in s*s parity classes (iy % s, ix % s).
// A pixel of class (py, px) receives only the taps ky = py + jy s, kx = px + jx s -- the same ones for every pixel
// of its class, with oy = iy/s - jy, ox = ix/s - jx -- so a tile of one class spends its MFMAs on those taps only
// (k*k / s*s of them; the generic gather above multiplies through all k*k and zeroes the rest).  Taps still go in
// ascending (ky, kx) order: same sums as the generic kernel, bit for bit.
// grid (persistent workgroups, class = blockIdx.x % s*s; ci / 32); class c has first[c+1] - first[c] tiles
// ---------------------------------------------------------------------------------------------
constexpr int DG_MAX_S = 4;
struct StridedTiles { int first[DG_MAX_S * DG_MAX_S + 1]; };

__global__ __launch_bounds__(CV_THREADS) void conv_dgrad_strided_kernel(ConvArgs g, StridedTiles tl, int max_chunks) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [this class's chunks][32 c][WL_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int oblocks = g.co >> 5, kk = g.k * g.k;
    const int c0 = blockIdx.y * 32;
    // a workgroup belongs to ONE class (blockIdx.x % s*s) and stages only that class's taps: k*k / s*s of the weights
    const int n_classes = g.s * g.s, cls = blockIdx.x % n_classes, wg = blockIdx.x / n_classes, n_wg = gridDim.x / n_classes;
    const int py = cls / g.s, px = cls - py * g.s;
    const int njy = max((g.k - py + g.s - 1) / g.s, 0), njx = max((g.k - px + g.s - 1) / g.s, 0);
    const int nch = njy * njx * oblocks;   // chunk (jy, jx, ob), ob fastest
    int *pix_off = reinterpret_cast<int *>(wl + (size_t)max_chunks * 32 * WL_LD);   // [4 waves][32]: output pixel index
    for (int ch = 0; ch < nch; ++ch) {
        const int ob = ch % oblocks, tj = ch / oblocks, jx = tj % njx, jy = tj / njx;
        const int ky = py + jy * g.s, kx = px + jx * g.s;
        const float *src = g.w + ((int64_t)(ob * 32) * g.ci + c0) * kk + ky * g.k + kx;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * CV_THREADS, o = i & 31, c = i >> 5;
            wl[wl_off(ch * 32 + c, o)] = src[((int64_t)o * g.ci + c) * kk];
        }
    }
    __syncthreads();
    const int Hy = (g.Hi - py + g.s - 1) / g.s, Wx = (g.Wi - px + g.s - 1) / g.s;
    const int npix = g.B * Hy * Wx;
    const int n_tiles = tl.first[cls + 1] - tl.first[cls];
    for (int tile = wg; tile < n_tiles; tile += n_wg) {
        const int p = tile * CV_PIX + wave * 32 + li;
        const bool ok = p < npix;
        const int pp = ok ? p : 0;
        const int xq = pp % Wx, t = pp / Wx;
        const int yq = t % Hy, b = t / Hy;
        if (lh == 0) pix_off[wave * 32 + li] = ok ? (b * g.Hi + yq * g.s + py) * g.Wi + xq * g.s + px : -1;
        // (the pixel offsets of this lane's 16 output rows, and with them the ReLU mask values, are fetched NOW: the mask
        // loads come back under the K loop instead of being an exposed round trip in the epilogue of every tile)
        lds_barrier();   // pix_off of this tile (a wave reads only the 32 entries it wrote itself)
        int po[16];
        float xm[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            po[r] = pix_off[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
            xm[r] = g.x[(int64_t)(po[r] >= 0 ? po[r] : 0) * g.ci + c0 + li];
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        const float *dyb = g.dy + (int64_t)b * g.Ho * g.Wo * g.co + lh * 16;
        // chunk order: jy, jx ascending, output-channel block fastest (uniform stepping; stays on the last chunk)
        int ijy = 0, ijx = 0, iob = 0, left = nch - 1;
        auto load = [&](f4 (&a)[4]) {
            const int oy = yq - ijy, ox = xq - ijx;
            const bool v = ok && nch > 0 && (unsigned)oy < (unsigned)g.Ho && (unsigned)ox < (unsigned)g.Wo;
            const float *q = dyb + (v ? (oy * g.Wo + ox) * g.co : 0) + iob * 32;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f4 x = *reinterpret_cast<const f4 *>(q + 4 * u);
                a[u] = v ? x : (f4){0.f, 0.f, 0.f, 0.f};
            }
            if (left > 0) {
                --left;
                if (++iob == oblocks) { iob = 0; if (++ijx == njx) { ijx = 0; ++ijy; } }
            }
        };
        auto compute = [&](const f4 (&a)[4], int ch) {
            if (ch < nch) {
                f4 bf[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    bf[u] = *reinterpret_cast<const f4 *>(wl + wl_off(ch * 32 + li, lh * 16 + 4 * u));
#pragma unroll
                for (int tt = 0; tt < 16; ++tt)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tt >> 2][tt & 3], bf[tt >> 2][tt & 3], acc, 0, 0, 0);
            }
        };
        f4 a0[4], a1[4], a2[4];
        load(a0);
        load(a1);
        for (int ch = 0; ch < nch; ch += 3) {
            load(a2); compute(a0, ch);
            load(a0); compute(a1, ch + 1);
            load(a1); compute(a2, ch + 2);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (po[r] >= 0) g.out[(int64_t)po[r] * g.ci + c0 + li] = xm[r] > 0.0f ? acc[r] : 0.0f;
        lds_barrier();   // before the next tile overwrites pix_off
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient: grid (pixel slices, ci / 32, co / 32); each workgroup owns the (32 co x 32 c x k x k) block of
// dW for its slice of output pixels.  Wave w takes the taps t = w, w+4, ... (one 32x32 accumulator per tap).
//   A = dy^T : lane (co, half) needs dy[pixel 16 half + t][co]      -> 16 scalar loads, 128 B coalesced per pixel
//   B = x    : lane (c,  half) needs x[pixel 16 half + t + tap][c]  -> same
// ---------------------------------------------------------------------------------------------
constexpr int WG_MAX_TAPS = 4;  // taps per wave held in registers (k*k <= 16)

// A workgroup's (32 co x 32 c x k k) block of partial_w[slice][co][c][ky][kx] leaves through LDS: in the accumulators a
// lane holds (co row, c = lane, ONE tap) -- stored directly, consecutive lanes are k k floats apart (a 64-byte stride at
// 4 x 4: every store instruction touched 64 different segments, the epilogue of a 256-slice launch cost ~45 us).  Staged
// as [16 co rows][32 c][k k + 1] (odd row length: conflict-free) a co row's 32 k k floats are one contiguous run in
// global memory; two halves of 16 rows.  tap_of(j) = the tap of accumulator j of this wave (>= k k: none).
template <int NACC, typename TapOf>
__device__ __forceinline__ void wgrad_store_block(float *stage, const f32x16 (&acc)[NACC], TapOf tap_of, int kk, float *pw,
                                                  int ci_total, int c0, int co0) {
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int ldk = kk + 1, row_floats = 32 * kk;
    for (int h = 0; h < 2; ++h) {
        __syncthreads();   // (the staging area is free: the main loop / the previous half is done with it)
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            const int tap = tap_of(j);
            if (tap < kk) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int rr = 8 * h + r;   // accumulator rows of this half: co rows 16 h .. 16 h + 15
                    const int row = (rr & 3) + 8 * ((rr >> 2) & 1) + 4 * lh;
                    stage[(row * 32 + li) * ldk + tap] = acc[j][rr];
                }
            }
        }
        __syncthreads();
        for (int row = 0; row < 16; ++row) {   // (per co row one contiguous run of 32 k k floats; no division by row length)
            float *dst = pw + ((int64_t)(co0 + 16 * h + row) * ci_total + c0) * kk;
            for (int rem = tid; rem < row_floats; rem += CV_THREADS) {
                const int c = kk == 16 ? rem >> 4 : kk == 9 ? rem / 9 : kk == 4 ? rem >> 2 : rem / kk;
                dst[rem] = stage[(row * 32 + c) * ldk + rem - c * kk];
            }
        }
    }
}

__global__ __launch_bounds__(CV_THREADS) void conv_wgrad_kernel(ConvArgs g, int64_t pix_per_slice, float *partial_w,
                                                                float *partial_b) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int kk = g.k * g.k;
    const int c0 = blockIdx.y * 32, co0 = blockIdx.z * 32;
    const int64_t M = (int64_t)g.B * g.Ho * g.Wo;
    const int64_t m_lo = (int64_t)blockIdx.x * pix_per_slice, m_hi = min(M, m_lo + pix_per_slice);
    f32x16 acc[WG_MAX_TAPS];
#pragma unroll
    for (int j = 0; j < WG_MAX_TAPS; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
    float bsum = 0.0f;
    for (int64_t mc = m_lo; mc < m_hi; mc += 32) {
        float av[16];
        int64_t xoff[16];
        {
            // (b, oy, ox) of this lane's first pixel by division, the other 15 by stepping
            const int64_t m_first = mc + lh * 16;
            const int64_t mf = m_first < M ? m_first : 0;
            int ox = (int)(mf % g.Wo);
            const int64_t q = mf / g.Wo;
            int oy = (int)(q % g.Ho), b = (int)(q / g.Ho);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int64_t m = m_first + t;
                const bool ok = m < m_hi;
                const float d = g.dy[(ok ? m : m_lo) * g.co + co0 + li];
                av[t] = ok ? d : 0.0f;
                xoff[t] = ok ? (((int64_t)b * g.Hi + oy * g.s) * g.Wi + ox * g.s) * g.ci + c0 + li : (int64_t)(c0 + li);
                if (++ox == g.Wo) { ox = 0; if (++oy == g.Ho) { oy = 0; ++b; } }
            }
        }
        if (wave == 0 && blockIdx.y == 0) {
#pragma unroll
            for (int t = 0; t < 16; ++t) bsum += av[t];
        }
#pragma unroll
        for (int j = 0; j < WG_MAX_TAPS; ++j) {
            const int tap = wave + 4 * j;
            if (tap < kk) {
                const int ky = tap / g.k, kx = tap - ky * g.k;
                const int64_t toff = ((int64_t)ky * g.Wi + kx) * g.ci;
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], g.x[xoff[t] + toff], acc[j], 0, 0, 0);
            }
        }
    }
    // partial_w[slice][co][c][ky][kx]
    float *pw = partial_w + (int64_t)blockIdx.x * g.co * g.ci * kk;
#pragma unroll
    for (int j = 0; j < WG_MAX_TAPS; ++j) {
        const int tap = wave + 4 * j;
        if (tap < kk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                pw[((int64_t)co * g.ci + c0 + li) * kk + tap] = acc[j][r];
            }
        }
    }
    if (wave == 0 && blockIdx.y == 0) {  // bias gradient: column sums of dy over this slice
        bsum += __shfl_xor(bsum, 32, 64);
        if (lh == 0) partial_b[(int64_t)blockIdx.x * g.co + co0 + li] = bsum;
    }
}

// The same weight gradient for SMALL feature maps (Atari's 20 x 20 and 9 x 9 layers): conv_wgrad_kernel issues 16 + 16 taps
// global loads per 32-pixel chunk and wave -- every wave re-reads dy, every tap re-reads x -- and runs at ~0.28 of the
// matrix peak.  Here a workgroup takes whole images: the image's 32-channel slice of x (Hi Wi x 32 floats, channels-last)
// and its 32-channel slice of dy (zero rows behind the last pixel) are staged in LDS once with 16-byte loads, and both
// MFMA operands are ds_read_b32 (lane = channel: consecutive dwords, conflict-free).  Wave w still takes the taps
// w, w + 4, ...; persistent workgroups accumulate over their images, one partial (= slice) per workgroup.
constexpr int WI_NX = 13, WI_ND = 4;   // 16-byte words of x / dy a thread carries for the next image (Hi Wi <= 416, Ho Wo <= 128)
__global__ __launch_bounds__(CV_THREADS, 2) void conv_wgrad_img_kernel(ConvArgs g, float *partial_w, float *partial_b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // xs [Hi Wi][32] | dys [chunks * 32][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int kk = g.k * g.k, HWi = g.Hi * g.Wi, npix = g.Ho * g.Wo, nch = (npix + 31) >> 5;
    const int c0 = blockIdx.y * 32, co0 = blockIdx.z * 32;
    float *xs = lds, *dys = lds + HWi * 32;
    f4 *xs4 = reinterpret_cast<f4 *>(xs), *dys4 = reinterpret_cast<f4 *>(dys);
    int toff[WG_MAX_TAPS];
#pragma unroll
    for (int j = 0; j < WG_MAX_TAPS; ++j) {
        const int tap = wave + 4 * j, ky = tap / g.k, kx = tap - ky * g.k;
        toff[j] = tap < kk ? (ky * g.Wi + kx) * 32 + li : li;
    }
    f32x16 acc[WG_MAX_TAPS];
#pragma unroll
    for (int j = 0; j < WG_MAX_TAPS; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
    float bsum = 0.0f;
    // the NEXT image's tiles are requested (registers) before this image's chunks and written to LDS after them: the
    // round trip hides under the MFMAs (two co-resident workgroups run in lockstep: they do not hide it for each other)
    f4 xr[WI_NX], dr[WI_ND];
    auto request = [&](int b) {
        const float *xb = g.x + (int64_t)b * HWi * g.ci + c0;
        const float *db = g.dy + (int64_t)b * npix * g.co + co0;
#pragma unroll
        for (int u = 0; u < WI_NX; ++u) {
            const int i = tid + u * CV_THREADS;
            if (i < HWi * 8) xr[u] = *reinterpret_cast<const f4 *>(xb + (int64_t)(i >> 3) * g.ci + 4 * (i & 7));
        }
#pragma unroll
        for (int u = 0; u < WI_ND; ++u) {
            const int i = tid + u * CV_THREADS;
            dr[u] = f4{0.0f, 0.0f, 0.0f, 0.0f};
            if (i < npix * 8) dr[u] = *reinterpret_cast<const f4 *>(db + (int64_t)(i >> 3) * g.co + 4 * (i & 7));
        }
    };
    if ((int)blockIdx.x < g.B) request(blockIdx.x);
    for (int b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();   // (the previous image's chunks have read the tiles)
#pragma unroll
        for (int u = 0; u < WI_NX; ++u) {
            const int i = tid + u * CV_THREADS;
            if (i < HWi * 8) xs4[i] = xr[u];
        }
#pragma unroll
        for (int u = 0; u < WI_ND; ++u) {
            const int i = tid + u * CV_THREADS;
            if (i < nch * 32 * 8) dys4[i] = dr[u];
        }
        __syncthreads();
        if (b + (int)gridDim.x < g.B) request(b + gridDim.x);
        for (int ch = 0; ch < nch; ++ch) {
            float av[16];
            int xo[16];
            {
                const int p0 = ch * 32 + 16 * lh, pf = p0 < npix ? p0 : 0;
                int oy = pf / g.Wo, ox = pf - oy * g.Wo;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int p = p0 + t;
                    av[t] = dys[p * 32 + li];   // (zero behind the last pixel)
                    xo[t] = p < npix ? ((oy * g.s) * g.Wi + ox * g.s) * 32 : 0;
                    if (++ox == g.Wo) { ox = 0; ++oy; }
                }
            }
            if (wave == 0 && blockIdx.y == 0) {
#pragma unroll
                for (int t = 0; t < 16; ++t) bsum += av[t];
            }
#pragma unroll
            for (int j = 0; j < WG_MAX_TAPS; ++j) {
                if (wave + 4 * j < kk) {
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], xs[xo[t] + toff[j]], acc[j], 0, 0, 0);
                }
            }
        }
    }
    // partial_w[slice][co][c][ky][kx], through LDS (the tiles are dead)
    wgrad_store_block<WG_MAX_TAPS>(lds, acc, [&](int j) { return wave + 4 * j; }, kk,
                                   partial_w + (int64_t)blockIdx.x * g.co * g.ci * kk, g.ci, c0, co0);
    if (wave == 0 && blockIdx.y == 0) {  // bias gradient: column sums of dy over this workgroup's images
        bsum += __shfl_xor(bsum, 32, 64);
        if (lh == 0) partial_b[(int64_t)blockIdx.x * g.co + co0 + li] = bsum;
    }
}

// ---------------------------------------------------------------------------------------------
// First layer (cnns.py:41, 76): the input is the fp32 NCHW image with the x/div + shift normalisation of the
// encoder's forward in front, few input channels (4 / 9) and a stride that divides the kernel.  Its column matrix
// is the largest buffer of the whole pixel update (419 MB at Atari batch 1024; written once, read twice), so the
// gather happens in the operand loads here too:
// The normalisation is linear, so the raw image goes through the MFMAs: the forward pass stages w/div and adds
// shift * sum(w) to the bias, the weight gradient divides the finished sums and adds shift * sum(dy).
//   forward        K runs over (c, ky); inside a run lane half h takes the KH consecutive taps kx = h*KH .. h*KH+KH-1
//                  of its pixel with ONE 8/16-byte load (rows of k < 2 KH taps are padded with zero WEIGHTS; the
//                  extra pixel read is inside the image row: (Wo-1) s + 2 KH <= Wi is a launch condition);
//   weight grad    N runs over the flattened (c, ky, kx) patch index (NB blocks of 32), the reduction over pixels;
//                  MFMA step t of a 32-pixel chunk takes the pixels 2t and 2t+1 (adjacent -> shared cache lines).
// ---------------------------------------------------------------------------------------------
struct FirstArgs {
    const float *img;    // (B, C, Hi, Wi) fp32, raw pixel values
    const float *w;      // (co, C, k, k)
    const float *bias;   // (co)
    const float *dy;     // (B, Ho, Wo, co)   [wgrad]
    float *out;          // (B, Ho, Wo, co)   [fwd]
    float div, shift;    // x / div + shift in front of the convolution
    int B, C, Hi, Wi, Ho, Wo, co, k, s;
};

template <int KH> struct TapVec;
template <> struct TapVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct TapVec<4> { typedef float type __attribute__((ext_vector_type(4))); };

constexpr int FG = 4;  // runs per load group (two groups in flight)

template <int KH>
__global__ __launch_bounds__(CV_THREADS) void conv_first_fwd_kernel(FirstArgs g, int n_tiles) {
    typedef typename TapVec<KH>::type vec;
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [run = (c, ky)][32 co][2 KH taps], zero-padded taps
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int nruns = g.C * g.k, kk = g.k * g.k, HW = g.Hi * g.Wi;
    const int co0 = blockIdx.y * 32;
    for (int i = tid; i < nruns * 32 * 2 * KH; i += CV_THREADS) {
        const int kx = i % (2 * KH), t = i / (2 * KH), co = t & 31, run = t >> 5;
        const int c = run / g.k, ky = run - c * g.k;
        // the input normalisation is linear, so it moves to the weights:  sum w (x/div + shift) = sum (w/div) x + shift sum w
        wl[i] = kx < g.k ? g.w[((int64_t)(co0 + co) * g.C + c) * kk + ky * g.k + kx] / g.div : 0.0f;
    }
    __syncthreads();
    float bias = g.bias[co0 + li];
    if (g.shift != 0.0f) {
        // shift * sum_n w[co][n] for this lane's output channel: fixed-order sum of the staged (w/div) rows, times div
        float ws = 0.0f;
        for (int run = 0; run < nruns; ++run)
#pragma unroll
            for (int j = 0; j < 2 * KH; ++j) ws += wl[(run * 32 + li) * 2 * KH + j];
        bias += g.shift * (ws * g.div);
    }
    const int M = g.B * g.Ho * g.Wo;
    const int ngroups = (nruns + FG - 1) / FG;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int m = tile * CV_PIX + wave * 32 + li;
        const int mm = m < M ? m : 0;
        const int ox = mm % g.Wo, t = mm / g.Wo;
        const int oy = t % g.Ho, b = t / g.Ho;
        const float *base = g.img + (int64_t)b * g.C * HW + (oy * g.s) * g.Wi + ox * g.s + lh * KH;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        // run -> image offset, stepped (c, ky) without divisions
        int l_run = 0, l_ky = 0, l_off = 0;
        auto load_group = [&](vec (&a)[FG]) {
#pragma unroll
            for (int u = 0; u < FG; ++u) {
                a[u] = *reinterpret_cast<const vec *>(base + l_off);
                if (l_run + 1 < nruns) {  // clamped: the groups past the end re-read the last run
                    ++l_run;
                    l_off += g.Wi;
                    if (++l_ky == g.k) { l_ky = 0; l_off += HW - g.k * g.Wi; }
                }
            }
        };
        auto compute_group = [&](const vec (&a)[FG], int grp) {
#pragma unroll
            for (int u = 0; u < FG; ++u) {
                const int run = grp * FG + u;
                if (run < nruns) {
                    const vec bw = *reinterpret_cast<const vec *>(wl + ((run * 32 + li) * 2 + lh) * KH);
#pragma unroll
                    for (int j = 0; j < KH; ++j)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][j], bw[j], acc, 0, 0, 0);
                }
            }
        };
        vec a0[FG], a1[FG];
        load_group(a0);
        for (int grp = 0; grp < ngroups; grp += 2) {
            load_group(a1);
            compute_group(a0, grp);
            load_group(a0);
            compute_group(a1, grp + 1);
        }
        const int m_wave = tile * CV_PIX + wave * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = m_wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (mr < M) g.out[(int64_t)mr * g.co + co0 + li] = fmaxf(acc[r] + bias, 0.0f);
        }
    }
}


// weight gradient of the first layer: grid (pixel slices, co / 32); the 4 waves of a workgroup split the slice's pixels
// and add their accumulators through LDS in a fixed order, so a slice is one workgroup's partial
template <int NB>
__global__ __launch_bounds__(CV_THREADS) void conv_first_wgrad_kernel(FirstArgs g, int pix_per_slice, float *partial_w,
                                                                      float *partial_b) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [NB][16][64] + [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int kk = g.k * g.k, ckk = g.C * kk, HW = g.Hi * g.Wi, CHW = g.C * HW;
    const int co0 = blockIdx.y * 32;
    int noff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = nb * 32 + li;
        const int c = n / kk, rem = n - c * kk, ky = rem / g.k, kx = rem - ky * g.k;
        noff[nb] = n < ckk ? c * HW + ky * g.Wi + kx : 0;
    }
    const int M = g.B * g.Ho * g.Wo;
    const int per_wave = pix_per_slice >> 2;
    const int m_lo = blockIdx.x * pix_per_slice + wave * per_wave;
    const int m_hi = min(M, m_lo + per_wave);
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.0f;
    float bsum = 0.0f;
    for (int mc = m_lo; mc < m_hi; mc += 32) {
        float av[16];
        int xo[16];
        {
            const int m_first = mc + lh;
            const int mf = m_first < M ? m_first : 0;
            int ox = mf % g.Wo;
            const int q = mf / g.Wo;
            int oy = q % g.Ho, b = q / g.Ho;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int m = m_first + 2 * t;
                const bool ok = m < m_hi;
                const float d = g.dy[(int64_t)(ok ? m : m_lo) * g.co + co0 + li];
                av[t] = ok ? d : 0.0f;
                xo[t] = ok ? b * CHW + (oy * g.s) * g.Wi + ox * g.s : 0;
                ox += 2;
                if (ox >= g.Wo) { ox -= g.Wo; if (++oy == g.Ho) { oy = 0; ++b; } }
            }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) bsum += av[t];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int t = 0; t < 16; ++t)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], g.img[xo[t] + noff[nb]], acc[nb], 0, 0, 0);
        }
    }
    bsum += __shfl_xor(bsum, 32, 64);
    float *bred = red + NB * 16 * 64;
    // fixed-order sum over the waves: 3 -> 2 -> 1 -> 0
    for (int w = 3; w >= 1; --w) {
        if (wave == w) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(nb * 16 + r) * 64 + lane] = acc[nb][r];
            if (lh == 0) bred[li] = bsum;
        }
        __syncthreads();
        if (wave == w - 1) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nb][r] += red[(nb * 16 + r) * 64 + lane];
            bsum += bred[li];
        }
        __syncthreads();
    }
    if (wave == 0) {
        float *pw = partial_w + (int64_t)blockIdx.x * g.co * ckk;   // partial_w[slice][co][c][ky][kx]
        float brow[16];  // sum of dy over the slice for the output channel of accumulator row r
#pragma unroll
        for (int r = 0; r < 16; ++r) brow[r] = __shfl(bsum, (r & 3) + 8 * (r >> 2) + 4 * lh, 64);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = nb * 32 + li;
            if (n < ckk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // sum dy (x/div + shift) = (sum dy x) / div + shift sum dy   (the raw image went through the MFMAs)
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    pw[(int64_t)(co0 + row) * ckk + n] = acc[nb][r] / g.div + g.shift * brow[r];
                }
            }
        }
        if (lh == 0) partial_b[(int64_t)blockIdx.x * g.co + co0 + li] = bsum;
    }
}

// The same weight gradient with the image operand staged through LDS.  conv_first_wgrad_kernel gathers its B operand from
// global memory -- per MFMA 64 scattered dwords (lane = patch entry (c, ky, kx), 3 / 8 consecutive floats per (c, ky) row):
// 48 (DMC) / 128 (Atari) gather instructions per 32-pixel chunk through the texture path.  Here a workgroup takes (image b,
// band of R output rows) items: it copies the band's input rows (C x Rin x Wi fp32, contiguous per channel) into LDS with
// 16-byte loads and reads the patch entries from there (ds_read_b32, 2-way conflicts at worst).  Persistent workgroups
// accumulate over their items; one partial (= slice) per workgroup.  Summation order over pixels differs from the gather
// form's (rounding only).
// 16-byte loads a thread keeps in flight while staging a band (DMC: 2457 per band / 256 threads); 5 .. 8 patch blocks hold
// 80 .. 128 accumulator registers per lane: fewer loads in flight there, and two waves per SIMD asked of the compiler
template <int NB>
__global__ __launch_bounds__(CV_THREADS, NB > 4 ? 2 : 1) void conv_first_wgrad_band_kernel(FirstArgs g, int R, int n_items, float *partial_w,
                                                                           float *partial_b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // band [C][Rin][Wi]; after the items: red [NB][16][64] + [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int kk = g.k * g.k, ckk = g.C * kk, HW = g.Hi * g.Wi;
    const int Rin = (R - 1) * g.s + g.k, plane = Rin * g.Wi;
    const int co0 = blockIdx.y * 32;
    int noff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = nb * 32 + li;
        const int c = n / kk, rem = n - c * kk, ky = rem / g.k, kx = rem - ky * g.k;
        noff[nb] = n < ckk ? c * plane + ky * g.Wi + kx : 0;
    }
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.0f;
    float bsum = 0.0f;
    const f4 *img4 = reinterpret_cast<const f4 *>(g.img);
    f4 *lds4 = reinterpret_cast<f4 *>(lds);
    constexpr int BAND_LOADS = NB > 4 ? 4 : 10;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        // item -> (image, band): the band index changes slowest, so a persistent workgroup meets every band height
        const int bi = item / g.B, b = item - bi * g.B;
        const int oy0 = bi * R, rows = min(R, g.Ho - oy0);
        const int iy0 = oy0 * g.s, rin = (rows - 1) * g.s + g.k;
        const int per_c = (rin * g.Wi) >> 2, total = g.C * per_c;
        __syncthreads();   // (the previous item's chunks have read the band)
        for (int base = tid; base < total; base += BAND_LOADS * CV_THREADS) {   // the band in (mostly) one round trip
            f4 v[BAND_LOADS];
#pragma unroll
            for (int u = 0; u < BAND_LOADS; ++u) {
                const int i = base + u * CV_THREADS;
                if (i < total) {
                    const int c = i / per_c, rem = i - c * per_c;
                    v[u] = img4[(((int64_t)b * g.C + c) * HW + iy0 * g.Wi) / 4 + rem];
                }
            }
#pragma unroll
            for (int u = 0; u < BAND_LOADS; ++u) {
                const int i = base + u * CV_THREADS;
                if (i < total) {
                    const int c = i / per_c, rem = i - c * per_c;
                    lds4[(c * plane) / 4 + rem] = v[u];
                }
            }
        }
        __syncthreads();
        const int npix = rows * g.Wo, nchunks = (npix + 31) >> 5;
        const int64_t mbase = ((int64_t)b * g.Ho + oy0) * g.Wo;
        for (int ch = wave; ch < nchunks; ch += 4) {
            float av[16];
            int xo[16];
            {
                const int p_first = ch * 32 + lh, pf = p_first < npix ? p_first : 0;
                int r = pf / g.Wo, ox = pf - r * g.Wo;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int p = p_first + 2 * t;
                    const bool ok = p < npix;
                    const float d = g.dy[(mbase + (ok ? p : 0)) * g.co + co0 + li];
                    av[t] = ok ? d : 0.0f;
                    xo[t] = ok ? (r * g.s) * g.Wi + ox * g.s : 0;
                    ox += 2;
                    if (ox >= g.Wo) { ox -= g.Wo; ++r; }
                }
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) bsum += av[t];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], lds[xo[t] + noff[nb]], acc[nb], 0, 0, 0);
            }
        }
    }
    __syncthreads();   // the band is dead: its LDS carries the waves' partial sums now
    float *red = lds;
    bsum += __shfl_xor(bsum, 32, 64);
    float *bred = red + NB * 16 * 64;
    // fixed-order sum over the waves: 3 -> 2 -> 1 -> 0
    for (int w = 3; w >= 1; --w) {
        if (wave == w) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(nb * 16 + r) * 64 + lane] = acc[nb][r];
            if (lh == 0) bred[li] = bsum;
        }
        __syncthreads();
        if (wave == w - 1) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nb][r] += red[(nb * 16 + r) * 64 + lane];
            bsum += bred[li];
        }
        __syncthreads();
    }
    if (wave == 0) {
        float *pw = partial_w + (int64_t)blockIdx.x * g.co * ckk;   // partial_w[slice][co][c][ky][kx]
        float brow[16];  // sum of dy over the slice for the output channel of accumulator row r
#pragma unroll
        for (int r = 0; r < 16; ++r) brow[r] = __shfl(bsum, (r & 3) + 8 * (r >> 2) + 4 * lh, 64);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = nb * 32 + li;
            if (n < ckk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // sum dy (x/div + shift) = (sum dy x) / div + shift sum dy   (the raw image went through the MFMAs)
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    pw[(int64_t)(co0 + row) * ckk + n] = acc[nb][r] / g.div + g.shift * brow[r];
                }
            }
        }
        if (lh == 0) partial_b[(int64_t)blockIdx.x * g.co + co0 + li] = bsum;
    }
}

// taps per lane half of the first-layer kernels for this geometry, 0 = stays on im2col
int first_kh(int C, int co, int k, int s, int Hi, int Wi, int64_t B) {
    if (co % 32 || k < 1 || s < 1 || Hi < k || Wi < k || C * k * k > 8 * 32) return 0;
    if (B * C * Hi * Wi >= (1ll << 31) || B * ((Hi - k) / s + 1) * ((Wi - k) / s + 1) >= (1ll << 31) - 65536) return 0;
    const int Wo = (Wi - k) / s + 1;
    for (int kh = 2; kh <= 4; kh += 2) {
        if (k <= 2 * kh && (Wo - 1) * s + 2 * kh <= Wi && Wi % kh == 0 && s % kh == 0) return kh;
    }
    return 0;
}

// The same weight gradient for k*k <= NT taps with every wave holding ALL taps (NT accumulators) over its own quarter
// of the slice's pixels: the taps are balanced over the waves (9 taps over 4 waves were 3:2:2:2 above), dy is loaded
// once per pixel instead of once per wave, MFMA step t of a 32-pixel chunk takes the ADJACENT pixels 2t / 2t+1 (one
// 256-byte segment per operand load), and the operand loads of tap j+1 are in flight under the MFMAs of tap j.  The
// four waves' accumulators are added through LDS in a fixed order (3 -> 2 -> 1 -> 0).
template <int NT>
__global__ __launch_bounds__(CV_THREADS) void conv_wgrad_taps_kernel(ConvArgs g, int pix_per_slice, float *partial_w,
                                                                     float *partial_b) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [NT][16][64] + [32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int kk = g.k * g.k;
    const int c0 = blockIdx.y * 32, co0 = blockIdx.z * 32;
    const int M = g.B * g.Ho * g.Wo;
    const int per_wave = pix_per_slice >> 2;
    const int m_lo = blockIdx.x * pix_per_slice + wave * per_wave;
    const int m_hi = min(M, m_lo + per_wave);
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
    float bsum = 0.0f;
    const float *xb = g.x + c0 + li;
    // One wave per SIMD (nine accumulators: 440 VGPRs), so nothing hides a round trip but the wave's own MFMAs: the dy values
    // and pixel offsets of the NEXT 32-pixel chunk are requested under the current chunk's 144 MFMAs, and so are the next
    // chunk's first-tap x operands (under the current chunk's last tap) -- per chunk two exposed round trips less.
    auto fetch = [&](float (&av)[16], int (&xo)[16], int mc) {
        const int m_first = mc + lh;
        const int mf = m_first < M ? m_first : 0;
        int ox = mf % g.Wo;
        const int q = mf / g.Wo;
        int oy = q % g.Ho, b = q / g.Ho;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int m = m_first + 2 * t;
            const bool ok = m < m_hi;
            const float d = g.dy[(int64_t)(ok ? m : m_lo) * g.co + co0 + li];
            av[t] = ok ? d : 0.0f;
            xo[t] = ok ? ((b * g.Hi + oy * g.s) * g.Wi + ox * g.s) * g.ci : 0;
            ox += 2;
            if (ox >= g.Wo) { ox -= g.Wo; if (++oy == g.Ho) { oy = 0; ++b; } }
        }
    };
    float avA[16], avB[16], bx0[16];
    int xoA[16], xoB[16];
    // one chunk: its first-tap operands are in bx0; (avn, xon) is the next chunk, fetched here, whose first-tap operands
    // replace bx0 under the last tap
    auto process = [&](const float (&av)[16], const int (&xo)[16], float (&avn)[16], int (&xon)[16], int mc_next) {
        fetch(avn, xon, mc_next);
        if (blockIdx.y == 0) {
#pragma unroll
            for (int t = 0; t < 16; ++t) bsum += av[t];
        }
        float bx[2][16];
        int ky = 0, kx = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) bx[0][t] = bx0[t];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (j + 1 < NT) {  // operands of the next tap (clamped to the last one: uniform, in bounds)
                if (j + 1 < kk) { if (++kx == g.k) { kx = 0; ++ky; } }
                const int toff = (ky * g.Wi + kx) * g.ci;
#pragma unroll
                for (int t = 0; t < 16; ++t) bx[(j + 1) & 1][t] = xb[xo[t] + toff];
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) bx0[t] = xb[xon[t]];   // the next chunk's tap (0, 0)
            }
            if (j < kk) {
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bx[j & 1][t], acc[j], 0, 0, 0);
            }
        }
    };
    if (m_lo < m_hi) {
        fetch(avA, xoA, m_lo);
#pragma unroll
        for (int t = 0; t < 16; ++t) bx0[t] = xb[xoA[t]];
        for (int mc = m_lo; mc < m_hi; mc += 64) {
            process(avA, xoA, avB, xoB, mc + 32);   // (a chunk beyond m_hi fetches zeros from valid addresses)
            if (mc + 32 >= m_hi) break;
            process(avB, xoB, avA, xoA, mc + 64);
        }
    }
    bsum += __shfl_xor(bsum, 32, 64);
    float *bred = red + NT * 16 * 64;
    for (int w = 3; w >= 1; --w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(j * 16 + r) * 64 + lane] = acc[j][r];
            if (lh == 0) bred[li] = bsum;
        }
        __syncthreads();
        if (wave == w - 1) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] += red[(j * 16 + r) * 64 + lane];
            bsum += bred[li];
        }
        __syncthreads();
    }
    if (wave == 0) {
        // (direct stores: consecutive lanes are k k floats apart -- staging this block through LDS as conv_wgrad_img_kernel
        //  does cost this kernel 9 us per launch at DMC: two more barrier rounds with three waves idle)
        float *pw = partial_w + (int64_t)blockIdx.x * g.co * g.ci * kk;   // partial_w[slice][co][c][ky][kx]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (j < kk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    pw[((int64_t)co * g.ci + c0 + li) * kk + j] = acc[j][r];
                }
            }
        }
        if (blockIdx.y == 0 && lh == 0) partial_b[(int64_t)blockIdx.x * g.co + co0 + li] = bsum;
    }
}

// persistent workgroups per CU: as many as the weight tile in LDS allows, at most 3 (three waves per SIMD hide the
// operand-load latency the two-chunk prefetch leaves; measured on the DMC update at 2 / 3 / 4: 3.13 / 3.09 / 3.28 ms --
// a fourth workgroup's operand stream no longer fits the L1 next to the others')
int persistent_per_cu(size_t lds) {
    const int fit = (int)((160 * 1024) / (lds + 512));
    return fit < 1 ? 1 : fit > 3 ? 3 : fit;
}

int conv_ok(int ci, int co, int k) { return ci % 32 == 0 && co % 32 == 0 && k >= 1 && k * k <= 4 * WG_MAX_TAPS; }

}  // namespace

extern "C" int ssac_conv_implicit_supported(int ci, int co, int k) { return conv_ok(ci, co, k) ? 1 : 0; }

extern "C" int ssac_conv_fwd(const float *x, const float *w, const float *bias, float *y, int B, int Hi, int Wi,
                             int ci, int co, int k, int s, void *stream) {
    if (!conv_ok(ci, co, k)) return ssac_fail("ssac_conv_fwd: needs ci % 32 == 0 and co % 32 == 0");
    ConvArgs g{};
    g.x = x; g.w = w; g.bias = bias; g.out = y; g.B = B; g.Hi = Hi; g.Wi = Wi; g.ci = ci; g.co = co; g.k = k; g.s = s;
    g.Ho = (Hi - k) / s + 1; g.Wo = (Wi - k) / s + 1;
    const int64_t M = (int64_t)B * g.Ho * g.Wo;
    const int n_tiles = (int)((M + CV_PIX - 1) / CV_PIX);
    const size_t lds = sizeof(float) * (size_t)k * k * (ci / 32) * 32 * WL_LD;
    if (lds > 160 * 1024) return ssac_fail("ssac_conv_fwd: weight tile does not fit LDS");
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)conv_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const int per_cu = persistent_per_cu(lds);
    const int cap = 256 * per_cu / (co / 32) > 0 ? 256 * per_cu / (co / 32) : 1;
    const int gx = n_tiles < cap ? n_tiles : cap;
    SSAC_LAUNCH(conv_fwd_kernel, dim3(gx, co / 32), dim3(CV_THREADS), lds, (hipStream_t)stream, g, n_tiles);
    return ssac_check_launch("conv_fwd");
}

extern "C" int ssac_conv_dgrad(const float *dy, const float *w, const float *x_mask, float *dx, int B, int Hi, int Wi,
                               int ci, int co, int k, int s, void *stream) {
    if (!conv_ok(ci, co, k)) return ssac_fail("ssac_conv_dgrad: needs ci % 32 == 0 and co % 32 == 0");
    ConvArgs g{};
    g.x = x_mask; g.w = w; g.dy = dy; g.out = dx; g.B = B; g.Hi = Hi; g.Wi = Wi; g.ci = ci; g.co = co; g.k = k; g.s = s;
    g.Ho = (Hi - k) / s + 1; g.Wo = (Wi - k) / s + 1;
    const int64_t M = (int64_t)B * Hi * Wi;
    const int n_tiles = (int)((M + CV_PIX - 1) / CV_PIX);
    const size_t lds = sizeof(float) * (size_t)k * k * (co / 32) * 32 * WL_LD;
    if (lds > 160 * 1024) return ssac_fail("ssac_conv_dgrad: weight tile does not fit LDS");
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)conv_dgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)conv_dgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const int per_cu = persistent_per_cu(lds);
    const int cap = 256 * per_cu / (ci / 32) > 0 ? 256 * per_cu / (ci / 32) : 1;
    const int gx = n_tiles < cap ? n_tiles : cap;
    if (s == 1) {
        SSAC_LAUNCH(conv_dgrad_kernel<true>, dim3(gx, ci / 32), dim3(CV_THREADS), lds, (hipStream_t)stream, g, n_tiles);
    } else if (s <= DG_MAX_S && M < (1ll << 31) - 65536) {
        // parity classes: only the taps a class of input pixels can receive
        StridedTiles tl{};
        int total = 0;
        for (int cls = 0; cls < s * s; ++cls) {
            const int py = cls / s, px = cls % s;
            const int64_t npix = (int64_t)B * ((Hi - py + s - 1) / s) * ((Wi - px + s - 1) / s);
            tl.first[cls] = total;
            total += (int)((npix + CV_PIX - 1) / CV_PIX);
        }
        tl.first[s * s] = total;
        const int tj = (k + s - 1) / s, max_chunks = tj * tj * (co / 32);   // taps of the richest class
        const size_t lds2 = sizeof(float) * (size_t)max_chunks * 32 * WL_LD + sizeof(int) * 128;
        static bool attr2 = false;
        if (!attr2) {
            (void)hipFuncSetAttribute((const void *)conv_dgrad_strided_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
            attr2 = true;
        }
        if (lds2 > 160 * 1024) return ssac_fail("ssac_conv_dgrad: weight tile does not fit LDS");
        // persistent workgroups, the same number for every class
        const int per_cu2 = persistent_per_cu(lds2), ncls = s * s;
        int per_class = 256 * per_cu2 / (ci / 32) / ncls;
        const int most = (total + ncls - 1) / ncls + 1;
        per_class = per_class < 1 ? 1 : per_class > most ? most : per_class;
        SSAC_LAUNCH(conv_dgrad_strided_kernel, dim3(per_class * ncls, ci / 32), dim3(CV_THREADS), lds2,
                    (hipStream_t)stream, g, tl, max_chunks);
    } else {
        SSAC_LAUNCH(conv_dgrad_kernel<false>, dim3(gx, ci / 32), dim3(CV_THREADS), lds, (hipStream_t)stream, g, n_tiles);
    }
    return ssac_check_launch("conv_dgrad");
}

extern "C" int ssac_conv_wgrad_slices(int B, int Ho, int Wo, int pix_per_slice) {
    const int64_t M = (int64_t)B * Ho * Wo;
    return (int)((M + pix_per_slice - 1) / pix_per_slice);
}

// the whole-image form: number of persistent workgroups per (ci / 32, co / 32) block pair = slices (0: not covered --
// both tiles of an image must fit LDS twice per CU, at most 16 taps)
static int wgrad_img_slices(int B, int Hi, int Wi, int ci, int co, int k, int s, size_t *lds_out) {
    if (!conv_ok(ci, co, k) || k * k > 4 * WG_MAX_TAPS || s < 1 || Hi < k || Wi < k || (ci & 3) || (co & 3)) return 0;
    const int Ho = (Hi - k) / s + 1, Wo = (Wi - k) / s + 1;
    size_t lds = sizeof(float) * 32 * ((size_t)Hi * Wi + (size_t)((Ho * Wo + 31) / 32) * 32);
    const size_t stage = sizeof(float) * 16 * 32 * (size_t)(k * k + 1);   // (the epilogue's staging area)
    if (lds < stage) lds = stage;
    if (lds > 80 * 1024 || Hi * Wi * 8 > WI_NX * CV_THREADS || ((Ho * Wo + 31) / 32) * 32 * 8 > WI_ND * CV_THREADS) return 0;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 2) per_cu = 2;   // (two waves per SIMD asked of the compiler: the accumulators + the next image's words)
    const int blocks = (ci / 32) * (co / 32);
    const int cap = 256 * per_cu / blocks > 0 ? 256 * per_cu / blocks : 1;
    if (lds_out) *lds_out = lds;
    return B < cap ? B : cap;
}

extern "C" int ssac_conv_wgrad_img_slices(int B, int Hi, int Wi, int ci, int co, int k, int s) {
    return wgrad_img_slices(B, Hi, Wi, ci, co, k, s, nullptr);
}

extern "C" int ssac_conv_wgrad_img(const float *dy, const float *x, float *partial_w, float *partial_b, int B, int Hi, int Wi,
                                   int ci, int co, int k, int s, void *stream) {
    size_t lds = 0;
    const int slices = wgrad_img_slices(B, Hi, Wi, ci, co, k, s, &lds);
    if (!slices) return ssac_fail("ssac_conv_wgrad_img: geometry not supported (see ssac_conv_wgrad_img_slices)");
    if (((uintptr_t)x | (uintptr_t)dy) & 15) return ssac_fail("ssac_conv_wgrad_img: operands must be 16-byte aligned");
    ConvArgs g{};
    g.x = x; g.dy = dy; g.B = B; g.Hi = Hi; g.Wi = Wi; g.ci = ci; g.co = co; g.k = k; g.s = s;
    g.Ho = (Hi - k) / s + 1; g.Wo = (Wi - k) / s + 1;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)conv_wgrad_img_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    SSAC_LAUNCH(conv_wgrad_img_kernel, dim3(slices, ci / 32, co / 32), dim3(CV_THREADS), lds, (hipStream_t)stream, g, partial_w,
                partial_b);
    return ssac_check_launch("conv_wgrad_img");
}

extern "C" int ssac_conv_wgrad(const float *dy, const float *x, float *partial_w, float *partial_b, int B, int Hi,
                               int Wi, int ci, int co, int k, int s, int pix_per_slice, void *stream) {
    if (!conv_ok(ci, co, k)) return ssac_fail("ssac_conv_wgrad: needs ci % 32 == 0 and co % 32 == 0");
    if (pix_per_slice <= 0 || (pix_per_slice & 31)) return ssac_fail("ssac_conv_wgrad: slice must be a multiple of 32");
    ConvArgs g{};
    g.x = x; g.dy = dy; g.B = B; g.Hi = Hi; g.Wi = Wi; g.ci = ci; g.co = co; g.k = k; g.s = s;
    g.Ho = (Hi - k) / s + 1; g.Wo = (Wi - k) / s + 1;
    const int slices = ssac_conv_wgrad_slices(B, g.Ho, g.Wo, pix_per_slice);
    const int64_t in_elems = (int64_t)B * Hi * Wi * ci, out_pix = (int64_t)B * g.Ho * g.Wo;
    if (k * k <= 9 && (pix_per_slice & 127) == 0 && in_elems < (1ll << 31) && out_pix < (1ll << 31) - 65536) {
        // every wave holds all taps (32-bit offsets)
        const size_t lds = sizeof(float) * (9 * 16 * 64 + 32);
        SSAC_LAUNCH(conv_wgrad_taps_kernel<9>, dim3(slices, ci / 32, co / 32), dim3(CV_THREADS), lds, (hipStream_t)stream,
                    g, pix_per_slice, partial_w, partial_b);
        return ssac_check_launch("conv_wgrad");
    }
    SSAC_LAUNCH(conv_wgrad_kernel, dim3(slices, ci / 32, co / 32), dim3(CV_THREADS), 0, (hipStream_t)stream, g,
                (int64_t)pix_per_slice, partial_w, partial_b);
    return ssac_check_launch("conv_wgrad");
}

extern "C" int ssac_conv_first_supported(int C, int co, int k, int s, int Hi, int Wi, int64_t B) {
    return first_kh(C, co, k, s, Hi, Wi, B);
}

extern "C" int ssac_conv_first_fwd(const float *img, const float *w, const float *bias, float *y, int B, int C, int Hi,
                                   int Wi, int co, int k, int s, float div, float shift, void *stream) {
    const int kh = first_kh(C, co, k, s, Hi, Wi, B);
    if (!kh) return ssac_fail("ssac_conv_first_fwd: geometry not supported (see ssac_conv_first_supported)");
    if ((uintptr_t)img & 15) return ssac_fail("ssac_conv_first_fwd: image must be 16-byte aligned");
    FirstArgs g{};
    g.img = img; g.w = w; g.bias = bias; g.out = y; g.div = div; g.shift = shift;
    g.B = B; g.C = C; g.Hi = Hi; g.Wi = Wi; g.co = co; g.k = k; g.s = s;
    g.Ho = (Hi - k) / s + 1; g.Wo = (Wi - k) / s + 1;
    const int M = B * g.Ho * g.Wo;
    const int n_tiles = (M + CV_PIX - 1) / CV_PIX;
    const size_t lds = sizeof(float) * (size_t)C * k * 32 * 2 * kh;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)conv_first_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)conv_first_fwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    if (lds > 160 * 1024) return ssac_fail("ssac_conv_first_fwd: weight tile does not fit LDS");
    const int cap = 1024 / (co / 32) > 0 ? 1024 / (co / 32) : 1;  // persistent: ~4 workgroups per CU in total
    const int gx = n_tiles < cap ? n_tiles : cap;
    if (kh == 2)
        SSAC_LAUNCH(conv_first_fwd_kernel<2>, dim3(gx, co / 32), dim3(CV_THREADS), lds, (hipStream_t)stream, g, n_tiles);
    else
        SSAC_LAUNCH(conv_first_fwd_kernel<4>, dim3(gx, co / 32), dim3(CV_THREADS), lds, (hipStream_t)stream, g, n_tiles);
    return ssac_check_launch("conv_first_fwd");
}

// ---- the LDS-staged first-layer weight gradient: band height R and the number of persistent workgroups (= slices)
static int first_band_rows(int C, int co, int k, int s, int Hi, int Wi, int64_t B, int *slices_out) {
    if (!first_kh(C, co, k, s, Hi, Wi, B) || (Wi & 3) || ((Hi * Wi) & 3)) return 0;
    const int Ho = (Hi - k) / s + 1, Wo = (Wi - k) / s + 1, nb = (C * k * k + 31) / 32;
    if (nb > 8) return 0;
    const size_t red = sizeof(float) * ((size_t)nb * 16 * 64 + 64);
    int best = 0;
    double best_cost = 0.0;
    for (int R = 1; R <= Ho; ++R) {
        const int Rin = (R - 1) * s + k;
        if (sizeof(float) * (size_t)C * Rin * Wi > 48 * 1024) break;
        double cost = 0.0;
        for (int oy0 = 0; oy0 < Ho; oy0 += R) {   // MFMA rounds of the four waves + the band's rows (clocks, roughly)
            const int rows = R < Ho - oy0 ? R : Ho - oy0, nch = (rows * Wo + 31) / 32;
            cost += (double)((nch + 3) / 4) * (nb * 16 * 64 + 400) + (double)((rows - 1) * s + k) * (C * Wi * 0.25) + 1500.0;
        }
        if (!best || cost <= best_cost) { best = R; best_cost = cost; }
    }
    if (best && slices_out) {
        const size_t band = sizeof(float) * (size_t)C * ((best - 1) * s + k) * Wi, lds = band > red ? band : red;
        int per_cu = (int)((160 * 1024) / lds);
        const int by_regs = nb <= 1 ? 4 : nb <= 3 ? 3 : 2;   // (116 / 160 / 148 / 172 registers per lane for 1 .. 4 blocks; <= 256 asked for 5 .. 8)
        per_cu = per_cu < by_regs ? per_cu : by_regs;
        if (per_cu < 1) per_cu = 1;
        const int64_t n_items = B * ((Ho + best - 1) / best);
        const int64_t cap = 256 * per_cu / (co / 32) > 0 ? 256 * per_cu / (co / 32) : 1;
        *slices_out = (int)(n_items < cap ? n_items : cap);
    }
    return best;
}

extern "C" int ssac_conv_first_wgrad_band_slices(int B, int C, int Hi, int Wi, int co, int k, int s) {
    int slices = 0;
    return first_band_rows(C, co, k, s, Hi, Wi, B, &slices) ? slices : 0;
}

extern "C" int ssac_conv_first_wgrad_band(const float *dy, const float *img, float *partial_w, float *partial_b, int B, int C,
                                          int Hi, int Wi, int co, int k, int s, float div, float shift, void *stream) {
    int slices = 0;
    const int R = first_band_rows(C, co, k, s, Hi, Wi, B, &slices);
    if (!R) return ssac_fail("ssac_conv_first_wgrad_band: geometry not supported (see ssac_conv_first_wgrad_band_slices)");
    if ((uintptr_t)img & 15) return ssac_fail("ssac_conv_first_wgrad_band: image must be 16-byte aligned");
    FirstArgs g{};
    g.img = img; g.dy = dy; g.div = div; g.shift = shift;
    g.B = B; g.C = C; g.Hi = Hi; g.Wi = Wi; g.co = co; g.k = k; g.s = s;
    g.Ho = (Hi - k) / s + 1; g.Wo = (Wi - k) / s + 1;
    const int nb = (C * k * k + 31) / 32;
    const size_t band = sizeof(float) * (size_t)C * ((R - 1) * s + k) * Wi, red = sizeof(float) * ((size_t)nb * 16 * 64 + 64);
    const size_t lds = band > red ? band : red;
    const int n_items = B * ((g.Ho + R - 1) / R);
    const dim3 grid(slices, co / 32), block(CV_THREADS);
#define SSAC_FIRST_WGRAD_BAND(NB) \
    case NB: { \
        static bool attr##NB = false; \
        if (!attr##NB) { (void)hipFuncSetAttribute((const void *)conv_first_wgrad_band_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr##NB = true; } \
        SSAC_LAUNCH(conv_first_wgrad_band_kernel<NB>, grid, block, lds, (hipStream_t)stream, g, R, n_items, partial_w, partial_b); } break;
    switch (nb) {
        SSAC_FIRST_WGRAD_BAND(1) SSAC_FIRST_WGRAD_BAND(2) SSAC_FIRST_WGRAD_BAND(3) SSAC_FIRST_WGRAD_BAND(4)
        SSAC_FIRST_WGRAD_BAND(5) SSAC_FIRST_WGRAD_BAND(6) SSAC_FIRST_WGRAD_BAND(7) SSAC_FIRST_WGRAD_BAND(8)
        default: return ssac_fail("ssac_conv_first_wgrad_band: patch too large");
    }
#undef SSAC_FIRST_WGRAD_BAND
    return ssac_check_launch("conv_first_wgrad_band");
}


extern "C" int ssac_conv_first_wgrad(const float *dy, const float *img, float *partial_w, float *partial_b, int B, int C,
                                     int Hi, int Wi, int co, int k, int s, float div, float shift, int pix_per_slice,
                                     void *stream) {
    if (!first_kh(C, co, k, s, Hi, Wi, B)) return ssac_fail("ssac_conv_first_wgrad: geometry not supported");
    if (pix_per_slice <= 0 || (pix_per_slice & 127)) return ssac_fail("ssac_conv_first_wgrad: slice must be a multiple of 128");
    FirstArgs g{};
    g.img = img; g.dy = dy; g.div = div; g.shift = shift;
    g.B = B; g.C = C; g.Hi = Hi; g.Wi = Wi; g.co = co; g.k = k; g.s = s;
    g.Ho = (Hi - k) / s + 1; g.Wo = (Wi - k) / s + 1;
    const int slices = ssac_conv_wgrad_slices(B, g.Ho, g.Wo, pix_per_slice);
    const int nb = (C * k * k + 31) / 32;
    const size_t lds = sizeof(float) * ((size_t)nb * 16 * 64 + 64);
    const dim3 grid(slices, co / 32), block(CV_THREADS);
#define SSAC_FIRST_WGRAD(NB) \
    case NB: SSAC_LAUNCH(conv_first_wgrad_kernel<NB>, grid, block, lds, (hipStream_t)stream, g, pix_per_slice, partial_w, partial_b); break;
    switch (nb) {
        SSAC_FIRST_WGRAD(1) SSAC_FIRST_WGRAD(2) SSAC_FIRST_WGRAD(3) SSAC_FIRST_WGRAD(4)
        SSAC_FIRST_WGRAD(5) SSAC_FIRST_WGRAD(6) SSAC_FIRST_WGRAD(7) SSAC_FIRST_WGRAD(8)
        default: return ssac_fail("ssac_conv_first_wgrad: patch too large");
    }
#undef SSAC_FIRST_WGRAD
    return ssac_check_launch("conv_first_wgrad");
}
