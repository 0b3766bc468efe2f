// micro-benchmark: what an MFMA K-loop loses to its operand traffic.  One workgroup (8 waves) per CU.
//   variant 0: 16 dependent 32x32x2 fp32 MFMAs per chunk, operands in registers
//   variant 1: + the A fragment of every chunk read from LDS (4 x ds_read_b128 per wave)
//   variant 2: + the B fragment of every chunk loaded from global memory (4 x dwordx4 per wave), 2 chunks ahead
//   variant 3: variant 2 with A double-buffered (next chunk's A issued before this chunk's MFMAs)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int LDA = 260;
template <int V, int ROT>
__global__ __launch_bounds__(512) void k(float *out, const float *W, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float As[];  // [32][LDA]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 32 * LDA; i += 512) As[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int li = lane & 31, lh = lane >> 5;
    const float *wp = W + (size_t)(wave * 32 + li) * 256 + lh * 16;
    const int rot = ROT == 0 ? 0 : (ROT == 1 ? wave : (ROT == 2 ? (int)blockIdx.x : (int)blockIdx.x + wave));  // start chunk
    f4 a[2][4], b[3][4];
    for (int q = 0; q < 4; ++q) { a[0][q] = (f4){1.f, 1.f, 1.f, 1.f}; a[1][q] = a[0][q]; b[0][q] = a[0][q]; b[1][q] = a[0][q]; b[2][q] = a[0][q]; }
    if (V >= 2) { for (int q = 0; q < 4; ++q) { b[0][q] = *(const f4 *)(wp + ((rot) & 7) * 32 + 4 * q); b[1][q] = *(const f4 *)(wp + ((rot + 1) & 7) * 32 + 4 * q); } }
    if (V == 3) { const f4 *ap = (const f4 *)(As + li * LDA + lh * 16); for (int q = 0; q < 4; ++q) a[0][q] = ap[q]; }
#pragma unroll 1
    for (int c0 = 0; c0 < chunks; c0 += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int c = (c0 + u + rot) & 7;  // 8 chunks of K = 256, cycled
            if (V >= 2) { const int cn = (c + 2) & 7; for (int q = 0; q < 4; ++q) b[(u + 2) % 3][q] = *(const f4 *)(wp + cn * 32 + 4 * q); }
            if (V == 1 || V == 2) { const f4 *ap = (const f4 *)(As + li * LDA + c * 32 + lh * 16); for (int q = 0; q < 4; ++q) a[0][q] = ap[q]; }
            if (V == 3) { const int cn = (c + 1) & 7; const f4 *ap = (const f4 *)(As + li * LDA + cn * 32 + lh * 16); for (int q = 0; q < 4; ++q) a[(u + 1) & 1][q] = ap[q]; }
            const int ai = V == 3 ? (u & 1) : 0;
#pragma unroll
            for (int t = 0; t < 16; ++t)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ai][t >> 2][t & 3], b[u % 3][t >> 2][t & 3], acc, 0, 0, 0);
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 512 + tid] = s;
}
template <int V, int ROT> void run(float *out, float *W, int wgs) {
    const int chunks = 6 * 200;
    const size_t lds = 32 * LDA * 4;
    k<V, ROT><<<wgs, 512, lds>>>(out, W, chunks); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); for (int r = 0; r < 10; ++r) k<V, ROT><<<wgs, 512, lds>>>(out, W, chunks); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 10, fl = (double)wgs * 8 * chunks * 16 * 4096.0;
    printf("rot %d variant %d, %d WGs x 8 waves: %.1f us, %.1f TFLOP/s (%.1f%% of 157.3), %.0f clk/chunk/SIMD at 2.38 GHz\n", ROT, V, wgs, us, fl / (us * 1e-6) / 1e12,
           100 * fl / (us * 1e-6) / 1e12 / 157.3, us * 1e-6 * 2.38e9 / chunks);
}
int main() {
    float *out, *W; hipMalloc(&out, 4 * 512 * 1024); hipMalloc(&W, 4 * 256 * 256 * 4); hipMemset(W, 0, 4 * 256 * 256 * 4);
    for (int wgs : {160, 256}) { run<2, 0>(out, W, wgs); run<2, 1>(out, W, wgs); run<2, 2>(out, W, wgs); run<2, 3>(out, W, wgs); }
    return 0;
}
