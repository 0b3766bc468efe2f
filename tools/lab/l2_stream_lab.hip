// How fast can ONE CU stream an L2-resident operand when every CU of the chip does the same?  (the K loops of the
// update's kernels stream 256-512 KB of weights / activations per workgroup: is 12-13 B/clk/CU a hardware bound?)
// Each workgroup reads `kb` KB (its net's slab: net = wg / 16, 2.6 MB in all) with DEPTH 16-byte loads in flight per thread.
// hipcc --offload-arch=gfx950 -O3 tools/lab/l2_stream_lab.hip -o tools/l2_stream_lab.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int DEPTH, int STRIDED>
__global__ void stream(const float *__restrict__ W, int kb, int slab_kb, float *out, long long *clk) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const float *base = W + (size_t)(blockIdx.x / 16 % 10) * slab_kb * 256;
    const int n4 = kb * 64;               // 16-byte elements to read
    f4 acc = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    // STRIDED = 0: a wave instruction reads 1 KB contiguous; 1: lane l reads 16 B at row (l/2) stride 1 KB (+16 B * (l&1)) like KcDirect
    for (int i = tid; i < n4; i += nthr * DEPTH) {
        f4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            int e = i + d * nthr;
            if (STRIDED) {   // permute within a 64 KB window: element index -> (row-major 64 rows x 1 KB) transposed walk
                const int w = e >> 12, r = e & 4095;            // 4096 f4 = 64 KB window
                const int lane2 = r & 1, row = (r >> 1) & 63, col = r >> 7;   // col in 0..31 (x2 lanes x 16 B = 1 KB per row)
                e = (w << 12) + row * 64 + col * 2 + lane2;
            }
            v[d] = *reinterpret_cast<const f4 *>(base + 4 * (size_t)(e % (slab_kb * 64)));
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.0f;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}
template <int DEPTH, int STRIDED>
void run(const float *W, float *out, long long *clk, int wgs, int thr, int kb) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream<DEPTH, STRIDED>), dim3(wgs), dim3(thr), 0, 0, W, kb, 256, out, clk);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((stream<DEPTH, STRIDED>), dim3(wgs), dim3(thr), 0, 0, W, kb, 256, out, clk);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(wgs);
    hipMemcpy(h.data(), clk, wgs * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[wgs / 2];   // s_memtime ticks at 100 MHz: 1 tick = 10 ns
    printf("wgs %3d thr %4d depth %2d strided %d: %6.1f us/launch, median WG %7.0f ns -> %5.1f B/ns/CU (%.2f TB/s chip)\n", wgs, thr,
           DEPTH, STRIDED, ms * 100.0, med * 10.0, kb * 1024.0 / (med * 10.0), wgs * kb * 1024.0 / (ms * 1e-4) / 1e12 * 1e-0 / 1.0);
}
int main() {
    float *W, *out; long long *clk;
    hipMalloc(&W, 10 * 256 * 1024 + 4096); hipMalloc(&out, 64); hipMalloc(&clk, 4096 * 8);
    hipMemset(W, 0, 10 * 256 * 1024);
    for (int wgs : {224, 32}) {
        for (int thr : {512, 1024}) {
            run<2, 0>(W, out, clk, wgs, thr, 512);
            run<4, 0>(W, out, clk, wgs, thr, 512);
            run<8, 0>(W, out, clk, wgs, thr, 512);
            run<16, 0>(W, out, clk, wgs, thr, 512);
            run<4, 1>(W, out, clk, wgs, thr, 512);
            run<8, 1>(W, out, clk, wgs, thr, 512);
        }
    }
    return 0;
}
