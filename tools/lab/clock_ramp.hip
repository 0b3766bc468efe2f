// Is the slower first ~8 ms after an idle period (bench.py --steps 20: 56 us per update against 52 steady) the core clock ramping?
// s_memtime counts CORE clocks, s_memrealtime a constant 100 MHz: their ratio inside a kernel is the core frequency while it ran.
//   burst: after `idle_ms` of host sleep, `n` launches back to back of a kernel with a FIXED amount of dependent FMA work on
//   every CU (256 workgroups x 256 threads); per launch: core MHz (wave 0 of workgroup 0), duration in us of constant time.
//   hipcc --offload-arch=gfx950 -O2 tools/lab/clock_ramp.hip -o tools/clock_ramp.bin
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdio>
#include <vector>

__global__ void work(long long *t, int iters, float *sink) {
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) {   // dependent chain: 8 FMAs per trip
        a = a * b + 0.5f; a = a * b + 0.25f; a = a * b + 0.125f; a = a * b + 0.0625f;
        a = a * b - 0.5f; a = a * b - 0.25f; a = a * b - 0.125f; a = a * b - 0.0625f;
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; t[2] = r0; }
    if (a == 12345.678f) sink[0] = a;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// the same with the matrix pipe busy on every SIMD (2 waves per SIMD issuing dependent 32x32x2 fp32 MFMAs): the update's load
__global__ __launch_bounds__(512) void work_mfma(long long *t, int iters, float *sink) {
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc, 0, 0, 0);
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; t[2] = r0; }
    if (acc[0] == 12345.678f) sink[0] = acc[0];
}

int main(int argc, char **argv) {
    const bool mfma = argc > 1;
    const int n = 400, iters = 1500;
    long long *t; float *sink;
    hipMalloc(&t, (size_t)n * 3 * sizeof(long long)); hipMalloc(&sink, 4);
    hipStream_t st; hipStreamCreate(&st);
    std::vector<long long> h(n * 3);
    for (int idle_ms : {0, 5, 100, 1000}) {
        for (int rep = 0; rep < 2; ++rep) {
            // keep the device busy first (a previous burst), then idle, then the measured burst
            for (int i = 0; i < 200; ++i) { if (mfma) work_mfma<<<256, 512, 0, st>>>(t, 300, sink); else work<<<256, 256, 0, st>>>(t, iters, sink); }
            hipStreamSynchronize(st);
            if (idle_ms) usleep(idle_ms * 1000);
            for (int i = 0; i < n; ++i) { if (mfma) work_mfma<<<256, 512, 0, st>>>(t + 3 * i, 300, sink); else work<<<256, 256, 0, st>>>(t + 3 * i, iters, sink); }
            hipStreamSynchronize(st);
            hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
            printf("%s idle %4d ms:", mfma ? "mfma" : "valu", idle_ms);
            for (int i : {0, 1, 2, 5, 10, 20, 40, 80, 160, 320, 399}) {
                const double mhz = 100.0 * (double)h[3 * i] / (double)h[3 * i + 1];
                printf("  [%3d] %4.0f MHz %5.1f us @%6.2f ms", i, mhz, h[3 * i + 1] / 100.0, (h[3 * i + 2] - h[2]) / 1e5);
            }
            printf("\n");
        }
    }
    return 0;
}
