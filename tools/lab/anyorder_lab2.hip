// What does hipExtAnyOrderLaunch skip?  anyorder_lab.hip: the successor starts 0.96 us earlier, but never before the predecessor's
// last workgroup has ended.  Is the data of the predecessor VISIBLE to it (per-XCD L2s: a release at the end of a kernel writes
// dirty L2 lines back, an acquire at the start invalidates L1 / non-local L2 lines)?
//   A(it): every workgroup spins a little, then writes `it` over ITS 16 KB slice of X as its last act.
//   B(it): workgroup b reads the slice of workgroup (b + 131) % n (another XCD), counts words != it; the lines it read stay in
//          its L1 / L2 for the next iteration (the stale copies an acquire would drop).
// 4000 iterations per mode: mismatches must be 0 if the kernel boundary keeps its release / acquire.
//   hipcc --offload-arch=gfx950 -O2 tools/lab/anyorder_lab2.hip -o tools/anyorder_lab2.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

constexpr int SLICE = 4096;   // floats per workgroup

__global__ void kern_a(float *x, int it, int spin_ticks) {
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks + (blockIdx.x & 7) * 20) __builtin_amdgcn_s_sleep(4);
    float4 v = make_float4((float)it, (float)it, (float)it, (float)it);
    float4 *p = reinterpret_cast<float4 *>(x + (size_t)blockIdx.x * SLICE);
    for (int i = threadIdx.x; i < SLICE / 4; i += blockDim.x) p[i] = v;
}
__global__ void kern_b(const float *x, int it, int n, unsigned *bad, int fence) {
    if (fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const float4 *p = reinterpret_cast<const float4 *>(x + (size_t)((blockIdx.x + 131) % n) * SLICE);
    unsigned cnt = 0;
    for (int i = threadIdx.x; i < SLICE / 4; i += blockDim.x) {
        const float4 v = p[i];
        cnt += (v.x != (float)it) + (v.y != (float)it) + (v.z != (float)it) + (v.w != (float)it);
    }
    if (cnt) atomicAdd(bad, cnt);
}

int main() {
    const int n = 256, iters = 4000;
    float *x; unsigned *bad;
    hipMalloc(&x, (size_t)n * SLICE * 4); hipMalloc(&bad, 4);
    hipMemset(x, 0, (size_t)n * SLICE * 4);
    hipStream_t st; hipStreamCreate(&st);
    for (int mode = 0; mode < 3; ++mode) {   // 0 in-order, 1 any-order, 2 any-order + acquire fence in B
        hipMemset(bad, 0, 4);
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 1; it <= iters; ++it) {
            int spin = 300, fence = mode == 2; const float *cx = x; int nn = n;
            void *aa[] = {&x, &it, &spin}; void *ab[] = {&cx, &it, &nn, &bad, &fence};
            hipLaunchKernel((const void *)kern_a, dim3(n), dim3(256), aa, 0, st);
            if (mode) hipExtLaunchKernel((const void *)kern_b, dim3(n), dim3(256), ab, 0, st, nullptr, nullptr, hipExtAnyOrderLaunch);
            else hipLaunchKernel((const void *)kern_b, dim3(n), dim3(256), ab, 0, st);
        }
        hipStreamSynchronize(st);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
        unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("%s: %u mismatching words in %d iterations (%.2f us per A+B pair)\n",
               mode == 0 ? "in-order            " : mode == 1 ? "any-order           " : "any-order + acquire ", h, iters, us);
    }
    // and the other boundary: A any-order behind B (A overwrites what B is reading?  B must have finished) -- WAR
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(bad, 0, 4); hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 1; it <= iters; ++it) {
            int spin = 300, fence = 0; const float *cx = x; int nn = n;
            void *aa[] = {&x, &it, &spin}; void *ab[] = {&cx, &it, &nn, &bad, &fence};
            if (mode) hipExtLaunchKernel((const void *)kern_a, dim3(n), dim3(256), aa, 0, st, nullptr, nullptr, hipExtAnyOrderLaunch);
            else hipLaunchKernel((const void *)kern_a, dim3(n), dim3(256), aa, 0, st);
            if (mode) hipExtLaunchKernel((const void *)kern_b, dim3(n), dim3(256), ab, 0, st, nullptr, nullptr, hipExtAnyOrderLaunch);
            else hipLaunchKernel((const void *)kern_b, dim3(n), dim3(256), ab, 0, st);
        }
        hipStreamSynchronize(st);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
        unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("both launches %s: %u mismatching words in %d iterations (%.2f us per A+B pair)\n", mode ? "any-order" : "in-order ", h, iters, us);
    }
    return 0;
}
