// How fast does STRAIGHT-LINE code run on gfx950 when every instruction is executed once (the prologues, epilogues and
// phase boundaries of the fused kernels: thousands of instructions per workgroup, none of them in a loop)?
//   straight<KB>: KB kilobytes of independent 8-byte VALU instructions, executed once by one wave per workgroup
//   looped:       the same number of instructions as a 64-instruction loop body
// Timed per workgroup with s_memtime; launched three times back to back (is the instruction cache warm across launches of
// the same kernel?), then alternating with a second kernel (does another kernel's code evict it?).
//   hipcc --offload-arch=gfx950 -O2 tools/lab/icache_lab.hip -o tools/icache_lab.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define I1 "v_add_f32_e64 %0, %0, 1.0\n"   /* 8-byte VOP3 encoding, dependent chain of 4-cycle ops would hide fetch: use 4 chains */
#define Q4(a, b, c, d) "v_add_f32_e64 %0, %0, 1.0\nv_add_f32_e64 %1, %1, 1.0\nv_add_f32_e64 %2, %2, 1.0\nv_add_f32_e64 %3, %3, 1.0\n"
#define R4 Q4(0, 1, 2, 3)
#define R16 R4 R4 R4 R4
#define R64 R16 R16 R16 R16
#define R256 R64 R64 R64 R64
#define R1024 R256 R256 R256 R256       /* 1024 instructions = 8 KB */

template <int KB8>   // KB8 blocks of 8 KB
__global__ void straight(float *out, long long *t) {
    float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < KB8; ++i) asm volatile(R1024 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}

__global__ void looped(float *out, long long *t, int iters) {
    float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) asm volatile(R64 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}

static void report(const char *name, long long *dt, int n, int instrs) {
    std::vector<long long> h(n);
    hipMemcpy(h.data(), dt, n * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-44s min %7lld  median %7lld  max %7lld clocks  (%.2f clk / instruction at the median)\n", name, h[0], h[n / 2],
           h[n - 1], (double)h[n / 2] / instrs);
}

int main() {
    float *out; long long *t;
    const int n = 256;
    hipMalloc(&out, n * 64 * sizeof(float)); hipMalloc(&t, n * sizeof(long long));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(looped, dim3(n), dim3(64), 0, 0, out, t, 64); hipDeviceSynchronize();
        report("looped, 4096 instructions (64 x 64)", t, n, 4096);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(straight<1>, dim3(n), dim3(64), 0, 0, out, t); hipDeviceSynchronize();
        report("straight 8 KB (1024 instr), back to back", t, n, 1024);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(straight<4>, dim3(n), dim3(64), 0, 0, out, t); hipDeviceSynchronize();
        report("straight 32 KB (4096 instr), back to back", t, n, 4096);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(straight<12>, dim3(n), dim3(64), 0, 0, out, t); hipDeviceSynchronize();
        report("straight 96 KB (12288 instr), back to back", t, n, 12288);
    }
    for (int rep = 0; rep < 3; ++rep) {   // alternating with another kernel of 96 KB
        hipLaunchKernelGGL(straight<12>, dim3(n), dim3(64), 0, 0, out, t); hipDeviceSynchronize();
        hipLaunchKernelGGL(straight<4>, dim3(n), dim3(64), 0, 0, out, t); hipDeviceSynchronize();
        report("straight 32 KB after a 96 KB kernel", t, n, 4096);
    }
    // 8 waves per workgroup all running the same straight code (a 512-thread workgroup's prologue)
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(straight<4>, dim3(n), dim3(512), 0, 0, out, t); hipDeviceSynchronize();
        report("straight 32 KB, 8 waves per workgroup", t, n, 4096);
    }
    return 0;
}
