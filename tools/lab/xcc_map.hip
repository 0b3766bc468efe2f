// Which XCD does workgroup b of a launch land on?  Prints XCC_ID (s_getreg HW_REG_XCC_ID) per blockIdx for launch
// shapes like the update's (225 x 512 threads with a big LDS carve, 241 x 1024).  hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int *out, long long *t) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * blockIdx.x] = (int)x;
        out[2 * blockIdx.x + 1] = (int)hw;
        t[blockIdx.x] = __builtin_amdgcn_s_memtime();
    }
    lds[threadIdx.x] = 1.0f;
    __syncthreads();
    // stay resident for a while so that the whole grid is co-resident
    long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
}
int main() {
    int *d; long long *t;
    hipMalloc(&d, 4096 * sizeof(int)); hipMalloc(&t, 2048 * sizeof(long long));
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int n = cfg == 0 ? 225 : (cfg == 1 ? 241 : 64), thr = cfg == 1 ? 1024 : 512;
        const size_t lds = cfg == 1 ? 140 * 1024 : 150 * 1024;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe, dim3(n), dim3(thr), lds, 0, d, t);
            hipDeviceSynchronize();
        }
        std::vector<int> h(2 * n);
        hipMemcpy(h.data(), d, 2 * n * sizeof(int), hipMemcpyDeviceToHost);
        printf("grid %d x %d threads: xcc id (low 4 bits) by blockIdx:\n", n, thr);
        int mism = 0;
        for (int b = 0; b < n; ++b) {
            printf("%d", h[2 * b] & 15);
            if ((h[2 * b] & 15) != (b & 7)) ++mism;
            if ((b & 63) == 63) printf("\n");
        }
        printf("\n  workgroups with xcc != blockIdx %% 8: %d of %d\n", mism, n);
    }
    return 0;
}
