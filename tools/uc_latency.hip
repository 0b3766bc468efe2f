// Latency of one dependent load: ordinary device memory (L2 hit / miss) vs the UNCACHED allocation the input ring lives
// in.   hipcc --offload-arch=gfx950 -O2 tools/uc_latency.hip -o /tmp/uc_latency && /tmp/uc_latency
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(const int *p, long long *out, int reps) {
    long long tot = 0;
    int idx = 0;
    for (int r = 0; r < reps; ++r) {
        const long long t0 = __builtin_amdgcn_s_memtime();
        idx = __builtin_nontemporal_load(p + idx);   // dependent chain: the next address comes from this load
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tot += __builtin_amdgcn_s_memtime() - t0 + (idx & 0);
    }
    out[0] = tot / reps;
    out[1] = idx;
}
__global__ void probe_plain(const int *p, long long *out, int reps) {
    long long tot = 0;
    int idx = 0;
    for (int r = 0; r < reps; ++r) {
        const long long t0 = __builtin_amdgcn_s_memtime();
        idx = p[idx];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tot += __builtin_amdgcn_s_memtime() - t0 + (idx & 0);
    }
    out[0] = tot / reps;
    out[1] = idx;
}

int main() {
    const int n = 1 << 22;   // 16 MiB of ints, stride walk of 4099 elements: no two probes share a cache line
    int *host = (int *)malloc(n * sizeof(int));
    for (int i = 0; i < n; ++i) host[i] = (int)(((long long)i + 4099 * 16) % n);
    int *cached, *uc;
    long long *out;
    hipMalloc(&cached, n * sizeof(int));
    hipExtMallocWithFlags((void **)&uc, n * sizeof(int), hipDeviceMallocUncached);
    hipMalloc(&out, 16);
    hipMemcpy(cached, host, n * sizeof(int), hipMemcpyHostToDevice);
    hipMemcpy(uc, host, n * sizeof(int), hipMemcpyHostToDevice);
    long long h[2];
    const char *names[] = {"device memory, first touch (HBM)", "device memory, second pass (L2 / MALL)", "uncached allocation",
                           "uncached allocation, second pass"};
    for (int v = 0; v < 4; ++v) {
        hipLaunchKernelGGL(probe_plain, dim3(1), dim3(64), 0, 0, v < 2 ? cached : uc, out, 200);
        hipDeviceSynchronize();
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("%-44s %lld clocks per dependent load (s_memtime, 100 MHz ticks x ... see note)\n", names[v], h[0]);
    }
    return 0;
}
