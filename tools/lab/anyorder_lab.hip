// Can a kernel start while its predecessor ON THE SAME STREAM is still running (hipExtLaunchKernel + hipExtAnyOrderLaunch: the AQL
// packet goes out without the barrier bit)?  hip_ext.h says "not supported on AMD GFX9xx boards" for one of the entry points;
// this measures it on gfx950.
//   A: `na` workgroups that spin `spin_us` each; B: `nb` workgroups that stamp their start.  s_memrealtime (100 MHz).
//   hipcc --offload-arch=gfx950 -O2 tools/lab/anyorder_lab.hip -o tools/anyorder_lab.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void kern_a(long long *t, int spin_ticks) {
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
}
__global__ void kern_b(long long *t, const int *flag_wait) {
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
}

int main() {
    const int na = 200, nb = 200, reps = 6;
    long long *ta, *tb;
    hipMalloc(&ta, 2 * na * sizeof(long long)); hipMalloc(&tb, 2 * nb * sizeof(long long));
    hipStream_t st; hipStreamCreate(&st);
    std::vector<long long> ha(2 * na), hb(2 * nb);
    for (int mode = 0; mode < 2; ++mode)
        for (int spin_us : {5, 20})
            for (int r = 0; r < reps; ++r) {
                int ticks = spin_us * 100; const int *nul = nullptr;
                void *aa[] = {&ta, &ticks}; void *ab[] = {&tb, &nul};
                hipLaunchKernel((const void *)kern_a, dim3(na), dim3(512), aa, 60 * 1024, st);   // (60 KB of LDS each)
                hipError_t e = mode ? hipExtLaunchKernel((const void *)kern_b, dim3(nb), dim3(512), ab, 60 * 1024, st, nullptr, nullptr, hipExtAnyOrderLaunch)
                                    : hipLaunchKernel((const void *)kern_b, dim3(nb), dim3(512), ab, 60 * 1024, st);
                hipStreamSynchronize(st);
                hipMemcpy(ha.data(), ta, ha.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), tb, hb.size() * 8, hipMemcpyDeviceToHost);
                long long a_end_max = 0, a_end_min = 1LL << 62, a_start_min = 1LL << 62, b_start_min = 1LL << 62, b_start_max = 0;
                for (int i = 0; i < na; ++i) { a_start_min = std::min(a_start_min, ha[2 * i]); a_end_max = std::max(a_end_max, ha[2 * i + 1]); a_end_min = std::min(a_end_min, ha[2 * i + 1]); }
                for (int i = 0; i < nb; ++i) { b_start_min = std::min(b_start_min, hb[2 * i]); b_start_max = std::max(b_start_max, hb[2 * i]); }
                if (r >= 2)
                    printf("%s spin %2d us (err %d): A runs %.2f us; first B start - last A end = %+.2f us; last B start - last A end = %+.2f us; first B start - first A end = %+.2f us\n",
                           mode ? "any-order" : "in-order ", spin_us, (int)e, (a_end_max - a_start_min) / 100.0, (b_start_min - a_end_max) / 100.0,
                           (b_start_max - a_end_max) / 100.0, (b_start_min - a_end_min) / 100.0);
            }
    return 0;
}
