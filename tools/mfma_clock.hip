// micro-benchmark: dependent-chain fp32 MFMA rate and s_memtime clock, for 1..N workgroups
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, int iters, long long* cyc) {
    f32x16 acc; for (int i=0;i<16;++i) acc[i]=0.f;
    float a = threadIdx.x*1e-3f, b = 1.0f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i=0;i<iters;++i) {
#pragma unroll
        for (int t=0;t<16;++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a,b,acc,0,0,0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s=0; for (int i=0;i<16;++i) s+=acc[i];
    out[blockIdx.x*blockDim.x+threadIdx.x]=s;
    if (threadIdx.x==0 && blockIdx.x==0) *cyc = t1-t0;
}
int main(){
    float* out; long long* cyc; hipMalloc(&out, 4*1024*1024); hipMalloc(&cyc, 8);
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {1, 16, 160, 256, 512}) for (int threads : {256, 512}) {
        int iters = 512;   // 8192 MFMAs per wave
        k<<<wgs, threads>>>(out, iters, cyc); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r=0;r<20;++r) k<<<wgs, threads>>>(out, iters, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        double us = ms*1e3/20; double mf = 8192.0;
        printf("wgs %4d thr %4d: %.2f us/launch, memtime ticks %lld -> %.1f ticks/MFMA/wave, %.1f ns/MFMA/wave, %.1f TFLOP/s\n", wgs, threads, us, c, (double)c/mf, us*1e3/mf, (double)wgs*(threads/64)*mf*4096.0/(us*1e-6)/1e12);
    }
    return 0;
}
