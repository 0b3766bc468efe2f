// K-loop laboratory (measurement tool, not product): the weight-gradient tile loop of csrc/ssac_gemm.hip reduced to its
// skeleton -- 64x64 tile, K = batch, KS K-split groups of 4 waves, operands (K x 64) row-major staged through LDS,
// v_mfma_f32_32x32x2_f32 -- with switches that take one ingredient out at a time, to see where the clocks between the
// 16.4 k MFMA floor and the measured 26 k go.     hipcc --offload-arch=gfx950 -O3 -o tools/kloop_lab.bin tools/lab/kloop_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int BK = 32, TILE = 64 * 33;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MODE bits: 1 = no barrier in the loop, 2 = no staging stores / global loads in the loop, 4 = no fragment reads,
//            8 = staging work BEHIND the first-half MFMAs (the order before round 3), 16 = no MFMA,
//            32 = pair-interleaved LDS layout [k/2][row][2] with ds_read_b64 fragment reads (a lane takes k = 4u+2lh,
//                 4u+2lh+1 for two MFMA steps), thread loads two CONSECUTIVE k rows and stores two b128
template <int KS, int MODE>
__global__ __launch_bounds__(256 * KS) void kloop(const float *__restrict__ A, const float *__restrict__ B, float *C,
                                                  int K, int ld, long long *dbg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid_all = threadIdx.x, kg = tid_all >> 8, tid = tid_all & 255;
    float *buf0 = lds + kg * 4 * TILE, *buf1 = buf0 + 2 * TILE;
    const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const float *Ab = A + (size_t)blockIdx.x * K * ld, *Bb = B + (size_t)blockIdx.x * K * ld;
    const int r4 = (tid & 15) * 4, kk = tid >> 4;
    constexpr bool PAIR = (MODE & 32) != 0;
    auto slot = [](int r) { return r ^ (((r >> 4) & 1) << 1); };
    f4 va[2], vb[2];
    auto load = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = PAIR ? k0 + 2 * kk + q : k0 + kk + 16 * q;
            va[q] = *reinterpret_cast<const f4 *>(Ab + (size_t)k * ld + r4);
            vb[q] = *reinterpret_cast<const f4 *>(Bb + (size_t)k * ld + r4);
        }
    };
    auto store = [&](float *d) {
        if (PAIR) {
            float *pa = d + (kk * 64 + slot(r4)) * 2, *pa2 = d + (kk * 64 + slot(r4 + 2)) * 2;
            *reinterpret_cast<f4 *>(pa) = (f4){va[0][0], va[1][0], va[0][1], va[1][1]};
            *reinterpret_cast<f4 *>(pa2) = (f4){va[0][2], va[1][2], va[0][3], va[1][3]};
            *reinterpret_cast<f4 *>(pa + TILE) = (f4){vb[0][0], vb[1][0], vb[0][1], vb[1][1]};
            *reinterpret_cast<f4 *>(pa2 + TILE) = (f4){vb[0][2], vb[1][2], vb[0][3], vb[1][3]};
            return;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *reinterpret_cast<f4 *>(d + (kk + 16 * q) * 64 + r4) = va[q];
            *reinterpret_cast<f4 *>(d + TILE + (kk + 16 * q) * 64 + r4) = vb[q];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    constexpr int HT = 8;
    float f0a[HT], f0b[HT], f1a[HT], f1b[HT];
    auto rd = [&](float (&fa)[HT], float (&fb)[HT], const float *buf, int half) {
        if (MODE & 4) return;
        if (PAIR) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            const int sa = slot(wm * 32 + li), sb = slot(wn * 32 + li);
#pragma unroll
            for (int u = 0; u < HT / 2; ++u) {
                const int p = 2 * (half * (HT / 2) + u) + lh;
                const f2 a = *reinterpret_cast<const f2 *>(buf + (p * 64 + sa) * 2);
                const f2 b = *reinterpret_cast<const f2 *>(buf + TILE + (p * 64 + sb) * 2);
                fa[2 * u] = a[0]; fa[2 * u + 1] = a[1]; fb[2 * u] = b[0]; fb[2 * u + 1] = b[1];
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            fa[t] = buf[(2 * (half * HT + t) + lh) * 64 + wm * 32 + li];
            fb[t] = buf[TILE + (2 * (half * HT + t) + lh) * 64 + wn * 32 + li];
        }
    };
    auto mm = [&](const float (&fa)[HT], const float (&fb)[HT]) {
        if (MODE & 16) {
#pragma unroll
            for (int t = 0; t < HT; ++t) acc[t] += fa[t] * fb[t];
            return;
        }
#pragma unroll
        for (int t = 0; t < HT; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc, 0, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < HT; ++t) { f0a[t] = f0b[t] = f1a[t] = f1b[t] = 1.0f + t + lane; }
    const int iters = K / BK / KS;
    const bool stamp = dbg && blockIdx.x == 0 && lane == 0;
    long long t0 = 0;
    load(kg * BK);
    store(buf0);
    if (iters > 1) load((KS + kg) * BK);
    lds_barrier();
    rd(f0a, f0b, buf0, 0);
    if (MODE & 128) {
        // DIRECT-TO-LDS staging (global_load_lds_dwordx4): no staging registers, no ds_write.  One wave-instruction lands
        // 4 k-rows x 64 floats (1 KiB) contiguously -- the [k][row] layout as it is.  Chunk c+2 is requested right
        // behind the barrier of iteration c (its buffer's last readers passed that barrier) and must have landed by the
        // barrier of iteration c+1: a whole iteration of slack.
        typedef __attribute__((address_space(3))) void lds_void;
        typedef const __attribute__((address_space(1))) void glb_void;
        const int wv = tid >> 6;   // wave inside the K-group: k-rows 8 wv .. 8 wv + 7 of a chunk (2 instructions per operand)
        auto dma = [&](float *d, int k0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int krow = 8 * wv + 4 * q;   // first of the 4 k-rows this instruction lands
                const int k = k0 + krow + (lane >> 4);
                const float *ga = Ab + (size_t)k * ld + (lane & 15) * 4, *gb = Bb + (size_t)k * ld + (lane & 15) * 4;
                __builtin_amdgcn_global_load_lds((glb_void *)ga, (lds_void *)(d + krow * 64), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_void *)gb, (lds_void *)(d + TILE + krow * 64), 16, 0, 0);
            }
        };
        __syncthreads();   // (the prologue above used the buffers)
        dma(buf0, kg * BK);
        if (iters > 1) dma(buf1, (KS + kg) * BK);
        if (iters > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        rd(f0a, f0b, buf0, 0);
        if (stamp) t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            float *cur = (it & 1) ? buf1 : buf0, *nxt = (it & 1) ? buf0 : buf1;
            rd(f1a, f1b, cur, 1);
            mm(f0a, f0b);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk it+1 has landed (this wave's part)
            lds_barrier();
            if (it + 2 < iters) dma(cur, ((it + 2) * KS + kg) * BK);
            if (it + 1 < iters) rd(f0a, f0b, nxt, 0);
            mm(f1a, f1b);
        }
    } else
    if (stamp) t0 = __builtin_amdgcn_s_memtime();
    if (MODE & 128) {
    } else if (MODE & 64) {
        // STAGGERED K-groups: every half chunk ends with a workgroup barrier; the staging work (X: waits for the loads
        // of an iteration ago, LDS stores, next loads) sits IN FRONT of the MFMAs of the half that carries it, and even
        // groups carry it in the first half of an iteration, odd groups in the second -- so on every SIMD (one wave
        // of each group) two waves open a half with MFMAs while the other two do their staging.
        const bool odd = kg & 1;
        for (int it = 0; it < iters; ++it) {
            float *cur = (it & 1) ? buf1 : buf0, *nxt = (it & 1) ? buf0 : buf1;
            // ---- half A: MFMAs of the first half of chunk `it`; fragments of its second half are read here
            rd(f1a, f1b, cur, 1);
            if (!odd) {
                if (it + 1 < iters) store(nxt);
                if (it + 2 < iters) load(((it + 2) * KS + kg) * BK);
                __builtin_amdgcn_sched_barrier(0);
            }
            mm(f0a, f0b);
            lds_barrier();
            // ---- half B: MFMAs of the second half; odd groups stage the next chunk in front of them
            if (odd) {
                if (it + 1 < iters) store(nxt);
                if (it + 2 < iters) load(((it + 2) * KS + kg) * BK);
                __builtin_amdgcn_sched_barrier(0);
                mm(f1a, f1b);
                lds_barrier();
                if (it + 1 < iters) rd(f0a, f0b, nxt, 0);
            } else {
                if (it + 1 < iters) rd(f0a, f0b, nxt, 0);
                mm(f1a, f1b);
                lds_barrier();
            }
        }
    } else
    for (int it = 0; it < iters; ++it) {
        float *cur = (it & 1) ? buf1 : buf0, *nxt = (it & 1) ? buf0 : buf1;
        rd(f1a, f1b, cur, 1);
        if (!(MODE & 8) && !(MODE & 2)) {
            if (it + 1 < iters) store(nxt);
            if (it + 2 < iters) load(((it + 2) * KS + kg) * BK);
            __builtin_amdgcn_sched_barrier(0);
        }
        mm(f0a, f0b);
        if ((MODE & 8) && !(MODE & 2)) {
            if (it + 1 < iters) store(nxt);
            if (it + 2 < iters) load(((it + 2) * KS + kg) * BK);
        }
        if (!(MODE & 1)) lds_barrier();
        if (it + 1 < iters) rd(f0a, f0b, nxt, 0);
        mm(f1a, f1b);
    }
    if (stamp) dbg[tid_all >> 6] = __builtin_amdgcn_s_memtime() - t0;
    float *c = C + (size_t)blockIdx.x * 4096 * KS + kg * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + wn * 32 + li] = acc[r];
}

template <int KS, int MODE>
void run(const char *name, const float *A, const float *B, float *C, int K, int ld, long long *dbg, int grid) {
    const size_t ldsb = sizeof(float) * KS * 4 * TILE;
    hipFuncSetAttribute((const void *)kloop<KS, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) kloop<KS, MODE><<<grid, 256 * KS, ldsb>>>(A, B, C, K, ld, nullptr);
    hipEventRecord(e0);
    for (int i = 0; i < 100; ++i) kloop<KS, MODE><<<grid, 256 * KS, ldsb>>>(A, B, C, K, ld, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemset(dbg, 0, 64 * 8);
    kloop<KS, MODE><<<grid, 256 * KS, ldsb>>>(A, B, C, K, ld, dbg);
    long long h[64];
    hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    long long mx = 0, mn = 1LL << 60;
    for (int w = 0; w < 4 * KS; ++w) { if (h[w] > mx) mx = h[w]; if (h[w] < mn) mn = h[w]; }
    printf("%-58s KS %d: %7.2f us/launch   loop clocks per wave min %lld max %lld\n", name, KS, ms * 10.f, mn, mx);
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 512, grid = argc > 2 ? atoi(argv[2]) : 200, ld = 256;
    float *A, *B, *C; long long *dbg;
    hipMalloc(&A, (size_t)grid * K * ld * 4); hipMalloc(&B, (size_t)grid * K * ld * 4);
    hipMalloc(&C, (size_t)grid * 4096 * 4 * 4); hipMalloc(&dbg, 64 * 8);
    hipMemset(A, 0, (size_t)grid * K * ld * 4); hipMemset(B, 0, (size_t)grid * K * ld * 4);
    printf("K %d, %d workgroups; MFMA floor %d clocks per SIMD\n", K, grid, K / 2 * 64 * 4 / 4 * 4 / 4);
    run<4, 0>("staging in front of first-half MFMAs (round 3)", A, B, C, K, ld, dbg, grid);
    run<4, 8>("staging behind first-half MFMAs (round 2)", A, B, C, K, ld, dbg, grid);
    run<4, 8 | 32>("round 2 order, pair-interleaved LDS + ds_read_b64", A, B, C, K, ld, dbg, grid);
    run<4, 32>("round 3 order, pair-interleaved LDS + ds_read_b64", A, B, C, K, ld, dbg, grid);
    run<4, 2 | 32>("pair-interleaved, no staging", A, B, C, K, ld, dbg, grid);
    run<4, 16 | 32>("pair-interleaved, everything but the MFMAs", A, B, C, K, ld, dbg, grid);
    run<2, 8 | 32>("8 waves, round 2 order, pair-interleaved", A, B, C, K, ld, dbg, grid);
    run<4, 64>("staggered K-groups (2 barriers per chunk)", A, B, C, K, ld, dbg, grid);
    run<4, 64 | 32>("staggered K-groups, pair-interleaved LDS", A, B, C, K, ld, dbg, grid);
    run<4, 128>("direct-to-LDS staging (global_load_lds_dwordx4)", A, B, C, K, ld, dbg, grid);
    run<2, 128>("direct-to-LDS staging, 8 waves", A, B, C, K, ld, dbg, grid);
    run<4, 1>("no loop barrier", A, B, C, K, ld, dbg, grid);
    run<4, 2>("no staging stores / global loads", A, B, C, K, ld, dbg, grid);
    run<4, 2 | 4>("no staging, no fragment reads (MFMA + barrier)", A, B, C, K, ld, dbg, grid);
    run<4, 1 | 2 | 4>("MFMA only", A, B, C, K, ld, dbg, grid);
    run<4, 16>("everything but the MFMAs (VALU fma instead)", A, B, C, K, ld, dbg, grid);
    run<2, 0>("8 waves (KS 2), staging in front", A, B, C, K, ld, dbg, grid);
    run<2, 1 | 2 | 4>("8 waves MFMA only", A, B, C, K, ld, dbg, grid);
    run<1, 0>("4 waves (KS 1), staging in front", A, B, C, K, ld, dbg, grid);
    run<1, 1 | 2 | 4>("4 waves MFMA only", A, B, C, K, ld, dbg, grid);
    return 0;
}
