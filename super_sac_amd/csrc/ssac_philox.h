// Philox4x32-10 + Box-Muller: the engine's counter-based normal stream (see ssac_rng in include/ssac_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct RngArgs { uint64_t seed; const int64_t *counter; int64_t offset; };

__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// standard normal for element (row, col) of draw `draw`: one Philox block per (row, col / 4, draw)
__device__ __forceinline__ float philox_normal(uint64_t seed, int64_t draw, int row, int col) {
    uint32_t c[4] = {(uint32_t)row, (uint32_t)(col >> 2), (uint32_t)draw, (uint32_t)((uint64_t)draw >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const int pair = (col >> 1) & 1;
    const float u1 = ((float)c[2 * pair] + 1.0f) * 2.3283064365386963e-10f;   // (0, 1]
    const float u2 = (float)c[2 * pair + 1] * 2.3283064365386963e-10f;        // [0, 1]
    const float r = sqrtf(-2.0f * logf(u1));
    const float th = 6.283185307179586f * u2;
    return (col & 1) ? r * sinf(th) : r * cosf(th);
}

__device__ __forceinline__ int64_t rng_draw(const RngArgs &r) { return r.offset + (r.counter ? *r.counter : 0); }
