// Start-of-update duties shared by ssac_begin_update, the replay gather and the merged actor / critic-forward
// launch: this update's input slot -> fixed device block, and the optimizer step advanced.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ssac_hip.h"

__device__ __forceinline__ void adam_refresh(ssac_adam_ctl *c, int t) {
    c->step = t;
    c->step_size = (float)(c->lr_d / (1.0 - pow(c->beta1_d, (double)t)));
    c->bc2_sqrt = (float)sqrt(1.0 - pow(c->beta2_d, (double)t));
}

// this update's slot of the input ring (ssac_feed)
__device__ __forceinline__ const uint32_t *feed_slot(const ssac_feed &f) {
    return f.host_ring + (int64_t)(f.tick % f.n_slots) * f.slot_words;
}

// slot -> fixed device block (read by the later launches of the update).  16 bytes per lane and every load issued
// before the first store (slot_words % 4 == 0 and 16-byte aligned slots are the host's contract).
__device__ __forceinline__ void feed_pull(const ssac_feed &f) {
    const uint4 *s4 = reinterpret_cast<const uint4 *>(feed_slot(f));
    uint4 *d4 = reinterpret_cast<uint4 *>(f.dst);
    const int n4 = f.slot_words >> 2;
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = threadIdx.x + u * blockDim.x;
        v[u] = s4[i < n4 ? i : 0];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = threadIdx.x + u * blockDim.x;
        if (i < n4) d4[i] = v[u];
    }
    for (int i = 4 * blockDim.x + threadIdx.x; i < n4; i += blockDim.x) d4[i] = s4[i];
}
