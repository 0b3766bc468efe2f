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
    // (native vector type and predicated stores of named values: with HIP's uint4 struct in an array the compiler
    // kept the staged values in scratch memory, which made every kernel that inlines this a scratch user)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *s4 = reinterpret_cast<const u32x4 *>(feed_slot(f));
    u32x4 *d4 = reinterpret_cast<u32x4 *>(f.dst);
    const int n4 = f.slot_words >> 2;
    const int t = threadIdx.x, nt = blockDim.x;
    const int i0 = t, i1 = t + nt, i2 = t + 2 * nt, i3 = t + 3 * nt;
    const u32x4 v0 = s4[i0 < n4 ? i0 : 0], v1 = s4[i1 < n4 ? i1 : 0], v2 = s4[i2 < n4 ? i2 : 0], v3 = s4[i3 < n4 ? i3 : 0];
    if (i0 < n4) d4[i0] = v0;
    if (i1 < n4) d4[i1] = v1;
    if (i2 < n4) d4[i2] = v2;
    if (i3 < n4) d4[i3] = v3;
    for (int i = 4 * nt + t; i < n4; i += nt) d4[i] = s4[i];
    // late-bound Polyak (include/ssac_hip.h): publish "begun", THEN look for this update's request -- one decider
    if (t == 0 && f.late_word) {
        uint32_t *tail = const_cast<uint32_t *>(f.host_ring) + (int64_t)f.n_slots * f.slot_words;
        __hip_atomic_store(reinterpret_cast<int64_t *>(tail), f.tick + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __atomic_thread_fence(__ATOMIC_SEQ_CST);   // (system scope: the store is out before the loads below are issued)
        const uint32_t *req = tail + 8 + 2 * (uint32_t)(f.tick % f.n_slots);
        const uint32_t tag = __hip_atomic_load(req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t bits = __hip_atomic_load(req + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t mine = (uint32_t)(f.tick & 0x7fffffff) + 1u;
        *f.late_word = tag == mine ? bits : 0u;
        if (tag == mine) __hip_atomic_store(tail + 4, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // "decided": a host that found the update already begun waits for this (microseconds), then reads last_served
        __hip_atomic_store(reinterpret_cast<int64_t *>(tail + 2), f.tick + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
