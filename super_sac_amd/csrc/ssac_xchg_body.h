// The one-shot exchange's device side (protocol: header of ssac_xchg.hip), shared by its stand-alone kernel (ssac_xchg.hip) and
// by the chained launch that runs it in a tail workgroup (ssac_fused.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int X_SLOTS = 4;
constexpr int X_MAX_WORLD = 8;
constexpr int X_THREADS = 256;
constexpr long long X_SPIN_LIMIT = 20000000000LL;   // shader clocks (~10 s): a missing peer raises an error, never a hang
// round 3's protocol / ack-less senders for the failing-first evidence (ssac_xchg_test_mode): LAB build only -- the product
// kernel does not read the field
#ifdef SSAC_LAB
#define X_TEST_MODE(a) ((a).test_mode)
#else
#define X_TEST_MODE(a) 0
#endif

struct XchgArgs {
    float *peer[X_MAX_WORLD];   // every rank's receive buffer, as mapped into THIS process (peer[rank] = own buffer)
    int rank, world, n, slot_floats, op;   // op 0 = MIN, 1 = SUM
    float *data;                // in: this rank's partial (n floats); out: the reduction over ranks
    unsigned long long *seq;    // device-resident exchange counter
    const int32_t *owners;      // != null: OWNERS-ONLY exchange (below), n_slots entries
    int n_slots;
    int *error;                 // HOST-pinned int (device view), set to 1 when a peer's flag did not arrive in time
    int *dead;                  // device int: once a spin gave up, later exchanges fail at once instead of spinning again
    unsigned *arrive;           // device counter behind `dead`: the workgroups whose output the in-launch form of the exchange
                                // waits for (fused_chain_pc_kernel's target-critic workgroups) bump it, the exchange zeroes it
    int test_mode;              // LAB build only (ssac_xchg_test_mode): bit 0 = senders skip step 0, bit 1 = accept flag >= seq,
                                // bit 2 = the in-launch form treats its wait for the launch's target-critic workgroups as timed out
    long long spin_limit;       // shader clocks a wait may take (X_SPIN_LIMIT; longer when the ranks time-slice ONE device)
    int n_parts, part_stride;   // > 1: element i of the payload is the SUM of n_parts partials (column-split target critics,
                                // ssac_td_spec.n_parts): data[(slot n_parts + s) part_stride + b]; the reduction lands in
                                // part 0 of every slot and the other parts are zeroed, so the sum stays the value
};

__device__ __forceinline__ float *slot_of(float *base, int src, int slot, int slot_floats) {
    return base + ((int64_t)src * X_SLOTS + slot) * (int64_t)(slot_floats + 4);
}

// "rank src has consumed every exchange up to this number": one 8-byte word per source rank, 16 bytes apart, behind
// the world x X_SLOTS slots of a receive buffer
__device__ __forceinline__ unsigned long long *ack_of(float *base, int world, int src, int slot_floats) {
    return reinterpret_cast<unsigned long long *>(slot_of(base, world, 0, slot_floats) + 4 * src);
}

// OWNERS-ONLY form (SURVEY 8(e) "Collective -- critic step": with n = 2 of N >= 10 subset members and 8 ranks, most ranks
// contribute +inf -- only the members' owners need to SEND): `owners` is the update's id block as every rank composes
// it from the same subset draw -- entry j >= 0: a member this rank owns, entry j = -(r + 1): rank r owns it.  Ranks that
// own no member write nothing; every rank waits for the owners' flags only and reduces over the owners' payloads (the
// others' would be +inf throughout).  `senders` = bit mask of owner ranks, the same on every rank.
// The exchange as a DEVICE function of one whole workgroup of NT threads (round 5): `xchg_kernel` is it as a launch of its own;
// the chained launch of a sharded rank runs it in a tail workgroup (fused_chain_pc_kernel, ssac_fused.hip) once the launch's
// target-critic workgroups have arrived.  DATA_AGENT: the payload in `data` was produced by OTHER workgroups of the same
// launch (agent-scope stores): read it with agent-scope loads, not through this CU's non-coherent caches.
// owners_now: the update's id block where the CALLER found it (the in-launch form reads it from the input slot, as the
// launch's other workgroups do); null = a.owners.
// force_fail: see below.
// lds16: 16 bytes of the CALLER's LDS (8-byte aligned) -- the body keeps no static LDS of its own: a kernel that asks for the
// CU's whole 160 KB as dynamic LDS (the chained launch) could not carry even 16 static bytes.
template <int NT, bool DATA_AGENT>
__device__ __forceinline__ void xchg_body(const XchgArgs &a, void *lds16, const int32_t *owners_now = nullptr,
                                          bool force_fail = false) {
    const int32_t *owners = owners_now ? owners_now : a.owners;
    unsigned long long &s_seq = *reinterpret_cast<unsigned long long *>(lds16);
    int &s_ok = *reinterpret_cast<int *>(reinterpret_cast<char *>(lds16) + 8);
    const int tid = threadIdx.x;
    // force_fail: the caller already knows the payload is not whole (the in-launch form's wait for its producers gave up):
    // take the !s_ok path from the start -- nothing is sent or polled, the result is poisoned, the error word raised
    if (tid == 0) { s_seq = *a.seq + 1; s_ok = force_fail ? 0 : 1; }
    __syncthreads();
    const unsigned long long seq = s_seq;
    const int slot = (int)(seq % X_SLOTS);
    unsigned senders = (1u << a.world) - 1u;
    if (owners) {
        senders = 0u;
        for (int j = 0; j < a.n_slots; ++j) {
            const int v = owners[j];
            senders |= 1u << (v >= 0 ? a.rank : -v - 1);
        }
    }
    const bool i_send = (senders >> a.rank) & 1u;
    // payload element i <-> where it lives in `data` (plain: data[i]; partial sums: slot i / stride, row i % stride)
    const int np = a.n_parts > 1 ? a.n_parts : 1;
    auto mine = [&](int i) {
        auto ld = [&](int64_t o) { return DATA_AGENT ? __hip_atomic_load(a.data + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.data[o]; };
        if (np == 1) return ld(i);
        const int j = i / a.part_stride, b = i - j * a.part_stride;
        float v = ld((int64_t)(j * np) * a.part_stride + b);
        for (int s_ = 1; s_ < np; ++s_) v += ld((int64_t)(j * np + s_) * a.part_stride + b);
        return v;
    };
    auto put = [&](int i, float v) {
        if (np == 1) { a.data[i] = v; return; }
        const int j = i / a.part_stride, b = i - j * a.part_stride;
        a.data[(int64_t)(j * np) * a.part_stride + b] = v;
        for (int s_ = 1; s_ < np; ++s_) a.data[(int64_t)(j * np + s_) * a.part_stride + b] = 0.0f;
    };
    // ---- 0. slot reuse: every rank must have consumed exchange seq - X_SLOTS before its slot is written again
    if (i_send && seq > (unsigned long long)X_SLOTS && !(X_TEST_MODE(a) & 1)) {
        if (tid < a.world) {
            const unsigned long long *ack = ack_of(a.peer[a.rank], a.world, tid, a.slot_floats);
            const long long t0 = __builtin_amdgcn_s_memtime();
            const long long limit = *a.dead ? 0 : a.spin_limit;
            while (__hip_atomic_load(ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + X_SLOTS < seq) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memtime() - t0 > limit) { s_ok = 0; break; }
            }
        }
        __syncthreads();
    }
    // (a reuse wait that gave up: nothing is written over the unread slot, nothing is polled, the result is poisoned)
    const bool go = s_ok != 0;
    // ---- 1. my partial -> every rank's recv[my rank][slot]
    for (int p = 0; p < (i_send && go ? a.world : 0); ++p) {
        float *dst = slot_of(a.peer[p], a.rank, slot, a.slot_floats);
        for (int i = tid; i < a.n; i += NT)
            __hip_atomic_store(dst + i, mine(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: payload before flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < a.world && i_send && go) {
        float *dst = slot_of(a.peer[tid], a.rank, slot, a.slot_floats);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst + a.slot_floats), seq, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- 2. wait for every sender's flag in MY buffer
    if (tid < a.world && ((senders >> tid) & 1u) && go) {
        const unsigned long long *flag =
            reinterpret_cast<const unsigned long long *>(slot_of(a.peer[a.rank], tid, slot, a.slot_floats) + a.slot_floats);
        const long long t0 = __builtin_amdgcn_s_memtime();
        const long long limit = *a.dead ? 0 : a.spin_limit;
        for (;;) {
            const unsigned long long f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (f == seq) break;
            if (f > seq) {   // the slot was LAPPED: it holds a later exchange's payload
                if (!(X_TEST_MODE(a) & 2)) s_ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memtime() - t0 > limit) { s_ok = 0; break; }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    // ---- 3. reduce over ranks in rank order (system-scope loads: the payload may have come from a peer device)
    if (s_ok) {
        for (int i = tid; i < a.n; i += NT) {
            bool first = true;
            float r = a.op == 0 ? __builtin_inff() : 0.0f;
            for (int p = 0; p < a.world; ++p) {
                if (!((senders >> p) & 1u)) continue;
                const float v = __hip_atomic_load(slot_of(a.peer[a.rank], p, slot, a.slot_floats) + i, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_SYSTEM);
                r = first ? v : (a.op == 0 ? fminf(r, v) : r + v);
                first = false;
            }
            put(i, r);
        }
    } else {
        // no reduction happened: poison the result (slots this rank does not own still hold +inf, which a TD target
        // would silently absorb) and tell the host -- the error word is pinned host memory, read without a device
        // synchronisation at the training loop's periodic slot-reuse wait (learning.py) and raised there
        for (int i = tid; i < a.n; i += NT) put(i, __builtin_nanf(""));
        if (tid == 0) {
            *a.dead = 1;
            __hip_atomic_store(a.error, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // ---- 4. consumed: every rank may reuse the slot (the payload loads above have returned -- their values were stored)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < a.world)
        __hip_atomic_store(ack_of(a.peer[tid], a.world, a.rank, a.slot_floats), seq, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid == 0) *a.seq = seq;
}

// host side (ssac_xchg.hip): fill the kernel-side view of an exchange over `data` (op 0 = MIN, 1 = SUM); non-zero + message on error
struct ssac_xchg;
int ssac_xchg_fill_args(ssac_xchg *x, float *data, int n, int op, const int32_t *owners, int n_slots, int n_parts, XchgArgs *out);
