// One-shot exchange between the ranks of a critic-sharded update (SURVEY.md 8(e) "Transport").
//
// The messages of the sharded update are tiny -- the (n_subset x B) partial min-Q of the critic step (4 KiB at the metric
// shape), the (B,) min-Q and the (B x A) action gradient of the actor step -- so a ring all-reduce would be pure hop
// latency.  Instead every rank owns a receive buffer in its HBM, exported over HIP IPC and mapped by every peer:
//
//   recv[src rank][slot][payload floats | flag]            (slot = sequence number mod N_SLOTS)
//
// and ONE kernel per exchange (a recordable launch: it sits inside the update's launch list, between the launch that
// produces the partial and the launch that consumes the result, with no host involvement):
//   1. writes this rank's partial into recv[rank][slot] of EVERY rank (its own included) with system-scope stores,
//      then, after a system-scope release, the slot's flag = sequence number;
//   2. polls the `world` flags of its OWN buffer (relaxed system-scope loads, bounded spin) until they carry the number;
//   3. reduces the `world` payloads (MIN or SUM, fixed rank order -> identical bits on every rank) in place of the input.
// The sequence number lives in device memory and is advanced by the kernel itself, so a replayed launch list needs no
// per-update argument.
//
// Flow control (round 4).  When EVERY rank sends, a rank can run at most one exchange ahead of the slowest peer (it needs
// that peer's flag of the current exchange).  The owners-only form breaks that bound: a rank that owns no subset member
// for several updates in a row writes no flag, so nothing held the senders back and they could lap it -- overwrite slot
// seq % 4 with exchange seq + 4 before the slow rank had read exchange seq, whose poll (flag >= seq) then accepted the
// later update's payload.  So every rank, sender or not, ACKNOWLEDGES what it has consumed:
//
//   ack[src rank]                                          (behind the slots of every receive buffer; 8-byte words)
//
//   4. after its reduction a rank stores ack[rank] = seq into EVERY rank's buffer (off the critical path: the result is
//      already in place);
//   0. before a sender writes slot seq % X_SLOTS it waits until every peer's ack in its OWN buffer has reached
//      seq - X_SLOTS, i.e. until everybody has consumed the exchange that used the slot last (normally true for a long
//      time: a local load per peer, no added latency).
// A sender can therefore run at most X_SLOTS exchanges ahead of the slowest rank, and a receiver accepts a flag only when
// it EQUALS the sequence number: a larger one means the slot was lapped after all -- the result is poisoned and the error
// word set (the same path as a peer that never arrives), never reduced silently.
// Over xGMI the writes are posted peer-to-peer stores; on a single device (two ranks sharing one GPU in the tests) the
// very same code runs through the local HBM.  torch.distributed (RCCL) remains the fallback path (parallel.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "ssac_internal.h"

namespace {

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
    int test_mode;              // LAB build only (ssac_xchg_test_mode): bit 0 = senders skip step 0, bit 1 = accept flag >= seq
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
__global__ __launch_bounds__(X_THREADS) void xchg_kernel(XchgArgs a) {
    __shared__ unsigned long long s_seq;
    __shared__ int s_ok;
    const int tid = threadIdx.x;
    if (tid == 0) { s_seq = *a.seq + 1; s_ok = 1; }
    __syncthreads();
    const unsigned long long seq = s_seq;
    const int slot = (int)(seq % X_SLOTS);
    unsigned senders = (1u << a.world) - 1u;
    if (a.owners) {
        senders = 0u;
        for (int j = 0; j < a.n_slots; ++j) {
            const int v = a.owners[j];
            senders |= 1u << (v >= 0 ? a.rank : -v - 1);
        }
    }
    const bool i_send = (senders >> a.rank) & 1u;
    // payload element i <-> where it lives in `data` (plain: data[i]; partial sums: slot i / stride, row i % stride)
    const int np = a.n_parts > 1 ? a.n_parts : 1;
    auto mine = [&](int i) {
        if (np == 1) return a.data[i];
        const int j = i / a.part_stride, b = i - j * a.part_stride;
        float v = a.data[(int64_t)(j * np) * a.part_stride + b];
        for (int s_ = 1; s_ < np; ++s_) v += a.data[(int64_t)(j * np + s_) * a.part_stride + b];
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
        for (int i = tid; i < a.n; i += X_THREADS)
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
        for (int i = tid; i < a.n; i += X_THREADS) {
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
        for (int i = tid; i < a.n; i += X_THREADS) put(i, __builtin_nanf(""));
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

}  // namespace

struct ssac_xchg {
    int rank, world, slot_floats;
    float *local;                       // own receive buffer
    std::vector<float *> peers;         // mapped views, peers[rank] == local
    std::vector<bool> opened;
    unsigned long long *seq;
    int *dead;                          // device int behind seq
    int *error_host, *error_dev;        // pinned host word and its device view
    int test_mode = 0;
    // ranks that share ONE device (allow_cached: the one-GPU test box) run their exchange kernels in turn -- a spinning
    // kernel is preempted only by the scheduler's quantum, and a wait can sit out several of them per peer: the bound grows
    // with the number of ranks there (three ranks once sat out 10 s on a loaded box; eight did in round 3).  One rank per
    // GPU -- the real layout -- keeps ~10 s.
    long long spin_limit = X_SPIN_LIMIT;
};

extern "C" ssac_xchg *ssac_xchg_create(int rank, int world, int slot_floats, int allow_cached) {
    if (rank < 0 || world < 1 || world > X_MAX_WORLD || rank >= world || slot_floats <= 0) {
        ssac_fail("ssac_xchg_create: bad arguments");
        return nullptr;
    }
    slot_floats = (slot_floats + 3) & ~3;   // 16-byte slots: the 8-byte flag behind the payload stays aligned
    ssac_xchg *x = new ssac_xchg();
    x->rank = rank; x->world = world; x->slot_floats = slot_floats;
    if (allow_cached && world > 1) x->spin_limit = X_SPIN_LIMIT * (long long)(2 * world);
    const size_t bytes = sizeof(float) * ((size_t)world * X_SLOTS * (slot_floats + 4) + 4 * (size_t)world);   // slots | acks
    // The receive buffer is written by PEER devices while this device polls it: uncached (fine-grained) device memory,
    // so that no stale line of it can sit in this device's L2 (what RCCL does for its flags and LL buffers).  Ordinary
    // (cached) device memory is accepted only when the caller says every rank shares ONE device (allow_cached: the
    // test box) -- there a single L2 serves all of them and the system-scope accesses of the kernel are coherent by
    // construction.  Across devices a cached buffer could serve the poll or the payload loads from a stale L2 line
    // (the loads are sc0 sc1, but whether a peer's xGMI write invalidates the owner's L2 copy of a coarse-grained line
    // is not something one probe exchange can prove): refused, the set-up vote then puts every rank on the collective.
    if (hipExtMallocWithFlags((void **)&x->local, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        x->local = nullptr;
        if (!allow_cached) {
            ssac_fail("ssac_xchg_create: no uncached device memory for the receive buffer (and the ranks do not share "
                      "one device): use the collective");
            delete x;
            return nullptr;
        }
    }
    if ((!x->local && hipMalloc((void **)&x->local, bytes) != hipSuccess) || hipMemset(x->local, 0, bytes) != hipSuccess ||
        hipMalloc((void **)&x->seq, 16) != hipSuccess || hipMemset(x->seq, 0, 16) != hipSuccess) {
        ssac_fail("ssac_xchg_create: allocation failed");
        delete x;
        return nullptr;
    }
    x->dead = reinterpret_cast<int *>(x->seq + 1);
    if (hipHostMalloc((void **)&x->error_host, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&x->error_dev, x->error_host, 0) != hipSuccess) {
        ssac_fail("ssac_xchg_create: cannot allocate the error word");
        delete x;
        return nullptr;
    }
    *x->error_host = 0;
    x->peers.assign(world, nullptr);
    x->opened.assign(world, false);
    x->peers[rank] = x->local;
    (void)hipDeviceSynchronize();
    return x;
}

extern "C" int ssac_xchg_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

extern "C" int ssac_xchg_handle(ssac_xchg *x, void *handle_out) {
    if (!x || !handle_out) return ssac_fail("ssac_xchg_handle: null argument");
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, x->local) != hipSuccess) return ssac_check_launch("ssac_xchg_handle: hipIpcGetMemHandle");
    memcpy(handle_out, &h, sizeof(h));
    return 0;
}

// handles: world x ssac_xchg_handle_bytes(), rank-major (as gathered from every rank)
extern "C" int ssac_xchg_connect(ssac_xchg *x, const void *handles) {
    if (!x || !handles) return ssac_fail("ssac_xchg_connect: null argument");
    for (int p = 0; p < x->world; ++p) {
        if (p == x->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + (size_t)p * sizeof(h), sizeof(h));
        void *ptr = nullptr;
        if (hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !ptr)
            return ssac_check_launch("ssac_xchg_connect: hipIpcOpenMemHandle");
        x->peers[p] = (float *)ptr;
        x->opened[p] = true;
    }
    return 0;
}

static int xchg_launch(ssac_xchg *x, float *data, int n, int op, const int32_t *owners, int n_slots, void *stream,
                       int n_parts = 1);

// in place over data[0 .. n): MIN (op 0) or SUM (op 1) over the ranks.  A recordable launch.
extern "C" int ssac_xchg_reduce(ssac_xchg *x, float *data, int n, int op, void *stream) {
    return xchg_launch(x, data, n, op, nullptr, 0, stream);
}

// MIN over the ranks of data[0 .. n) where only the OWNERS of the update's subset members send (see xchg_kernel):
// owners = the update's id block in device memory (n_slots int32: >= 0 a member of this rank, -(r + 1) rank r's).
extern "C" int ssac_xchg_reduce_owned(ssac_xchg *x, float *data, int n, const int32_t *owners, int n_slots, int n_parts,
                                      void *stream) {
    if (!owners || n_slots <= 0 || n_slots > 64) return ssac_fail("ssac_xchg_reduce_owned: bad owner block");
    if (n_parts < 1 || n_parts > 8 || n % n_slots) return ssac_fail("ssac_xchg_reduce_owned: bad partial-sum layout");
    return xchg_launch(x, data, n, 0, owners, n_slots, stream, n_parts);
}

static int xchg_launch(ssac_xchg *x, float *data, int n, int op, const int32_t *owners, int n_slots, void *stream,
                       int n_parts) {
    if (!x || !data || n <= 0 || n > x->slot_floats || (op != 0 && op != 1))
        return ssac_fail("ssac_xchg_reduce: bad arguments");
    XchgArgs a{};
    a.owners = owners; a.n_slots = n_slots;
    for (int p = 0; p < x->world; ++p) {
        if (!x->peers[p]) return ssac_fail("ssac_xchg_reduce: not connected");
        a.peer[p] = x->peers[p];
    }
    a.rank = x->rank; a.world = x->world; a.n = n; a.slot_floats = x->slot_floats; a.op = op;
    a.data = data; a.seq = x->seq; a.error = x->error_dev; a.dead = x->dead;
    a.test_mode = x->test_mode;
    a.spin_limit = x->spin_limit;
    a.n_parts = n_parts; a.part_stride = n_slots > 0 ? n / n_slots : n;
    SSAC_LAUNCH(xchg_kernel, dim3(1), dim3(X_THREADS), 0, (hipStream_t)stream, a);
    return ssac_check_launch("xchg");
}

// Tests only.  Bit 0: this rank's later exchanges skip the slot-reuse wait (step 0 of xchg_kernel); bit 1: its receivers
// accept flag >= seq.  Mode 3 is the protocol of round 3; tests/test_hip_sharded.py uses 3 to show the hazard (a delayed
// non-owner silently reduces a later update's payload) and 1 to show that the lap DETECTION (flag > seq) fires.
extern "C" int ssac_xchg_test_mode(ssac_xchg *x, int mode) {
#ifndef SSAC_LAB
    // (the product library's exchange cannot be put back on round 3's unsafe protocol: the kernel does not read the field)
    if (mode != 0) return ssac_fail("ssac_xchg_test_mode: " SSAC_LAB_REFUSAL);
#endif
    if (!x || mode < 0 || mode > 3) return ssac_fail("ssac_xchg_test_mode: bad argument");
    x->test_mode = mode;
    return 0;
}

// 1 when a peer's flag failed to arrive within the spin bound since the last call (cleared by this read).  A plain
// host load of a pinned word: no device synchronisation, cheap enough for the training loop's periodic check; a caller
// that wants the verdict of a particular exchange synchronises the stream first.
extern "C" int ssac_xchg_error(ssac_xchg *x) {
    if (!x) return 1;
    return __atomic_exchange_n(x->error_host, 0, __ATOMIC_ACQ_REL) ? 1 : 0;
}

extern "C" void ssac_xchg_destroy(ssac_xchg *x) {
    if (!x) return;
    for (int p = 0; p < x->world; ++p)
        if (x->opened[p]) (void)hipIpcCloseMemHandle(x->peers[p]);
    (void)hipFree(x->local);
    (void)hipFree(x->seq);
    (void)hipHostFree(x->error_host);
    delete x;
}
