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

#include "ssac_xchg_body.h"

namespace {

__global__ __launch_bounds__(X_THREADS) void xchg_kernel(XchgArgs a) {
    __shared__ unsigned long long scratch[2];
    xchg_body<X_THREADS, false>(a, scratch);
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

// the kernel-side view of an exchange (ssac_xchg_body.h); also used by ssac_chain_update, whose tail workgroup runs the
// exchange of a sharded rank's target Q inside the chained launch
int ssac_xchg_fill_args(ssac_xchg *x, float *data, int n, int op, const int32_t *owners, int n_slots, int n_parts,
                        XchgArgs *out) {
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
    a.arrive = reinterpret_cast<unsigned *>(x->dead + 1);   // (the fourth word of the 16-byte counter block)
    a.test_mode = x->test_mode;
    a.spin_limit = x->spin_limit;
    a.n_parts = n_parts; a.part_stride = n_slots > 0 ? n / n_slots : n;
    *out = a;
    return 0;
}

static int xchg_launch(ssac_xchg *x, float *data, int n, int op, const int32_t *owners, int n_slots, void *stream,
                       int n_parts) {
    XchgArgs a{};
    if (ssac_xchg_fill_args(x, data, n, op, owners, n_slots, n_parts, &a)) return 1;
    SSAC_LAUNCH(xchg_kernel, dim3(1), dim3(X_THREADS), 0, (hipStream_t)stream, a);
    return ssac_check_launch("xchg");
}

// Tests only.  Bit 0: this rank's later exchanges skip the slot-reuse wait (step 0 of xchg_kernel); bit 1: its receivers
// accept flag >= seq.  Mode 3 is the protocol of round 3; tests/test_hip_sharded.py uses 3 to show the hazard (a delayed
// non-owner silently reduces a later update's payload) and 1 to show that the lap DETECTION (flag > seq) fires.
// (LAB build only -- ssac_hip_test.h: the product library does not define the symbol, and its kernel does not read the field:
//  a production exchange cannot be put back on round 3's unsafe protocol)
#ifdef SSAC_LAB
extern "C" int ssac_xchg_test_mode(ssac_xchg *x, int mode) {
    if (!x || mode < 0 || mode > 7) return ssac_fail("ssac_xchg_test_mode: bad argument");
    x->test_mode = mode;
    return 0;
}
#endif

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
