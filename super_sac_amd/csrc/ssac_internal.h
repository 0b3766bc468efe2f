// Internal helpers shared by the .hip translation units of libssac_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>
#include <tuple>
#include <vector>

#include "ssac_hip.h"

// record an error message (thread-local) and return a non-zero status
int ssac_fail(const char *msg);
// hipGetLastError() after a launch; non-zero + message on failure
int ssac_check_launch(const char *what);

// ---------------------------------------------------------------------------------------------
// Launch lists (ssac_record_begin / ssac_record_end / ssac_replay in include/ssac_hip.h).
// Every kernel of the library is launched through SSAC_LAUNCH.  While a recording is open, the launch is
// ALSO appended -- kernel address, geometry and a byte copy of its arguments -- to a list, which a later
// ssac_replay() re-issues from one C loop: an update whose inputs live at fixed device addresses costs one
// host call instead of one Python->C->HIP trip per kernel, without the ~13 us idle tail a hipGraph launch
// leaves on the queue on this platform.
// ---------------------------------------------------------------------------------------------
struct SsacLaunchRec {
    const void *func;
    dim3 grid, block;
    size_t lds;
    std::vector<char> blob;        // argument values, each at its natural alignment
    std::vector<size_t> offsets;   // byte offset of argument i in `blob`
    // byte offsets in `blob` of 8-byte pointers that a replay through ssac_step_run overwrites with the address of THIS
    // update's slot of the input ring (ssac_record_slot_patch below); a plain ssac_replay writes null there
    std::vector<size_t> slot_patches;
    // ... and of integers that a replay through ssac_replay_value overwrites with (value + addend): the per-update number
    // a launch derives its hand-off tag and its noise draw from, when the update has no device-resident counter
    // (ssac_record_value_patch below).  kind 0: uint32, (value + addend) & 0x7fffffff; kind 1: int64 value + addend;
    // kind 2: a POINTER, addend + value2 * stride -- the replay's SECOND number (ssac_replay_value2) picks a slot of a ring
    // (the online actor update's log block goes straight to its slot of the log ring: no copy behind the replay)
    struct ValuePatch { size_t off; int kind; long long addend; long long stride; };
    std::vector<ValuePatch> value_patches;
};

extern thread_local std::vector<SsacLaunchRec> *g_ssac_recording;

template <typename T>
inline void ssac_pack_arg(SsacLaunchRec &r, const T &v) {
    size_t off = (r.blob.size() + alignof(T) - 1) / alignof(T) * alignof(T);
    r.blob.resize(off + sizeof(T));
    std::memcpy(r.blob.data() + off, &v, sizeof(T));
    r.offsets.push_back(off);
}

template <typename... KArgs, typename... Args>
inline void ssac_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st,
                        Args... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count mismatch");
    std::tuple<KArgs...> vals{static_cast<KArgs>(args)...};
    void *argv[sizeof...(KArgs) > 0 ? sizeof...(KArgs) : 1];
    size_t i = 0;
    std::apply([&](auto &...v) { ((argv[i++] = (void *)&v), ...); }, vals);
    if (g_ssac_recording) {
        SsacLaunchRec r;
        r.func = (const void *)kernel;
        r.grid = grid; r.block = block; r.lds = lds;
        std::apply([&](auto &...v) { (ssac_pack_arg(r, v), ...); }, vals);
        g_ssac_recording->push_back(std::move(r));
    }
    (void)hipLaunchKernel((const void *)kernel, grid, block, argv, lds, st);
}

#define SSAC_LAUNCH(kernel, grid, block, lds, st, ...) ssac_launch(kernel, grid, block, lds, st, __VA_ARGS__)

// "This update's slot, by value".  A recorded launch finds its inputs through device memory: kernel arguments -> the
// ssac_feed block (its update counter) -> the slot of the input ring (replay indices, subset ids) -> the replay rows: three
// DEPENDENT loads from cold memory at the front of every workgroup.  The host that replays the list through
// ssac_step_run knows the slot -- it has just written it -- so the recorded argument bytes carry a pointer member that
// the replay fills in: the feed block leaves the address chain of every workgroup (one of them still reads it for the
// start-of-update duties).  Call right behind the SSAC_LAUNCH of a kernel whose argument `arg_index` is a struct with such
// a member at byte `member_off`; the member is null in the launch that is being recorded and in plain replays
// (ssac_replay, hipGraph captures), where the kernels take the path through the feed block.
inline void ssac_record_slot_patch(int arg_index, size_t member_off) {
    if (!g_ssac_recording || g_ssac_recording->empty()) return;
    SsacLaunchRec &r = g_ssac_recording->back();
    r.slot_patches.push_back(r.offsets[(size_t)arg_index] + member_off);
}
inline void ssac_record_value_patch(int arg_index, size_t member_off, int kind, long long addend, long long stride = 0) {
    if (!g_ssac_recording || g_ssac_recording->empty()) return;
    SsacLaunchRec &r = g_ssac_recording->back();
    r.value_patches.push_back(SsacLaunchRec::ValuePatch{r.offsets[(size_t)arg_index] + member_off, kind, addend, stride});
}

// ---------------------------------------------------------------------------------------------
// XCD-contiguous workgroup order.  The dispatcher is observed to place workgroup b on XCD b % 8 (a speed
// assumption only -- nothing here depends on it for correctness), and each XCD has its own 4 MiB L2.  The
// kernels whose neighbouring tiles share operands (the row tiles of one net share its weights; the tiles of
// one net's weight gradient share its activations) therefore take their tile from this LOGICAL id: XCD x
// works on one contiguous range of logical ids, so a net's operands are pulled into one or two L2s instead
// of all eight.  Bijective for any number of workgroups.  g_ssac_xcd (ssac_xcd_order) = 0 keeps the
// hardware order.
// ---------------------------------------------------------------------------------------------
extern int g_ssac_xcd;
extern long long *g_ssac_timeline;   // ssac_debug_timeline (ssac_elementwise.hip)
// Measurement scaffolding -- s_memtime phase stamps of one workgroup, per-workgroup (start, end) timelines, builds in which a
// workgroup class returns at once -- is compiled only into the LAB build (`./build.sh --lab`, -DSSAC_LAB): the product
// library carries none of it, and does not define the ssac_*_debug_stamps / ssac_debug_timeline / ssac_xchg_test_mode entry
// points at all (include/ssac_hip_test.h, "lab hooks").
#ifdef SSAC_LAB
#define SSAC_LAB_ONLY(...) __VA_ARGS__
#else
#define SSAC_LAB_ONLY(...)
#endif
#ifdef __HIPCC__
// min over the REDQ subset slots of the target Q of row b (ssac_td_spec): a slot's value is q_t[j][b], or -- n_parts > 1,
// column-split target critics -- the sum of its partials q_t[(j n_parts + s)][b] in index order.  Every reader of
// ssac_td_spec::q_t goes through this (n_parts <= 1: the plain min, operation by operation as before).
__device__ __forceinline__ float ssac_td_min_q(const ssac_td_spec &t, int b, int n_rows) {
    const int np = t.n_parts > 1 ? t.n_parts : 1;
    float mq = 0.0f;
    for (int j = 0; j < t.n_sel; ++j) {
        float v = t.q_t[(int64_t)(j * np) * n_rows + b];
        for (int s = 1; s < np; ++s) v += t.q_t[(int64_t)(j * np + s) * n_rows + b];
        mq = j == 0 ? v : fminf(mq, v);
    }
    return mq;
}

// Hand-off of the sampled action a' from the ACTOR workgroup of a 16-row tile to the tile's TARGET-CRITIC workgroups
// (fused_chain_pc_kernel): one 8-byte granule {a' bits, tag} per (row, action dimension), written with ONE write-through
// (agent-scope) store and polled with agent-scope loads -- no fence on either side (a granule is complete or absent).  tag =
// base + the update counter of a recorded launch list (ssac_feed.tick) or the host's launch counter (eager launches):
// distinct for every launch, so a stale granule of an earlier update is never taken for this update's.
struct Handoff {
    unsigned long long *pub;    // [n_rows][A]; null = no hand-off
    const long long *tick;      // device-resident update counter, or null
    unsigned base;
    int S, A;                   // the consumer's input columns [S, S + A) arrive through pub
    int nsplit;                 // consumers per (slot, tile): > 1 = each takes hidden / nsplit columns of fc2 (below)
    // the chained ACTOR update (fused_actor_chain_kernel): the critics' results travel back the same way --
    unsigned long long *qpub;   // [n_critics][n_rows]: Q_j(s, a_theta)           (MODE_CRITIC_U publishes, MODE_ACTOR_BWD polls)
    unsigned long long *dxpub;  // [n_critics][n_rows][A]: unscaled dQ_j / da
};

constexpr long long HANDOFF_SPIN_LIMIT = 4000000000LL;   // shader clocks (~2 s): a producer that never arrives poisons, never hangs
// one granule: spin until its tag is this launch's (bounded: a producer that never arrives yields NaN, never a hang)
__device__ __forceinline__ float handoff_poll(const unsigned long long *gp, unsigned tag) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long w;
    for (;;) {
        w = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(w >> 32) == tag) break;
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memtime() - t0 > HANDOFF_SPIN_LIMIT) { w = 0x7fc00000ull; break; }   // (NaN: never silently stale)
    }
    return __uint_as_float((unsigned)w);
}
__device__ __forceinline__ void handoff_publish(unsigned long long *gp, unsigned tag, float v) {
    __hip_atomic_store(gp, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// ---- DrQv2 random shift (augmentations.py:214-269): the sampling position of output index i along one axis.
// torch.linspace(start, end, steps) element i in fp32 (two-sided evaluation: each element is one fused multiply-add in
// ATen -- CPU build and nvcc contract it alike); then grid_sample's un-normalisation (align_corners=False)
// ((g + 1) * size - 1) / 2 in ATen's separate fp32 ops.  Shared by the shift kernels (ssac_elementwise.hip) and the
// first convolution that applies the shift in its operand staging (ssac_conv_implicit.hip): the same bits in both.
__device__ __forceinline__ float linspace_f32(float start, float end, float step, int steps, int i) {
    return (i < steps / 2) ? __fmaf_rn(step, (float)i, start) : __fmaf_rn(-step, (float)(steps - 1 - i), end);
}
struct ShiftAxis {
    float w0, w1;   // bilinear weights of the taps p0 and p0 + 1
    int p0;         // first tap, in PADDED coordinates (valid inside [0, hp); source index = clamp(p - pad, 0, h - 1))
};
struct ShiftGrid {  // the constants of one padded size hp (two double divisions: evaluate once per kernel, not per position)
    float start, end, step, sscale;
    int hp;
};
__device__ __forceinline__ ShiftGrid drqv2_shift_grid(int hp) {
#pragma clang fp contract(off)
    ShiftGrid g;
    g.start = (float)(-1.0 + 1.0 / (double)hp); g.end = (float)(1.0 - 1.0 / (double)hp);
    g.step = __fdiv_rn(__fsub_rn(g.end, g.start), (float)(hp - 1));
    g.sscale = (float)(2.0 / (double)hp);
    g.hp = hp;
    return g;
}
__device__ __forceinline__ ShiftAxis drqv2_shift_axis(int i, int64_t shift, const ShiftGrid &g) {
#pragma clang fp contract(off)
    const float gpos = __fadd_rn(linspace_f32(g.start, g.end, g.step, g.hp, i), __fmul_rn((float)shift, g.sscale));
    const float ip = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gpos, 1.0f), (float)g.hp), 1.0f), 2.0f);
    const float fl = floorf(ip);
    ShiftAxis a;
    a.w1 = __fsub_rn(ip, fl);
    a.w0 = __fsub_rn(1.0f, a.w1);
    a.p0 = (int)fl;
    return a;
}
__device__ __forceinline__ ShiftAxis drqv2_shift_axis(int i, int64_t shift, int hp) {
    return drqv2_shift_axis(i, shift, drqv2_shift_grid(hp));
}

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic, NOT for its global loads / stores, and
// is opaque to the compiler's fence lowering.  With this toolchain (ROCm 7.2, gfx950, probed in round 6) __syncthreads()
// emits the same two instructions -- s_waitcnt lgkmcnt(0); s_barrier, NO vmcnt(0): a workgroup-scope release does not
// drain vector-memory stores on a CU in non-tgsplit mode -- so neither barrier makes one wave's GLOBAL stores visible
// beyond the CU.  A cross-workgroup hand-off therefore needs `s_waitcnt vmcnt(0)` in EVERY storing wave in front of the
// barrier that precedes the arrival (consumer_arrive in ssac_fused.hip, xchg_body, the weight-gradient epilogue).
// The waitcnt pass still inserts the vmcnt wait in front of the first USE of a loaded register.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int ssac_xcd_contiguous(int bid, int nwg, int on) {
    if (!on) return bid;
    const int xcd = bid & 7, slot = bid >> 3, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// The same order for a sub-range [t0, n) of a launch's workgroup ids (the halves of a merged launch): logical id
// (0-based within the range) of workgroup `bid`, such that the range's workgroups on XCD x -- those with
// bid % 8 == x -- hold one contiguous run of logical ids.
__device__ __forceinline__ int ssac_xcd_contiguous_range(int bid, int t0, int n, int on) {
    if (!on) return bid - t0;
    const int x = bid & 7;
    int before = 0;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const int upto_n = n > y ? (n - y + 7) >> 3 : 0, upto_t0 = t0 > y ? (t0 - y + 7) >> 3 : 0;
        if (y < x) before += upto_n - upto_t0;
    }
    const int first_slot = t0 > x ? (t0 - x + 7) >> 3 : 0;   // ids below t0 on this XCD
    return before + (bid >> 3) - first_slot;
}
#endif
