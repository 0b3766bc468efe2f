/*
 * ssac_hip.h -- C ABI of libssac_hip.so, the MI355X (gfx950) update-path kernels.
 *
 * The reference (jakegrigsby/super_sac) is pure Python: it has no FFI of its own, so
 * there is nothing to match symbol-for-symbol.  Each entry point below replaces the
 * arithmetic of one group of reference call sites (cited per function); the Python
 * host layer (super_sac_amd/) keeps the reference's function names and signatures
 * (learning.critic_update, ...) and calls these through ctypes.  INTEGRATION.md shows
 * the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless the
 *     parameter name ends in _host.  No torch types, no exceptions.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*).
 *   - return value: 0 on success, non-zero on error; ssac_last_error() gives the text.
 *   - all floating point data is fp32 (the reference computes in fp32 throughout).
 *   - weight tensors use the torch.nn.Linear layout: W is (out, in) row-major.
 *
 * Packed ensemble-MLP arena ("ssac_mlp"): `n_nets` identically shaped 3-layer ReLU MLPs
 * (reference nets/mlps.py:11-41, 78-93, 113-129, 132-149, 170-185), each stored as
 *   [ W1 (hidden x in) | b1 (hidden) | W2 (hidden x hidden) | b2 (hidden) |
 *     W3 (out x hidden) | b3 (out) | pad to a multiple of 4 floats ]
 * with `net_stride` floats between consecutive nets.  Adam moments and Polyak targets
 * use arenas of the same layout.
 */
#ifndef SSAC_HIP_H
#define SSAC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSAC_ABI_VERSION 7
#define SSAC_MAX_NETS 64

typedef struct ssac_mlp {
    float *params;      /* arena base */
    int64_t net_stride; /* floats between nets */
    int32_t n_nets;
    int32_t in_dim, hidden, out_dim;
} ssac_mlp;

/* Adam hyper-parameters + per-step scalars, resident in DEVICE memory so that a captured
 * hipGraph can be replayed while the step count advances (ssac_adam_advance). */
typedef struct ssac_adam_ctl {
    float lr, beta1, beta2, eps, weight_decay;
    float step_size;    /* lr / (1 - beta1^t)           (torch.optim.Adam, main.py:188-239) */
    float bc2_sqrt;     /* sqrt(1 - beta2^t) */
    float clip_coef;    /* clip_grad_norm_ coefficient, 1.0 when clipping is off */
    int32_t step;       /* t */
    int32_t _pad[3];
    double lr_d, beta1_d, beta2_d; /* the host's exact doubles, used for the bias corrections */
} ssac_adam_ctl;

/* PopArt layer state (popart.py:8-20), resident in device memory. */
typedef struct ssac_popart {
    float mu, nu, w, b;
    int32_t t, min_steps, stable, _pad;
    double beta;
} ssac_popart;

/* Per-update host inputs of a captured (hipGraph) update, fed WITHOUT a copy node: the host writes replay
 * indices / REDQ subset ids / the log-ring slot of update k into slot k % n_slots of a pinned host ring;
 * ssac_begin_update pulls that slot over PCIe into `dst` (fixed device address, read by the captured gather /
 * target-critic launches) and ssac_critic_logs, the last launch of the update, publishes the log block into
 * log_ring[dst_words[log_slot_word]] and advances `tick`.  Lives in DEVICE memory. */
typedef struct ssac_feed {
    const uint32_t *host_ring;  /* the input ring (ssac_feed_ring_alloc): n_slots x slot_words 4-byte words;
                                   16-byte aligned, slot_words % 4 == 0 */
    uint32_t *dst;              /* device copy of the current slot (slot_words words) */
    float *log_ring;            /* device ring of log blocks, log_width floats each */
    int64_t tick;               /* updates consumed so far */
    int32_t n_slots, slot_words, log_slot_word, log_width;
    uint32_t *late_word;        /* late-bound Polyak (ssac_late_polyak below): where the update's first launch leaves
                                   its decision (tau bits, 0 = no request); NULL = the ring has no tail / feature off */
} ssac_feed;

/* The input ring of ssac_feed.  On a large-BAR system (the MI355X boxes: all of HBM is CPU-mappable) it is UNCACHED
 * DEVICE memory the host stores into directly -- posted PCIe writes issued many updates ahead -- so the update's
 * first launch reads its slot from local HBM instead of paying a PCIe read round trip (~4 us) on the critical
 * path; otherwise pinned host memory the launch reads over PCIe.  *device_resident tells which.  The host writes a
 * slot with ssac_feed_write (copy + store fence: write-combining buffers are drained before the launch that
 * reads the slot is submitted). */
int ssac_feed_ring_alloc(size_t bytes, void **ring, int *device_resident); /* + SSAC_FEED_TAIL_BYTES behind, zeroed */
int ssac_feed_ring_free(void *ring, int device_resident);
int ssac_feed_write(void *ring_slot, const void *src, size_t bytes);
int ssac_feed_ring_mode(int device_ok); /* 0: always pinned host memory (default 1: device memory when large-BAR) */

/* The replay gather (replay.py:131-161: rows idx of s, a, r, s', d) folded into ssac_actor_sample_critic_fwd: each
 * workgroup fetches ITS tile's rows straight from the replay arrays -- the actor half s' (and writes [s'|.], r, d out
 * for the later launches), the critic half [s|a] (net slot 0 writes it out for the weight-gradient launch) -- so the
 * update has no gather launch.  float32 vector observations.  Host-side struct, copied at launch. */
typedef struct ssac_gather {
    const float *s, *s1;           /* replay observation arrays (rows x s_elems) */
    const float *act;              /* (rows x a_elems) */
    const float *rew;              /* (rows) */
    const uint8_t *done;           /* (rows) */
    int64_t s_elems, a_elems;
    const int64_t *idx;            /* (n_rows) replay row of every batch row, or NULL: the current slot of `feed` */
    const ssac_feed *feed;         /* recorded update: indices from the input ring; the launch then also does the
                                      work of ssac_begin_update (slot -> device block, logs cleared, step advanced) */
    float *xsa; int64_t ld_x;      /* out: [s | a] (n_rows x ld_x) */
    float *x1sa; int64_t ld_x1;    /* out: [s' | .] */
    float *rew_out, *done_out;     /* out: (n_rows) */
    float *logs; int32_t n_logs;   /* with feed: log block to clear */
    int32_t rng_word;              /* with feed: 4-byte word offset of the int64 noise draw number in the slot, -1 = none */
    ssac_adam_ctl *ctl;            /* with feed: optimizer control block to advance (may be NULL) */
    int32_t ids_word, _pad;        /* with feed (ssac_chain_update): word offset of the int32 REDQ subset ids in the slot
                                      (the target critics run in the launch that mirrors the slot: they read the ids
                                      from the slot itself), -1 = use net_ids */
} ssac_gather;

/* TD target evaluated INSIDE the critic launch instead of by ssac_td_target (continuous actions, no PopArt):
 * td[b] = rew[b] + gamma (1 - done[b]) (min_j q_t[j][b] - alpha logp[b])   (learning_utils.py:298-354),
 * the same arithmetic in the same order as ssac_td_target.  Host-side struct, copied at launch. */
typedef struct ssac_td_spec {
    const float *q_t;        /* (n_sel x n_rows) target-critic outputs of the REDQ subset */
    const float *logp;       /* (n_rows) log pi(a'|s'), read when use_entropy */
    const float *rew, *done; /* (n_rows) */
    const float *log_alpha;  /* scalar, read when use_entropy */
    float *td_out;           /* (n_rows) the targets, written by the launch (for logs / replay dicts) */
    float gamma;
    int32_t n_sel, use_entropy;
    int32_t n_parts;         /* 0 / 1: q_t is (n_sel x n_rows).  > 1: every slot's value arrives in n_parts partial sums,
                                q_t[(j n_parts + s)][b], s < n_parts (column-split target critics, ssac_chain_update); the
                                slot's value is their sum in index order (part 0 carries the head's bias) */
} ssac_td_spec;

/* Engine noise stream (SURVEY 8(b) RNG contract: "device noise from an engine Philox stream"): standard normals
 * from Philox4x32-10 + Box-Muller, element (row b, dim i) of draw number d = offset + *counter (counter may be
 * NULL) under `seed`.  Counter-based, so a replayed launch list advances it with the device-resident update count
 * and an eager launch passes the same number from the host.  ssac_philox_normal fills a buffer with that stream
 * (tests; and the definition of the stream: out[b*cols + i]). */
typedef struct ssac_rng {
    uint64_t seed;
    const int64_t *counter;  /* device pointer or NULL */
    int64_t offset;
} ssac_rng;
int ssac_philox_normal(float *out, int n_rows, int cols, const ssac_rng *rng, void *stream);

int ssac_abi_version(void);
const char *ssac_last_error(void);

/* ---- launch lists: record the kernel launches of one update (they still execute while being recorded),
 * then re-issue the whole sequence with ONE call.  Valid while every pointer argument of the recorded
 * calls stays alive at the same address -- the same contract as a hipGraph capture, which this replaces on
 * the update path (a hipGraph launch leaves a ~13 us idle tail on the queue on MI355X; a replayed list
 * does not).  Recording is per host thread; calls issued between begin and end on that thread are
 * recorded in issue order, whatever stream they were given, and replayed on the one stream passed. */
typedef struct ssac_launch_list ssac_launch_list;
int ssac_record_begin(void);
ssac_launch_list *ssac_record_end(void);          /* NULL (+ ssac_last_error) when no recording is open */
int ssac_launch_list_size(const ssac_launch_list *list);
int ssac_replay(ssac_launch_list *list, void *stream);
/* the same for a list with launches that are NUMBERED per update (ssac_actor_chain_fused's update_no): `value` >= 0 takes
 * the place of the number they were recorded with */
int ssac_replay_value(ssac_launch_list *list, void *stream, long long value);
/* ... and a second number, >= 0: the slot of a ring that a recorded launch writes its result to (ssac_actor_logs) */
int ssac_replay_value2(ssac_launch_list *list, void *stream, long long value, long long value2);
void ssac_launch_list_free(ssac_launch_list *list);

/* ---- one host call per recorded update.  A step owns the host side of a recorded critic update whose per-update
 * inputs travel through an ssac_feed ring: ssac_step_run composes slot k % n_slots -- [n_rows int64 replay indices |
 * n_ids int32 REDQ subset ids at ids_off | int32 log-ring slot at logslot_off | int64 noise draw number at draw_off
 * (-1: none)] (byte offsets inside a slot of slot_bytes, a multiple of 16) -- copies it into the ring, re-issues the
 * launch lists added with ssac_step_add_list in order (borrowed, not freed), and records / waits the slot-reuse events
 * (one per event_every updates; event_every divides n_slots).  What remains above the C ABI per update is drawing the
 * indices (torch CPU generator, replay.py:122) and the subset (Python random, agent.py:29). */
typedef struct ssac_step ssac_step;
ssac_step *ssac_step_create(void *ring, int n_slots, int slot_bytes, int n_rows, int n_ids, int ids_off,
                            int logslot_off, int draw_off, int event_every);   /* NULL + ssac_last_error on error */
int ssac_step_add_list(ssac_step *step, ssac_launch_list *list);
int ssac_step_run(ssac_step *step, const int64_t *idx_host, const int32_t *ids_host, int32_t log_slot, int64_t draw,
                  void *stream);
/* ssac_step_run hands the address of the slot it has just written to the replayed launches BY VALUE (a pointer member of
 * their recorded argument bytes is overwritten before each re-issue), so no workgroup reads the ssac_feed block to find
 * its inputs: one dependent load from cold memory less at the front of every workgroup of the update's first launch.
 * (A/B switch: ssac_slot_by_value in ssac_hip_test.h; results are bit-identical either way.) */
int64_t ssac_step_count(const ssac_step *step);
int ssac_step_seek(ssac_step *step, int64_t k);   /* updates the ring's device-side counter has consumed so far */
/* soft_update right after ssac_step_run (late-bound Polyak): leaves the request for the update just issued in the ring
 * tail.  Returns 1 when the request is or will be served by that update's weight-gradient launch (nothing to launch):
 * either the update's first launch had provably not started when the request landed, or -- the device keeping up with
 * the host -- it had, and its decision (waited for: microseconds) was "served".  0: not served, the caller launches
 * ssac_polyak.  -1: the decision did not arrive within 2 ms (the caller synchronises and asks ssac_step_polyak_done). */
int ssac_step_polyak(ssac_step *step, float tau);
int ssac_step_polyak_done(ssac_step *step);
void ssac_step_destroy(ssac_step *step);

/* ---- one-shot exchange between the ranks of a critic-sharded update (csrc/ssac_xchg.hip; SURVEY 8(e) "Transport"):
 * every rank owns a receive buffer, exported over HIP IPC and mapped by all peers; ssac_xchg_reduce is ONE recordable
 * launch that writes this rank's partial into every rank's buffer (posted peer-to-peer stores over xGMI), raises a
 * sequence flag, waits (bounded) for the peers' flags in its own buffer and reduces the `world` payloads in rank order,
 * in place: op 0 = MIN (the per-shard min-Q, agent.py:37-38 / learning.py:402), op 1 = SUM (the action gradient).
 * Set-up: create on every rank, exchange the ssac_xchg_handle bytes among the ranks (any host channel), connect. */
typedef struct ssac_xchg ssac_xchg;
/* allow_cached: 1 only when every rank shares ONE device (a single L2): the receive buffer may then fall back to
 * ordinary device memory; across devices it must be uncached (fine-grained) memory or the call fails. */
ssac_xchg *ssac_xchg_create(int rank, int world, int slot_floats, int allow_cached);   /* NULL + ssac_last_error on error */
int ssac_xchg_handle_bytes(void);
int ssac_xchg_handle(ssac_xchg *x, void *handle_out);
int ssac_xchg_connect(ssac_xchg *x, const void *handles /* world x ssac_xchg_handle_bytes(), rank-major */);
int ssac_xchg_reduce(ssac_xchg *x, float *data, int n, int op, void *stream);
/* MIN where only the OWNERS of the update's REDQ subset members send: owners = the update's id block in
 * device memory, n_slots int32 entries as every rank composes them from the same draw (agent.py:29): >= 0 a member this
 * rank owns (its local index), -(r + 1) a member rank r owns.  Ranks that own none write nothing; every rank waits for
 * the owners' flags only.  Same bits as ssac_xchg_reduce on partials that are +inf wherever a rank owns nothing. */
int ssac_xchg_reduce_owned(ssac_xchg *x, float *data, int n, const int32_t *owners, int n_slots,
                           int n_parts /* 1; > 1: data holds every slot's n / n_slots values as n_parts partial sums
                                          (ssac_td_spec.n_parts): summed before they are sent, the result in part 0 */,
                           void *stream);
int ssac_xchg_error(ssac_xchg *x);   /* 1: a peer's flag did not arrive within the spin bound since the last call (the result
                                        was poisoned with NaN); a pinned host word, cleared by the read, no synchronisation */
/* Flow control: every rank acknowledges the exchanges it has consumed and a sender reuses slot seq % 4 only when every
 * rank has consumed exchange seq - 4; a receiver accepts a flag only when it EQUALS its sequence number (a larger one
 * = the slot was lapped: poisoned result + error word).  (The protocol's failing-first evidence switch, ssac_xchg_test_mode,
 * exists in the LAB build only: ssac_hip_test.h.) */
void ssac_xchg_destroy(ssac_xchg *x);

/* ---- prioritised replay on the device (replaces super_sac/replay.py:140-190 sample / update_priorities and the
 * SumSegmentTree / MinSegmentTree of :207-353).  sum_tree / min_tree: float64 [2 cap], implicit heaps over cap = next
 * power of two >= capacity (replay.py:147-152), root at 1, leaves at [cap, 2 cap).
 * ssac_per_assign: leaves[rows] = prio^alpha (prio NULL: the current *max_priority, replay.py:156-161 push) and every
 * ancestor re-derived; a row named twice takes the LAST entry (numpy fancy assignment); update_max: also
 * *max_priority = max(*max_priority, max(prio)) and rows must be < n_filled (replay.py:183-190).  Violations of the
 * reference's assertions (priority <= 0 -> 1, row out of range -> 2) are written to the pinned host word and raised
 * by the Python layer at its next call.  winner_scratch: int32 [cap], all -1 between calls.
 * ssac_per_sample: mass_b = u_b * sum(0, n_filled - 1) in the reference's association order (replay.py:163-168,
 * 229-258), prefix-sum descent (:297-336), w_b = (p_b n)^-beta / max_weight (:171-177).  u: the B float64 uniforms the
 * host drew from numpy's global generator. */
int ssac_per_assign(double *sum_tree, double *min_tree, int64_t cap, const int64_t *rows, int n, const void *prio,
                    int prio_is_f64, double alpha, double *max_priority, int update_max, int64_t n_filled,
                    int *err_host_word, int32_t *winner_scratch, void *stream);
int ssac_per_sample(const double *sum_tree, const double *min_tree, int64_t cap, int64_t n_filled, const double *u,
                    int n_draws, double beta, int64_t *idx_out, double *weights_out, void *stream);

/* floats per net and the six segment offsets {W1,b1,W2,b2,W3,b3}. */
int64_t ssac_mlp_layout(int in_dim, int hidden, int out_dim, int64_t offsets[6]);

/* ---- replay sample path: replay.py:66-84 (fancy-index gather) + learning_utils.py:186-197
 * (.float()).  Gathers `n_rows` rows `idx[i]` from a row-major source of `row_elems`
 * elements per row into dst (row stride `ld_dst`, starting at column dst_col0).
 * src_dtype: 0 = fp32, 1 = uint8 (cast to fp32). */
int ssac_gather_rows(const void *src, int src_dtype, int64_t row_elems, const int64_t *idx,
                     int n_rows, float *dst, int64_t ld_dst, int64_t dst_col0, void *stream);

/* Whole-transition gather in ONE launch (replay.py:66-84 gathers five arrays separately):
 *   xsa [i, 0:s_elems]           <- float(s [idx[i]])      (row stride ld_x)
 *   xsa [i, s_elems:+a_elems]    <- act[idx[i]]
 *   x1sa[i, 0:s_elems]           <- float(s1[idx[i]])      (row stride ld_x1)
 *   rew_out[i], done_out[i]      <- rew[idx[i]], float(done[idx[i]])   (done is uint8)
 * for vector observations (or flattened images without augmentation). */
int ssac_gather_transition(const void *s, const void *s1, int s_dtype, int64_t s_elems,
                           const float *act, int64_t a_elems, const float *rew, const uint8_t *done,
                           const int64_t *idx, int n_rows, float *xsa, int64_t ld_x, float *x1sa,
                           int64_t ld_x1, float *rew_out, float *done_out, void *stream);
/* The same gather as the FIRST launch of a captured update: row indices are the first n_rows int64 of the
 * current slot of feed's pinned host ring (read over PCIe by the gather itself), and the launch also does
 * the work of ssac_begin_update(logs, n_logs, ctl, feed): one launch fewer on the update's critical chain. */
int ssac_gather_transition_begin(const void *s, const void *s1, int s_dtype, int64_t s_elems, const float *act,
                                 int64_t a_elems, const float *rew, const uint8_t *done, int n_rows, float *xsa,
                                 int64_t ld_x, float *x1sa, int64_t ld_x1, float *rew_out, float *done_out,
                                 const ssac_feed *feed, float *logs, int n_logs, ssac_adam_ctl *ctl,
                                 void *stream);

/* ---- replay push: ReplayBufferStorage.add (replay.py:48-60) for n transitions in ONE launch.  `packed` (device) holds
 * the fields one after the other, each as n rows of row_bytes payload bytes (the result of ONE async host-to-device copy
 * of a pinned staging buffer); field i's rows go to dst + ((start_row + r) % capacity) * row_bytes, verbatim. */
#define SSAC_MAX_PUSH_FIELDS 12
typedef struct ssac_push_field {
    void *dst;            /* the field's ring array (capacity x row_bytes) */
    int64_t row_bytes;
    int64_t src_offset;   /* byte offset of the field's n rows inside `packed` */
} ssac_push_field;
int ssac_replay_push(const ssac_push_field *fields, int n_fields, const void *packed, int n_rows, int64_t start_row,
                     int64_t capacity, void *stream);

/* ---- one layer of every selected net: Y[e] = act(X[e] W_l[id_e]^T + b_l[id_e])
 * (mlps.py:33-35,125-129; agent.py:34 runs this once per net in a Python loop).
 * layer 0/1/2 = fc1/fc2/out.  net_ids: device int32[n_sel] or NULL (=0..n_sel-1).
 * x_net_stride = 0 shares X across nets.  relu != 0 applies ReLU. */
int ssac_mlp_layer_fwd(const ssac_mlp *nets, int layer, const int32_t *net_ids, int n_sel,
                       const float *X, int64_t ldx, int64_t x_net_stride, int n_rows,
                       float *Y, int64_t ldy, int64_t y_net_stride, int relu, void *stream);

/* ---- backward-data of one layer: dX[e] = (dY[e] W_l[id_e]) (.) [mask[e] > 0]
 * (what autograd does for loss.backward(), learning.py:121,411).  mask may be NULL. */
int ssac_mlp_layer_dgrad(const ssac_mlp *nets, int layer, const int32_t *net_ids, int n_sel,
                         const float *dY, int64_t ldy, int64_t y_net_stride,
                         const float *mask, int64_t ldmask, int64_t mask_net_stride, int n_rows,
                         float *dX, int64_t ldx, int64_t x_net_stride, void *stream);

/* ---- backward-weights of one layer fused with the optimizer step:
 * g_W = dY^T X, g_b = colsum(dY); then either
 *   (grads == NULL)  Adam in place on params/m/v (torch.optim.Adam.step, learning.py:129-130,416)
 *                    and, when `target` != NULL, the Polyak update target <- (1-tau) target + tau p
 *                    (learning_utils.py:160-162) on the freshly updated values, or
 *   (grads != NULL)  store the gradients into a same-layout arena (for clip_grad_norm_).
 * sumsq: device float array, one slot per (net, tile) of this launch, receives sum(g^2)
 * partials (for clip_grad_norm_ / get_grad_norm, learning_utils.py:95-106); layout
 * sumsq[e * sumsq_net_stride + tile], tile < ssac_wgrad_tiles(nets, layer).  May be NULL. */
int ssac_wgrad_tiles(const ssac_mlp *nets, int layer);
int ssac_mlp_layer_wgrad(const ssac_mlp *nets, int layer, const int32_t *net_ids, int n_sel,
                         const float *X, int64_t ldx, int64_t x_net_stride,
                         const float *dY, int64_t ldy, int64_t y_net_stride, int n_rows,
                         float *adam_m, float *adam_v, const ssac_adam_ctl *ctl,
                         float *grads, float *sumsq, int64_t sumsq_net_stride,
                         float *target, float tau, void *stream);

/* fc2 AND fc1 weight gradients of every selected net in ONE launch (the small fc1 problem fills the CUs
 * the fc2 tiles leave idle): H1/DZ2 feed fc2, X/DZ1 feed fc1 (all (n_sel x n_rows x hidden) except X).
 * sumsq1 / sumsq0 point at the two layers' slots of the per-net sumsq row. */
int ssac_mlp_wgrad_fc12(const ssac_mlp *nets, const int32_t *net_ids, int n_sel, const float *X, int64_t ldx,
                        int64_t x_net_stride, const float *H1, const float *DZ2, const float *DZ1, int n_rows,
                        float *adam_m, float *adam_v, const ssac_adam_ctl *ctl, float *grads, float *sumsq1,
                        float *sumsq0, int64_t sumsq_net_stride, float *target, float tau, void *stream);

/* ssac_mlp_wgrad_fc12 plus the head layer's weight gradient (ssac_head_wgrad: out_dim <= 16, H2 / DQ) as extra
 * workgroups of the SAME launch; sumsq2 points at the head's slots of the per-net sumsq row. */
int ssac_mlp_wgrad_all(const ssac_mlp *nets, const int32_t *net_ids, int n_sel, const float *X, int64_t ldx,
                       int64_t x_net_stride, const float *H1, const float *H2, const float *DZ2, const float *DZ1,
                       const float *DQ, int n_rows, float *adam_m, float *adam_v, const ssac_adam_ctl *ctl,
                       float *grads, float *sumsq2, float *sumsq1, float *sumsq0, int64_t sumsq_net_stride,
                       float *target, float tau, void *stream);

/* ssac_mlp_wgrad_all for single-output heads when the backward pass was run UNSCALED (ssac_target_fwd_critic_bwdu):
 * row_scale (n_sel x n_rows) = dL/dq of every net and row (ssac_critic_loss_bwd[_lazy]'s dq) multiplies the rows of
 * DZ2u / DZ1u while they are loaded, and is the head layer's output gradient itself. */
int ssac_mlp_wgrad_all_scaled(const ssac_mlp *nets, const int32_t *net_ids, int n_sel, const float *X, int64_t ldx,
                              int64_t x_net_stride, const float *H1, const float *H2, const float *DZ2u,
                              const float *DZ1u, const float *row_scale, int n_rows, float *adam_m, float *adam_v,
                              const ssac_adam_ctl *ctl, float *grads, float *sumsq2, float *sumsq1, float *sumsq0,
                              int64_t sumsq_net_stride, float *target, float tau, void *stream);
/* ssac_mlp_wgrad_all_scaled with the loss gradient itself folded in (all nets of the arena, single-output heads,
 * n_rows <= 4096): every workgroup evaluates dL/dq = -2 pw w (td - (pw Q + pb)) / (denom n_rows) of ITS net's rows
 * into LDS (learning.py:90-98, 112) instead of reading a row_scale array a separate launch wrote.  td (n_rows), or
 * lazy_td (evaluated here, td_target_kernel's arithmetic; net slot 0's first workgroup writes lazy_td->td_out).
 * partials[n_nets][2] receives sum_b w err^2 and sum_b err per net: ssac_critic_logs(partials, n_nets, tiles = 1, ...)
 * finishes "losses/critic_overall_loss" and "losses/last_member_critic_td_error". */
/* logfold (nullable; Adam mode, every sumsq slot given): the update's log finalisation -- ssac_critic_logs' work --
 * rides in this launch: the workgroup that evaluated the TD targets also computes their statistics, every workgroup
 * draws an arrival ticket after its last cross-workgroup store, and the last arriver sums the partials in index order,
 * writes logs[0] += loss, logs[1] = TD error of the last net, logs[2] = gradient norm, publishes the block to its ring
 * slot and advances the input ring.  No fence and no second launch (csrc/ssac_critic_logs.h). */
/* Late-bound Polyak of a RECORDED update.  The reference calls soft_update AFTER critic_update returned
 * (main.py:409-414), i.e. after the update's launches were queued -- but, with the host running many updates ahead of
 * the GPU, long before they EXECUTE.  So the host leaves a request {tag = update number + 1, tau} in the tail of the
 * input ring (ssac_step_polyak); ONE thread of the update's first launch (the one that pulls the input slot) publishes
 * "begun", then looks for the request of its update and writes the decision (tau bits or 0) to feed->late_word; the
 * update's weight-gradient launch reads that word and, when it is set, applies
 * theta_bar <- (1 - tau) theta_bar + tau theta' in its Adam epilogue (the new parameters are in registers) -- no Polyak
 * launch.  One decider, so every workgroup acts alike; "begun" is stored (and fenced) BEFORE the request is read, so a
 * host that reads begun <= k after writing the request knows the request will be seen.
 * Tail of the input ring (behind the n_slots slots; SSAC_FEED_TAIL_BYTES, zeroed by ssac_feed_ring_alloc):
 *   int64 begun | int64 decided | uint32 last_served (tag) | 12 bytes pad | n_slots x {uint32 tag, uint32 tau bits}. */
#define SSAC_FEED_TAIL_BYTES 512

typedef struct ssac_logfold {
    unsigned *done_counter;   /* one zero-initialised uint32 in device memory, reset by the launch itself */
    float *logs;              /* the update's log block */
    float *td_logs;           /* mean / std / entropy bonus of the TD targets (3 floats inside the block), or NULL */
    ssac_feed *feed;          /* recorded update: publish + advance the ring; NULL otherwise */
    float *deferred_stats;    /* != NULL (recorded updates): DEFERRED finalisation -- this launch leaves the partials and
                                 the three TD statistics (here) behind and advances the input ring; the next update's
                                 first launch, or ssac_deferred_logs_flush, writes the ring slot (ssac_deferred_logs) */
    const uint32_t *late_word;     /* != NULL (= feed->late_word): `target` is updated only when the word is set, with
                                      the tau it holds (the `tau` argument is ignored) */
} ssac_logfold;

/* Deferred log finalisation of a recorded update (see ssac_logfold.deferred_stats): what the finishing workgroup reads.
 * Host-side struct, copied at launch.  Passed to ssac_chain_update / ssac_bf16_chain_update (one extra workgroup of the
 * launch finishes the PREVIOUS update's log block into its ring slot) and to ssac_deferred_logs_flush. */
typedef struct ssac_deferred_logs {
    const float *partials; int32_t n_nets, n_ss;   /* loss partials [n_nets][2]; gradient-norm partials (n_ss) */
    const float *sumsq;
    const float *td_stats;                         /* the 3 TD statistics the weight-gradient launch left, or NULL */
    int32_t td_off, n_rows;                        /* index of the TD statistics inside a log block; batch rows */
    float denom; int32_t _pad;
    const ssac_feed *feed;
} ssac_deferred_logs;
int ssac_deferred_logs_flush(const ssac_deferred_logs *d, int ring_slot, void *stream);
int ssac_mlp_wgrad_all_lossfold(const ssac_mlp *nets, const float *X, int64_t ldx, int64_t x_net_stride,
                                const float *H1, const float *H2,
                                const float *DZ2u /* NULL: rebuilt as W3_snapshot (.) [H2 > 0] while the fc2 tiles stage
                                                     their operand (the chained launch then never wrote it) */,
                                const float *DZ1u,
                                const float *W3_snapshot /* (n_nets x hidden) head rows as ssac_chain_update saved them;
                                                            NOT the arena's W3, which this launch's head workgroups
                                                            update concurrently */,
                                const float *Q, const float *td, const ssac_td_spec *lazy_td, const float *weight,
                                const ssac_popart *popart, int pop, float denom, float *partials, int n_rows,
                                float *adam_m, float *adam_v, const ssac_adam_ctl *ctl, float *grads, float *sumsq2,
                                float *sumsq1, float *sumsq0, int64_t sumsq_net_stride, float *target, float tau,
                                const ssac_logfold *logfold, void *stream);
/* ssac_critic_loss_bwd with the TD targets evaluated in the same launch (ssac_td_spec; they are also written to
 * lazy_td->td_out). */
int ssac_critic_loss_bwd_lazy(const float *q, int n_nets, int n_rows, int q_dim, const float *act, int64_t ld_act,
                              const ssac_td_spec *lazy_td, const float *weight, const ssac_popart *popart, int pop,
                              float denom, float *dq, float *logs, void *stream);
/* second merged launch of an update: the target critics' forward on the REDQ subset (Qt: n_sel x n_rows x out) and, as
 * extra workgroups, the TD-independent half of the online critics' backward pass on the saved forward H1 / H2:
 * DZ2u[b,:] = W3[a_b,:] (.) [h2 > 0], DZ1u[b,:] = (DZ2u[b,:] W2) (.) [h1 > 0]  (act: action index, discrete only). */
int ssac_target_fwd_critic_bwdu(const ssac_mlp *targets, const int32_t *net_ids, int n_sel, const float *X1,
                                int64_t ldx1, int n_rows, float *Qt, const ssac_mlp *critics, const float *H1,
                                const float *H2, const float *act, int64_t ld_act, float *DZ2u, float *DZ1u,
                                void *stream);

/* ---- elementwise Adam over a whole arena from stored gradients (clip path):
 * g *= ctl->clip_coef first (torch.nn.utils.clip_grad_norm_, learning.py:122-128). */
int ssac_adam_step(float *params, float *adam_m, float *adam_v, const float *grads, int64_t n,
                   const ssac_adam_ctl *ctl, void *stream);

/* one launch at the start of an update: zero the n_logs (<= 256) floats of the log block and, when
 * ctl != NULL, advance the optimizer step like ssac_adam_advance; when feed != NULL, also pull this
 * update's host inputs (ssac_feed) into feed->dst. */
int ssac_begin_update(float *logs, int n_logs, ssac_adam_ctl *ctl, const ssac_feed *feed, void *stream);

/* advance ctl->step by one and refresh step_size / bc2_sqrt (double precision on device). */
int ssac_adam_advance(ssac_adam_ctl *ctl, void *stream);
/* clip_coef = min(1, max_norm / (sqrt(sum(sumsq[0..n))) + 1e-6)); max_norm <= 0 -> 1.
 * Also writes the total norm to *norm_out when not NULL. */
int ssac_clip_coef(ssac_adam_ctl *ctl, const float *sumsq, int n, float max_norm, float *norm_out,
                   void *stream);

/* out[g] = sqrt(sum(sumsq[g*group_size .. (g+1)*group_size))): per-net gradient norms for the
 * "gradients/..." logs (learning_utils.py:95-106); multiplied by scale_by_clip->clip_coef when
 * given (the reference logs the norm after clip_grad_norm_ rescaled the gradients). */
int ssac_group_norms(const float *sumsq, int n_groups, int group_size, const ssac_adam_ctl *scale_by_clip,
                     float *out, void *stream);

/* ---- Polyak / hard update over n floats: learning_utils.py:160-167 */
int ssac_polyak(float *target, const float *source, int64_t n, float tau, void *stream);
/* the same over several (target, source, count) tensors in one launch (a module whose parameters are not a packed arena:
 * the pixel encoders' conv / fc / norm tensors); per element the arithmetic of ssac_polyak */
#define SSAC_MAX_POLYAK_SEGS 24
int ssac_polyak_multi(float *const *targets, const float *const *sources, const int64_t *counts, int n_tensors, float tau,
                      void *stream);

/* ---- tanh-squashed normal head: distributions.py:9-15, 64-104.
 * out (n_rows x 2A) -> a = tanh(mu + sigma eps) written to act_dst (row stride ld_act, column
 * offset act_col0) and log pi (n_rows) using the cached pre-tanh value. */
int ssac_tanh_normal_fwd(const float *out, int64_t ld_out, const float *eps, int n_rows, int act_dim,
                         float log_std_lo, float log_std_hi, float *act_dst, int64_t ld_act,
                         int64_t act_col0, float *logp, void *stream);

/* deterministic actor: a = tanh(out) + sample_std*eps (distributions.py:107-114, eps may be NULL),
 * then optional exploration noise (learning_utils.py:48-59): a += clamp(scale*noise, +-clip),
 * clamp to [-1+1e-6, 1-1e-6].  noise may be NULL; noise_clip <= 0 disables the clip. */
int ssac_det_action_fwd(const float *out, int64_t ld_out, const float *eps, float sample_std,
                        const float *noise, float noise_scale, float noise_clip, int n_rows,
                        int act_dim, float *act_dst, int64_t ld_act, int64_t act_col0, void *stream);

/* ---- TD target: learning_utils.py:298-354 (+ popart.py:35-59).  One workgroup.
 * q_t: (n_sel x n_rows x q_dim) target-critic outputs; min over n_sel (agent.py:37-38).
 * continuous (q_dim == 1): val = minq - exp(log_alpha)*logp  (use_entropy != 0), else val = minq.
 * discrete  (q_dim  > 1): logits (n_rows x q_dim): val = sum_a pi (minq_a - alpha log pi_a).
 * popart: device ssac_popart* or NULL; pop != 0 de-normalises.
 * td <- r + gamma (1-d) val, then PopArt stat update + normalise when popart != NULL.
 * logs[0..2] = mean(td), unbiased std(td), mean(entropy bonus). */
int ssac_td_target(const float *q_t, int n_sel, int n_rows, int q_dim, const float *logp_or_logits,
                   const float *rew, const float *done, const float *log_alpha, int use_entropy,
                   float gamma, ssac_popart *popart, int pop, float *td, float *logs, void *stream);

/* ---- critic loss gradient: learning.py:90-98,112.
 * q: (n_nets x n_rows x q_dim).  For q_dim > 1 the taken action column is gathered
 * (learning.py:92, act holds the index as float).  PopArt (w,b) applied to q when pop.
 * dq = -2 * weight_b * (td - q') * popart_w / (denom * n_rows); weight may be NULL (=1).
 * logs[0] += sum over nets of mean(weight*(td-q')^2)/denom   (critic_overall_loss)
 * logs[1]  = mean(td - q') of the LAST net                   (last_member_critic_td_error) */
int ssac_critic_loss_bwd(const float *q, int n_nets, int n_rows, int q_dim, const float *act,
                         int64_t ld_act, const float *td, const float *weight, const ssac_popart *popart,
                         int pop, float denom, float *dq, float *logs, void *stream);

/* ---- advantage-filtered behavioural cloning (learning.py:144-219)
 * ssac_adv_filter: A(s,a) = Q(s,a) - V(s) of adv_estimator.py:58-79 from the critics' outputs q on the stacked
 * batch [data | sample 1 .. sample n] ((n_nets) x (1+n_samples)*n_rows): Q = min over the nets (then w*q+b
 * when popart != NULL), V = mean (use_max: max) over the sampled actions.  Optional outputs: adv, the binary
 * filter mask (A >= 0), PER priorities relu(A)+1e-4 (learning_utils.py:287-295), logs[0] = mean(mask). */
int ssac_adv_filter(const float *q, int n_nets, int n_rows, int n_samples, const ssac_popart *popart, int use_max,
                    float *adv, float *mask, float *prio, float *logs, void *stream);
/* DR3 regulariser (learning.py:100-108) on a stacked batch [ (s,a) rows 0..B-1 | (s',a') rows B..2B-1 ]:
 * dz2 += coef * h2[partner row] through the ReLU mask, for both halves; partial[0..ssac_dr3_blocks()) receive
 * per-block sums of sum_h h2[b][h] * h2[B+b][h] (the "dr3_dotproduct" log is their total / (n_nets * B)). */
int ssac_dr3_blocks(void);
int ssac_dr3_add(float *dz2, const float *h2, int n_nets, int batch, int hidden, float coef, float *partial,
                 void *stream);
/* discrete counterparts (adv_estimator.py:41-56 "indirect"; learning_utils.py:257-268): q (n_nets x n_rows x A),
 * logits (n_actors x n_rows x A) of ALL ensemble actors (V uses their mean probabilities), act = action index. */
int ssac_adv_filter_discrete(const float *q, int n_nets, int n_rows, int n_actions, const float *logits, int n_actors,
                             const float *act, int64_t ld_act, const ssac_popart *popart, float *adv, float *mask,
                             float *prio, float *logs, void *stream);
int ssac_bc_discrete_bwd(const float *logits, const float *act, int64_t ld_act, const float *mask, int n_rows,
                         int n_actions, float inv_members, float *d_logits, float *logs_member, float *logs_total,
                         void *stream);
/* filtered BC loss of one member and its gradient w.r.t. the actor output (learning_utils.py:241-269):
 * loss_i = -mean(log pi(a_data|s) * mask) with the data action's pre-tanh value atanh(clamp(a, +-0.99))
 * (distributions.py:74-84); d_out = dL/d(out) for L = sum_i loss_i * inv_members; logs_member[0] = loss_i,
 * logs_total[0] += loss_i * inv_members.  mask == NULL: plain behavioural cloning (filter_=False). */
int ssac_bc_logprob_bwd(const float *out, int64_t ld_out, const float *act, int64_t ld_act, const float *mask,
                        int n_rows, int act_dim, float log_std_lo, float log_std_hi, float inv_members,
                        float *d_out, int64_t ld_dout, float *logs_member, float *logs_total, void *stream);

/* ---- Markov state-abstraction update (learning.py:266-341): the loss heads beside the MLP / encoder kernels.
 * The inverse model's head is ssac_bc_logprob_bwd / ssac_bc_discrete_bwd above (log-probability of the DATA action).
 *
 * contrastive head (learning.py:300-308): rows [0, n_pos) of z are real transitions (label 1), the rest shuffled ones
 * (label 0); loss_out[0] = F.binary_cross_entropy(sigmoid(z), labels) (logs clamped at -100 like torch) and
 * dz = coeff * d loss / d z (through torch's binary_cross_entropy_backward and the sigmoid).  One workgroup. */
int ssac_bce_sigmoid_bwd(const float *z, int n_pos, int n, float coeff, float *dz, float *loss_out, void *stream);
/* smoothness term (learning.py:310-311): loss_out[0] = mean_b relu(||s1_b - s_b||_2 / sqrt(dim) - max_dist)^2;
 * ds1 (and ds = its negative), nullable, receive coeff * d loss / d s1 -- written when accumulate == 0, added to
 * otherwise (0 where the norm is 0, as torch.norm's backward). */
int ssac_markov_smoothness_bwd(const float *s, int64_t lds, const float *s1, int64_t lds1, int n_rows, int dim,
                               float max_dist, float coeff, float *ds, int64_t ldds, float *ds1, int64_t ldds1,
                               int accumulate, float *loss_out, void *stream);
/* logs[0..4) = inverse loss (inverse_raw[0] * inverse_scale), contrastive loss, smoothness loss and
 * markov_loss = ic * inverse + cc * contrastive + sc * smoothness (learning.py:313-317, 337-340) */
int ssac_markov_logs(const float *inverse_raw, float inverse_scale, const float *contrastive, const float *smoothness,
                     float inverse_coeff, float contrastive_coeff, float smoothness_coeff, float *logs, void *stream);

/* encoder invariance constraint (learning_utils.py:401-409; learning.py:114-117): loss_out[0] = ||a - b||_F over the
 * (n_rows x dim) batch (a = encoder(augmented obs), b = encoder(original obs), no gradient to b); d_a (nullable)
 * receives coeff * d loss / d a -- written when accumulate == 0, added to otherwise; add_to[0] (nullable) += coeff *
 * loss (the critic update's overall-loss log includes the regulariser, learning.py:133). */
int ssac_frobenius_diff_bwd(const float *a, int64_t lda, const float *b, int64_t ldb, int n_rows, int dim, float coeff,
                            float *d_a, int64_t ldd, int accumulate, float *loss_out, float *add_to, void *stream);

/* action invariance constraint (learning_utils.py:272-285; offline_actor_update's actor_lambda term).  act: the action
 * sampled from the actor's distribution at the ORIGINAL observation, olp (n_rows) its log-probability there; out_a
 * (n_rows x 2 act_dim): the actor's output at the AUGMENTED observation.  loss_out[0] = F.mse_loss(olp, alp) with
 * alp_b = log pi_a(act_b) through atanh(clamp(act, +-0.99)); d_out = coeff * d loss / d out_a; add_to[0] += coeff*loss. */
int ssac_action_invariance_bwd(const float *out_a, int64_t ld_out, const float *act, int64_t ld_act, const float *olp,
                               int n_rows, int act_dim, float log_std_lo, float log_std_hi, float coeff, float *d_out,
                               int64_t ld_dout, float *loss_out, float *add_to, void *stream);
/* ... categorical actors: logits at the original / augmented observation, act (n_rows) the sampled class as float;
 * the reference sums the (B,) log-probabilities before the mse: loss = (sum_b olp_b - sum_b alp_b)^2. */
int ssac_action_invariance_discrete_bwd(const float *logits_o, const float *logits_a, const float *act, int n_rows,
                                        int n_actions, float coeff, float *d_logits, float *loss_out, float *add_to,
                                        void *stream);
/* deterministic actors (nets/mlps.py:78-93) in the offline actor update: the reference's Normal(tanh(out), 1e-4)
 * (distributions.py:107-114).  ssac_bc_det_logprob_bwd: filtered BC loss of learning_utils.py:241-269 and its gradient;
 * ssac_action_invariance_det_bwd: the constraint of :272-285 with a = loc at the original observation. */
int ssac_bc_det_logprob_bwd(const float *out, int64_t ld_out, const float *act, int64_t ld_act, const float *mask,
                            int n_rows, int act_dim, float inv_members, float *d_out, int64_t ld_dout,
                            float *logs_member, float *logs_total, void *stream);
int ssac_action_invariance_det_bwd(const float *out_o, int64_t ld_o, const float *out_a, int64_t ld_a, int n_rows,
                                   int act_dim, float coeff, float *d_out, int64_t ld_dout, float *loss_out,
                                   float *add_to, void *stream);

/* GaussianExplorationNoise.sample on a device action (learning_utils.py:48-60), in place over act[:, col0:col0+A]:
 * a <- clamp(a + clamp(scale * noise, +-clip), -1 + 1e-6, 1 - 1e-6)  (clip <= 0: no noise clipping). */
int ssac_exploration_noise(float *act, int64_t ld_act, int64_t act_col0, const float *noise, float noise_scale,
                           float noise_clip, int n_rows, int act_dim, void *stream);
/* log-probability of a ContinuousDeterministic action under its Normal(loc, 1e-4) (distributions.py:107-114), summed over
 * the action dimensions; eps (n_rows x act_dim, nullable): the rsample draw (NULL: the action is loc itself). */
int ssac_det_logprob(const float *eps, int n_rows, int act_dim, float *logp, void *stream);

/* ---- actor loss gradient, continuous: learning.py:392-408.
 * q (n_nets x n_rows): min over ALL nets (learning.py:402), arg-min routing.
 * dq[j][b] = -(popart_w) / (n_rows*E) for j = argmin_b else 0;  logs[0] += -mean(minq' - bonus)/E,
 * bonus = exp(log_alpha)*logp when use_entropy.  qmin_global (nullable): when the ensemble is sharded
 * across ranks, the all-reduced min over ALL critics; the gradient is routed only where the local min
 * equals it. */
int ssac_actor_loss_bwd(const float *q, int n_nets, int n_rows, const float *logp,
                        const float *log_alpha, int use_entropy, const ssac_popart *popart, int pop,
                        float inv_members, const float *qmin_global, float *dq, float *logs, void *stream);

/* use_baseline (learning.py:359, 401): the objective is the advantage A(s, a_theta) = Q'(s, a_theta) - V(s)
 * (adv_estimator.py:58-79; V has no gradient): dq as above, logs[0] += -mean(adv - bonus)/E with adv (n_rows) from
 * ssac_adv_filter on the same actions. */
int ssac_actor_loss_bwd_adv(const float *q, int n_nets, int n_rows, const float *logp, const float *log_alpha,
                            int use_entropy, const ssac_popart *popart, int pop, float inv_members,
                            const float *adv, float *dq, float *logs, void *stream);

/* ---- backward of the tanh-normal head: given dL/da summed from the critics' input
 * gradients dX (n_nets x n_rows x ldx, action columns start at act_col0) and the entropy
 * term alpha/(n_rows*E) * log pi, produce dL/d(out) (n_rows x 2A). */
int ssac_tanh_normal_bwd(const float *dX, int n_nets, int64_t ldx, int64_t x_net_stride,
                         int64_t act_col0, const float *out, int64_t ld_out, const float *eps,
                         int n_rows, int act_dim, float log_std_lo, float log_std_hi,
                         const float *log_alpha, int use_entropy, float inv_members, float *d_out,
                         int64_t ld_dout, void *stream);
/* deterministic actor: d_out = (sum_j dX_j[:, act cols]) * (1 - tanh(out)^2) (straight-through clamp). */
int ssac_det_action_bwd(const float *dX, int n_nets, int64_t ldx, int64_t x_net_stride,
                        int64_t act_col0, const float *out, int64_t ld_out, int n_rows, int act_dim,
                        float *d_out, int64_t ld_dout, void *stream);

/* ---- SAC-Discrete actor loss gradient: learning.py:382-390,407-408.
 * logits (n_rows x A); q (n_nets x n_rows x A) elementwise min (agent.py:38), no grad.
 * loss = -(1/E) mean_b sum_a pi (minq' - alpha log pi);  d_logits written. */
int ssac_discrete_actor_loss_bwd(const float *logits, const float *q, int n_nets, int n_rows,
                                 int n_act, const float *log_alpha, const ssac_popart *popart, int pop,
                                 float inv_members, float *d_logits, float *logs, void *stream);

/* ---- temperature update: learning.py:222-263 (loss uses log_alpha itself).
 * continuous: logp (n_rows); discrete: logits (n_rows x n_act) -> sum_a pi log pi.
 * One Adam step (betas from ctl) on the device scalar *log_alpha.
 * logs[0] = alpha_loss, logs[1] = exp(log_alpha) after the step. */
int ssac_alpha_update(float *log_alpha, float *adam_m, float *adam_v, ssac_adam_ctl *ctl,
                      const float *logp_or_logits, int n_rows, int n_act, float target_entropy,
                      float *logs, void *stream);

/* ---- SUNRISE backup weights: learning_utils.py:372-382.  q (n_members x n_rows) ->
 * w = sigmoid(-std_unbiased(q) * temp) + 0.5; logs[0..3] = mean,max,min,std(w). */
int ssac_sunrise_weights(const float *q, int n_members, int n_rows, float temp, float *w, float *logs,
                         void *stream);

/* ---- "softmax" backup weights: learning_utils.py:383-393.  q (n_members x n_rows): member k's critics (min over its
 * nets) on (s', a'_k ~ pi_k(.|s')) -> w = n_rows * softmax over the BATCH of (-std_unbiased(q) * temp); logs as above. */
int ssac_softmax_weights(const float *q, int n_members, int n_rows, float temp, float *w, float *logs,
                         void *stream);

/* ---- agent.Critic.forward(return_min=True) (agent.py:37-38): q (n_nets x n_rows x q_dim) -> elementwise min over
 * the nets; with act != NULL the column (int)act[b*ld_act] is selected (q.gather(-1, a.long()),
 * learning_utils.py:375, 389) and out is (n_rows), otherwise out is (n_rows x q_dim). */
int ssac_ensemble_min_select(const float *q, int n_nets, int n_rows, int q_dim, const float *act, int64_t ld_act,
                             float *out, void *stream);

/* ---- DrQ augmentations: augmentations.py:214-263 (Drqv2Aug) and :165-204 (DrqAug).
 * src: uint8 or fp32 images (n x c x h x h) gathered through idx (may be NULL = identity);
 * rows >= n_aug are copied un-augmented (aug_mix, learning_utils.py:200-206).
 * shift: int64 (n x 2) = (x, y) per sample.  mode 0: Drqv2 (replicate pad + fp32 bilinear grid),
 * mode 1: DrQ v1 (reflection pad + integer crop, optional additive noise, clamp 0..255). */
int ssac_drq_shift(const void *src, int src_dtype, const int64_t *idx, int n, int c, int h, int pad,
                   const int64_t *shift, int mode, const float *noise, int n_aug, float *dst,
                   void *stream);

/* ==== fused kernels (csrc/ssac_fused.hip): the same arithmetic as the per-layer entry points above,
 * with the activations of a 32-row tile kept in LDS across fc1 -> fc2 -> head.  Supported when
 * ssac_fused_supported() (hidden % 32 == 0, hidden <= 256, out_dim <= 64, LDS carve fits 160 KB);
 * callers fall back to the per-layer entry points otherwise.  Returns 1 when the double-buffered weight
 * staging fits (required of the CRITIC halves of the merged launches below), 2 when only the single-buffer
 * carve fits (wide input + wide head, e.g. 376 -> 34), 0 when unsupported. ==== */
int ssac_fused_supported(const ssac_mlp *nets);
/* The merged weight-gradient launch (ssac_mlp_wgrad_all / _scaled / _lossfold / _fc12) picks its FORM by itself: 64 x 64 tiles
 * with the K loop staged through LDS, or -- while all of its workgroups are resident at once (<= 512), i.e. for the under-
 * filled launches of small ensembles (SAC's 2 critics, a rank that holds 2-4 of 16, the actor) -- the LATENCY form: 32 x 32
 * tiles whose 8 waves split the batch (K) and buffer-load their MFMA operands straight from memory.  Same sums in another
 * order (fp32 rounding).  (Forcing a form, for the parity tests: ssac_wgrad_variant in ssac_hip_test.h.) */
/* row tiles the fused critic launch uses for (n_rows, n_nets): the `partials` buffer holds
 * n_nets * tiles * 2 floats (the tile size is the library's automatic choice unless ssac_fused_tile_rows of
 * ssac_hip_test.h overrides it). */
int ssac_fused_row_tiles(const ssac_mlp *nets, int n_rows, int n_nets);

/* y = MLP(x) for every selected net in ONE launch (agent.py:34 loop + mlps.py:123-129).
 * H1/H2 (n_sel x n_rows x hidden) are written when not NULL (needed by a later backward).  A negative entry of
 * net_ids marks an empty slot: its rows of Y are set to +inf (the sharded update forwards only the REDQ subset
 * members a rank owns but keeps a fixed launch shape). */
int ssac_mlp3_fwd_fused(const ssac_mlp *nets, const int32_t *net_ids, int n_sel, const float *X,
                        int64_t ldx, int64_t x_net_stride, int n_rows, float *H1, float *H2, float *Y,
                        void *stream);

/* actor forward + tanh-normal sample + log pi in ONE launch (mlps.py:32-39, distributions.py:9-15).
 * out (n_rows x 2A), H1, H2 may be NULL. */
int ssac_actor_sample_fused(const ssac_mlp *actor, const float *X, int64_t ldx, int n_rows,
                            const float *eps, float log_std_lo, float log_std_hi, float *act_dst,
                            int64_t ld_act, int64_t act_col0, float *logp, float *H1, float *H2,
                            float *out, const ssac_rng *rng /* used when eps == NULL */, void *stream);

/* ssac_actor_sample_fused and the online critics' forward (ssac_mlp3_fwd_fused over all nets, H1 / H2 / Q saved) as
 * ONE launch: the critic forward does not depend on the sampled action and fills the CUs the small actor launch
 * leaves idle; follow it with ssac_critic_bwd_fused instead of ssac_critic_fwd_bwd_fused. */
int ssac_actor_sample_critic_fwd(const ssac_mlp *actor, const float *Xa, int64_t ldxa, int n_rows, const float *eps,
                                 float log_std_lo, float log_std_hi, float *act_dst, int64_t ld_act,
                                 int64_t act_col0, float *logp, const ssac_rng *rng, const ssac_mlp *critics,
                                 const float *Xc, int64_t ldxc, float *H1, float *H2, float *Q,
                                 const ssac_gather *gather /* NULL: inputs are Xa / Xc */, void *stream);

/* Everything of a critic update that does not need the TD target, as ONE launch (continuous actions, single-output
 * critics): per REDQ subset slot j and 16-row tile a TARGET CHAIN workgroup -- actor forward on s' + tanh-normal sample
 * (a' -> x1sa[:, act_col0:], log pi -> logp; every slot recomputes the actor for its rows) and then target critic
 * net_ids[j] on [s'|a'] (-> Qt, n_sel x n_rows) -- beside the online critics' forward AND the TD-independent half of
 * their backward pass in the same workgroup (H1 / H2 / Q saved, DZ2u / DZ1u as ssac_target_fwd_critic_bwdu writes
 * them).  Replaces ssac_actor_sample_critic_fwd + ssac_target_fwd_critic_bwdu; `gather` as there. */
int ssac_chain_update(const ssac_mlp *actor, const float *Xa, int64_t ldxa, int n_rows, const float *eps,
                      float log_std_lo, float log_std_hi, float *x1sa, int64_t ld_x1, int64_t act_col0, float *logp,
                      const ssac_rng *rng, const ssac_mlp *targets, const int32_t *net_ids, int n_sel, float *Qt,
                      const ssac_mlp *critics, const float *Xc, int64_t ldxc, float *H1, float *H2, float *Q,
                      float *DZ2u /* NULL: not written; W3_snapshot (n_nets x hidden) is filled instead */, float *DZ1u,
                      float *W3_snapshot, const ssac_gather *gather,
                      const ssac_deferred_logs *deferred /* nullable */,
                      unsigned long long *handoff /* nullable: n_rows x A words, zeroed once */,
                      int target_splits /* 1, 2, 4: ssac_chain_target_splits() */,
                      ssac_xchg *xchg /* nullable; critic-sharded rank, with handoff: the MIN exchange of Qt over the ranks
                                         (what ssac_xchg_reduce_owned(xchg, Qt, n_sel * n_rows, net_ids, n_sel, target_splits)
                                         would do as a launch of its own) runs in a tail workgroup of THIS launch, behind
                                         its target-critic workgroups */,
                      void *stream);
/* handoff != NULL selects the PRODUCER / CONSUMER form of the launch: the actor runs ONCE per 16-row tile (not once per
 * subset slot) and publishes a' as tagged 8-byte granules in `handoff`; the target-critic workgroups of the tile gather
 * their own s' rows, run fc1 on the state columns while the actor is still working, poll the granules and add
 * a' W1[:, S:S+A]^T -- their critical path behind the actor is fc2 + head instead of a whole MLP pass.  Same outputs up
 * to fp32 association of fc1's sum (the action columns enter last).
 * target_splits > 1 (hidden 256, with handoff): every (slot, tile) gets that many consumers, each computing hidden /
 * target_splits columns of fc2 from register-resident weight fragments and a PARTIAL head dot product; Qt is then
 * (n_sel x target_splits x n_rows), to be read through an ssac_td_spec with n_parts = target_splits. */
int ssac_chain_target_splits(const ssac_mlp *actor, const ssac_mlp *targets, const ssac_mlp *critics, int n_rows, int n_sel);

/* ---- the online actor update (learning.py:344-421) in four launches:
 *   ssac_actor_sample_concat_fused   actor forward (h1 / h2 / head output saved) + tanh-normal rsample + log pi, the rows
 *                                    [s | a_theta] written out as the critics' input                 (mlps.py:32-39, 124)
 *   ssac_critic_fwd_dx_fused         all critics' forward on [s | a_theta] and the UNSCALED input gradient of the action
 *                                    columns, DXu[j][b][:] = dQ_j/da  (dz2 = W3 (.) [h2>0], dz1 = (dz2 W2) (.) [h1>0],
 *                                    dx = dz1 W1[:, act cols]) -- what loss.backward() sends back to the actor
 *   ssac_actor_bwd_fused             per row: arg-min critic (learning.py:402 takes the min over ALL critics), the
 *                                    policy gradient -pw/(B E) dQ_argmin/da and the entropy term through the tanh-normal
 *                                    head (d_out, B x 2A), then the actor's head backward and fc2 backward-data on the
 *                                    saved forward (DZ2, DZ1); partials[tile] = sum of (Q' - alpha log pi) over the tile
 *   ssac_mlp_wgrad_all               the actor's weight gradients + Adam (above), then ssac_actor_logs for the two logs */
int ssac_actor_sample_concat_fused(const ssac_mlp *actor, const float *X, int64_t ldx, int n_rows, const float *eps,
                                   float log_std_lo, float log_std_hi, float *xsa, int64_t ld_xsa, float *logp,
                                   float *H1, float *H2, float *out, const ssac_rng *rng, void *stream);
int ssac_critic_fwd_dx_fused(const ssac_mlp *nets, const float *X, int64_t ldx, int n_rows, int dx_col0, int dx_cols,
                             float *Q, float *DXu, void *stream);
int ssac_actor_bwd_fused(const ssac_mlp *actor, const float *H1, const float *H2, int n_rows, const float *Qc,
                         int n_critics, const float *DXu, const float *aout, const float *eps, const float *logp,
                         const float *log_alpha, int use_entropy, float log_std_lo, float log_std_hi, float inv_members,
                         const ssac_popart *popart, int pop, float *d_out, float *DZ2, float *DZ1, float *partials,
                         void *stream);
/* The same two logs folded into the actor's weight-gradient launch (round 6): ssac_mlp_wgrad_all for ONE net in Adam mode
 * whose last workgroup to finish (arrival tickets, as ssac_logfold) adds -inv_members sum(partials) / n_rows to logs_loss[0],
 * writes sqrt(sum of the launch's sumsq slots) to logs_gn[0] and -- ring != NULL -- the finished block to slot ring_slot of
 * the log ring (recorded: the second number of ssac_replay_value2).  No log launch behind the update. */
typedef struct ssac_actor_logfold {
    unsigned *done_counter;   /* one zero-initialised uint32 in device memory, reset by the launch itself */
    const float *partials;    /* the tiles' loss terms (ssac_actor_bwd_fused / ssac_actor_chain_fused) */
    int32_t n_tiles, n_rows;
    float inv_members;
    int32_t width;            /* floats of the log block (with ring) */
    float *logs_loss, *logs_gn;   /* logs_gn may be NULL */
    const float *block;       /* the log block logs_loss / logs_gn point into (with ring) */
    float *ring;              /* NULL: no publication */
    int64_t ring_slot;
} ssac_actor_logfold;
int ssac_mlp_wgrad_all_actor(const ssac_mlp *nets, const float *X, int64_t ldx, const float *H1, const float *H2,
                             const float *DZ2, const float *DZ1, const float *DQ, int n_rows, float *adam_m, float *adam_v,
                             const ssac_adam_ctl *ctl, float *sumsq2, float *sumsq1, float *sumsq0, int64_t sumsq_net_stride,
                             const ssac_actor_logfold *fold, void *stream);
/* ring != NULL (round 6): the finished block (`width` floats at `block`, logs_loss / logs_gn pointing into it) is also written
 * to slot `ring_slot` of the log ring (ring + ring_slot * width) by this launch; in a RECORDED launch the slot is the
 * second number of ssac_replay_value2 -- no copy behind the replay. */
int ssac_actor_logs(const float *partials, int n_tiles, int n_rows, float inv_members, const float *sumsq, int n_sumsq,
                    float *logs_loss, float *logs_gn, const float *block, int width, float *ring, long long ring_slot,
                    void *stream);
/* The fused actor update on a CRITIC-SHARDED rank (SURVEY 8(e) "Collective -- actor step": a MIN over (B,) and a SUM over
 * (B x A)): between ssac_critic_fwd_dx_fused on the rank's own critics and ssac_actor_bwd_fused with n_critics = 1 --
 *   ssac_actor_route_local  q_local[b] = min over the local critics (and a copy, q_reduce, for the MIN all-reduce),
 *                           d_sel[b][:] = dQ/da of the local arg-min critic (first index on ties)
 *   (MIN all-reduce of q_reduce)
 *   ssac_actor_route_claim  claim[b] = rank if the local minimum IS the global one, +inf otherwise
 *   (MIN all-reduce of claim: bit-equal minima on several ranks go to the lowest rank -- torch.min's first index)
 *   ssac_actor_route_mask   rows this rank did not win are zeroed in d_sel
 *   (SUM all-reduce of d_sel) */
int ssac_actor_route_local(const float *q, const float *dxu, int n_local, int n_rows, int action_dim, float *q_local,
                           float *q_reduce, float *d_sel, void *stream);
int ssac_actor_route_claim(const float *q_local, const float *q_global, int n_rows, int rank, float *claim, void *stream);
int ssac_actor_route_mask(const float *claim, int rank, int n_rows, int action_dim, float *d_sel, void *stream);

/* The first three launches above as ONE (round 4): the actor's workgroups (16-row tiles, lowest workgroup ids) run the
 * forward + rsample, publish a_theta as tagged 8-byte granules, WAIT for their rows' Q_j and dQ_j/da from every critic and
 * run the backward half on the forward they have just saved; the critics' tiles take [s | a_theta] as the chained critic
 * update's target critics do (state columns gathered and multiplied first, the action columns' rank-A term added when
 * a_theta arrives) and send Q / dQ/da back as granules.  Same outputs as the three launches (fc1 of the critics sums the
 * state and action columns separately: fp32 association only).
 *   handoff    ssac_actor_chain_handoff_words(n_rows, n_critics, A) 8-byte words, zeroed ONCE; never reset: a granule
 *              carries the launch's tag
 *   update_no  >= 0: the caller's number of this update -- the tag is 1 + update_no, and with eps == NULL the noise of row
 *              b, dimension i is the engine's Philox stream at draw rng->offset (rng->counter NULL), evaluated identically
 *              by the forward and the backward half.  A RECORDED launch must be numbered; its replays go through
 *              ssac_replay_value(list, stream, n), which renumbers it: tag 1 + n, draw rng->offset - update_no + n (a
 *              plain ssac_replay of such a list is refused).  Numbers must not repeat on one handoff buffer.
 *              < 0: an eager launch nobody numbers (tag: a host counter with bit 31 set)
 *   n_rows     at most SSAC_ACTOR_CHAIN_MAX_ROWS (refused beyond: see the define below)
 *   begin_logs / n_logs / begin_ctl   nullable: ssac_begin_update's duties (log block cleared, optimizer step advanced)
 *            done by the first actor workgroup, so that a recorded actor update needs no launch in front */
int ssac_actor_chain_fused(const ssac_mlp *actor, const float *X, int64_t ldx, int n_rows, const float *eps,
                           const ssac_rng *rng, float log_std_lo, float log_std_hi, float *xsa, int64_t ld_xsa,
                           float *logp, float *H1, float *H2, float *out, const ssac_mlp *critics, float *Q, float *DXu,
                           const float *log_alpha, int use_entropy, float inv_members, const ssac_popart *popart, int pop,
                           float *d_out, float *DZ2, float *DZ1, float *partials, unsigned long long *handoff,
                           long long update_no, float *begin_logs, int n_logs, ssac_adam_ctl *begin_ctl, void *stream);
int64_t ssac_actor_chain_handoff_words(int n_rows, int n_critics, int action_dim);
/* batch rows the chained launch takes: its actor workgroups (one per 16 rows) wait for critic tiles dispatched behind them,
 * so they may occupy at most half of the 256 CUs; larger batches run the three launches */
#define SSAC_ACTOR_CHAIN_MAX_ROWS 2048

/* critic forward of ALL nets + loss gradient + backward-data in ONE launch (learning.py:83-98,112,121):
 * writes H1, H2, Q (n_nets x n_rows x out), DQ, DZ2 = dL/d(pre-activation of fc2), DZ1, and per-(net,
 * row-tile) partial sums partials[(e*tiles + tile)*2 + {sum w*err^2, sum err}] for ssac_critic_logs. */
int ssac_critic_fwd_bwd_fused(const ssac_mlp *nets, const float *X, int64_t ldx, int n_rows,
                              const float *td, const float *weight, const float *act, int64_t ld_act,
                              const ssac_popart *popart, int pop, float denom, float *H1, float *H2,
                              float *Q, float *DQ, float *DZ2, float *DZ1, float *partials,
                              const ssac_td_spec *lazy_td /* NULL: read `td` */, void *stream);

/* the second half of ssac_critic_fwd_bwd_fused alone: H1, H2, Q come from an earlier ssac_mlp3_fwd_fused
 * launch over all nets (same layouts); results are bit-identical to the one-launch form.  The forward does
 * not depend on the TD target, so the host runs it on a second stream beside the actor / target-critic /
 * TD-target chain (learning_utils.py:298-354) and joins before this launch. */
int ssac_critic_bwd_fused(const ssac_mlp *nets, int n_rows, const float *td, const float *weight,
                          const float *act, int64_t ld_act, const ssac_popart *popart, int pop, float denom,
                          const float *H1, const float *H2, const float *Q, float *DQ, float *DZ2,
                          float *DZ1, float *partials, const ssac_td_spec *lazy_td, void *stream);

/* weight gradient of the head layer (out_dim <= 16) + Adam/Polyak, VALU: dW3 = DQ^T H2, db3 = colsum(DQ).
 * Same grads/sumsq/target conventions as ssac_mlp_layer_wgrad; sumsq slots: ssac_head_wgrad_tiles(). */
int ssac_head_wgrad_tiles(const ssac_mlp *nets);
int ssac_head_wgrad(const ssac_mlp *nets, const int32_t *net_ids, int n_sel, const float *H2,
                    const float *DQ, int n_rows, float *adam_m, float *adam_v, const ssac_adam_ctl *ctl,
                    float *grads, float *sumsq, int64_t sumsq_net_stride, float *target, float tau,
                    void *stream);

/* reduce the fused critic kernel's partials into the log block: logs[0] += loss, logs[1] = mean td
 * error of the last net, logs[2] = sqrt(sum(sumsq[0..n_sumsq))) (x clip_coef when given).  With feed != NULL
 * the launch also does what ssac_publish_logs does. */
int ssac_critic_logs(const float *partials, int n_nets, int tiles, int n_rows, float denom,
                     const float *sumsq, int n_sumsq, const ssac_adam_ctl *scale_by_clip, float *logs,
                     const ssac_td_spec *lazy_td /* with td_logs: mean, std, entropy bonus of the targets */,
                     float *td_logs, ssac_feed *feed, void *stream);
/* last launch of a captured update: copy the finished log block to its ring slot and advance feed->tick. */
int ssac_publish_logs(const float *logs, ssac_feed *feed, void *stream);

/* ==== pixel encoders (nets/cnns.py:37-103): convolution = im2col + GEMM, activations channels-last ====
 * ssac_im2col: col[(b,oy,ox)][(c,ky,kx)] = float(src[b,c,oy*s+ky,ox*s+kx]) / div + shift, source addressed by
 * element strides (so NCHW images and channels-last feature maps both work; src_u8 reads uint8).
 * The nn.Linear over the flattened NCHW feature map is the same gather with k = the whole map. */
int ssac_im2col(const void *src, int src_u8, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int B, int C,
                int Hi, int Wi, int k, int stride, float div, float shift, float *col, void *stream);
/* adjoint of im2col: dx[b,c,y,x] = sum of the dcol entries that read it, times [mask > 0] when mask != NULL
 * (ReLU backward of the producing layer), destination / mask addressed by element strides. */
int ssac_col2im(const float *dcol, float *dx, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                const float *mask, int64_t mb, int64_t mc, int64_t my, int64_t mx, int B, int C, int Hi,
                int Wi, int k, int stride, void *stream);
/* the same adjoint for a column matrix with columns in (ky, kx, c) order (dY times the weight permuted by
 * ssac_permute_cp(W, co, ci, k*k, 1)) into a contiguous channels-last dx (B, Hi, Wi, C), C % 4 == 0; mask (same
 * shape, may be NULL) as above.  Sums the same taps in the same order as ssac_col2im. */
int ssac_col2im_cl(const float *dcol, float *dx, const float *mask, int B, int C, int Hi, int Wi, int k, int stride,
                   void *stream);
/* Y = act(X W^T + b), X (M x K), W (N x K): F.conv2d on patches / nn.Linear (cnns.py:61-66,98-102). */
int ssac_linear_fwd(const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, float *Y,
                    int64_t ldy, int M, int N, int K, int relu, void *stream);
/* ---- implicit-GEMM convolutions (csrc/ssac_conv_implicit.hip) for layers with ci % 32 == 0 and co % 32 == 0:
 * channels-last fp32 activations (B, H, W, C), nn.Conv2d weights (co, ci, k, k) used in place, no padding,
 * stride s; the patch gather happens in the operand loads, no column matrix is materialised.
 *   ssac_conv_fwd   y = relu(conv(x) + bias)                                  (cnns.py:59-66, 96-100)
 *   ssac_conv_dgrad dx = [x_mask > 0] * conv_transpose(dy)    (x_mask = this layer's input = previous ReLU output;
 *                   stride 1: every tap; stride 2..4: per parity class (iy % s, ix % s) of input pixels, only the taps
 *                   that class can receive; larger strides: generic gather through all taps)
 *   ssac_conv_wgrad partial_w[slice] (co,ci,k,k), partial_b[slice] (co) over slices of pix_per_slice (multiple
 *                   of 32) output pixels; sum them with ssac_reduce_slices (fixed order).  With k*k <= 9 and slices
 *                   that are multiples of 128 pixels every wave keeps all taps' accumulators (balanced, dy read once)
 *                   and the four waves are summed through LDS in a fixed order; otherwise the taps are split over
 *                   the waves. */
int ssac_conv_implicit_supported(int ci, int co, int k);
int ssac_conv_fwd(const float *x, const float *w, const float *bias, float *y, int B, int Hi, int Wi, int ci, int co,
                  int k, int s, void *stream);
int ssac_conv_dgrad(const float *dy, const float *w, const float *x_mask, float *dx, int B, int Hi, int Wi, int ci,
                    int co, int k, int s, void *stream);
int ssac_conv_wgrad_slices(int B, int Ho, int Wo, int pix_per_slice);
int ssac_conv_wgrad(const float *dy, const float *x, float *partial_w, float *partial_b, int B, int Hi, int Wi,
                    int ci, int co, int k, int s, int pix_per_slice, void *stream);
/* The weight gradient of a layer with SMALL feature maps (an image's 32-channel slices of x and dy fit LDS twice per CU;
 * k k <= 16): both operands staged in LDS per image, persistent workgroups, ONE partial (slice) per workgroup and
 * (ci / 32, co / 32) block pair.  ssac_conv_wgrad_img_slices returns that count (0: use ssac_conv_wgrad); partial_w
 * (slices, co, ci, k, k), partial_b (slices, co) as ssac_conv_wgrad.  Same values up to the order of the sum over pixels. */
int ssac_conv_wgrad_img_slices(int B, int Hi, int Wi, int ci, int co, int k, int s);
int ssac_conv_wgrad_img(const float *dy, const float *x, float *partial_w, float *partial_b, int B, int Hi, int Wi, int ci,
                        int co, int k, int s, void *stream);
/* ---- the FIRST layer as an implicit GEMM too (cnns.py:41/76 conv1 with the x/div + shift input normalisation of
 * cnns.py:58/95 in front): fp32 NCHW image (B, C, Hi, Wi) of raw pixel values, nn.Conv2d weight (co, C, k, k),
 * channels-last output.  ssac_conv_first_supported returns the taps per lane half (2 or 4) when the geometry is
 * covered -- co % 32 == 0, C k k <= 256, k <= 8, stride a multiple of 2 (k <= 4) or 4, (Wo-1) s + 2 taps <= Wi,
 * fewer than 2^31 elements -- and 0 when the layer has to stay on ssac_im2col + ssac_linear_fwd.
 *   ssac_conv_first_fwd    y = relu(conv(img / div + shift) + bias)
 *   ssac_conv_first_wgrad  partial_w[slice] (co,C,k,k), partial_b[slice] (co) per slice of pix_per_slice (multiple of
 *                          128) output pixels = one workgroup; slices as ssac_conv_wgrad_slices, sum with
 *                          ssac_reduce_slices. */
int ssac_conv_first_supported(int C, int co, int k, int s, int Hi, int Wi, int64_t B);
int ssac_conv_first_fwd(const float *img, const float *w, const float *bias, float *y, int B, int C, int Hi, int Wi,
                        int co, int k, int s, float div, float shift, void *stream);
int ssac_conv_first_wgrad(const float *dy, const float *img, float *partial_w, float *partial_b, int B, int C, int Hi,
                          int Wi, int co, int k, int s, float div, float shift, int pix_per_slice, void *stream);
/* The same weight gradient with the image staged through LDS in bands of output rows (16-byte loads; the patch entries are
 * LDS reads instead of global gathers); persistent workgroups, ONE partial (slice) per workgroup:
 * ssac_conv_first_wgrad_band_slices returns that count (0: geometry not covered, use ssac_conv_first_wgrad); partial_w
 * (slices, co, C, k, k) and partial_b (slices, co) are summed with ssac_reduce_slices(_pair) as before.  Same values up to
 * the order of the sum over pixels. */
int ssac_conv_first_wgrad_band_slices(int B, int C, int Hi, int Wi, int co, int k, int s);
int ssac_conv_first_wgrad_band(const float *dy, const float *img, float *partial_w, float *partial_b, int B, int C, int Hi,
                               int Wi, int co, int k, int s, float div, float shift, void *stream);
/* split-K forward for short, very deep problems (the pixel encoders' fc over the flattened feature map):
 * partial (slices x M x N) = X[:, slice] W[:, slice]^T per K slice of k_per_slice (multiple of 32) columns;
 * ssac_reduce_slices_bias then writes Y[m*ld_out + n] = bias[n] + sum over slices, in a fixed order. */
int ssac_linear_fwd_splitk(const float *X, int64_t ldx, const float *W, int64_t ldw, float *partial, int M, int N,
                           int K, int k_per_slice, void *stream);
/* the same partials for N <= 64 outputs as an operand stream (no LDS: each lane reads its row 16 bytes at a time into the
 * MFMA operand registers, 8 groups in flight); K and k_per_slice multiples of 8, 16-byte aligned rows.  Same values up to
 * the k order inside a slice. */
int ssac_linear_fwd_stream_supported(int M, int N, int K, int k_per_slice, int64_t ldx, int64_t ldw);
int ssac_linear_fwd_stream(const float *X, int64_t ldx, const float *W, int64_t ldw, float *partial, int M, int N, int K,
                           int k_per_slice, void *stream);
int ssac_reduce_slices_bias(const float *partial, int slices, int M, int N, const float *bias, float *out,
                            int64_t ld_out, void *stream);
/* dX (M x N_in) = dY (M x K_out) W (K_out x N_in) */
int ssac_linear_dgrad(const float *dY, int64_t ldy, const float *W, int64_t ldw, float *dX, int64_t ldx,
                      int M, int N_in, int K_out, void *stream);
/* the same with the ReLU derivative of the producing layer in the epilogue: dX = [mask > 0] * (dY W), mask (M x N_in)
 * with row stride ldmask (the layer's saved output) -- the fc of the pixel encoders over the last feature map
 * (cnns.py:63-66, 98-100), one pass over dX instead of a GEMM store + a mask pass */
int ssac_linear_dgrad_masked(const float *dY, int64_t ldy, const float *W, int64_t ldw, const float *mask,
                             int64_t ldmask, float *dX, int64_t ldx, int M, int N_in, int K_out, void *stream);
/* split-K weight gradient: slice z covers rows [z*rows_per_slice, ...): partial_w[z] = dY_z^T X_z
 * (M_out x N_in), partial_b[z] = colsum(dY_z); reduce with ssac_reduce_slices. */
int ssac_linear_wgrad_splitk(const float *dY, int64_t ldy, const float *X, int64_t ldx, float *partial_w,
                             float *partial_b, int M_out, int N_in, int n_rows, int rows_per_slice,
                             void *stream);
int ssac_reduce_slices(const float *partial, int slices, int64_t n, float *out, void *stream);
/* two such reductions with the same slice count in one launch (a layer's weight and bias slices); per output the same
 * sums in the same order as ssac_reduce_slices */
int ssac_reduce_slices_pair(const float *partial0, int64_t n0, float *out0, const float *partial1, int64_t n1, float *out1,
                            int slices, void *stream);
int ssac_relu_mask(float *dy, const float *y, int64_t n, void *stream);
int ssac_relu_mask_to(const float *dy, const float *y, int64_t n, float *out, void *stream);
/* (n x channels x pixels) <-> (n x pixels x channels): the encoders' fc weight is stored in the reference's NCHW
 * flatten order (cnns.py:63,98); a channels-last copy lets the fc read the last feature map in place. */
int ssac_permute_cp(const float *src, float *dst, int n, int channels, int pixels, int to_channels_last,
                    void *stream);
/* sum of squares of x as ssac_sumsq_blocks() partials (feed ssac_clip_coef / ssac_group_norms) */
int ssac_sumsq_blocks(void);
int ssac_sumsq(const float *x, int64_t n, float *out_partials, void *stream);
/* LayerNorm(eps 1e-5) + tanh (cnns.py:66-68) and its backward (a wave per row, then a workgroup per feature for
 * dgamma / dbeta; dy_scratch n_rows*dim). */
int ssac_ln_tanh_fwd(const float *x, int64_t ldx, const float *gamma, const float *beta, int n_rows, int dim,
                     float *out, int64_t ldo, float *xhat, float *rstd, void *stream);
int ssac_ln_tanh_bwd(const float *d_out, int64_t ldd, const float *out, int64_t ldo, const float *xhat,
                     const float *rstd, const float *gamma, int n_rows, int dim, float *dx, int64_t ldx,
                     float *dy_scratch, float *dgamma, float *dbeta, void *stream);

/* ==== bf16-operand mode (csrc/ssac_bf16.hip): BASELINE.json config 2.  No reference counterpart (the reference is
 * fp32 only, super_sac/__init__.py:3): fp32 master weights / Adam moments / Polyak targets as above, matrix products
 * on v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulate) fed from a bf16 SHADOW arena per net
 *   [ W1 (hidden x K1P, K zero-padded to 16) | W2 | W2^T | W3 ]        (ssac_bf16_layout: stride + the 4 offsets)
 * that the Adam epilogue keeps current.  W1, W2 and W2^T are stored FRAGMENT-MAJOR -- element (n, k) of a matrix with
 * K = 16 * steps columns at (((n / 32) * steps + k / 16) * 64 + n % 32 + 32 * (k / 8 % 2)) * 8 + k % 8 -- so that K-step
 * t of a 32-row block, i.e. one wave-level load of MFMA B operands, is one contiguous KiB; W3 is row-major.
 * Saved activations are bf16 and TRANSPOSED ((n_nets x) hidden x Bp: the batch is the K of the weight-gradient products,
 * Bp = n_rows rounded up to 16, zero padded) and fragment-major in the same way (features = rows n, batch = k; XT holds
 * K1P rounded up to 32 feature rows), so those products read both operands with fully used 16-byte loads.  Covers the chained critic update of continuous single-output critics (hidden % 32 == 0, <= 256). ==== */
int64_t ssac_bf16_layout(int in_dim, int hidden, int out_dim, int64_t offsets[4]);
int ssac_bf16_supported(const ssac_mlp *nets);
/* shadow <- bf16(master) for every net (after construction, load_state_dict, or an fp32 update of the arena) */
int ssac_bf16_sync(const ssac_mlp *nets, uint16_t *shadow, void *stream);
/* learning_utils.py:160-162 on the fp32 masters, and the target's shadow refreshed in the same launch */
int ssac_bf16_polyak(const ssac_mlp *target, const ssac_mlp *source, float tau, uint16_t *target_shadow, void *stream);
/* ensemble-Q forward (agent.py:34 loop + mlps.py:123-129) in bf16: Y (n_sel x n_rows x out) fp32 */
int ssac_bf16_mlp3_fwd(const ssac_mlp *nets, const uint16_t *shadow, const int32_t *net_ids, int n_sel, const float *X,
                       int64_t ldx, int n_rows, float *Y, void *stream);
/* (large batches -- >= 512 row tiles x nets, single-output critics, hidden 256, 17 <= in_dim <= 32 -- take the register-chained
 * kernel: weights in LDS, activations never leave the registers; otherwise the streaming kernel.  Same operands, same rounding
 * points; forcing one for the parity tests: ssac_bf16_fwd_form in ssac_hip_test.h.) */
/* ssac_chain_update in bf16: same roles and arguments; the saved forward / unscaled backward leave as H1T, H2T, DZ2uT,
 * DZ1uT (n_nets x hidden x Bp) and XT (K1P x Bp, the [s|a] tile transposed) instead of fp32 row-major buffers. */
int ssac_bf16_chain_update(const ssac_mlp *actor, const uint16_t *actor_shadow, const float *Xa, int64_t ldxa, int n_rows,
                           const float *eps, float log_std_lo, float log_std_hi, float *x1sa, int64_t ld_x1,
                           int64_t act_col0, float *logp, const ssac_rng *rng, const ssac_mlp *targets,
                           const uint16_t *target_shadow, const int32_t *net_ids, int n_sel, float *Qt,
                           const ssac_mlp *critics, const uint16_t *critic_shadow, const float *Xc, int64_t ldxc,
                           float *Q, uint16_t *H1T, uint16_t *H2T, uint16_t *DZ2uT, uint16_t *DZ1uT, uint16_t *XT,
                           const ssac_gather *gather, const ssac_deferred_logs *deferred /* nullable */,
                           unsigned long long *handoff /* nullable: n_rows x A words, zeroed once -- the producer / consumer
                                                          form of the launch, as ssac_chain_update's (A <= 8) */,
                           void *stream);
/* ssac_mlp_wgrad_all_lossfold in bf16 (all three layers, loss gradient per workgroup, Adam on the fp32 masters, shadow
 * refreshed from the new values, optional Polyak of `target` + its shadow).  sumsq: ssac_bf16_wgrad_tiles() slots per
 * net.  No split-K: every gradient element is accumulated by one wave in a fixed order. */
int ssac_bf16_wgrad_tiles(const ssac_mlp *nets);
int ssac_bf16_wgrad_lossfold(const ssac_mlp *nets, uint16_t *shadow, const uint16_t *XT, const uint16_t *H1T,
                             const uint16_t *H2T, const uint16_t *DZ2uT, const uint16_t *DZ1uT, const float *Q,
                             const float *td, const ssac_td_spec *lazy_td, const float *weight,
                             const ssac_popart *popart /* nullable */, int pop, float denom,
                             float *partials, int n_rows, float *adam_m, float *adam_v, const ssac_adam_ctl *ctl,
                             float *grads /* != NULL: store the fp32 gradients there (arena layout), update nothing:
                                             the clip_grad_norm_ path, followed by ssac_clip_coef + ssac_adam_step +
                                             ssac_bf16_sync */,
                             float *sumsq, int64_t sumsq_net_stride, float *target, uint16_t *target_shadow, float tau,
                             const ssac_logfold *logfold, void *stream);

/* ==== the ACTING path (agent.py:204-315: Agent.forward / Agent.sample_action; csrc/ssac_act.hip) ====
 * One observation per environment in, one action out: the cost of a call is its launches and its two trips over PCIe, not
 * its arithmetic.  ssac_act bundles what removes them: an observation buffer the HOST writes directly (uncached device
 * memory behind the large BAR, else pinned host memory), a result buffer the DEVICE writes directly (pinned host memory, a
 * sequence word behind it), a device-resident call counter -- the draw number of the engine's noise stream
 * (ssac_rng.counter = ssac_act_counter) -- and recorded launch lists.
 *   plan:  a = ssac_act_create(obs_bytes, out_floats); ssac_record_begin(); <the rule's launches reading ssac_act_obs(a),
 *          ssac_rng.counter = ssac_act_counter(a)>; ssac_act_publish(a, result, n, stream) LAST; which = ssac_act_add_list(a,
 *          ssac_record_end())            (an agent that draws a random actor per call records one list per actor)
 *   call:  ssac_act_run(a, which, obs_host, obs_bytes, out_host, out_floats, stream) = memcpy + sfence, re-issue the list,
 *          spin on the sequence word (bounded: 5 s -> error), memcpy.  No hipMemcpy, no stream synchronisation.
 * The lists are owned by the plan (freed by ssac_act_destroy). */
typedef struct ssac_act ssac_act;
ssac_act *ssac_act_create(int obs_bytes, int out_floats);
void *ssac_act_obs(ssac_act *a);                 /* DEVICE pointer of the observation buffer */
const int64_t *ssac_act_counter(ssac_act *a);    /* DEVICE pointer of the call counter (advanced by ssac_act_publish) */
int ssac_act_publish(ssac_act *a, const float *src, int n, void *stream);
int ssac_act_add_list(ssac_act *a, ssac_launch_list *list);   /* returns the list's index, < 0 on error */
int ssac_act_run(ssac_act *a, int which, const void *obs_host, int obs_bytes, float *out_host, int out_floats, void *stream);
long long ssac_act_calls(const ssac_act *a);
void ssac_act_destroy(ssac_act *a);
/* SUNRISE's UCB rule (agent.py:262-300) on stacked candidates: q_members[c] (HOST array of n_members <= 8 device pointers) =
 * member c's critics on row (a n_rows + b) of X, (n_nets x n_cand n_rows); a candidate's value per member = min over its nets
 * (agent.Critic.forward), score = mean over the members + bonus * unbiased std, best = first arg-max over the candidates;
 * act (n_rows x act_dim) = columns [col0, col0 + act_dim) of X's row (best n_rows + b), clamped to [-1, 1]. */
int ssac_ucb_select(const float *const *q_members, int n_members, int n_nets, int n_cand, int n_rows, float bonus,
                    const float *X, int64_t ldx, int col0, int act_dim, float *act, void *stream);
/* the UCB rule's candidates from the head outputs of n_actors PACKED actors (outs: n_actors x n_rows x 2 act_dim, one
 * ssac_mlp3_fwd_fused over the pack): row (e n_rows + b) of X = [S_rows[b] | tanh(mu + sd eps)], eps = element (b, i) of `rng`'s
 * stream at draw rng->offset + e member_stride (+ *rng->counter) -- what ssac_actor_sample_concat_fused with that offset draws */
int ssac_act_candidates(const float *outs, int n_actors, int n_rows, int act_dim, const float *S_rows, int64_t lds,
                        int state_dim, float log_std_lo, float log_std_hi, const ssac_rng *rng, long long member_stride,
                        float *X, int64_t ldx, void *stream);
/* greedy continuous action (agent.py:204-246): act = clamp(mean over the actors of tanh(outs[e][b][k]), -1, 1) */
int ssac_act_mean_tanh(const float *const *outs, int n_actors, int64_t ld_out, int n_rows, int act_dim, float *act,
                       void *stream);
/* act (n_rows x act_dim) = clamp(columns [col0, col0 + act_dim) of src, lo, hi) */
int ssac_act_take_clamp(const float *src, int64_t ld, int col0, int n_rows, int act_dim, float lo, float hi, float *act,
                        void *stream);
/* discrete actors (agent.py:218-226, 301-309): sample = 0: arg-max of the mean over the actors of softmax(outs[e]); sample = 1
 * (one actor): Categorical(logits).sample() by inversion with one uniform per row from `rng`'s Philox stream.  act (n_rows):
 * the action index as a float. */
int ssac_act_discrete(const float *const *outs, int n_actors, int64_t ld_out, int n_rows, int n_actions, int sample,
                      const ssac_rng *rng, float *act, void *stream);

/* zero a float buffer (log accumulators) */
int ssac_zero(float *p, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SSAC_HIP_H */
