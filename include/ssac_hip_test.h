/*
 * ssac_hip_test.h -- NOT part of the drop-in boundary (that is ssac_hip.h).
 *
 * Two groups of entry points that libssac_hip.so carries for the repository's own parity tests and measurement scripts:
 *
 *   1. FORM SELECTION (always exported).  Where the library has more than one kernel form for the same arithmetic it picks
 *      one by itself from the shapes; these process-global switches force a form so that tests/ can run EVERY form a
 *      configuration can take against the reference fixtures, and tools/ can time one against the other.  They are not
 *      thread-safe, change nothing but the launch selection / workgroup placement, and no product code path
 *      (super_sac_amd/learning*.py, agent.py, replay.py) calls them.
 *
 *   2. LAB HOOKS (#ifdef SSAC_LAB: `./build.sh --lab` -> libssac_hip_lab.so).  Phase stamps, per-workgroup timelines and the
 *      exchange protocol's failing-first switch.  The product library does not define these symbols at all.
 */
#ifndef SSAC_HIP_TEST_H
#define SSAC_HIP_TEST_H

#include "ssac_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- 1. form selection ------------------------------------------------------------------------------------------- */

/* ssac_step_run hands the replayed launches their input slot by value (ssac_hip.h); 0 switches that off (the launches go
 * through the feed block, as plain ssac_replay and hipGraph captures always do).  Bit-identical results
 * (tests/test_hip_cases.py). */
int ssac_slot_by_value(int on);
/* default 1: the merged weight-gradient launch uses its lean kernel (16-byte operand loader only) whenever both problems
 * qualify; 0 = always the general kernel.  Bit-identical results. */
int ssac_gemm_lean(int on);
/* Form of the merged weight-gradient launch (ssac_mlp_wgrad_all / _scaled / _lossfold / _fc12): 0 = automatic, 1 = 64 x 64
 * tiles with the K loop staged through LDS, 2 = the LATENCY form whenever the shapes allow (32 x 32 tiles, K split over the
 * 8 waves, operands straight from memory).  Same sums in another order (fp32 rounding). */
int ssac_wgrad_variant(int variant);
/* row tile of the fused critic launches: 0 = automatic, 16 | 17 | 32 force one (17 = 16 rows with a single weight-staging
 * buffer, two workgroups per CU). */
int ssac_fused_tile_rows(int rows);
/* bit mask, default 2: kernels take their tile in XCD-contiguous order (workgroup b runs on XCD b % 8 --
 * tools/lab/xcc_map.hip; each XCD then works on one contiguous range of (net, tile) ids, so a net's weights / saved
 * activations are pulled into one or two of the eight L2s instead of all of them).  bit 0: the stand-alone fused MLP
 * launches, bit 1: the GEMM / weight-gradient launches.  On unless disabled: the merged weight-gradient launch orders
 * PER WORKGROUP CLASS (an eighth of the fc2 tiles, of the fc1 tiles and of the head workgroups per XCD, so that every
 * XCD carries the same mix of long and short workgroups; bit 2 = off), the chained launch orders each of its halves
 * (bit 3 = off).  0 | 12 = hardware order everywhere.  Placement only: results are bit-identical either way. */
int ssac_xcd_order(int mask);
/* Form of the producer / consumer launch (ssac_chain_update): 1 = when 16-row tiles of the three roles are more than 256 but
 * at most 512 workgroups and every role's co-resident LDS carve fits 80 KB, the launch runs TWO workgroups per CU (16-row
 * tiles, <= 128 VGPRs); 0 = always one workgroup per CU; -1 = the library's default (0: the co-resident form measured
 * slower, profiles/r5_chain_coresident.md).  Outputs are bit-identical to the 16-row tiles of the one-per-CU form
 * (ssac_fused_tile_rows(16)); against its 32-row tiles they differ by fp32 association of the K sums. */
int ssac_chain_form(int form);
/* large-batch form of ssac_bf16_mlp3_fwd: 1 (default) = the register-chained kernel where it applies, 0 = the streaming
 * kernel everywhere.  Same operands, same rounding points. */
int ssac_bf16_fwd_form(int form);

/* ---- 2. lab hooks ------------------------------------------------------------------------------------------------ */
#ifdef SSAC_LAB
/* a device buffer of >= 16 int64: workgroup (0,0) of every fused launch records s_memtime() at its phase boundaries there;
 * NULL (default) disables it. */
int ssac_fused_debug_stamps(long long *dev_buf);
int ssac_gemm_debug_stamps(long long *dev_buf); /* same for the weight-gradient GEMM launches */
/* a device buffer of >= 64 int64: phase stamps of tile 0 of the bf16 launches (slots 0.. actor pass, 16.. target-critic
 * pass, 32.. critic workgroup, 48.. weight-gradient tile); NULL disables */
int ssac_bf16_debug_stamps(long long *dev_buf);
/* dev_buf: 2048 x int64 or NULL.  (start, end) of EVERY workgroup of the chained launch [0, 1024) and of the merged
 * weight-gradient launch [1024, 2048), s_memrealtime ticks (100 MHz): dispatch skew and the slowest workgroup class of a
 * launch (tools/wg_timeline.py). */
int ssac_debug_timeline(long long *dev_buf);
/* the one-shot exchange's protocol switches for the failing-first evidence (tests/test_hip_sharded.py): bit 0 makes this
 * rank's senders skip the slot-reuse wait, bit 1 makes its receivers accept flag >= seq -- 3 is the protocol of round 3,
 * whose owners-only form let senders lap a rank that owned no subset member; bit 2 makes the exchange that runs inside the
 * chained launch treat its wait for the launch's target-critic workgroups as TIMED OUT (the failure path: nothing is sent,
 * the result is NaN, the error word is raised). */
int ssac_xchg_test_mode(ssac_xchg *x, int mode);
#endif /* SSAC_LAB */

#ifdef __cplusplus
}
#endif
#endif /* SSAC_HIP_TEST_H */
