"""Headline benchmark: gradient updates/sec of the REDQ critic update (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one ``learning.critic_update`` call on a synthetic replay batch (obs 17, act 6,
batch 512, N=10 critics, n=2 target subset, hidden 256, fp32 -- the shape the metric is
quoted on) followed, every ``target_delay``=2 updates, by the Polyak update of the target
critics, exactly as the reference's UTD loop does (main.py:379-414).  The replay buffer
(100k transitions) is resident in HBM before the timed region starts; the per-update host
work that remains (index draw from the torch CPU generator, REDQ subset draw, a 4 KB index
upload) is part of the path and is inside the timed region.

For N > 1 (launched by torch.distributed.run, one rank per GPU) the critic ensemble is
sharded across ranks (super_sac_amd.parallel): one MIN all-reduce of the (B,) partial
min-Q per update over RCCL; total work is fixed, so scaling is "strong".

Prints ONE JSON line (rank 0) with `roofline` for the ensemble-Q GEMM and `cpu_baseline`
(the CPU oracle -- a port of the reference's update -- timed on this host's cores).
"""
import argparse
import copy
import json
import math
import os
import sys
import time
from itertools import chain

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

OBS, ACT, BATCH, NCRIT, NSUB, HID = 17, 6, 512, 10, 2, 256
ROWS, CAP = 100_000, 1_000_000
GAMMA, LR, TAU, TARGET_DELAY = 0.99, 3e-4, 0.005, 2
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
# HBM bytes per launch from the PMC passes in profiles/r1_pmc_counters.md (FETCH_SIZE doubled per the guide's gfx950
# note for 16-byte streaming reads + WRITE_SIZE, KiB -> bytes); not collected live, N=10 single-GPU shape only
TRAFFIC_FUSED_CRITIC_BYTES = (2 * 13165 + 20527) * 1024
TRAFFIC_DUAL_BYTES = (2 * 13067 + 10403) * 1024  # merged actor + ensemble-Q forward launch (profiles/r1_final_kernel_stats.md)
TRAFFIC_CHAIN_BYTES = (2 * 18909 + 20647) * 1024  # chained launch (profiles/r1_final_kernel_stats.md)
TRAFFIC_FWD_BYTES = None  # the two-launch form (SSAC_SPLIT_FORWARD=1) has no PMC pass yet


def synth_data():
    import synth
    return synth.synth_transitions(ROWS, OBS, ACT, seed=1)


def build_engine(device, n_local, shard=None):
    import super_sac_amd as ssa
    torch.manual_seed(0)
    np.random.seed(0)
    import random
    random.seed(0)
    agent = ssa.Agent(act_space_size=ACT, encoder=ssa.nets.IdentityEncoder(OBS),
                      actor_network_cls=ssa.nets.ContinuousStochasticActor,
                      critic_network_cls=ssa.nets.ContinuousCritic, discrete=False, ensemble_size=1,
                      num_critics=n_local, ucb_bonus=0.0, hidden_size=HID, auto_rescale_targets=False,
                      log_std_low=-5.0, log_std_high=2.0)
    agent.to(device)
    agent.train()
    target = copy.deepcopy(agent)
    buf = ssa.replay.ReplayBuffer(CAP, device=device)
    buf.load_experience(*synth_data())
    copt = torch.optim.Adam(chain(*(c.parameters() for c in agent.critics)), lr=LR, betas=(0.9, 0.999))
    eopt = torch.optim.Adam(agent.encoder.parameters(), lr=1e-4)
    la = torch.Tensor([math.log(0.1)]).to(device)
    la.requires_grad = True
    aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.IdentityAug(BATCH)])
    if shard is not None:
        ssa.parallel.install(agent, target, shard)
    state = {"k": 0}

    def step():
        ssa.learning.critic_update(
            buffer=buf, agent=agent, target_agent=target, critic_optimizer=copt, encoder_optimizer=eopt,
            log_alphas=[la], batch_size=BATCH, gamma=GAMMA, critic_clip=None, encoder_clip=None,
            target_critic_ensemble_n=NSUB, weighted_bellman_temp=None, weight_type=None, pop=False,
            augmenter=aug, encoder_lambda=0, aug_mix=0.0, discrete=False, random_process=None,
            noise_clip=None, per=False, update_priorities=False, dr3_coeff=0.0)
        if state["k"] % TARGET_DELAY == 0:
            for ac, tc in zip(agent.critics, target.critics):
                ssa.learning_utils.soft_update(tc, ac, TAU)
        state["k"] += 1
    return step, ssa


def cpu_baseline(budget_s=12.0):
    """The oracle's critic_update (+Polyak) on the host cores: same shape, same 100k-row buffer."""
    import ssac_oracle as orc
    torch.manual_seed(0)
    buf = orc.ReplayOracle(CAP)
    buf.load_experience(*synth_data())
    oa = orc.AgentOracle(state_dim=OBS, act_dim=ACT, hidden=HID, num_critics=NCRIT, ensemble_size=1,
                         log_std_low=-5.0, log_std_high=2.0, seed=0).requires_grad_(True)
    ot = oa.clone()
    copt = orc.AdamOracle(oa.critic_params(), lr=LR)
    eopt = orc.AdamOracle([], lr=1e-4)
    la = [torch.tensor([math.log(0.1)], requires_grad=True)]
    aug = orc.AugOracle("identity", BATCH)

    def one(k):
        orc.critic_update(buf, oa, ot, copt, eopt, la, BATCH, GAMMA, None, None, NSUB, None, None, False, aug)
        if k % TARGET_DELAY == 0:
            orc.soft_update(ot.critic_params(), oa.critic_params(), TAU)
    # pick the thread count that is FASTEST for this workload on this host (tiny GEMMs do not
    # scale to hundreds of threads; the baseline should be the CPU's best, not its worst)
    best = (float("inf"), 1)
    for th in sorted({1, 4, 8, 16, 32, min(64, os.cpu_count() or 1)}):
        if th > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(th)
        one(0)
        t1 = time.perf_counter()
        for k in range(3):
            one(k)
        best = min(best, ((time.perf_counter() - t1) / 3, th))
    threads = best[1]
    torch.set_num_threads(threads)
    for k in range(3):
        one(k)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        one(n)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 2), "unit": "updates/s", "cores": threads, "kind": "port",
            "sample": f"{n} critic updates (+Polyak every 2nd) of the same workload in {dt:.1f} s, "
                      f"torch {torch.__version__} CPU, {threads} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py measures the HIP path; it needs an MI355X"
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)  # (the single-GPU smoke test of the N>1 path runs 2 ranks on 1 device)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    shard = None
    n_local = NCRIT
    # SSAC_BENCH_FORCE_DIST=1: take the sharded path (process group, MIN all-reduce between the recorded segments) with
    # a single rank too -- the RCCL check that can run on a 1-GPU box
    if world > 1 or os.environ.get("SSAC_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SSAC_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        from super_sac_amd import parallel
        shard = parallel.Shard(rank, world, NCRIT)
        n_local = shard.n_local
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    step, ssa = build_engine(device, n_local, shard)
    # Python's cyclic GC otherwise runs a full (generation-2) collection over the whole torch object graph every few
    # hundred updates -- a 40-80 ms pause, i.e. hundreds of updates: park the start-up objects in the permanent
    # generation (host runtime hygiene of a long-running training loop; nothing the update path allocates is cyclic)
    import gc
    gc.collect()
    gc.freeze()
    for _ in range(args.warmup):
        step()

    # ---- timed region: EXACTLY --steps steps between barrier+sync brackets
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)

    # ---- roofline of the dominant kernel: the ensemble-Q kernel = fused forward (fc1+fc2+head, activations in
    # LDS) of all local critics on the sampled batch, one launch per update.  Algorithmic FLOPs per launch
    # (SURVEY 8(d)): 2*B*N*(in*H + H*H + H*out).  Its backward half (loss gradient, head backward, fc2
    # backward-data: 2*B*N*(out*H + H*H) FLOPs) is a second launch of the same kernel template and is reported
    # beside it.  The timed region re-issues the update from ONE recorded launch list, inside which a single kernel is not
    # bracketed by events; so the same update is run again right here with plain launches and the launches are
    # bracketed by HIP events recorded on the stream each one is launched on (same shapes, buffers, binary).
    graphs_were_on = ssa.learning.USE_GRAPHS
    ssa.learning.USE_GRAPHS = False
    ssa.engine.PROFILE["tag"] = ("critic_fwd", "critic_bwd", "critic_fused", "dual_fwd", "dual_bwd", "chain")
    ssa.engine.PROFILE["events"] = []
    ssa.engine.PROFILE["reps"] = 8   # the bracketed (idempotent) launch is issued 8x per event pair
    for _ in range(min(args.steps, 300)):
        step()
    torch.cuda.synchronize()
    ssa.engine.PROFILE["tag"] = None
    ssa.engine.PROFILE["reps"] = 1
    ssa.learning.USE_GRAPHS = graphs_were_on
    by_tag = {}
    for a, b, tag, reps in ssa.engine.PROFILE["events"]:
        by_tag.setdefault(tag, []).append(a.elapsed_time(b) / reps)
    IN = OBS + ACT
    f_fwd = 2.0 * BATCH * n_local * (IN * HID + HID * HID + HID)
    f_bwd = 2.0 * BATCH * n_local * (HID + HID * HID)
    f_actor = 2.0 * BATCH * (OBS * HID + HID * HID + HID * 2 * ACT)
    f_tgt_ = 2.0 * BATCH * NSUB * (IN * HID + HID * HID + HID)
    if "chain" in by_tag:
        ms = by_tag["chain"]
        flops, kname = f_fwd + f_bwd + f_tgt_ + NSUB * f_actor, (
            "fused_chain_kernel: ensemble-Q forward (fc1+fc2+head, h1/h2/q saved) AND the TD-independent half of the "
            "backward pass (head backward + fc2 backward-data) of all local critics as 32-row workgroups, beside the "
            "target chains (actor forward + tanh-normal sample -> target critic, per REDQ subset slot) as 16-row "
            "workgroups; every workgroup gathers its own replay rows; ONE launch per update")
    elif "dual_fwd" in by_tag:
        ms = by_tag["dual_fwd"]
        flops, kname = f_fwd + f_actor, (
            "fused_dual_kernel: ensemble-Q forward (fc1+fc2+head, h1/h2/q saved) of all local critics as 32-row "
            "workgroups + the actor forward with tanh-normal sample as 16-row workgroups, each workgroup gathering its own "
            "replay rows, ONE launch per update")
    elif "critic_fwd" in by_tag:
        ms = by_tag["critic_fwd"]
        flops, kname = f_fwd, ("fused_mlp_kernel<plain>: ensemble-Q forward (fc1+fc2+head) of all local critics, "
                               "h1/h2/q stored for the backward launches (one launch per update)")
    else:  # the one-launch forward+backward form (learning.SPLIT_FORWARD off / chip-filling ensembles)
        ms = by_tag["critic_fused"]
        flops, kname = f_fwd + f_bwd, ("fused_mlp_kernel<critic>: forward of all local critics + loss gradient + "
                                       "head backward + fc2 backward-data (one launch per update)")
    avg_ms = sum(ms) / len(ms)
    achieved = flops / (avg_ms * 1e-3) / 1e12
    roofline = {"kernel": kname,
                "bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                "avg_launch_us": round(avg_ms * 1e3, 3), "launches_timed": len(ms),
                "timing": "HIP events on the launch stream around 8 back-to-back issues of the (idempotent) launch, "
                          "in an eager pass right after the timed (replayed) region",
                "flops_per_launch": flops,
                "traffic": (None if not (world == 1 and n_local == NCRIT) else
                            (TRAFFIC_CHAIN_BYTES if "chain" in by_tag else TRAFFIC_DUAL_BYTES if "dual_fwd" in by_tag else
                             (TRAFFIC_FWD_BYTES if "critic_fwd" in by_tag else TRAFFIC_FUSED_CRITIC_BYTES)))}
    if "dual_bwd" in by_tag:
        mb = by_tag["dual_bwd"]
        avg_b = sum(mb) / len(mb)
        f_tgt = 2.0 * BATCH * NSUB * (IN * HID + HID * HID + HID)
        roofline["backward_launch"] = {
            "kernel": "fused_dual2_kernel: TD-independent half of the critics' backward pass (head backward + fc2 "
                      "backward-data, unscaled) as 32-row workgroups + the target critics' forward on the REDQ subset",
            "avg_launch_us": round(avg_b * 1e3, 3), "flops_per_launch": f_bwd + f_tgt,
            "achieved": round((f_bwd + f_tgt) / (avg_b * 1e-3) / 1e12, 3)}
    elif "critic_bwd" in by_tag:
        mb = by_tag["critic_bwd"]
        avg_b = sum(mb) / len(mb)
        roofline["backward_launch"] = {"kernel": "fused_mlp_kernel<critic-bwd>: loss gradient + head backward + "
                                                 "fc2 backward-data on the saved forward",
                                       "avg_launch_us": round(avg_b * 1e3, 3), "flops_per_launch": f_bwd,
                                       "achieved": round(f_bwd / (avg_b * 1e-3) / 1e12, 3)}

    if rank == 0:
        out = {"metric": "gradient updates/sec (REDQ N=10, batch 512)", "value": round(args.steps / dt, 2),
               "unit": "updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 5), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "REDQ critic_update + Polyak/2: obs 17, act 6, batch 512, "
                                      "N=10 critics (n=2 target subset), hidden 256, replay 100k rows in HBM",
                          "global_batch": BATCH, "num_critics": NCRIT,
                          "launch": ("plain launches" if not graphs_were_on else
                                     ("recorded launch list (ssac_replay)" + ("" if world == 1 else
                                      ", MIN all-reduce between two segments")
                                      if ssa.learning.LAUNCH_MODE == "list" and (world == 1 or ssa.learning.SHARDED_LISTS)
                                      else ("HIP graph replay" if world == 1 else "plain launches"))),
                          "parallelism": "single GPU" if world == 1 else f"critic-ensemble sharded x{world}"},
               "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
