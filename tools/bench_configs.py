"""Secondary rows of SURVEY.md 8(d): the vector-observation configurations of BASELINE.json on one MI355X.

    python tools/bench_configs.py        (GPU box; prints a markdown table)

Rows: critic_update (+Polyak/2) at B in {256, 512}, N in {2, 10, 16}, the Humanoid shape (obs 376, act 17, N 16),
and the full REDQ environment step of redq.gin (20 critic updates + 10 Polyak + 1 actor + 1 temperature update).
"""
import copy, math, os, random, sys, time
from itertools import chain
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import super_sac_amd as ssa
import synth
if os.environ.get("SSAC_CHAIN_SPLIT"):
    ssa.learning_utils.CHAIN_SPLIT = os.environ["SSAC_CHAIN_SPLIT"] == "1"
if os.environ.get("SSAC_CHAIN_PC"):   # A/B of the chained launch's two forms (tools only)
    ssa.learning_utils.CHAIN_PC = os.environ["SSAC_CHAIN_PC"] == "1"
if os.environ.get("SSAC_SLOT_BY_VALUE"):   # A/B: replayed launches find their input slot through the feed block (tools only)
    ssa._lib.lib.ssac_slot_by_value(int(os.environ["SSAC_SLOT_BY_VALUE"]))
if os.environ.get("SSAC_WGRAD_VARIANT"):   # A/B of the weight-gradient launch's forms (tools only)
    ssa.engine.set_wgrad_variant(int(os.environ["SSAC_WGRAD_VARIANT"]))
if os.environ.get("SSAC_CHAIN_FORM"):   # A/B: one workgroup per CU (0) / co-resident 16-row tiles where they apply (1) (tools only)
    ssa._lib.check(ssa._lib.lib.ssac_chain_form(int(os.environ["SSAC_CHAIN_FORM"])))

dev = torch.device("cuda")


def build(obs, act, B, N, n, hidden=256, rows=100_000, precision="fp32", as_rank=False):
    """as_rank: the agent is installed as a critic-sharded RANK of a one-rank job (parallel.install + Shard(0, 1, N)): its
    updates take the sharded code path -- the launch list cut at the exchange, the one-shot exchange kernel between the
    chained launch and the weight-gradient launch -- with every subset member owned locally (tools/shard_budget.py)"""
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    agent = ssa.Agent(act_space_size=act, encoder=ssa.nets.IdentityEncoder(obs),
                      actor_network_cls=ssa.nets.ContinuousStochasticActor,
                      critic_network_cls=ssa.nets.ContinuousCritic, ensemble_size=1, num_critics=N,
                      hidden_size=hidden, auto_rescale_targets=False, log_std_low=-5.0, log_std_high=2.0)
    agent.to(dev)
    ssa.set_precision(agent, precision)
    target = copy.deepcopy(agent)
    if as_rank:
        ssa.parallel.install(agent, target, ssa.parallel.Shard(0, 1, N))
    buf = ssa.replay.ReplayBuffer(rows + 1000, device=dev)
    buf.load_experience(*synth.synth_transitions(rows, obs, act, seed=1))
    copt = torch.optim.Adam(chain(*(c.parameters() for c in agent.critics)), lr=3e-4)
    aopt = torch.optim.Adam(chain(*(a.parameters() for a in agent.actors)), lr=3e-4)
    eopt = torch.optim.Adam(agent.encoder.parameters(), lr=1e-4)
    la = torch.Tensor([math.log(0.1)]).to(dev); la.requires_grad = True
    lopt = torch.optim.Adam([la], lr=1e-4, betas=(0.5, 0.999))
    aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.IdentityAug(B)])
    st = {"k": 0}

    def critic():
        logs, dicts = ssa.learning.critic_update(
            buffer=buf, agent=agent, target_agent=target, critic_optimizer=copt, encoder_optimizer=eopt,
            log_alphas=[la], batch_size=B, gamma=0.99, critic_clip=None, encoder_clip=None,
            target_critic_ensemble_n=n, weighted_bellman_temp=None, weight_type=None, pop=False, augmenter=aug,
            encoder_lambda=0, aug_mix=0.0, discrete=False, random_process=None, noise_clip=None, per=False,
            update_priorities=False, dr3_coeff=0.0)
        if st["k"] % 2 == 0:
            ssa.learning_utils.soft_update(target.critics[0], agent.critics[0], 0.005)
        st["k"] += 1
        return dicts

    def env_step():  # redq.gin: UTD 20, then one actor and one temperature update on the last batch
        for _ in range(20):
            dicts = critic()
        ssa.learning.online_actor_update(buffer=buf, agent=agent, pop=False, actor_optimizer=aopt, log_alphas=[la],
                                         batch_size=B, clip=None, random_process=None, noise_clip=None, augmenter=aug,
                                         aug_mix=0.0, premade_replay_dicts=dicts)
        ssa.learning.alpha_update(buffer=buf, agent=agent, optimizers=[lopt], batch_size=B, log_alphas=[la],
                                  augmenter=aug, aug_mix=0.0, target_entropy=-float(act), premade_replay_dicts=dicts,
                                  discrete=False)
    critic.objects = dict(agent=agent, target=target)
    return critic, env_step


def timed(fn, n, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


print("| configuration | precision | obs / act | B | N (n) | us per critic update | critic updates/s |")
print("|---|---|---|---|---|---|---|")
ROWS = [("SAC (sac.gin shape)", 3, 1, 256, 2, 2), ("REDQ", 17, 6, 256, 10, 2),
        ("REDQ (headline)", 17, 6, 512, 10, 2), ("REDQ", 17, 6, 512, 16, 2),
        ("Humanoid", 376, 17, 512, 16, 2), ("Humanoid", 376, 17, 256, 10, 2)]
for precision in ("fp32", "bf16"):
  for name, obs, act, B, N, n in ROWS:
    if len(sys.argv) > 1 and sys.argv[1] not in name:  # optional row filter: python tools/bench_configs.py Humanoid
        continue
    critic, env_step = build(obs, act, B, N, n, precision=precision)
    import gc
    gc.collect(); gc.freeze()   # (as bench.py and INTEGRATION.md: no generation-2 collection pause inside a timed row)
    t = timed(critic, 1500, 200)
    print(f"| {name} | {precision} | {obs} / {act} | {B} | {N} ({n}) | {t * 1e6:.1f} | {1 / t:.0f} |")
    if name in ("REDQ (headline)", "REDQ") and N == 10:
        te = timed(env_step, 60, 5)
        print(f"| full REDQ env step: 20 critic updates + 10 Polyak + actor + temperature | {precision} | {obs} / {act} | {B} | {N} ({n}) "
              f"| {te * 1e6 / 20:.1f} (x20 = {te * 1e3:.2f} ms per env step) | {20 / te:.0f} |")
