"""Seeded synthetic inputs shared by the golden generator, the parity tests and bench.py.

Everything is derived from numpy ``RandomState`` seeds so the fixtures under
``tests/golden`` only need to store the reference's *outputs* (plus the host-RNG
draws that the reference consumed), never the 100k-row buffers or the weights.
"""
import numpy as np

# name -> configuration of one parity case.  Values follow SURVEY.md Appendix C.
# the cases that run at BASELINE.json's full pixel sizes (seconds per update on the CPU): the GPU suite runs them once each,
# not once per launch form
FULL_SIZE = ("drqv2_pixels_full", "atari_pixels_full")

CASES = {
    # small REDQ: full post-update parameter dump is stored
    "redq_small": dict(obs=17, act=6, hidden=64, N=4, n=2, E=1, B=128, rows=2000, cap=4096,
                       lo=-5.0, hi=2.0, popart=False, pop=False, discrete=False,
                       actor="stochastic", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                       clip=None, tau=0.005, weight_type=None, temp=None, noise=None,
                       cycles=2, utd=3, target_delay=2, seed=11),
    # the metric shape (BASELINE.json): obs 17 / act 6 / B 512 / N 10 / n 2 / H 256
    "redq_M": dict(obs=17, act=6, hidden=256, N=10, n=2, E=1, B=512, rows=5000, cap=8192,
                   lo=-5.0, hi=2.0, popart=False, pop=False, discrete=False,
                   actor="stochastic", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                   clip=None, tau=0.005, weight_type=None, temp=None, noise=None,
                   cycles=2, utd=2, target_delay=2, seed=12),
    # BASELINE config 2's exact shape (HalfCheetah REDQ N=10, batch 256): run in fp32 against the reference and in the
    # bf16-operand mode at the stated bf16 tolerance (tests/test_hip_bf16.py)
    "redq_c2": dict(obs=17, act=6, hidden=256, N=10, n=2, E=1, B=256, rows=5000, cap=8192,
                    lo=-5.0, hi=2.0, popart=False, pop=False, discrete=False,
                    actor="stochastic", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                    clip=None, tau=0.005, weight_type=None, temp=None, noise=None,
                    cycles=2, utd=3, target_delay=2, seed=91),
    # PopArt + pop + gradient clipping on a SAC shape (Agent defaults: ART on, lo=-10)
    "sac_popart": dict(obs=11, act=3, hidden=64, N=2, n=2, E=1, B=64, rows=1000, cap=1024,
                       lo=-10.0, hi=2.0, popart=True, pop=True, discrete=False,
                       actor="stochastic", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                       clip=40.0, tau=0.005, weight_type=None, temp=None, noise=None,
                       cycles=2, utd=2, target_delay=1, seed=13, popart_min_steps=2),
    # SAC-Discrete with vector observations (experiments/gym/sac_discrete.gin shape)
    "sac_discrete": dict(obs=8, act=4, hidden=64, N=2, n=2, E=1, B=64, rows=1000, cap=1024,
                         lo=-10.0, hi=2.0, popart=False, pop=False, discrete=True,
                         actor="discrete", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                         clip=40.0, tau=0.005, weight_type=None, temp=None, noise=None,
                         cycles=2, utd=2, target_delay=2, seed=14),
    # TD3/DrQv2-style: deterministic actor + Gaussian exploration process inside the updates
    "td3_noise": dict(obs=9, act=4, hidden=64, N=2, n=2, E=1, B=64, rows=1000, cap=1024,
                      lo=-10.0, hi=2.0, popart=False, pop=False, discrete=False,
                      actor="deterministic", gamma=0.99 ** 3, lr=1e-4, alpha_lr=0.0,
                      init_alpha=0.0, clip=None, tau=0.01, weight_type=None, temp=None,
                      noise=dict(scale=0.7, clip=0.3), cycles=2, utd=1, target_delay=1, seed=15),
    # SUNRISE: 3 members x 2 critics, sigmoid backup weights
    "sunrise": dict(obs=11, act=3, hidden=64, N=2, n=2, E=3, B=64, rows=1000, cap=1024,
                    lo=-5.0, hi=2.0, popart=False, pop=False, discrete=False,
                    actor="stochastic", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                    clip=None, tau=0.005, weight_type="sunrise", temp=20.0, noise=None,
                    cycles=1, utd=2, target_delay=2, seed=16),
    # DrQv2 on pixels (dmc/drqv2.gin shape): BigPixelEncoder, deterministic actor + exploration
    # noise, Drqv2Aug with aug_mix 1.0, "no target encoder" (encoder_tau 1.0, target_delay 1), n-step 3
    "drqv2_pixels": dict(obs=50, act=4, hidden=64, N=2, n=2, E=1, B=8, rows=48, cap=64,
                         lo=-10.0, hi=2.0, popart=False, pop=False, discrete=False,
                         actor="deterministic", gamma=0.99 ** 3, lr=1e-4, alpha_lr=0.0, init_alpha=0.0,
                         clip=None, tau=0.01, weight_type=None, temp=None,
                         noise=dict(scale=0.5, clip=0.3), cycles=2, utd=1, target_delay=1, seed=17,
                         pixels=dict(kind="big", channels=9, hw=84, emb=50, enc_lr=1e-4, enc_tau=1.0,
                                     aug="drqv2", aug_mix=1.0)),
    # Atari-style SAC-Discrete on pixels (atari/basic_online.gin shape): SmallPixelEncoder, clips 40,
    # Drqv2Aug with aug_mix 0.9, Polyak'd target encoder
    "atari_pixels": dict(obs=128, act=6, hidden=64, N=2, n=2, E=1, B=8, rows=48, cap=64,
                         lo=-10.0, hi=2.0, popart=False, pop=False, discrete=True,
                         actor="discrete", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                         clip=40.0, tau=0.005, weight_type=None, temp=None, noise=None,
                         cycles=2, utd=2, target_delay=2, seed=18,
                         pixels=dict(kind="small", channels=4, hw=84, emb=128, enc_lr=3e-4, enc_tau=0.01,
                                     aug="drqv2", aug_mix=0.9)),
    # ---- round 3: a TRAINABLE pixel encoder shared by the members of an ensemble (learning.py:47-117 loops the members
    # over one encoder; its gradient is the sum over members, clipped and stepped once): SUNRISE-style E = 2 on the
    # Atari-shaped encoder (discrete, clips 40) and on the DrQ encoder (deterministic actor + exploration noise)
    "ens_pixels_atari": dict(obs=128, act=6, hidden=64, N=2, n=2, E=2, B=8, rows=48, cap=64,
                             lo=-10.0, hi=2.0, popart=False, pop=False, discrete=True,
                             actor="discrete", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                             clip=40.0, tau=0.005, weight_type=None, temp=None, noise=None,
                             cycles=2, utd=2, target_delay=2, seed=92,
                             pixels=dict(kind="small", channels=4, hw=84, emb=128, enc_lr=3e-4, enc_tau=0.01,
                                         aug="drqv2", aug_mix=0.9)),
    "ens_pixels_drq": dict(obs=50, act=4, hidden=64, N=2, n=2, E=2, B=8, rows=48, cap=64,
                           lo=-10.0, hi=2.0, popart=False, pop=False, discrete=False,
                           actor="deterministic", gamma=0.99 ** 3, lr=1e-4, alpha_lr=0.0, init_alpha=0.0,
                           clip=None, tau=0.01, weight_type=None, temp=None,
                           noise=dict(scale=0.5, clip=0.3), cycles=2, utd=1, target_delay=1, seed=93,
                           pixels=dict(kind="big", channels=9, hw=84, emb=50, enc_lr=1e-4, enc_tau=1.0,
                                       aug="drqv2", aug_mix=1.0)),
    # ---- round 5: the two pixel configurations at BASELINE.json's FULL sizes (configs 3 and 4): until now the reference
    # fixtures for pixels were B 8 and the full-size GPU tests compared the implicit-GEMM path with the im2col path
    # (round-4 review, weak 1d).  One environment step each (the CPU reference needs ~10 s per update at these sizes).
    "drqv2_pixels_full": dict(obs=50, act=6, hidden=1024, N=2, n=2, E=1, B=512, rows=640, cap=704,
                              lo=-10.0, hi=2.0, popart=False, pop=False, discrete=False,
                              actor="deterministic", gamma=0.99 ** 3, lr=1e-4, alpha_lr=0.0, init_alpha=0.0,
                              clip=None, tau=0.01, weight_type=None, temp=None,
                              noise=dict(scale=0.5, clip=0.3), cycles=1, utd=2, target_delay=1, seed=117,
                              pixels=dict(kind="big", channels=9, hw=84, emb=50, enc_lr=1e-4, enc_tau=1.0,
                                          aug="drqv2", aug_mix=1.0)),
    "atari_pixels_full": dict(obs=128, act=4, hidden=256, N=2, n=2, E=1, B=1024, rows=1200, cap=1280,
                              lo=-10.0, hi=2.0, popart=False, pop=False, discrete=True,
                              actor="discrete", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                              clip=40.0, tau=0.005, weight_type=None, temp=None, noise=None,
                              cycles=1, utd=2, target_delay=2, seed=118,
                              pixels=dict(kind="small", channels=4, hw=84, emb=128, enc_lr=3e-4, enc_tau=0.01,
                                          aug="drqv2", aug_mix=0.9)),
    # ---- round 2 ----
    # BASELINE config 1: Pendulum-v1 SAC (gym/sac.gin), 2 critics, batch 256, hidden 256
    "pendulum_sac": dict(obs=3, act=1, hidden=256, N=2, n=2, E=1, B=256, rows=2000, cap=4096,
                         lo=-5.0, hi=2.0, popart=False, pop=False, discrete=False,
                         actor="stochastic", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                         clip=None, tau=0.005, weight_type=None, temp=None, noise=None,
                         cycles=2, utd=1, target_delay=2, seed=31),
    # BASELINE config 5's shape: Humanoid obs 376 / act 17, N 16 critics, batch 512 (also the sharded tests)
    "redq_S": dict(obs=376, act=17, hidden=256, N=16, n=2, E=1, B=512, rows=3000, cap=4096,
                   lo=-5.0, hi=2.0, popart=False, pop=False, discrete=False,
                   actor="stochastic", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                   clip=None, tau=0.005, weight_type=None, temp=None, noise=None,
                   cycles=1, utd=2, target_delay=2, seed=32),
    # config 3's MLP stack on its own (DrQv2: embedding 50 + act 6 -> 1024 -> 1024 -> 1, batch 512): the per-layer
    # kernel family at its production size, deterministic actor + exploration noise, n-step-3 discount
    "drqv2_mlp1024": dict(obs=50, act=6, hidden=1024, N=2, n=2, E=1, B=512, rows=2000, cap=4096,
                          lo=-10.0, hi=2.0, popart=False, pop=False, discrete=False,
                          actor="deterministic", gamma=0.99 ** 3, lr=1e-4, alpha_lr=0.0, init_alpha=0.0,
                          clip=None, tau=0.01, weight_type=None, temp=None,
                          noise=dict(scale=0.5, clip=0.3), cycles=1, utd=2, target_delay=1, seed=33),
    # config 4's MLP stack on its own (Atari: embedding 128 -> 256 -> 256 -> |A| = 4, batch 1024), clips 40
    "atari_mlp_b1024": dict(obs=128, act=4, hidden=256, N=2, n=2, E=1, B=1024, rows=3000, cap=4096,
                            lo=-10.0, hi=2.0, popart=False, pop=False, discrete=True,
                            actor="discrete", gamma=0.99 ** 3, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                            clip=40.0, tau=0.005, weight_type=None, temp=None, noise=None,
                            cycles=1, utd=2, target_delay=2, seed=34),
    # "softmax" backup weights (learning_utils.py:383-393; super_sac()'s DEFAULT weight_type, main.py:37-88)
    "softmax_weights": dict(obs=11, act=3, hidden=64, N=2, n=2, E=3, B=64, rows=1000, cap=1024,
                            lo=-5.0, hi=2.0, popart=False, pop=False, discrete=False,
                            actor="stochastic", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                            clip=None, tau=0.005, weight_type="softmax", temp=20.0, noise=None,
                            cycles=1, utd=2, target_delay=2, seed=35),
    # SUNRISE weights for a discrete agent (learning_utils.py:372-377)
    "sunrise_discrete": dict(obs=8, act=4, hidden=64, N=2, n=2, E=3, B=64, rows=1000, cap=1024,
                             lo=-10.0, hi=2.0, popart=False, pop=False, discrete=True,
                             actor="discrete", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                             clip=40.0, tau=0.005, weight_type="sunrise", temp=20.0, noise=None,
                             cycles=1, utd=2, target_delay=2, seed=36),
    # softmax weights for a discrete agent: Categorical.sample() draws are recorded in the fixture (u*_cat*)
    "softmax_discrete": dict(obs=8, act=4, hidden=64, N=2, n=2, E=2, B=64, rows=1000, cap=1024,
                             lo=-10.0, hi=2.0, popart=False, pop=False, discrete=True,
                             actor="discrete", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                             clip=None, tau=0.005, weight_type="softmax", temp=20.0, noise=None,
                             cycles=1, utd=2, target_delay=2, seed=37),
    # online_actor_update(use_baseline=True): the advantage estimator as the actor objective (learning.py:401)
    "sac_baseline": dict(obs=11, act=3, hidden=64, N=2, n=2, E=1, B=64, rows=1000, cap=1024,
                         lo=-5.0, hi=2.0, popart=False, pop=False, discrete=False,
                         actor="stochastic", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                         clip=None, tau=0.005, weight_type=None, temp=None, noise=None,
                         cycles=2, utd=1, target_delay=1, seed=38, use_baseline=True),
    # encoder invariance constraint (learning.py:114-117): half the batch augmented (the fully augmented batch needs
    # its own encoder pass), and the atari shape with every row augmented (the constraint shares the critic's pass)
    "drqv2_pixels_inv": dict(obs=50, act=4, hidden=64, N=2, n=2, E=1, B=8, rows=48, cap=64,
                             lo=-10.0, hi=2.0, popart=False, pop=False, discrete=False,
                             actor="deterministic", gamma=0.99 ** 3, lr=1e-4, alpha_lr=0.0, init_alpha=0.0,
                             clip=None, tau=0.01, weight_type=None, temp=None,
                             noise=dict(scale=0.5, clip=0.3), cycles=1, utd=2, target_delay=1, seed=51,
                             encoder_lambda=0.1,
                             pixels=dict(kind="big", channels=9, hw=84, emb=50, enc_lr=1e-4, enc_tau=1.0,
                                         aug="drqv2", aug_mix=0.5)),
    "atari_pixels_inv": dict(obs=128, act=6, hidden=64, N=2, n=2, E=1, B=8, rows=48, cap=64,
                             lo=-10.0, hi=2.0, popart=False, pop=False, discrete=True,
                             actor="discrete", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1,
                             clip=40.0, tau=0.005, weight_type=None, temp=None, noise=None,
                             cycles=1, utd=2, target_delay=2, seed=52, encoder_lambda=0.5,
                             pixels=dict(kind="small", channels=4, hw=84, emb=128, enc_lr=3e-4, enc_tau=0.01,
                                         aug="drqv2", aug_mix=1.0)),
    # the two remaining actor / exploration combinations of compute_td_targets and online_actor_update
    # (learning_utils.py:329-341, learning.py:392-399): exploration noise on a STOCHASTIC actor (the noise replaces the
    # entropy term) and a deterministic actor WITHOUT a process (entropy of Normal(loc, 1e-4))
    "sac_noise": dict(obs=11, act=3, hidden=64, N=2, n=2, E=1, B=64, rows=1000, cap=1024,
                      lo=-5.0, hi=2.0, popart=False, pop=False, discrete=False,
                      actor="stochastic", gamma=0.99, lr=3e-4, alpha_lr=1e-4, init_alpha=0.1, clip=None, tau=0.01,
                      weight_type=None, temp=None, noise=dict(scale=0.3, clip=0.5), cycles=2, utd=2,
                      target_delay=1, seed=81),
    "ddpg_plain": dict(obs=9, act=4, hidden=64, N=2, n=2, E=1, B=64, rows=1000, cap=1024,
                       lo=-10.0, hi=2.0, popart=False, pop=False, discrete=False,
                       actor="deterministic", gamma=0.99, lr=1e-4, alpha_lr=1e-4, init_alpha=0.05, clip=10.0,
                       tau=0.01, weight_type=None, temp=None, noise=None, cycles=2, utd=2, target_delay=2, seed=85),
}


# ---- round 6: a checkpoint written by the REFERENCE (Agent.save after the base case's whole update sequence, agent.py:172-
# 195) is loaded into a freshly initialised agent of ANOTHER seed (Agent.load, agent.py:196-202), the target agent is a deepcopy
# of the loaded one and the optimizers are fresh (what main.super_sac does with a loaded agent, main.py:188-239, 321) -- then
# the base case's update schedule runs on.  The fixture holds the checkpoint's state dicts as arrays ("ckpt|<file>|<key>")
# next to the usual records, so the GPU test loads what the reference wrote, not what this package wrote.
for _base, _seed in (("redq_small", 211), ("sac_popart", 213), ("sac_discrete", 214), ("atari_pixels", 218)):
    CASES[f"ckpt_{_base}"] = dict(CASES[_base], seed=_seed, resume=_base)


# advantage-filtered BC (AWAC/AFBC actor update) with prioritised replay: SURVEY.md 8(f) rank 1.
# steps: (per, filter_) of consecutive learning.offline_actor_update calls (learning.py:144-219)
AFBC_CASES = {
    "afbc_awac": dict(obs=11, act=3, hidden=64, N=2, n=2, E=1, B=64, rows=600, cap=1024, lo=-10.0, hi=2.0,
                      popart=False, discrete=False, actor="stochastic", lr=3e-4, clip=40.0, seed=21,
                      steps=[(False, False), (True, True), (True, True), (False, True), (True, True)]),
    # AWAC-style offline phase: critic updates refresh the priorities too (main.py:403-404 update_priorities)
    "awac_offline": dict(obs=11, act=3, hidden=64, N=2, n=2, E=1, B=64, rows=600, cap=1024, lo=-10.0, hi=2.0,
                         popart=False, discrete=False, actor="stochastic", lr=3e-4, clip=None, seed=23,
                         gamma=0.99, tau=0.005, init_alpha=0.1,
                         steps=[(False, False), "critic", (True, True), "critic", (True, True)]),
    # gym/awac_discrete.gin shape: SAC-Discrete critics, categorical actor, "indirect" advantage
    "afbc_discrete": dict(obs=8, act=4, hidden=64, N=2, n=2, E=1, B=64, rows=600, cap=1024, lo=-10.0, hi=2.0,
                          popart=False, discrete=True, actor="discrete", lr=3e-4, clip=40.0, seed=24,
                          steps=[(False, False), (True, True), (True, True), (False, True)]),
    # d4rl/basic_afbc.gin: critic updates with the DR3 regulariser (dr3_coeff 0.01) between filtered-BC updates
    "d4rl_afbc_dr3": dict(obs=11, act=3, hidden=64, N=2, n=2, E=1, B=64, rows=600, cap=1024, lo=-10.0, hi=2.0,
                          popart=False, discrete=False, actor="stochastic", lr=3e-4, clip=None, seed=25,
                          gamma=0.99, tau=0.005, init_alpha=0.1, dr3=0.01,
                          steps=["critic", (True, True), "critic", "critic", (True, True)]),
    "afbc_noclip": dict(obs=17, act=6, hidden=64, N=3, n=2, E=1, B=128, rows=900, cap=1024, lo=-5.0, hi=2.0,
                        popart=False, discrete=False, actor="stochastic", lr=1e-3, clip=None, seed=22,
                        steps=[(True, True), (True, True), (True, False)]),
}

# BC warm-up on pixels (fixtures written by oracle/gen_golden.py::run_bc_pixels_case)
BC_PIXEL_CASES = {
    # the behavioural-cloning warm-up of dmc/bc_from_pixels.gin (main.py:292-312): offline_actor_update with
    # update_encoder=True, filter_=False, per=False -- the pixel encoder is trained through the BC loss
    "bc_pixels": dict(obs=50, act=4, hidden=64, N=2, n=2, E=1, B=8, rows=48, cap=64, lo=-10.0, hi=2.0,
                      popart=False, discrete=False, actor="stochastic", lr=1e-4, clip=None, seed=61, bc_pixels=True,
                      steps=[(False, False)] * 3, enc_clip=(None, 0.5, 0.5),
                      pixels=dict(kind="big", channels=9, hw=84, emb=50, enc_lr=1e-4, enc_tau=1.0, aug="drqv2",
                                  aug_mix=0.9)),
    # round 3: two ensemble members behind ONE trainable encoder (its gradient is the sum over the members' BC losses)
    "bc_pixels_ens": dict(obs=64, act=5, hidden=64, N=2, n=2, E=2, B=8, rows=48, cap=64, lo=-10.0, hi=2.0,
                          popart=False, discrete=True, actor="discrete", lr=3e-4, clip=5.0, seed=94,
                          bc_pixels=True, steps=[(False, False)] * 3, enc_clip=(5.0, None, 5.0),
                          pixels=dict(kind="small", channels=4, hw=84, emb=64, enc_lr=3e-4, enc_tau=1.0,
                                      aug="drqv2", aug_mix=0.9)),
    "bc_pixels_discrete": dict(obs=64, act=5, hidden=64, N=2, n=2, E=1, B=8, rows=48, cap=64, lo=-10.0, hi=2.0,
                               popart=False, discrete=True, actor="discrete", lr=3e-4, clip=5.0, seed=62,
                               bc_pixels=True, steps=[(False, False)] * 3, enc_clip=(5.0, 5.0, 5.0),
                               pixels=dict(kind="small", channels=4, hw=84, emb=64, enc_lr=3e-4, enc_tau=1.0,
                                           aug="drqv2", aug_mix=0.9)),
}

# action invariance constraint in offline_actor_update (fixtures written by oracle/gen_golden.py::run_actor_inv_case)
ACTOR_INV_CASES = {
    "actinv_vec": dict(obs=17, act=6, hidden=64, N=2, n=2, E=2, B=128, rows=1500, cap=2048, lo=-5.0, hi=2.0,
                       popart=False, discrete=False, actor="stochastic", lr=1e-3, seed=71, actor_lambda=0.5,
                       steps=[dict(clip=None, update_encoder=False), dict(clip=0.5, update_encoder=False),
                              dict(clip=0.5, update_encoder=True)]),
    "actinv_discrete": dict(obs=12, act=5, hidden=64, N=2, n=2, E=1, B=96, rows=1000, cap=1024, lo=-10.0, hi=2.0,
                            popart=False, discrete=True, actor="discrete", lr=1e-3, seed=72, actor_lambda=0.01,
                            steps=[dict(clip=None, update_encoder=False)] * 3),
    # round 3: the constraint (and the BC loss) on a DETERMINISTIC actor -- the reference's Normal(tanh(out), 1e-4),
    # distributions.py:107-114; a = loc at the original observation, no draw; two members behind one trainable encoder
    "actinv_det_pixels": dict(obs=50, act=4, hidden=64, N=2, n=2, E=2, B=8, rows=48, cap=64, lo=-5.0, hi=2.0,
                              popart=False, discrete=False, actor="deterministic", lr=1e-4, seed=95, actor_lambda=1e-9,
                              steps=[dict(clip=None, update_encoder=True, enc_clip=None),
                                     dict(clip=None, update_encoder=False, enc_clip=1.0),
                                     dict(clip=1.0, update_encoder=True, enc_clip=1.0)],
                              pixels=dict(kind="big", channels=9, hw=84, emb=50, enc_lr=1e-4, enc_tau=1.0, aug="drqv2",
                                          aug_mix=0.5)),
    "actinv_pixels": dict(obs=50, act=4, hidden=64, N=2, n=2, E=1, B=8, rows=48, cap=64, lo=-5.0, hi=2.0,
                          popart=False, discrete=False, actor="stochastic", lr=1e-4, seed=73, actor_lambda=1.0,
                          steps=[dict(clip=None, update_encoder=True, enc_clip=None),
                                 dict(clip=None, update_encoder=False, enc_clip=1.0),
                                 dict(clip=1.0, update_encoder=True, enc_clip=1.0)],
                          pixels=dict(kind="big", channels=9, hw=84, emb=50, enc_lr=1e-4, enc_tau=1.0, aug="drqv2",
                                      aug_mix=0.5)),
}

# Markov state-abstraction update (fixtures written by oracle/gen_golden.py::run_markov_case)
MARKOV_CASES = {
    # ---- Markov state-abstraction update (learning.py:266-341): steps of markov_state_abstraction_update
    "markov_vec": dict(obs=17, act=6, hidden=64, N=2, n=2, E=1, B=128, rows=1500, cap=2048, lo=-10.0, hi=2.0,
                       popart=False, pop=False, discrete=False, actor="stochastic", seed=41, lr=1e-3,
                       markov=dict(steps=4, coeffs=(1.0, 1.0, 10.0), max_dist=0.01, clip=(None, None, 0.5, 0.5))),
    "markov_discrete": dict(obs=12, act=5, hidden=64, N=2, n=2, E=1, B=96, rows=1000, cap=1024, lo=-10.0, hi=2.0,
                            popart=False, pop=False, discrete=True, actor="discrete", seed=42, lr=1e-3,
                            markov=dict(steps=3, coeffs=(1.0, 0.5, 0.0), max_dist=0.0, clip=(10.0, 10.0, 10.0))),
    # dmc/drqv2_markov.gin's shape in small: BigPixelEncoder trained through the Markov loss, DrQv2 shift, coeffs 1/1/10
    "markov_pixels": dict(obs=50, act=4, hidden=64, N=2, n=2, E=1, B=8, rows=48, cap=64, lo=-10.0, hi=2.0,
                          popart=False, pop=False, discrete=False, actor="deterministic", seed=43, lr=1e-4,
                          markov=dict(steps=3, coeffs=(1.0, 1.0, 10.0), max_dist=0.01, clip=(None, 1.0, 1.0)),
                          pixels=dict(kind="big", channels=9, hw=84, emb=50, enc_lr=1e-4, enc_tau=1.0,
                                      aug="drqv2", aug_mix=1.0)),
    # visgrid/train.py's shape in small: SmallPixelEncoder, discrete actions, inverse + contrastive only, no augmentation
    "markov_pixels_discrete": dict(obs=64, act=4, hidden=64, N=2, n=2, E=1, B=8, rows=48, cap=64, lo=-10.0, hi=2.0,
                                   popart=False, pop=False, discrete=True, actor="discrete", seed=44, lr=3e-4,
                                   markov=dict(steps=3, coeffs=(1.0, 1.0, 0.0), max_dist=0.0, clip=(None, None, 2.0)),
                                   pixels=dict(kind="small", channels=4, hw=84, emb=64, enc_lr=3e-4, enc_tau=1.0,
                                               aug="identity", aug_mix=0.0)),
}


def synth_transitions(rows, obs_dim, act_dim, discrete=False, seed=1, n_actions=None):
    """obs,next_obs ~ N(0,1); act ~ U(-1,1); rew ~ N(0,1); done ~ Bern(0.01)."""
    rng = np.random.RandomState(seed)
    s = rng.standard_normal((rows, obs_dim)).astype(np.float32)
    s1 = rng.standard_normal((rows, obs_dim)).astype(np.float32)
    if discrete:
        a = rng.randint(0, n_actions, size=(rows, 1)).astype(np.float32)
    else:
        a = rng.uniform(-1.0, 1.0, size=(rows, act_dim)).astype(np.float32)
    r = rng.standard_normal((rows,)).astype(np.float32)
    d = (rng.uniform(size=(rows,)) < 0.01)
    return {"obs": s}, a, r, {"obs": s1}, d


def synth_pixel_transitions(rows, channels, hw, n_actions=None, act_dim=None, seed=1):
    rng = np.random.RandomState(seed)
    s = rng.randint(0, 256, size=(rows, channels, hw, hw)).astype(np.uint8)
    s1 = rng.randint(0, 256, size=(rows, channels, hw, hw)).astype(np.uint8)
    if n_actions is not None:
        a = rng.randint(0, n_actions, size=(rows, 1)).astype(np.float32)
    else:
        a = rng.uniform(-1.0, 1.0, size=(rows, act_dim)).astype(np.float32)
    r = rng.standard_normal((rows,)).astype(np.float32)
    d = (rng.uniform(size=(rows,)) < 0.01)
    return {"obs": s}, a, r, {"obs": s1}, d


def fingerprint_indices(numel, k=48, seed=1234):
    """Deterministic sample positions used to fingerprint big tensors in fixtures."""
    rng = np.random.RandomState(seed + numel % 9973)
    return rng.randint(0, numel, size=min(k, numel))
