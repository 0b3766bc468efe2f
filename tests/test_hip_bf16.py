"""GPU: the bf16-operand mode (BASELINE.json config 2; csrc/ssac_bf16.hip).

The reference has no bf16 path (fp32 only, super_sac/__init__.py:3), so these tests pin the mode two ways:
  * kernel level, against an emulation with the SAME rounding points (weights, inputs and hidden activations rounded to
    bf16, products accumulated in fp32): agreement to accumulation-order noise;
  * update sequences, against the REFERENCE fixtures at a separately stated bf16 tolerance (below), plus recorded-list
    replay == eager launches bit for bit.

Stated bf16 tolerances (vs the fp32 reference): TD targets 3e-2 * max(1, |x|); scalar logs and gradient norms 2e-2
relative; parameters / Polyak targets: worst element 2.5 * lr * n_updates (Adam moves a weight by ~lr per update whatever
the gradient's size, so a near-zero gradient whose sign flips under bf16 rounding displaces it by up to 2 lr per update)
with the MEDIAN element within 2e-5; Adam first moments 2e-2 of their largest magnitude, second moments 5e-2.
"""
import copy
import ctypes as C
import math
import random
from itertools import chain

import numpy as np
import pytest
import torch

import case_runner
import synth

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _arena(n_nets, in_dim, hidden, out_dim, seed=0):
    import super_sac_amd as ssa
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(seed)
    ar = ssa.engine.MlpArena(n_nets, in_dim, hidden, out_dim, dev)
    for j in range(n_nets):
        for seg, scale in (("w1", in_dim ** -0.5), ("b1", 0.05), ("w2", hidden ** -0.5), ("b2", 0.05),
                           ("w3", hidden ** -0.5), ("b3", 0.05)):
            v = ar.view(j, seg)
            v.copy_((torch.randn(v.shape, generator=g) * scale).to(dev))
    return ar


@pytest.mark.parametrize("in_dim,hidden,out_dim", [(23, 256, 1), (17, 64, 12), (393, 256, 1), (376, 256, 34), (4, 32, 2)])
def test_shadow_layout_and_sync(in_dim, hidden, out_dim):
    from super_sac_amd._lib import lib
    ar = _arena(3, in_dim, hidden, out_dim).enable_bf16()
    offs = (C.c_int64 * 4)()
    stride = int(lib.ssac_bf16_layout(in_dim, hidden, out_dim, offs))
    k1p = (in_dim + 15) // 16 * 16
    assert list(offs) == [0, hidden * k1p, hidden * k1p + hidden * hidden, hidden * k1p + 2 * hidden * hidden]
    assert stride % 8 == 0 and stride >= offs[3] + out_dim * hidden and ar.shadow.numel() == 3 * stride
    torch.cuda.synchronize()

    def unfrag(flat, n_rows, n_cols):
        """fragment-major -> row-major: element (n, k) lives at (((n // 32) * steps + k // 16) * 64 + n % 32 +
        32 * (k // 8 % 2)) * 8 + k % 8 -- K-step t of a 32-row block is one contiguous KiB (csrc/ssac_bf16.hip)"""
        steps = n_cols // 16
        t = flat.view(n_rows // 32, steps, 2, 32, 8)            # [row block][K step][half][row in block][8 k]
        return t.permute(0, 3, 1, 2, 4).reshape(n_rows, n_cols)  # -> [block, row, step, half, k]
    for j in range(3):
        sh = ar.shadow[j * stride:(j + 1) * stride]
        w1 = unfrag(sh[:hidden * k1p], hidden, k1p)
        assert torch.equal(w1[:, :in_dim], ar.view(j, "w1").to(torch.bfloat16)) and not w1[:, in_dim:].any()
        w2 = ar.view(j, "w2").to(torch.bfloat16)
        assert torch.equal(unfrag(sh[offs[1]:offs[2]], hidden, hidden), w2)
        assert torch.equal(unfrag(sh[offs[2]:offs[3]], hidden, hidden), w2.t())
        assert torch.equal(sh[offs[3]:offs[3] + out_dim * hidden].view(out_dim, hidden), ar.view(j, "w3").to(torch.bfloat16))


def _emulate(ar, j, x):
    """the kernel's rounding points: bf16 weights / inputs / hidden activations, exact (float64) accumulation"""
    W1, W2, W3 = (_bf(ar.view(j, s)).double() for s in ("w1", "w2", "w3"))
    b1, b2, b3 = (ar.view(j, s).double() for s in ("b1", "b2", "b3"))
    h1 = _bf(torch.relu(_bf(x).double() @ W1.t() + b1).float()).double()
    h2 = _bf(torch.relu(h1 @ W2.t() + b2).float()).double()
    return (h2 @ W3.t() + b3).float()


@pytest.mark.parametrize("in_dim,hidden,out_dim,B", [(23, 256, 1, 512), (23, 256, 1, 100), (17, 64, 12, 128),
                                                      (393, 256, 1, 512), (376, 256, 34, 64), (9, 96, 3, 33)])
def test_ensemble_q_forward_matches_bf16_emulation(in_dim, hidden, out_dim, B):
    from super_sac_amd import engine
    from super_sac_amd._lib import check, lib
    N = 4
    ar = _arena(N, in_dim, hidden, out_dim, seed=1).enable_bf16()
    x = torch.randn(B, in_dim, device="cuda")
    ids = torch.tensor([2, 0, 3], dtype=torch.int32, device="cuda")
    y = torch.empty(3, B, out_dim, device="cuda")
    check(lib.ssac_bf16_mlp3_fwd(C.byref(ar.desc()), ar.shadow.data_ptr(), ids.data_ptr(), 3, x.data_ptr(), in_dim, B,
                                 y.data_ptr(), engine.stream()))
    torch.cuda.synchronize()
    for e, j in enumerate([2, 0, 3]):
        ref = _emulate(ar, j, x)
        # a hidden activation whose fp32 pre-rounding value sits on a bf16 tie may round the other way under a
        # different accumulation order: one bf16 ulp (2^-8 relative) of one of `hidden` terms
        tol = 4e-3 * float(ref.abs().max()) + 1e-4
        assert float((y[e] - ref).abs().max()) <= tol, (e, float((y[e] - ref).abs().max()), tol)
    # and against the fp32 kernel family: bf16-level agreement
    _, _, y32 = engine.mlp_forward(ar, x, in_dim, 0, B, engine.Workspace(x.device), "t", net_ids=ids, n_sel=3, save=False)
    torch.cuda.synchronize()
    assert float((y - y32).abs().max()) <= 4e-2 * float(y32.abs().max()) + 1e-3


@pytest.mark.parametrize("in_dim,B,form", [(23, 8192 + 37, 1), (23, 65536, 1), (17, 40000, 1), (32, 12345, 1), (29, 6600, 1),
                                             (23, 8192 + 37, 0), (23, 4000, 1), (56, 20000, 1)])
def test_large_batch_ensemble_q_streaming_kernel(in_dim, B, form):
    """batches of >= 512 row tiles x nets take a persistent kernel -- form 1: register-chained (hidden 256, 17 <= in_dim <=
    32: weights in LDS, a wave owns 64 rows end to end, no activation leaves the registers), else / form 0: streaming
    (weight fragments in registers, 64-row tiles through LDS): against the bf16 emulation, against the per-tile kernel on
    the same rows, ragged last tile, a net list with an empty (-1) slot"""
    from super_sac_amd import engine
    from super_sac_amd._lib import check, lib
    check(lib.ssac_bf16_fwd_form(form))
    try:
        _large_batch_case(in_dim, B, form)
    finally:
        check(lib.ssac_bf16_fwd_form(1))


def _large_batch_case(in_dim, B, form):
    from super_sac_amd import engine
    from super_sac_amd._lib import check, lib
    N = 10
    torch.manual_seed(in_dim * 100003 + B)
    ar = _arena(N, in_dim, 256, 1, seed=2).enable_bf16()
    x = torch.randn(B, in_dim, device="cuda")
    sel = [3, -1, 0, 9, 5, 1, 2, 8]
    ids = torch.tensor(sel, dtype=torch.int32, device="cuda")
    y = torch.zeros(len(sel), B, 1, device="cuda")
    check(lib.ssac_bf16_mlp3_fwd(C.byref(ar.desc()), ar.shadow.data_ptr(), ids.data_ptr(), len(sel), x.data_ptr(), in_dim,
                                 B, y.data_ptr(), engine.stream()))
    y_tile = torch.zeros_like(y)
    for b0 in range(0, B, 1024):   # calls of <= 1024 rows stay on the per-tile kernel
        n = min(1024, B - b0)
        part = torch.zeros(len(sel), n, 1, device="cuda")
        check(lib.ssac_bf16_mlp3_fwd(C.byref(ar.desc()), ar.shadow.data_ptr(), ids.data_ptr(), len(sel),
                                     x[b0:b0 + n].data_ptr(), in_dim, n, part.data_ptr(), engine.stream()))
        y_tile[:, b0:b0 + n] = part
    torch.cuda.synchronize()
    assert torch.isinf(y[1]).all() and (y[1] > 0).all()
    live = [e for e, j in enumerate(sel) if j >= 0]
    # same bf16 operands, same fp32 products; summation order of the head differs (8 vs 16 lanes per row)
    if form == 1 and 17 <= in_dim <= 32:
        # register-chained kernel: fc2's K-steps carry a permutation of the k values, so the fp32 sums differ in their last
        # bits and a hidden activation on a bf16 tie may round the other way (one bf16 ulp of one term)
        assert float((y[live] - y_tile[live]).abs().max()) <= 4e-3 * float(y_tile[live].abs().max()) + 1e-4
    else:
        assert float((y[live] - y_tile[live]).abs().max()) <= 1e-6 * max(1.0, float(y_tile[live].abs().max()))
    for e in live[:3]:
        ref = _emulate(ar, sel[e], x[:4096])
        tol = 4e-3 * float(ref.abs().max()) + 1e-4
        assert float((y[e, :4096] - ref).abs().max()) <= tol


@pytest.mark.parametrize("in_dim,B,pad,n_sel", [(23, 30000, 9, 3), (20, 70001, 1, 1), (32, 9000, 32, 10)])
def test_register_chained_forward_on_strided_rows_and_other_net_counts(in_dim, B, pad, n_sel):
    """the register-chained kernel reads x with a leading dimension of its own (rows of a wider tensor, not 16-byte aligned),
    for 1 / 3 / 10 selected nets (1 net: 256 workgroups of one net; the last row block of the batch ragged): against the
    float64 emulation and against the streaming kernel on a contiguous copy"""
    from super_sac_amd import engine
    from super_sac_amd._lib import check, lib
    torch.manual_seed(B + pad)
    ar = _arena(10, in_dim, 256, 1, seed=4).enable_bf16()
    big = torch.randn(B, in_dim + pad, device="cuda")
    big[:, in_dim:] = float("nan")   # nothing beyond a row's in_dim floats may reach the result
    big[:, :3] = float("nan") if pad > 8 else big[:, :3]
    off = 3 if pad > 8 else 0
    if off:
        big[:, off:off + in_dim] = torch.randn(B, in_dim, device="cuda")
    x = big[:, off:off + in_dim]
    sel = list(range(10))[:n_sel]
    ids = torch.tensor(sel, dtype=torch.int32, device="cuda")
    y = torch.zeros(n_sel, B, 1, device="cuda")
    check(lib.ssac_bf16_fwd_form(1))
    check(lib.ssac_bf16_mlp3_fwd(C.byref(ar.desc()), ar.shadow.data_ptr(), ids.data_ptr(), n_sel,
                                 big.data_ptr() + 4 * off, in_dim + pad, B, y.data_ptr(), engine.stream()))
    xc = x.contiguous()
    y0 = torch.zeros_like(y)
    check(lib.ssac_bf16_fwd_form(0))
    try:
        check(lib.ssac_bf16_mlp3_fwd(C.byref(ar.desc()), ar.shadow.data_ptr(), ids.data_ptr(), n_sel, xc.data_ptr(), in_dim,
                                     B, y0.data_ptr(), engine.stream()))
    finally:
        check(lib.ssac_bf16_fwd_form(1))
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    assert float((y - y0).abs().max()) <= 4e-3 * float(y0.abs().max()) + 1e-4
    for e in range(min(n_sel, 2)):
        ref = _emulate(ar, sel[e], xc[-3000:])
        assert float((y[e, -3000:] - ref).abs().max()) <= 4e-3 * float(ref.abs().max()) + 1e-4


def test_empty_subset_slot_is_plus_infinity():
    from super_sac_amd import engine
    from super_sac_amd._lib import check, lib
    ar = _arena(2, 23, 64, 1).enable_bf16()
    x = torch.randn(40, 23, device="cuda")
    ids = torch.tensor([1, -1], dtype=torch.int32, device="cuda")
    y = torch.zeros(2, 40, 1, device="cuda")
    check(lib.ssac_bf16_mlp3_fwd(C.byref(ar.desc()), ar.shadow.data_ptr(), ids.data_ptr(), 2, x.data_ptr(), 23, 40,
                                 y.data_ptr(), engine.stream()))
    torch.cuda.synchronize()
    assert torch.isfinite(y[0]).all() and torch.isinf(y[1]).all() and (y[1] > 0).all()


def compare_bf16(rec, fx, cfg, who):
    n_upd = int(fx["n_updates"])
    par_tol = 2.5 * cfg["lr"] * max(n_upd, 1)
    worst = {}
    for key, ref in fx.items():
        if case_runner._INPUT_KEY.fullmatch(key):
            continue
        # EVERY output key of the fixture must be in the record (as case_runner.compare insists for fp32 since round 4: a
        # dropped key used to be skipped silently here unless it was a TD / log key)
        assert key in rec, f"{who}: the record lacks the fixture's output {key}"
        got, ref = np.asarray(rec[key], np.float64), np.asarray(ref, np.float64)
        if "_td" in key:
            dv = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
            assert dv <= 3e-2, f"{who}: {key} deviates {dv:.3e}"
            worst["td"] = max(worst.get("td", 0.0), dv)
        elif "_log:" in key:
            scale = max(1.0, abs(ref)) if "gradients/" not in key else max(1e-6, abs(ref))
            dv = float(abs(got - ref) / scale) if abs(ref) > 1e-12 else float(abs(got))
            assert dv <= 2e-2, f"{who}: {key} = {got} vs reference {ref}"
            worst["log"] = max(worst.get("log", 0.0), dv)
        elif key.endswith("_m") or key.endswith("_v"):
            dv = float(np.max(np.abs(got - ref)) / max(1e-12, np.max(np.abs(ref))))
            assert dv <= (2e-2 if key.endswith("_m") else 5e-2), f"{who}: {key} deviates {dv:.3e} of its scale"
        elif key.startswith("final") and key != "final_log_alpha":
            err = np.abs(got - ref)
            assert float(err.max()) <= par_tol and float(np.median(err)) <= 2e-5, \
                f"{who}: {key} max {err.max():.3e} (tol {par_tol:.1e}) median {np.median(err):.3e}"
            worst["param"] = max(worst.get("param", 0.0), float(err.max()))
        elif key == "final_log_alpha":
            assert np.max(np.abs(got - ref)) <= 1e-4
        elif "_popart" in key:
            assert np.allclose(got, ref, rtol=2e-2, atol=2e-3), f"{who}: {key} {got} vs {ref}"
        else:
            raise AssertionError(f"{who}: fixture key {key} is neither a declared input nor an output compare_bf16 knows")
    return worst


@pytest.mark.parametrize("name", ["redq_small", "pendulum_sac", "redq_M", "redq_c2", "redq_S"])
def test_bf16_update_sequences_against_reference_fixtures(name):
    """critic updates (bf16 chain + bf16 weight gradients + Adam on fp32 masters), Polyak with shadow refresh, fp32
    actor / temperature updates with the actor's shadow re-synced -- against the fp32 reference's outputs."""
    rec = case_runner.run_engine(name, precision="bf16")
    worst = compare_bf16(rec, case_runner.load_fixture(name), synth.CASES[name], f"hip-bf16[{name}]")
    print(f"{name}: worst bf16 deviations vs the fp32 reference {worst}")


@pytest.mark.parametrize("name", ["redq_small", "redq_c2"])
def test_bf16_chained_launch_forms_agree(name):
    """the bf16 chained launch as producer / consumer workgroups (bf_chain_pc_kernel: the actor once per tile, a' handed to
    the tile's target critics as tagged granules, fc1 on the state columns meanwhile) against one workgroup per target
    chain (bf_chain_kernel): both land on the reference fixture, and their first TD targets agree far inside the bf16
    tolerance -- the two forms differ by the fp32 association of fc1's sum only (the products are the same bf16 x bf16)"""
    import super_sac_amd as ssa
    fx, cfg = case_runner.load_fixture(name), synth.CASES[name]
    recs = {}
    old = ssa.learning_utils.CHAIN_PC
    try:
        for pc in (False, True):
            ssa.learning_utils.CHAIN_PC = pc
            recs[pc] = case_runner.run_engine(name, precision="bf16")
            compare_bf16(recs[pc], fx, cfg, f"hip-bf16[{name}, pc={pc}]")
    finally:
        ssa.learning_utils.CHAIN_PC = old
    k0 = sorted(k for k in fx if k.startswith("u0") and "_td" in k)[0]
    a, b = np.asarray(recs[False][k0], np.float64), np.asarray(recs[True][k0], np.float64)
    assert float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(a)))) <= 2e-3, "the two forms' first TD targets"


def test_bf16_with_popart_and_gradient_clipping():
    """sac_popart (PopArt statistics + pop + clip_grad_norm_ 40): the bf16 weight-gradient launch stores fp32 gradients,
    clip + Adam run on the masters, the shadows are refreshed; the PopArt scalars follow the (bf16-perturbed) targets"""
    name = "sac_popart"
    rec = case_runner.run_engine(name, precision="bf16")
    fx = case_runner.load_fixture(name)
    worst = compare_bf16(rec, fx, synth.CASES[name], f"hip-bf16[{name}]")
    for key, ref in fx.items():
        if "_popart" in key:   # mu, nu, w, b, sigma, t
            np.testing.assert_allclose(rec[key], ref, rtol=3e-2, atol=3e-3, err_msg=key)
    print(f"{name}: worst bf16 deviations vs the fp32 reference {worst}")


def test_bf16_mode_refuses_what_it_does_not_cover():
    with pytest.raises(NotImplementedError):
        case_runner.run_engine("drqv2_mlp1024", precision="bf16")  # hidden 1024


def _run20(ssa, use_lists, precision):
    L = ssa.learning
    old = L.USE_GRAPHS
    L.USE_GRAPHS = use_lists
    try:
        torch.manual_seed(3); np.random.seed(3); random.seed(3)
        dev = torch.device("cuda")
        agent = ssa.Agent(act_space_size=6, encoder=ssa.nets.IdentityEncoder(17),
                          actor_network_cls=ssa.nets.ContinuousStochasticActor,
                          critic_network_cls=ssa.nets.ContinuousCritic, ensemble_size=1, num_critics=4,
                          hidden_size=64, auto_rescale_targets=False, log_std_low=-5.0, log_std_high=2.0)
        agent.to(dev)
        ssa.set_precision(agent, precision)
        target = copy.deepcopy(agent)
        buf = ssa.replay.ReplayBuffer(4096, device=dev)
        buf.load_experience(*synth.synth_transitions(2000, 17, 6, seed=5))
        copt = torch.optim.Adam(chain(*(c.parameters() for c in agent.critics)), lr=3e-4)
        eopt = torch.optim.Adam(agent.encoder.parameters(), lr=1e-4)
        la = torch.Tensor([math.log(0.1)]).to(dev); la.requires_grad = True
        aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.IdentityAug(128)])
        for k in range(20):
            logs, dicts = ssa.learning.critic_update(
                buffer=buf, agent=agent, target_agent=target, critic_optimizer=copt, encoder_optimizer=eopt,
                log_alphas=[la], batch_size=128, gamma=0.99, critic_clip=None, encoder_clip=None,
                target_critic_ensemble_n=2, weighted_bellman_temp=None, weight_type=None, pop=False,
                augmenter=aug, encoder_lambda=0, aug_mix=0.0, discrete=False, random_process=None,
                noise_clip=None, per=False, update_priorities=False, dr3_coeff=0.0)
            if k % 2 == 0:
                ssa.learning_utils.soft_update(target.critics[0], agent.critics[0], 0.005)
        last = (float(logs["losses/critic_overall_loss"]), float(logs["gradients/critic_random_grad"]))
        ar, tar = agent.critics[0].arena(dev), target.critics[0].arena(dev)
        torch.cuda.synchronize()
        return (ar.params.cpu().numpy().copy(), tar.params.cpu().numpy().copy(),
                None if ar.shadow is None else (ar.shadow.float().cpu().numpy().copy(), tar.shadow.float().cpu().numpy().copy()),
                last, ar, tar)
    finally:
        L.USE_GRAPHS = old


def test_bf16_recorded_list_equals_eager_and_shadows_stay_current():
    import super_sac_amd as ssa
    pe, te, se, le, _, _ = _run20(ssa, False, "bf16")
    pl, tl, sl, ll, ar, tar = _run20(ssa, True, "bf16")
    assert np.array_equal(pe, pl) and np.array_equal(te, tl) and le == ll
    assert np.array_equal(se[0], sl[0]) and np.array_equal(se[1], sl[1])
    # after 20 updates + 10 Polyak steps every shadow equals bf16(master): the Adam epilogue / Polyak kernel keep them
    # current without a sync launch
    for a in (ar, tar):
        before = a.shadow.clone()
        a.sync_shadow()
        torch.cuda.synchronize()
        assert torch.equal(before, a.shadow)
    # and the mode is close to fp32 on the same run (sanity: same sign structure, lr-scale differences)
    p32, t32, _, l32, _, _ = _run20(ssa, True, "fp32")
    assert float(np.abs(p32 - pl).max()) <= 2.5 * 3e-4 * 20 and float(np.median(np.abs(p32 - pl))) <= 2e-5
    assert abs(l32[0] - ll[0]) <= 2e-2 * max(1.0, abs(l32[0]))
