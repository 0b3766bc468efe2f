"""CPU, build container only: the REFERENCE's own objects against the engine's adoption logic and checkpoint layout.

Needs /root/reference (imported unmodified through oracle/ref_harness.py); skipped where it does not exist (the GPU
box).  Nothing here launches a kernel: arenas are built on CPU tensors, which exercises exactly the attribute names,
shapes and parameter views the update path relies on (super_sac/agent.py:13-130, nets/mlps.py:11-185).
"""
import copy
import os

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "super_sac")),
                                reason="the reference tree is only present in the build container")


@pytest.fixture(scope="module")
def ref():
    import ref_harness
    return ref_harness.import_reference()


class _RefIdentityEncoder:
    """experiments/gym/train_gym.py:18-28 rebuilt on the reference's Encoder base (the script needs gym)"""

    def __new__(cls, ref, dim):
        class Enc(ref.nets.Encoder):
            def __init__(self):
                super().__init__()
                self._dim = dim

            @property
            def embedding_dim(self):
                return self._dim

            def forward(self, obs_dict):
                return obs_dict["obs"]
        return Enc()


def _ref_agent(ref, discrete=False, popart=True, E=2, N=3):
    torch.manual_seed(0)
    actor = ref.nets.mlps.DiscreteActor if discrete else ref.nets.mlps.ContinuousStochasticActor
    critic = ref.nets.mlps.DiscreteCritic if discrete else ref.nets.mlps.ContinuousCritic
    return ref.Agent(act_space_size=4, encoder=_RefIdentityEncoder(ref, 9), actor_network_cls=actor,
                     critic_network_cls=critic, discrete=discrete, ensemble_size=E, num_critics=N, hidden_size=32,
                     auto_rescale_targets=popart, log_std_low=-5.0, log_std_high=2.0)


@pytest.mark.parametrize("discrete", [False, True], ids=["continuous", "discrete"])
def test_reference_agent_is_adopted_in_place(ref, discrete):
    import super_sac_amd as ssa
    from super_sac_amd import engine
    agent = _ref_agent(ref, discrete)
    before = [p.detach().clone() for c in agent.critics for p in c.parameters()]
    ssa.adopt_agent(agent, device="cpu")
    in_dim = 9 if discrete else 9 + 4
    out_dim = 4 if discrete else 1
    for critic in agent.critics:
        arena = critic.arena("cpu")
        assert (arena.n_nets, arena.in_dim, arena.hidden, arena.out_dim) == (3, in_dim, 32, out_dim)
        stride, offs = engine.mlp_layout(in_dim, 32, out_dim)
        assert arena.stride == stride and arena.params.numel() == 3 * stride
        for j, net in enumerate(critic.nets):
            for k, lin in enumerate((net.fc1, net.fc2, net.out)):  # nets/mlps.py:113-129, 170-185
                for seg, p in ((engine.SEGS[2 * k], lin.weight), (engine.SEGS[2 * k + 1], lin.bias)):
                    view = arena.view(j, seg)
                    assert p.data_ptr() == view.data_ptr() and p.shape == view.shape, (j, seg)
                    assert view.data_ptr() == arena.params.data_ptr() + 4 * (j * stride + offs[engine.SEGS.index(seg)])
        assert arena.is_bound(list(critic.nets))
        assert critic.arena("cpu") is arena  # cached, re-validated by pointer
    after = [p.detach() for c in agent.critics for p in c.parameters()]
    assert all(torch.equal(a, b) for a, b in zip(before, after)), "adoption must not change a single weight"
    # the torch modules still compute with the adopted storage (acting path of the reference keeps working)
    s = torch.randn(5, 9)
    with torch.no_grad():  # (the reference's nets keep `.features`; a graph-carrying tensor would block deepcopy)
        q = agent.critics[0].nets[0](s) if discrete else agent.critics[0].nets[0](s, torch.randn(5, 4))
    assert q.shape == (5, out_dim)
    # actors: head name per class (fc3 / act_p), action size read off the head
    for actor in agent.actors:
        arena = engine.bind_arena(actor, "self", [actor], "cpu")
        assert actor.action_size == 4 and arena.out_dim == (4 if discrete else 8)
    assert agent.act_space_size == 4
    # deepcopy (the target network, main.py:321) re-packs on first use instead of aliasing the source arena
    tgt = copy.deepcopy(agent)
    ssa.adopt_agent(tgt, device="cpu")
    t_arena = tgt.critics[0].arena("cpu")
    assert t_arena.params.data_ptr() != agent.critics[0].arena("cpu").params.data_ptr()
    assert torch.equal(t_arena.params, agent.critics[0].arena("cpu").params)
    # PopArt: the reference's plain-tensor layer became the device-struct layer with the same statistics
    for p in agent.popart:
        assert hasattr(p, "ptr") and (p.mu, p.nu, p.w, p.b, p._t) == (0.0, 0.0, 1.0, 0.0, 1)
    assert hasattr(agent.adv_estimator, "evaluate")


def test_identity_probe_on_a_reference_encoder(ref):
    from super_sac_amd import adopt
    enc = _RefIdentityEncoder(ref, 9)
    key = adopt.probe_identity(enc, {"obs": torch.randn(1, 9)})
    assert key == "obs" and enc.ssac_identity_key == "obs"


@pytest.mark.parametrize("discrete", [False, True], ids=["continuous", "discrete"])
def test_checkpoints_are_interchangeable_with_the_reference(ref, tmp_path, discrete):
    """agent.save here -> the reference's Agent.load, and back (agent.py:172-202: encoder.pt, popart{i}.pt,
    critic{i}.pt, actor{i}.pt, inverse.pt, contrastive.pt)."""
    import super_sac_amd as ssa
    torch.manual_seed(1)
    mine = ssa.Agent(act_space_size=4, encoder=ssa.nets.IdentityEncoder(9),
                     actor_network_cls=ssa.nets.DiscreteActor if discrete else ssa.nets.ContinuousStochasticActor,
                     critic_network_cls=ssa.nets.DiscreteCritic if discrete else ssa.nets.ContinuousCritic,
                     discrete=discrete, ensemble_size=2, num_critics=3, hidden_size=32, auto_rescale_targets=True,
                     log_std_low=-5.0, log_std_high=2.0)
    d1 = tmp_path / "from_engine"
    d1.mkdir()
    mine.save(str(d1))
    theirs = _ref_agent(ref, discrete)
    theirs.load(str(d1))  # raises on any missing file / key mismatch

    def flat(agent):
        mods = [agent.encoder, *agent.actors, *agent.critics, agent.inverse_model, agent.contrastive_model]
        return np.concatenate([p.detach().numpy().ravel() for m in mods for p in m.parameters()])
    assert np.array_equal(flat(mine), flat(theirs))
    # ... and a directory written by the reference loads here
    torch.manual_seed(2)
    theirs2 = _ref_agent(ref, discrete)
    d2 = tmp_path / "from_reference"
    d2.mkdir()
    theirs2.save(str(d2))
    mine.load(str(d2))
    assert np.array_equal(flat(mine), flat(theirs2))
    assert sorted(os.listdir(d2)) == sorted(f for f in os.listdir(d1) if not f.endswith("_stats.pt"))


def test_install_rebinds_the_real_reference_modules(ref):
    import super_sac_amd as ssa
    saved = {n: getattr(ref.learning, n) for n in ("critic_update", "online_actor_update", "alpha_update",
                                                    "offline_actor_update")}
    saved_lu = {n: getattr(ref.learning_utils, n) for n in ("soft_update", "hard_update", "sample_move_and_augment",
                                                           "compute_td_targets", "compute_backup_weights",
                                                           "adjust_priorities", "compute_filter_stats")}
    try:
        ssa.install(ref)
        import inspect
        for n, fn in saved.items():
            new = getattr(ref.learning, n)
            assert new is getattr(ssa.learning, n)
            # same keyword arguments as the reference definitions (main.py passes everything by keyword)
            assert list(inspect.signature(new).parameters) == list(inspect.signature(fn).parameters), n
    finally:
        for n, fn in saved.items():
            setattr(ref.learning, n, fn)
        for n, fn in saved_lu.items():
            setattr(ref.learning_utils, n, fn)
