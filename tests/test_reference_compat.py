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
    saved_cls = [(ref.replay, "ReplayBuffer", ref.replay.ReplayBuffer)] + \
        [(ref.augmentations, n, getattr(ref.augmentations, n)) for n in ssa.INSTALLED_AUGMENTATIONS]
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
        for m, n, c in saved_cls:
            setattr(m, n, c)


def test_install_rebinds_the_buffer_and_augmentation_classes(ref):
    """the classes the shipped scripts construct through the module objects (experiments/gym/train_gym.py:84,
    main.py:138-141) resolve to this package's after install()"""
    import super_sac_amd as ssa
    names = ssa.INSTALLED_AUGMENTATIONS
    saved_aug = {n: getattr(ref.augmentations, n) for n in names}
    saved_buf = ref.replay.ReplayBuffer
    saved_fn = {(m, n): getattr(m, n) for m in (ref.learning, ref.learning_utils) for n in dir(m)
                if callable(getattr(m, n)) and hasattr(getattr(ssa, m.__name__.rsplit(".", 1)[1]), n)}
    try:
        ssa.install(ref)
        assert ref.replay.ReplayBuffer is ssa.replay.ReplayBuffer
        import inspect
        assert (list(inspect.signature(saved_buf.__init__).parameters)
                == list(inspect.signature(ssa.replay.ReplayBuffer.__init__).parameters)[:4])  # (+ optional device)
        for n in names:
            assert getattr(ref.augmentations, n) is getattr(ssa.augmentations, n), n
            # constructor arguments of the reference class are accepted positionally and by name
            a, b = inspect.signature(saved_aug[n].__init__), inspect.signature(getattr(ssa.augmentations, n).__init__)
            named = [p for p in a.parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD]
            mine = [p for p in b.parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD]
            assert [p.name for p in named] == [p.name for p in mine][:len(named)], n
            assert [p.default for p in named] == [p.default for p in mine][:len(named)], n
        # main.py:138-141 builds the default augmenter through the module attributes
        aug = ref.augmentations.AugmentationSequence([ref.augmentations.IdentityAug(32)])
        assert isinstance(aug, ssa.augmentations.AugmentationSequence) and aug.is_identity()
    finally:
        ref.replay.ReplayBuffer = saved_buf
        for n, c in saved_aug.items():
            setattr(ref.augmentations, n, c)
        for (m, n), fn in saved_fn.items():
            setattr(m, n, fn)


@pytest.mark.parametrize("cls,kw", [("Drqv2Aug", {}), ("DrqAug", {}), ("DrqNoNoiseAug", {}), ("LargeDrqAug", {}),
                                    ("LargeDrqNoNoiseAug", {}), ("Drqv2Aug", {"pad": 6})])
def test_reference_built_augmenter_is_adopted_in_place(ref, cls, kw):
    """augmentations.py:20-41, 165-293: class, batch size, pad, noise flag and the CURRENT randomisation carry over;
    adoption draws nothing from the host generators"""
    import super_sac_amd as ssa
    torch.manual_seed(5)
    theirs = getattr(ref.augmentations, cls)(16, **kw)
    seq = ref.augmentations.AugmentationSequence([theirs, ref.augmentations.IdentityAug(16)])
    want = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in vars(theirs).items()}
    state = torch.get_rng_state()
    out = ssa.adopt_augmenter(seq)
    assert out is seq and type(seq) is ssa.augmentations.AugmentationSequence
    assert torch.equal(torch.get_rng_state(), state), "adoption must not consume the torch CPU generator"
    mine = seq.aug_list[0]
    assert mine is theirs and type(mine) is getattr(ssa.augmentations, cls)
    assert type(seq.aug_list[1]) is ssa.augmentations.IdentityAug
    assert (mine.batch_size, mine.pad, mine.noise) == (16, want["pad"], want["noise"])
    assert repr(mine) == repr(getattr(ref.augmentations, cls)(16, **kw)) and "pad_func" not in vars(mine)
    if cls == "Drqv2Aug":
        assert torch.equal(mine._shift_host, want["shift"].reshape(16, 2))   # (x, y) per image, augmentations.py:226-231
    else:
        assert torch.equal(mine._shift_host, torch.stack([want["w1"], want["h1"]], 1))
    assert seq.single_shift() is mine and not seq.is_identity()
    # the next randomisation is the same draw the reference object would have made
    torch.manual_seed(9)
    mine.change_randomization_params()
    torch.manual_seed(9)
    again = getattr(ref.augmentations, cls)(16, **kw)   # (the constructor draws once, augmentations.py:178,222)
    if cls == "Drqv2Aug":
        assert torch.equal(mine.shift, again.shift)
    else:
        assert torch.equal(mine.w1, again.w1) and torch.equal(mine.h1, again.h1)
    assert ssa.adopt_augmenter(seq) is seq   # idempotent
    with pytest.raises(NotImplementedError, match="no HIP path"):
        ssa.adopt_augmenter(ref.augmentations.AugmentationSequence([ref.augmentations.GrayscaleAug(16)]))


def test_reference_built_replay_buffer_is_adopted_in_place(ref):
    """replay.py:10-190: rows, cursor, fill level, priorities, trees and counters carry over attribute by attribute
    (on CPU tensors here; the same code puts them in HBM on the GPU box)"""
    import super_sac_amd as ssa
    rs = np.random.RandomState(0)
    theirs = ref.replay.ReplayBuffer(size=100, alpha=0.7, beta=0.9)
    n = 130   # wraps around the ring once
    s = {"obs": rs.randn(n, 5).astype(np.float32), "img": rs.randint(0, 256, (n, 2, 4, 4)).astype(np.uint8)}
    s1 = {"obs": rs.randn(n, 5).astype(np.float32), "img": rs.randint(0, 256, (n, 2, 4, 4)).astype(np.uint8)}
    a, r, d = rs.uniform(-1, 1, (n, 3)).astype(np.float32), rs.randn(n, 1).astype(np.float32), rs.rand(n, 1) < 0.1
    theirs.push({k: v[:70] for k, v in s.items()}, a[:70], r[:70], {k: v[:70] for k, v in s1.items()}, d[:70])
    theirs.push({k: v[70:] for k, v in s.items()}, a[70:], r[70:], {k: v[70:] for k, v in s1.items()}, d[70:])
    theirs.update_priorities(np.arange(10), rs.rand(10) + 0.5)
    np.random.seed(3)
    _, w_ref, idx_ref = theirs.sample(32)
    calls = theirs.total_sample_calls
    old = theirs._storage
    want_sum, want_min = theirs._it_sum._value.copy(), theirs._it_min._value.copy()
    out = ssa.adopt_buffer(theirs, "cpu")
    assert out is theirs and type(theirs) is ssa.replay.ReplayBuffer
    st = theirs._storage
    assert (len(theirs), st._next_idx, st._max_filled, st.size) == (100, 30, 100, 100)
    assert theirs.total_sample_calls == calls and (theirs.alpha, theirs.beta, theirs._maxsize) == (0.7, 0.9, 100)
    for k in s:
        assert st.s_stack[k].dtype == (torch.uint8 if k == "img" else torch.float32)
        assert np.array_equal(st.s_stack[k].numpy(), old.s_stack[k]) and np.array_equal(st.s1_stack[k].numpy(), old.s1_stack[k])
    assert np.array_equal(st.action_stack.numpy(), old.action_stack)
    assert np.array_equal(st.reward_stack.numpy(), old.reward_stack) and np.array_equal(st.done_stack.numpy(), old.done_stack)
    assert np.array_equal(theirs._per.sum_tree, want_sum) and np.array_equal(theirs._per.min_tree, want_min)
    assert theirs._per._max_priority == pytest.approx(max(1.0, float(np.max(theirs._per._max_priority))))
    assert not hasattr(theirs, "_it_sum")
    # the prioritised draw of the adopted buffer is the reference's (numpy global generator, float64 trees)
    np.random.seed(3)
    idx, w = theirs._per.sample(len(theirs), 32)
    assert np.array_equal(idx, idx_ref) and np.array_equal(w, w_ref.numpy())
    assert ssa.adopt_buffer(theirs) is theirs   # idempotent
    # an empty reference buffer (the scripts hand main.super_sac a buffer that the warm-up fills) is adopted too
    empty = ssa.adopt_buffer(ref.replay.ReplayBuffer(size=64), "cpu")
    assert type(empty) is ssa.replay.ReplayBuffer and len(empty) == 0 and empty._storage is None
    with pytest.raises(TypeError, match="install"):
        ssa.adopt_buffer(ref.replay._BasicReplayBuffer(10))
