"""CPU: the in-place adoption of a replay buffer / augmentation sequence that carry only the reference classes'
attributes (tests/foreign_agent.py stand-ins; the reference's own objects are checked in test_reference_compat.py
where /root/reference exists).  No kernel runs here: storage is built on CPU tensors."""
import numpy as np
import pytest
import torch

import foreign_agent
import synth


def test_foreign_buffer_becomes_a_product_buffer():
    import super_sac_amd as ssa
    s, a, r, s1, d = synth.synth_transitions(200, 5, 2, seed=3)
    buf = foreign_agent.ForeignReplayBuffer(256, s, a, r, s1, d, alpha=0.6, beta=1.0)
    ssa.adopt_buffer(buf, "cpu")
    assert type(buf) is ssa.replay.ReplayBuffer and len(buf) == 200
    st = buf._storage
    assert np.array_equal(st.s_stack["obs"][:200].numpy(), s["obs"]) and np.array_equal(st.s1_stack["obs"][:200].numpy(), s1["obs"])
    assert np.array_equal(st.action_stack[:200].numpy(), a) and np.array_equal(st.reward_stack[:200, 0].numpy(), r)
    assert st.done_stack.dtype == torch.uint8 and int(st.done_stack.sum()) == int(np.sum(d))
    assert buf._per.sum_tree[1] == 200.0 and buf._per.min_tree[1] == 1.0 and buf._per.cap == 256
    # the uniform index draw of the adopted buffer is the reference's stream (replay.py:121-126)
    torch.manual_seed(0)
    want = torch.randint(200, (64,))
    torch.manual_seed(0)
    assert torch.equal(ssa.rng.draw_indices(len(buf), 64), want)


def test_foreign_augmenter_becomes_a_product_augmenter():
    import super_sac_amd as ssa
    v2 = foreign_agent.Drqv2Aug(8, pad=4)
    v2.shift = torch.arange(16).reshape(8, 1, 1, 2) % 9
    seq = foreign_agent.ForeignAugmentationSequence([v2])
    state = torch.get_rng_state()
    ssa.adopt_augmenter(seq)
    assert torch.equal(torch.get_rng_state(), state)
    assert type(seq) is ssa.augmentations.AugmentationSequence and type(v2) is ssa.augmentations.Drqv2Aug
    assert seq.single_shift() is v2 and torch.equal(v2._shift_host, (torch.arange(16) % 9).reshape(8, 2))
    v1 = foreign_agent.DrqNoNoiseAug(8)
    v1.w1, v1.h1 = torch.arange(8) % 8, (torch.arange(8) + 3) % 8
    seq1 = ssa.adopt_augmenter(foreign_agent.ForeignAugmentationSequence([v1]))
    assert type(v1) is ssa.augmentations.DrqNoNoiseAug and "pad_func" not in vars(v1) and v1.noise is False
    assert torch.equal(v1._shift_host, torch.stack([v1.w1, v1.h1], 1)) and seq1.single_shift() is v1
    ident = ssa.adopt_augmenter(foreign_agent.ForeignAugmentationSequence([foreign_agent.IdentityAug(8)]))
    assert ident.is_identity()

    class CutoutAug:   # (a reference augmentation outside the DrQ family)
        batch_size = 8
    with pytest.raises(NotImplementedError, match="no HIP path"):
        ssa.adopt_augmenter(foreign_agent.ForeignAugmentationSequence([CutoutAug()]))
