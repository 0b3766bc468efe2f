"""CPU: the reference-written checkpoints held by the resume fixtures (tests/golden/ckpt_*.npz, "ckpt|<file>|<key>" arrays;
oracle/gen_golden.py::reference_checkpoint) load through this package's Agent.load -- no kernel runs: the modules and
arenas are built on CPU tensors -- and, where /root/reference exists (the build container), the REFERENCE's Agent.load
accepts what this package's Agent.save writes for the loaded agent (agent.py:172-202)."""
import os

import numpy as np
import pytest
import torch

import case_runner
import synth

RESUME_CASES = [n_ for n_, c_ in synth.CASES.items() if c_.get("resume")]
REF = "/root/reference"


@pytest.mark.parametrize("name", RESUME_CASES)
def test_fixture_checkpoint_loads_bit_exactly(tmp_path, name):
    cfg, fx = synth.CASES[name], case_runner.load_fixture(name)
    files = case_runner.checkpoint_arrays(fx)
    agent = case_runner.build_engine_agent(cfg, "cpu")
    before = torch.cat([p.detach().flatten() for p in agent.critics[0].parameters()]).clone()
    agent.load(case_runner.write_checkpoint_dir(fx, str(tmp_path)))
    after = torch.cat([p.detach().flatten() for p in agent.critics[0].parameters()])
    assert not torch.equal(before, after), "the resume case's own seed must differ from the checkpoint's"
    mods = {"encoder.pt": agent.encoder, "inverse.pt": agent.inverse_model, "contrastive.pt": agent.contrastive_model}
    mods.update({f"critic{i}.pt": c for i, c in enumerate(agent.critics)})
    mods.update({f"actor{i}.pt": a for i, a in enumerate(agent.actors)})
    for fname, mod in mods.items():
        sd = mod.state_dict()
        assert set(sd) == set(files[fname]), fname
        for k, v in files[fname].items():
            assert np.array_equal(sd[k].numpy(), v), (fname, k)
    if cfg["popart"]:
        # popart.py:11-16: the reference's layer has an EMPTY state dict -- a loaded agent starts from fresh statistics
        assert files["popart0.pt"] == {}
        p = agent.popart[0]
        assert (p.mu, p.nu, p.w, p.b) == (0.0, 0.0, 1.0, 0.0)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "super_sac")), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("name", [n_ for n_ in RESUME_CASES if "pixels" not in n_])
def test_reference_loads_what_the_package_saves_for_the_loaded_agent(tmp_path, name):
    import ref_harness
    ref = ref_harness.import_reference()
    cfg, fx = synth.CASES[name], case_runner.load_fixture(name)
    agent = case_runner.build_engine_agent(cfg, "cpu")
    agent.load(case_runner.write_checkpoint_dir(fx, str(tmp_path / "a")))
    out = tmp_path / "b"
    out.mkdir()
    agent.save(str(out))

    class Enc(ref.nets.Encoder):
        def __init__(self):
            super().__init__()

        @property
        def embedding_dim(self):
            return cfg["obs"]

        def forward(self, obs_dict):
            return obs_dict["obs"]
    actor_cls = {"stochastic": ref.nets.mlps.ContinuousStochasticActor, "deterministic": ref.nets.mlps.ContinuousDeterministicActor,
                 "discrete": ref.nets.mlps.DiscreteActor}[cfg["actor"]]
    critic_cls = ref.nets.mlps.DiscreteCritic if cfg["discrete"] else ref.nets.mlps.ContinuousCritic
    theirs = ref.Agent(act_space_size=cfg["act"], encoder=Enc(), actor_network_cls=actor_cls, critic_network_cls=critic_cls,
                       discrete=cfg["discrete"], ensemble_size=cfg["E"], num_critics=cfg["N"], ucb_bonus=0.0,
                       hidden_size=cfg["hidden"], auto_rescale_targets=cfg["popart"], log_std_low=cfg["lo"], log_std_high=cfg["hi"])
    theirs.load(str(out))   # raises on a missing file or a key / shape mismatch
    files = case_runner.checkpoint_arrays(fx)
    for i, c in enumerate(theirs.critics):
        for k, v in files[f"critic{i}.pt"].items():
            assert np.array_equal(c.state_dict()[k].numpy(), v), (i, k)
