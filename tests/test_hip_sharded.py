"""GPU: the critic-sharded update on the real kernels.  Two ranks share the one GPU of the test box
(RCCL refuses two ranks on one device, so the collective runs over gloo with device tensors); every
rank replays the reference fixture with HALF of the critic ensemble and must reproduce the
reference's TD targets, its own critics' final parameters and the (replicated) actor."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _rank_main(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    import case_runner
    import synth
    import torch.distributed as dist
    from super_sac_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cfg = synth.CASES["redq_small"]
    shard = parallel.Shard(rank, world, cfg["N"])
    rec = case_runner.run_engine("redq_small", device="cuda:0", shard=shard)
    fx = case_runner.slice_fixture(case_runner.load_fixture("redq_small"), cfg, shard)
    worst = case_runner.compare(rec, fx, who=f"hip-sharded[rank {rank}]")
    np.savez(os.path.join(out_dir, f"ok{rank}.npz"), **{k: np.float64(v) for k, v in worst.items()})
    dist.destroy_process_group()


def test_two_rank_sharded_sequence_matches_reference(tmp_path):
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        assert (tmp_path / f"ok{rank}.npz").exists()
