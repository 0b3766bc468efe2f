import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p_ in ("tests", "oracle", ""):
    sys.path.insert(0, os.path.join(ROOT, p_))
import numpy as np, torch, torch.multiprocessing as mp

def main(rank, world, port, one_shot):
    import case_runner, synth
    import torch.distributed as dist
    from super_sac_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    if one_shot:
        parallel.enable_one_shot(torch.device("cuda:0"))
    cfg = synth.CASES["redq_M"]
    shard = parallel.Shard(rank, world, cfg["N"])
    rec = case_runner.run_engine("redq_M", device="cuda:0", shard=shard)
    fx = case_runner.slice_fixture(case_runner.load_fixture("redq_M"), cfg, shard)
    try:
        worst = case_runner.compare(rec, fx, who=f"rank {rank}")
        print("rank", rank, "ok", worst, flush=True)
    except AssertionError as e:
        print("rank", rank, "FAIL", str(e)[:300], flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    world, one_shot = int(sys.argv[1]), int(sys.argv[2])
    mp.spawn(main, args=(world, 29500 + os.getpid() % 1000, one_shot), nprocs=world, join=True)
