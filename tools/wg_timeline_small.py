"""(needs the LAB build of the library: ./build.sh --lab)
Per-workgroup (start, end) of the two launches of a critic update for UNDER-FILLED configurations (the latency form
of the weight-gradient launch: 32 x 32 tiles):      python tools/wg_timeline_small.py [obs] [act] [B] [N]"""
import os, sys
os.environ.setdefault("SSAC_LAB_BUILD", "1")   # the lab library (./build.sh --lab -> libssac_hip_lab.so)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv, args = sys.argv[:1] + ["__none__"], sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location("bc", os.path.join(ROOT, "tools", "bench_configs.py"))
bc = importlib.util.module_from_spec(spec); spec.loader.exec_module(bc)
import numpy as np
import torch
import super_sac_amd as ssa
obs, act, B, N = (int(v) for v in (args + ["17", "6", "512", "2"][len(args):])[:4])
if os.environ.get("SSAC_WGRAD_VARIANT"):
    ssa.engine.set_wgrad_variant(int(os.environ["SSAC_WGRAD_VARIANT"]))
ssa.learning.USE_GRAPHS = False
critic, _ = bc.build(obs, act, B, N, 2)
for _ in range(5):
    critic()
tl = torch.zeros(2048, dtype=torch.int64, device="cuda")
ssa._lib.check(ssa._lib.lib.ssac_debug_timeline(tl.data_ptr()))
H = 256
in_dim = obs + act
small = (((H // 32) * (H // 32), "fc2 tiles"), ((H // 32) * ((in_dim + 31) // 32), "fc1 tiles"), (H // 16, "head"))
for rep in range(3):
    tl.zero_()
    critic()
    torch.cuda.synchronize()
    t = tl.cpu().numpy().reshape(2, 512, 2) * 10e-3   # us
    for name, a in (("chained launch", t[0]), ("weight-gradient launch", t[1])):
        n = int((a[:, 1] > 0).sum())
        a = a[:n]
        t0 = a[:, 0].min()
        s, e = a[:, 0] - t0, a[:, 1] - t0
        print(f"[{rep}] {name}: {n} workgroups; starts {s.min():.2f} .. {s.max():.2f} us (median {np.median(s):.2f}); "
              f"ends {e.min():.2f} .. {e.max():.2f}; durations min {np.min(e - s):.2f} median {np.median(e - s):.2f} max {np.max(e - s):.2f}")
        if rep == 2 and name.startswith("chained") and ssa.learning_utils.CHAIN_PC:
            tiles = (B + 15) // 16
            import ctypes as C
            ag = critic.objects["agent"] if hasattr(critic, "objects") else None
            splits = 1
            if ssa.learning_utils.CHAIN_SPLIT and ag is not None:
                dev = torch.device("cuda")
                aa = ssa.engine.bind_arena(ag.actors[0], "self", [ag.actors[0]], dev)
                ca = ag.critics[0].arena(dev)
                splits = int(ssa._lib.lib.ssac_chain_target_splits(C.byref(aa.desc()), C.byref(ca.desc()), C.byref(ca.desc()), B, 2))
            ncons = 2 * splits * tiles
            crit16 = (n - tiles - ncons) == tiles * N   # 16-row critic tiles: [producers | consumers | critics]
            if crit16:
                groups = (("actor producers", 0, tiles), (f"target consumers (x{splits} column splits)", tiles, tiles + ncons),
                          ("critic tiles (16 rows)", tiles + ncons, n))
            else:
                nc = n - tiles - ncons
                groups = (("actor producers", 0, tiles), ("critic tiles (32 rows)", tiles, tiles + nc),
                          (f"target consumers (x{splits} column splits)", tiles + nc, n))
            for gname, lo, hi in groups:
                print(f"      {gname:40s} start {s[lo:hi].min():6.2f}..{s[lo:hi].max():6.2f}  end {e[lo:hi].min():6.2f}..{e[lo:hi].max():6.2f}  "
                      f"duration {np.min((e - s)[lo:hi]):6.2f}..{np.max((e - s)[lo:hi]):6.2f} (median {np.median((e - s)[lo:hi]):6.2f})")
        if rep == 2 and name.startswith("weight") and n == sum(c for c, _ in small) * N + 1:
            # (hardware order: classes are interleaved per XCD; report by duration clusters instead)
            d = np.sort(e - s)
            print("      duration deciles:", " ".join(f"{np.percentile(d, q):.2f}" for q in range(0, 101, 10)))
            print("      end-time deciles:", " ".join(f"{np.percentile(e, q):.2f}" for q in range(0, 101, 10)))
    gap = t[1][:, 0][t[1][:, 1] > 0].min() - t[0][:, 1].max()
    print(f"[{rep}] chained last end -> weight-gradient first start: {gap:.2f} us")
