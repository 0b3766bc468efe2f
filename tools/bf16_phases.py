"""(needs the LAB build of the library: ./build.sh --lab)
s_memtime phase stamps inside the bf16 chain / weight-gradient launches (eager launches, tile 0 of each role)
    python tools/bf16_phases.py [B] [N]"""
import os, sys
os.environ.setdefault("SSAC_LAB_BUILD", "1")   # the lab library (./build.sh --lab -> libssac_hip_lab.so)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv, args = sys.argv[:1] + ["__none__"], sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location("bc", os.path.join(ROOT, "tools", "bench_configs.py"))
bc = importlib.util.module_from_spec(spec); spec.loader.exec_module(bc)
import torch
import super_sac_amd as ssa
B = int(args[0]) if args else 256
N = int(args[1]) if len(args) > 1 else 10
ssa.learning.USE_GRAPHS = False
critic, _ = bc.build(17, 6, B, N, 2, precision="bf16")
for _ in range(5):
    critic()
dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
ssa._lib.check(ssa._lib.lib.ssac_bf16_debug_stamps(dbg.data_ptr()))
for _ in range(3):
    critic()
torch.cuda.synchronize()
ssa._lib.lib.ssac_bf16_debug_stamps(0)
t = dbg.cpu().numpy()
def row(name, base, labels):
    print(name, "total", t[base + len(labels)] - t[base], "clk:", ", ".join(f"{l} {t[base+i+1]-t[base+i]}" for i, l in enumerate(labels)))
row("actor pass   ", 0, ["gather+prefetch", "fc1", "fc1-epi+sync", "fc2", "fc2-epi+sync", "head", "sample"])
row("target pass  ", 16, ["x+prefetch", "fc1", "fc1-epi+sync", "fc2", "fc2-epi+sync", "head"])
print("actor start -> target pass end:", t[22] - t[0])
row("critic WG    ", 32, ["gather+prefetch", "fc1", "fc1-epi+sync", "fc2", "fc2-epi+sync", "head", "dz2u+sync", "dgrad", "dz1 store"])
row("wgrad tile 0 ", 48, ["prefetch+lossfold", "K loop", "adam epilogue"])
for name, base in (("actor", 0), ("target", 16), ("critic", 32)):
    print(name, "prologue (serialised by the debug stamps): index", t[base + 12] - t[base], " x rows", t[base + 13] - t[base + 12],
          " small operands", t[base + 14] - t[base + 13], " 18 fragment loads", t[base + 15] - t[base + 14],
          " rest to barrier", t[base + 1] - t[base + 15])
