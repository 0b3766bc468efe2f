"""(needs the LAB build of the library: ./build.sh --lab)
s_memtime phase stamps inside the fp32 chained launch and the merged weight-gradient launch (eager launches of the
real update at the headline shape; tile 0 of each role)       python tools/fp32_phases.py [B] [N]"""
import os, sys
os.environ.setdefault("SSAC_LAB_BUILD", "1")   # the lab library (./build.sh --lab -> libssac_hip_lab.so)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv, args = sys.argv[:1] + ["__none__"], sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location("bc", os.path.join(ROOT, "tools", "bench_configs.py"))
bc = importlib.util.module_from_spec(spec); spec.loader.exec_module(bc)
import torch
import super_sac_amd as ssa
B = int(args[0]) if args else 512
N = int(args[1]) if len(args) > 1 else 10
ssa.learning.USE_GRAPHS = False
critic, _ = bc.build(17, 6, B, N, 2)
for _ in range(5):
    critic()
dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
gdbg = torch.zeros(16, dtype=torch.int64, device="cuda")
ssa._lib.check(ssa._lib.lib.ssac_fused_debug_stamps(dbg.data_ptr()))
ssa._lib.check(ssa._lib.lib.ssac_gemm_debug_stamps(gdbg.data_ptr()))
for _ in range(3):
    critic()
torch.cuda.synchronize()
ssa._lib.lib.ssac_fused_debug_stamps(0)
ssa._lib.lib.ssac_gemm_debug_stamps(0)
t, g = dbg.cpu().numpy(), gdbg.cpu().numpy()
def row(name, tt, base, labels):
    print(name, "total", tt[base + len(labels)] - tt[base], "clk:", ", ".join(f"{l} {tt[base+i+1]-tt[base+i]}" for i, l in enumerate(labels)))
names = ["prologue", "fc1", "fc1-epi", "fc2", "fc2-epi+sync", "w3stage", "head"]
row("actor pass  ", t, 0, names)
print("actor pass: tanh-normal sample per (row, dim) incl. publish", t[9] - t[7], " barrier + log pi sum + store", t[8] - t[9])
row("target pass ", t, 16, names)
print("actor start -> target pass end:", t[23] - t[0])
row("critic WG   ", t, 32, names + ["loss", "head-bwd", "dgrad", "dz1-store"])
row("wgrad tile 0", g, 0, ["operand issue + loss fold + first chunk", "K loop", "partials -> LDS", "adam epilogue"])
print("wgrad tile 0 prologue (serialised by the debug stamps): first operand chunk", g[5] - g[0], " loss fold (pre)", g[6] - g[5],
      " LDS stores + second chunk", g[7] - g[6], " barrier + first fragment reads", g[1] - g[7])
print("wgrad tile 0 epilogue: partial sums from LDS", g[8] - g[3], " gradient-norm partial (2 barriers)", g[9] - g[8], " Adam + stores issued", g[4] - g[9])
print("   gradient-norm partial: squares + wave sum", g[10] - g[8], " first barrier", g[11] - g[10], " second barrier", g[12] - g[11], " sum of 16 + slot stores", g[9] - g[12])
