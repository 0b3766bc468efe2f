"""(needs the LAB build of the library: ./build.sh --lab)
Phase stamps (s_memtime) of fc2 tile (0, 0, 0) of the LATENCY form of the weight-gradient launch:
    python tools/small_phases.py [obs] [act] [B] [N]"""
import os, sys
os.environ.setdefault("SSAC_LAB_BUILD", "1")   # the lab library (./build.sh --lab -> libssac_hip_lab.so)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv, args = sys.argv[:1] + ["__none__"], sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location("bc", os.path.join(ROOT, "tools", "bench_configs.py"))
bc = importlib.util.module_from_spec(spec); spec.loader.exec_module(bc)
import torch
import super_sac_amd as ssa
obs, act, B, N = (int(v) for v in (args + ["17", "6", "512", "2"][len(args):])[:4])
ssa.engine.set_wgrad_variant(2)
ssa.learning.USE_GRAPHS = False
critic, _ = bc.build(obs, act, B, N, 2)
for _ in range(5):
    critic()
gdbg = torch.zeros(16, dtype=torch.int64, device="cuda")
ssa._lib.check(ssa._lib.lib.ssac_gemm_debug_stamps(gdbg.data_ptr()))
for rep in range(4):
    gdbg.zero_()
    critic()
    torch.cuda.synchronize()
    g = gdbg.cpu().numpy()
    print(f"[{rep}] fc2 tile 0: total {g[5] - g[0]} clk: fold table {g[1] - g[0]}, K loop (operand waits + 32 MFMAs) {g[2] - g[1]}, "
          f"partials through LDS {g[3] - g[2]}, gradient-norm partial {g[4] - g[3]}, Adam + stores issued {g[5] - g[4]}")
ssa._lib.lib.ssac_gemm_debug_stamps(0)
