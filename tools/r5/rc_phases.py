"""(LAB build) s_memtime stamps of wave 0 of workgroup (0, 0) of bf_regchain_kernel at B 65 536, N 10:  python tools/r5/rc_phases.py"""
import os, sys
os.environ.setdefault("SSAC_LAB_BUILD", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from super_sac_amd import engine
from super_sac_amd._lib import check, lib
dev = torch.device("cuda")
N, IN, B = 10, 23, int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ar = engine.MlpArena(N, IN, 256, 1, dev)
torch.manual_seed(1)
ar.params.copy_(torch.randn_like(ar.params) * 0.05)
ar.enable_bf16()
x = torch.randn(B, IN, device=dev)
y = torch.empty(N, B, 1, device=dev)
run = lambda: check(lib.ssac_bf16_mlp3_fwd(C.byref(ar.desc()), ar.shadow.data_ptr(), 0, N, x.data_ptr(), IN, B, y.data_ptr(), engine.stream()))
for _ in range(5):
    run()
dbg = torch.zeros(64, dtype=torch.int64, device=dev)
check(lib.ssac_bf16_debug_stamps(dbg.data_ptr()))
for _ in range(3):
    run()
torch.cuda.synchronize()
lib.ssac_bf16_debug_stamps(0)
t = dbg.cpu().numpy()
print(f"units of the wave {t[8]};  kernel entry -> weights in LDS + first x requested {t[6]-t[5]} clk;  whole loop {t[7]-t[6]} clk "
      f"({(t[7]-t[6]) / max(1, (t[8] + 1) // 2):.0f} per iteration)")
print(f"second iteration: x pack + fc1 {t[1]-t[0]}, next x issue {t[2]-t[1]}, fc2 + head {t[3]-t[2]}, reduce + store {t[4]-t[3]}; total {t[4]-t[0]}")
