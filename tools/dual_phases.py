"""phase stamps of critic workgroup (0,0) inside the merged actor + critic-forward launch vs the stand-alone forward"""
import ctypes as C, os, sys
os.environ.setdefault("SSAC_LAB_BUILD", "1")   # the lab library (./build.sh --lab -> libssac_hip_lab.so)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import super_sac_amd as ssa
dev = torch.device("cuda")
B, S, A, H, N = 512, 17, 6, 256, 10
aa = ssa.engine.MlpArena(1, S, H, 2 * A, dev); aa.params.normal_(std=0.05)
ca = ssa.engine.MlpArena(N, S + A, H, 1, dev); ca.params.normal_(std=0.05)
x1 = torch.randn(B, S + A, device=dev); xc = torch.randn(B, S + A, device=dev); eps = torch.randn(B, A, device=dev)
h1 = torch.zeros(N, B, H, device=dev); h2 = torch.zeros_like(h1); q = torch.zeros(N, B, 1, device=dev); lp = torch.zeros(B, device=dev)
lib, st = ssa._lib.lib, ssa.engine.stream()
dbg = torch.zeros(16, dtype=torch.int64, device=dev)
names = ["xstage", "fc1", "fc1-epi", "fc2", "fc2-epi+sync", "w3stage", "head"]
def dual():
    ssa._lib.check(lib.ssac_actor_sample_critic_fwd(C.byref(aa.desc()), x1.data_ptr(), S + A, B, eps.data_ptr(), -5.0, 2.0, x1.data_ptr(), S + A, S,
        lp.data_ptr(), 0, C.byref(ca.desc()), xc.data_ptr(), S + A, h1.data_ptr(), h2.data_ptr(), q.data_ptr(), 0, st))
ws = ssa.engine.Workspace(dev)
def solo():
    ssa.engine.mlp_forward(ca, xc, S + A, 0, B, ws, "t")
for name, fn in (("dual", dual),):
    lib.ssac_fused_debug_stamps(dbg.data_ptr())
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = dbg.cpu().numpy(); lib.ssac_fused_debug_stamps(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: total {t[7]-t[0]} clk;", ", ".join(f"{n} {t[i+1]-t[i]}" for i, n in enumerate(names)), f"; {e0.elapsed_time(e1)*10:.2f} us/launch")
