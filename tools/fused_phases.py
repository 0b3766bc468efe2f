"""(needs the LAB build of the library: ./build.sh --lab)
phase timing (shader clocks) inside the fused MLP kernel, workgroup (0,0)"""
import ctypes as C, os, sys
os.environ.setdefault("SSAC_LAB_BUILD", "1")   # the lab library (./build.sh --lab -> libssac_hip_lab.so)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import super_sac_amd as ssa
import ssac_oracle as orc
dev = torch.device("cuda")
ssa._lib.check(ssa._lib.lib.ssac_fused_tile_rows(int(os.environ.get("SSAC_TILE", "0"))))
rng = np.random.RandomState(0)
for (B, in_dim, H, out, N) in [(512, 23, 256, 1, 2), (512, 17, 256, 12, 1), (512, 23, 256, 1, 10)]:
    mlps = [orc.make_mlp(rng, in_dim, H, out) for _ in range(N)]
    ar = ssa.engine.MlpArena(N, in_dim, H, out, dev)
    for j, p in enumerate(mlps):
        for seg in ssa.engine.SEGS:
            ar.view(j, seg).copy_(p[seg])
    x = torch.randn(B, in_dim, device=dev)
    ws = ssa.engine.Workspace(dev)
    dbg = torch.zeros(16, dtype=torch.int64, device=dev)
    ssa._lib.check(ssa._lib.lib.ssac_fused_debug_stamps(dbg.data_ptr()))
    for _ in range(3):
        ssa.engine.mlp_forward(ar, x, in_dim, 0, B, ws, "t", save=False)
    torch.cuda.synchronize()
    t = dbg.cpu().numpy()
    names = ["xstage", "fc1", "fc1-epi", "fc2", "fc2-epi+sync", "w3stage", "head"]
    print(f"B{B} in{in_dim} H{H} out{out} N{N}: total {t[7]-t[0]} clk;", ", ".join(f"{n} {t[i+1]-t[i]}" for i, n in enumerate(names)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ssa._lib.lib.ssac_fused_debug_stamps(0)
    e0.record()
    for _ in range(50):
        ssa.engine.mlp_forward(ar, x, in_dim, 0, B, ws, "t", save=False)
    e1.record(); torch.cuda.synchronize()
    print(f"   {e0.elapsed_time(e1)*1e3/50:.2f} us per launch (back-to-back)")

# ---- critic fused kernel phases
B, in_dim, H, N = 512, 23, 256, 10
mlps = [orc.make_mlp(rng, in_dim, H, 1) for _ in range(N)]
ar = ssa.engine.MlpArena(N, in_dim, H, 1, dev)
x = torch.randn(B, in_dim, device=dev); td = torch.randn(B, 1, device=dev)
h1 = torch.zeros(N, B, H, device=dev); h2 = torch.zeros_like(h1); dz2 = torch.zeros_like(h1); dz1 = torch.zeros_like(h1)
q = torch.zeros(N, B, 1, device=dev); dq = torch.zeros_like(q)
tiles = int(ssa._lib.lib.ssac_fused_row_tiles(C.byref(ar.desc()), B, N)); parts = torch.zeros(N * tiles * 2, device=dev)
dbg = torch.zeros(16, dtype=torch.int64, device=dev)
ssa._lib.check(ssa._lib.lib.ssac_fused_debug_stamps(dbg.data_ptr()))
def run():
    ssa._lib.check(ssa._lib.lib.ssac_critic_fwd_bwd_fused(C.byref(ar.desc()), x.data_ptr(), in_dim, B, td.data_ptr(), 0, 0, 1, 0, 0,
        float(N), h1.data_ptr(), h2.data_ptr(), q.data_ptr(), dq.data_ptr(), dz2.data_ptr(), dz1.data_ptr(), parts.data_ptr(), 0, ssa.engine.stream()))
for _ in range(3): run()
torch.cuda.synchronize()
t = dbg.cpu().numpy()
names = ["xstage", "fc1", "fc1-epi", "fc2", "fc2-epi+sync", "w3stage", "head", "loss", "head-bwd", "dgrad", "dz1-store"]
print("critic fused: total", t[11]-t[0], "clk;", ", ".join(f"{n} {t[i+1]-t[i]}" for i, n in enumerate(names)))
ssa._lib.lib.ssac_fused_debug_stamps(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(f"   {e0.elapsed_time(e1)*1e3/50:.2f} us per launch (back-to-back)")

# ---- weight-gradient GEMM phases (layer 1: dW2 = dz2^T h1, K = batch)
dbg2 = torch.zeros(16, dtype=torch.int64, device=dev)
topt = torch.optim.Adam([torch.zeros(1)], lr=3e-4)
grp = ssa.engine.AdamGroup(topt, dev)
grp.advance()
m_, v_ = grp.moments_for("k", ar.params)
for layer, (xin, ldi, sxi, dy, ldy, sy) in {1: (h1, H, B * H, dz2, H, B * H), 0: (x, in_dim, 0, dz1, H, B * H)}.items():
    ssa._lib.check(ssa._lib.lib.ssac_gemm_debug_stamps(dbg2.data_ptr()))
    def runw():
        ssa._lib.check(ssa._lib.lib.ssac_mlp_layer_wgrad(C.byref(ar.desc()), layer, 0, N, xin.data_ptr(), ldi, sxi, dy.data_ptr(), ldy, sy, B,
            m_.data_ptr(), v_.data_ptr(), grp.ctl.ptr, 0, 0, 0, 0, 0.0, ssa.engine.stream()))
    for _ in range(3): runw()
    torch.cuda.synchronize()
    t = dbg2.cpu().numpy()
    print(f"wgrad layer {layer}: prologue {t[1]-t[0]}, kloop {t[2]-t[1]}, ksplit-reduce {t[3]-t[2]}, adam-epilogue {t[4]-t[3]} clk")
    ssa._lib.lib.ssac_gemm_debug_stamps(0)
    e0.record()
    for _ in range(50): runw()
    e1.record(); torch.cuda.synchronize()
    print(f"   {e0.elapsed_time(e1)*1e3/50:.2f} us per launch (back-to-back)")
