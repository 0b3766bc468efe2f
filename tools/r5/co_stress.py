"""stress: ssac_chain_update alone (no gather, no recording), co-resident form, many launches; checks outputs stay bit-equal to launch 0"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import super_sac_amd as ssa
import ssac_oracle as orc
DEV = "cuda:0"
B, N, H, S, A = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 10, 256, 17, 6
n_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rng = np.random.RandomState(3)
def arena_from(nets):
    ar = ssa.engine.MlpArena(len(nets), nets[0]["w1"].shape[1], nets[0]["w1"].shape[0], nets[0]["w3"].shape[0], torch.device(DEV))
    for j, n in enumerate(nets):
        for seg in ssa.engine.SEGS:
            ar.view(j, seg).copy_(n[seg])
    return ar
actor = orc.make_mlp(rng, S, H, 2 * A)
crit = [orc.make_mlp(rng, S + A, H, 1) for _ in range(N)]
tgt = [orc.make_mlp(rng, S + A, H, 1) for _ in range(N)]
aa, ca, ta = arena_from([actor]), arena_from(crit), arena_from(tgt)
x1 = torch.from_numpy(rng.standard_normal((B, S + A)).astype(np.float32)).to(DEV)
xc = torch.from_numpy(rng.standard_normal((B, S + A)).astype(np.float32)).to(DEV)
eps = torch.from_numpy(rng.standard_normal((B, A)).astype(np.float32)).to(DEV)
ids = torch.tensor([N - 1, 0], dtype=torch.int32, device=DEV)
lib, st = ssa._lib.lib, ssa.engine.stream()
ssa._lib.check(lib.ssac_chain_form(int(os.environ.get("SSAC_CHAIN_FORM", "1"))))
h1 = torch.zeros(N, B, H, device=DEV); h2 = torch.zeros_like(h1); q = torch.zeros(N, B, 1, device=DEV)
qt = torch.zeros(2, B, 1, device=DEV); dz2 = torch.zeros_like(h1); dz1 = torch.zeros_like(h1)
ho = torch.zeros(B * A, dtype=torch.int64, device=DEV)
xp, lpp = x1.clone(), torch.zeros(B, device=DEV)
ref = None
for k in range(n_launch):
    ssa._lib.check(lib.ssac_chain_update(
        C.byref(aa.desc()), xp.data_ptr(), S + A, B, eps.data_ptr(), -5.0, 2.0, xp.data_ptr(), S + A, S, lpp.data_ptr(),
        0, C.byref(ta.desc()), ids.data_ptr(), 2, qt.data_ptr(), C.byref(ca.desc()), xc.data_ptr(), S + A,
        h1.data_ptr(), h2.data_ptr(), q.data_ptr(), dz2.data_ptr(), dz1.data_ptr(), 0, 0, 0, ho.data_ptr(), 1, 0, st))
    if k % 2000 == 0:
        torch.cuda.synchronize()
        cur = [t.clone() for t in (h1, h2, q, qt, dz2, dz1, lpp)]
        if ref is None:
            ref = cur
        else:
            bad = [i for i, (a_, b_) in enumerate(zip(ref, cur)) if not torch.equal(a_, b_)]
            print("launch", k, "differs" if bad else "same", bad, flush=True)
torch.cuda.synchronize()
print("done", n_launch)
