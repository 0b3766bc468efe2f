"""which torch-level copies / uploads one pixel critic update issues (call sites, shapes):
    python tools/count_copies.py [dmc|atari]"""
import os, sys, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_pixels
which = sys.argv[1] if len(sys.argv) > 1 else "dmc"
step, B = bench_pixels.build(which, torch.device("cuda:0"))
for _ in range(3):
    step()
torch.cuda.synchronize()
seen = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "super_sac_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"
orig_copy, orig_to = torch.Tensor.copy_, torch.Tensor.to
def copy_(self, src, *a, **k):
    seen[("copy_", site(), tuple(self.shape), str(src.device), self.is_contiguous() and src.is_contiguous())] += 1
    return orig_copy(self, src, *a, **k)
def to(self, *a, **k):
    r = orig_to(self, *a, **k)
    if r.device != self.device:
        seen[("to", site(), tuple(self.shape), str(self.device), True)] += 1
    return r
torch.Tensor.copy_, torch.Tensor.to = copy_, to
step()
torch.cuda.synchronize()
torch.Tensor.copy_, torch.Tensor.to = orig_copy, orig_to
for k, v in sorted(seen.items(), key=lambda kv: kv[0][1]):
    print(v, k)
