"""bench.py's `secondary.acting` rows on their own (GPU box):  python tools/acting_rows.py [--general]
--general: with the one-call path switched off (super_sac_amd.acting.ENABLED = False): the eager path of agent.py."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import super_sac_amd as ssa
from super_sac_amd import acting
if "--general" in sys.argv:
    acting.ENABLED = False
torch.set_num_threads(8)
out = bench.acting_rows(torch.device("cuda"))
for k, v in out["rows"].items():
    print(f"{k:40s} {v['us_per_call_median']:8.1f} us   p90 {v['us_per_call_p90']:8.1f}")
print(json.dumps(out))
