"""debug: eager critic updates in the co-resident form with a synchronisation after every update"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, args = sys.argv[:1] + ["__none__"], sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location("bc", os.path.join(ROOT, "tools", "bench_configs.py"))
bc = importlib.util.module_from_spec(spec); spec.loader.exec_module(bc)
import torch
import super_sac_amd as ssa
B, N = int(args[0]), int(args[1])
graphs = int(args[2]) if len(args) > 2 else 0
ssa.learning.USE_GRAPHS = bool(graphs)
if len(args) > 3:
    ssa.learning_utils.FOLD_GATHER = bool(int(args[3]))
critic, _ = bc.build(17, 6, B, N, 2)
for k in range(40):
    critic()
    torch.cuda.synchronize()
    print("update", k, "ok", flush=True)
