"""time ONE configuration of tools/bench_configs.py (for rocprofv3 runs):
    python tools/one_config.py <obs> <act> <B> <N> <n> <fp32|bf16> [updates]"""
import sys, os
sys.argv, args = sys.argv[:1], sys.argv[1:]
sys.argv.append("__none__")
import importlib.util
spec = importlib.util.spec_from_file_location("bc", os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_configs.py"))
bc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bc)
obs, act, B, N, n = (int(v) for v in args[:5])
if os.environ.get("SSAC_WGRAD_VARIANT"):   # A/B of the weight-gradient launch's two forms (tools only)
    bc.ssa.engine.set_wgrad_variant(int(os.environ["SSAC_WGRAD_VARIANT"]))
critic, env_step = bc.build(obs, act, B, N, n, precision=args[5])
t = bc.timed(critic, int(args[6]) if len(args) > 6 else 1500, 200)
print(f"{args}: {t * 1e6:.1f} us per critic update")
