"""deviations of the full-size pixel cases against the reference fixtures, per key (no assertions)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import case_runner, synth
for name in synth.FULL_SIZE:
    rec = case_runner.run_engine(name)
    fx = case_runner.load_fixture(name)
    print("==", name)
    for key, ref in fx.items():
        if case_runner._INPUT_KEY.fullmatch(key) or key not in rec:
            continue
        got, ref = np.asarray(rec[key], np.float64), np.asarray(ref, np.float64)
        if "_td" in key:
            print(f"  {key}: rel dev {np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))):.3e}  |ref| max {np.abs(ref).max():.3f}")
        elif "_log:" in key:
            print(f"  {key}: {float(got):.6g} vs {float(ref):.6g}  rel {abs(got - ref) / max(1e-6, abs(ref)):.2e}")
        else:
            err = np.abs(got - ref)
            print(f"  {key}: max {err.max():.3e} median {np.median(err):.3e}  > 3e-5: {int((err > 3e-5).sum())} of {err.size}")
