"""does hipDeviceScheduleSpin shorten the driver-style 20-step bursts?   python tools/spin_test.py [0|1]"""
import ctypes, os, sys, subprocess
flag = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if flag:
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipSetDeviceFlags(1)   # hipDeviceScheduleSpin
    print("hipSetDeviceFlags(spin) ->", rc)
sys.argv = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), "--steps", "20", "--warmup", "5",
            "--no-cpu-baseline", "--no-secondary"]
sys.path.insert(0, os.path.dirname(sys.argv[0]))
import runpy
runpy.run_path(sys.argv[0], run_name="__main__")
