import ctypes as C, subprocess, sys
if len(sys.argv) > 1:
    import torch
    torch.cuda.init(); x = torch.zeros(1, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    p = C.c_void_p()
    for flag, name in ((0x1, "finegrained"), (0x3, "uncached"), (0x0, "default")):
        if int(sys.argv[1]) != flag: continue
        r = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(1 << 16), C.c_uint(flag))
        print(name, "alloc rc", r, hex(p.value or 0), flush=True)
        src = (C.c_int32 * 16)(*range(100, 116))
        C.memmove(p.value, src, 64)   # CPU store into device memory (needs large BAR)
        print("cpu store ok", flush=True)
        t = torch.zeros(16, dtype=torch.int32, device="cuda")
        hip.hipMemcpy(C.c_void_p(t.data_ptr()), p, C.c_size_t(64), C.c_int(3))
        torch.cuda.synchronize()
        print("gpu sees", t.cpu().tolist(), flush=True)
else:
    for f in ("1", "3", "0"):
        r = subprocess.run([sys.executable, __file__, f], capture_output=True, text=True, timeout=120)
        print("flag", f, "rc", r.returncode, r.stdout.strip().replace("\n", " | "), r.stderr.strip()[-200:])
