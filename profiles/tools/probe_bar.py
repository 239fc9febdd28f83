"""Is fine-grained device memory (hipExtMallocWithFlags) writable from the CPU on this box (large BAR)?"""
import ctypes as C, subprocess, sys
if len(sys.argv) > 1:
    hip = C.CDLL("libamdhip64.so")
    flag = int(sys.argv[1], 0)
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(4096), C.c_uint(flag))
    print("alloc rc", rc, hex(p.value or 0), flush=True)
    hip.hipMemset(p, 0, C.c_size_t(4096)); hip.hipDeviceSynchronize()
    attr = (C.c_byte * 256)()
    a = C.cast(p.value, C.POINTER(C.c_uint))
    print("cpu read", a[0], flush=True)
    a[0] = 1234
    print("cpu write ok", a[0], flush=True)
    out = C.c_uint(0)
    hip.hipMemcpy(C.byref(out), p, C.c_size_t(4), C.c_int(2)); print("device sees", out.value, flush=True)
else:
    for flag in ("0x1", "0x3", "0x0"):   # finegrained, uncached, default
        r = subprocess.run([sys.executable, __file__, flag], capture_output=True, text=True)
        print("flag", flag, "rc", r.returncode, r.stdout.strip().replace("\n", " | "), r.stderr.strip()[-200:])
