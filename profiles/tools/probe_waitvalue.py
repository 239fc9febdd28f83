"""Does hipStreamWaitValue32 work here, and how long after the host's store does the stream go on?  (GPU box, repository root)"""
import ctypes as C, time, sys
import torch
hip = C.CDLL("libamdhip64.so")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
res = {}
for name, alloc in (("pinned host", lambda p: hip.hipHostMalloc(C.byref(p), C.c_size_t(4096), C.c_uint(0x40000000 | 0x2))),
                    ("fine-grained device (BAR)", lambda p: hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(4096), C.c_uint(0x1))),
                    ("signal memory", lambda p: hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(8), C.c_uint(0x2)))):
    p = C.c_void_p()
    rc = alloc(p)
    if rc != 0:
        print(name, "alloc rc", rc); hip.hipGetLastError(); continue
    if name == "pinned host":
        C.memset(p, 0, 4096)
    else:
        assert hip.hipMemset(p, 0, 8) == 0 and hip.hipDeviceSynchronize() == 0
    flag = C.cast(p, C.POINTER(C.c_uint32))
    x = torch.zeros(1, device="cuda")
    lat = []
    ok = True
    for i in range(1, 40):
        rc = hip.hipStreamWaitValue32(st, p, C.c_uint32(i), C.c_uint(0), C.c_uint32(0xFFFFFFFF))   # hipStreamWaitValueGte = 0
        if rc != 0:
            print(name, "hipStreamWaitValue32 rc", rc); hip.hipGetLastError(); ok = False; break
        x.add_(1.0)
        ev = torch.cuda.Event(); ev.record()
        time.sleep(0.002)
        if ev.query():
            print(name, "the stream did not wait"); ok = False; break
        t0 = time.perf_counter()
        try:
            flag[0] = i
        except Exception as e:
            print(name, "host store failed", e); ok = False; break
        while not ev.query():
            pass
        lat.append((time.perf_counter() - t0) * 1e6)
    if ok:
        lat.sort()
        print(f"{name}: store -> the kernel behind the wait has finished: median {lat[len(lat)//2]:.1f} us, min {lat[0]:.1f} us")
