import importlib, sys
sys.path.insert(0, '.')
import torch
capi = importlib.import_module('x-slam_amd.capi')
s = torch.cuda.current_stream()
for nb in (1000000, 64 << 20):
    ba = torch.empty((nb, 2), dtype=torch.float32, device="cuda").uniform_(-2, 2); bb = torch.empty((nb, 2), dtype=torch.float32, device="cuda").uniform_(0.05, 2)
    ba[:, 1] = 1e-6; bb[:, 1] = 1e-6
    bo = torch.empty_like(ba)
    row = {}
    for name in ("mul", "exp", "sin", "pow"):
        for variant in ("our", "raw"):
            reps = 50 if nb < (1 << 22) else 8
            capi.csfd_array_op(name, variant, ba, bb, bo, nb, stream=s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(reps):
                capi.csfd_array_op(name, variant, ba, bb, bo, nb, stream=s)
            e1.record(s); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            row[name + "_" + variant] = (round(ms, 5), round(24.0 * nb / ms / 1e6, 1))
    print(nb, row, flush=True)
