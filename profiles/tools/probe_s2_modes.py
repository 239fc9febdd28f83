"""Scene S2's integrate launch in its three regimes, with what the card was doing meanwhile (GPU box, repository root).

Round 4 left two things unexplained: `every_word_stored` and `first_touch` read 0.175-0.180 ms in the first processes on
a fresh box and 0.215 ms in later ones, and both are timed right behind a kernel that has just written hundreds of MB (xs_init_volume's
1.5 GiB, or the previous repetition's own 474 MB of stores), whose dirty cache lines the timed launch then pays for.  This probe
  * times each regime back to back (as bench.py did) AND with a read sweep of 1 GiB of unrelated memory between the predecessor and the
    timed launch (the sweep evicts the predecessor's dirty lines from L2 / Infinity Cache: the launch then pays only for its own bytes),
  * samples the card's sclk / mclk / fclk, socket power and temperatures from sysfs every ~2 ms while each regime runs in a loop of >= 0.4 s,
  * prints per-launch min / median / max kernel time (event pair on the dispatch packet).
Run it several times in a row in one gpurun call (fresh processes) to see the process-to-process split; profiles/tools/pmc_s2_modes.sh
collects the address-translation counters of the same launches.
    python profiles/tools/probe_s2_modes.py [tag]"""
import ctypes as C, glob, importlib, json, os, sys, threading, time
sys.path.insert(0, '.')
import numpy as np, torch
capi = importlib.import_module('x-slam_amd.capi'); synth = importlib.import_module('x-slam_amd.synth')
H, W = synth.HEIGHT, synth.WIDTH
tag = sys.argv[1] if len(sys.argv) > 1 else "run"
ONLY = os.environ.get("S2_ONLY")        # one regime only (the counter passes)


class Smi(threading.Thread):
    """sysfs poller: current sclk (freq1_input), mclk (freq2_input), power (power1_input / power1_average), junction / memory temperature"""

    def __init__(self):
        super().__init__(daemon=True)
        hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        self.hw = hw[0] if hw else None
        self.files = {}
        if self.hw:
            for key, names in dict(sclk=["freq1_input"], mclk=["freq2_input"], power=["power1_input", "power1_average"], t_junction=["temp2_input", "temp1_input"],
                                   t_mem=["temp3_input"]).items():
                for nm in names:
                    p = os.path.join(self.hw, nm)
                    if os.path.exists(p):
                        self.files[key] = p
                        break
        dev = os.path.dirname(os.path.dirname(self.hw)) if self.hw else None
        self.dpm = {k: os.path.join(dev, f"pp_dpm_{k}") for k in ("sclk", "mclk", "fclk")} if dev else {}
        self.samples = []
        self.on = False
        self.stop = False

    def read(self):
        s = {}
        for k, p in self.files.items():
            try:
                s[k] = float(open(p).read().strip())
            except (OSError, ValueError):
                pass
        for k, p in self.dpm.items():
            try:
                for line in open(p).read().splitlines():
                    if line.rstrip().endswith("*"):
                        s["dpm_" + k] = float(''.join(ch for ch in line.split(":")[1] if ch.isdigit() or ch == '.'))
            except (OSError, ValueError, IndexError):
                pass
        return s

    def run(self):
        while not self.stop:
            if self.on:
                self.samples.append(self.read())
            time.sleep(0.002)

    def window(self):
        self.samples = []
        self.on = True

    def close_window(self):
        self.on = False
        out = {}
        for k in sorted({k for s in self.samples for k in s}):
            v = np.array([s[k] for s in self.samples if k in s])
            scale = 1e-6 if k in ("sclk", "mclk", "power") else (1e-3 if k.startswith("t_") else 1.0)   # Hz -> MHz, uW -> W, mC -> C
            out[k] = [round(float(v.min() * scale), 1), round(float(v.mean() * scale), 1), round(float(v.max() * scale), 1)]
        out["samples"] = len(self.samples)
        return out


def main():
    n = 512
    prm = synth.s2_params(n); res = [n, n, n]; vs = np.float32(prm["tsdf_voxel_size"]); trunc = synth.tranc_dist(prm)
    pad = int(os.environ.get("PROBE_PITCH_PAD_FLOATS", "0"))     # round 6: a row pitch that is no power of two (plane stride 1 MiB + pad * 2 KiB)
    pitch = n + pad
    value = torch.empty((n * n, pitch), dtype=torch.float32, device="cuda"); weight = torch.empty((n * n, pitch), dtype=torch.int32, device="cuda")
    grad = torch.empty((n * n, pitch), dtype=torch.float32, device="cuda")
    capi.init_volume(value, weight, grad, pitch * 4, res)
    depth = torch.from_numpy(synth.render_s2().view(np.int16)).cuda()
    scaled = torch.empty((H, W), dtype=torch.float32, device="cuda"); dmax = torch.zeros(1, dtype=torch.float32, device="cuda")
    capi.scale_depth_max(depth, W * 2, H, W, scaled, W * 4, dmax)
    ws = torch.zeros(capi.integrate_workspace_bytes(res), dtype=torch.uint8, device="cuda")
    R = np.zeros((3, 3, 2), np.float32); R[[0, 1, 2], [0, 1, 2], 0] = 1
    t = np.zeros((3, 2), np.float32); t[:, 0] = [-prm["init_x"], -prm["init_y"], -prm["init_z"]]; t[0, 1] = 1e-7
    intr = np.array([synth.FX, synth.FY, synth.CX, synth.CY], np.float32)
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream()
    args = (scaled, W * 4, H, W, intr, 100, res, float(vs), R, t, trunc, value, weight, grad, pitch * 4)
    capi.integrate_scaled(*args, updated=counter, depth_max=dmax, workspace=ws, stream=s)
    torch.cuda.synchronize()
    U = int(counter.item()); nbytes = 24.0 * U + 2.0 * W * H
    sweep_buf = torch.ones(1 << 28, dtype=torch.float32, device="cuda")     # 1 GiB of unrelated memory
    hip = C.CDLL("libamdhip64.so"); ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        assert hip.hipEventCreate(C.byref(e)) == 0
    smi = Smi(); smi.start()
    fresh = lambda: capi.init_volume(value, weight, grad, pitch * 4, res, stream=s)
    sweep = lambda: sweep_buf.sum()

    def regime(name, flags, before=(), min_s=0.4, max_reps=400):
        ms = []
        smi.window()
        t0 = time.perf_counter()
        while (time.perf_counter() - t0 < min_s or len(ms) < 8) and len(ms) < max_reps:
            for f in before:
                f()
            capi.integrate_scaled_ex(*args, flags, depth_max=dmax, workspace=ws, stream=s, start_event=ev[0], stop_event=ev[1])
            torch.cuda.synchronize()
            dt = C.c_float(0); assert hip.hipEventElapsedTime(C.byref(dt), ev[0], ev[1]) == 0
            ms.append(dt.value)
        card = smi.close_window()
        ms = np.array(ms)
        med = float(np.median(ms))
        rec = {"regime": name, "launches": len(ms), "kernel_ms": {"min": round(float(ms.min()), 4), "median": round(med, 4), "max": round(float(ms.max()), 4)},
               "frac_of_8TBs_algorithmic": round(nbytes / med / 1e6 / 8000.0, 4), "card": card}
        print(json.dumps(rec), flush=True)
        return rec

    print(json.dumps({"tag": tag, "pid": os.getpid(), "U": U, "algorithmic_MB": round(nbytes / 1e6, 1), "hwmon": smi.hw, "idle": smi.read(),
                      "array_addresses": [hex(value.data_ptr()), hex(weight.data_ptr()), hex(grad.data_ptr())]}), flush=True)
    todo = [("warm_up (steady state, clocks settle)", 0, ()),
            ("steady (only changed words stored)", 0, ()),
            ("steady + sweep", 0, (sweep,)),
            ("every_word_stored, back to back", 8, ()),
            ("every_word_stored + sweep between repetitions", 8, (sweep,)),
            ("first_touch (init_volume, then the launch)", 0, (fresh,)),
            ("first_touch + sweep between init and launch", 0, (fresh, sweep)),
            ("first_touch every word, back to back", 8, (fresh,)),
            ("first_touch every word + sweep", 8, (fresh, sweep)),
            ("steady again", 0, ())]
    for name, flags, before in todo:
        if ONLY and not name.startswith(ONLY) and not name.startswith("warm_up"):
            continue
        regime(name, flags, before, min_s=0.1 if ONLY else 0.4, max_reps=12 if ONLY else 400)
    smi.stop = True


if __name__ == "__main__":
    main()
