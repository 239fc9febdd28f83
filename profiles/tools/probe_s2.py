import importlib, sys, json
sys.path.insert(0, '.')
import numpy as np, torch
capi = importlib.import_module('x-slam_amd.capi'); synth = importlib.import_module('x-slam_amd.synth')
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
r = bench.integrate_s2_probe(torch, capi, synth, 512, reps=reps)
print(json.dumps({'scene': 'S2 512^3', 'U': r['U'], 'ms': r['kernel_ms'], 'call_ms': r['whole_call_ms'], 'achieved_GBs': r['achieved'], 'frac': r['frac']}))
