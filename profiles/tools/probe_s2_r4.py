import importlib, sys, json
sys.path.insert(0, '.')
import numpy as np, torch
capi = importlib.import_module('x-slam_amd.capi'); synth = importlib.import_module('x-slam_amd.synth')
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
r = bench.integrate_s2_probe(torch, capi, synth, 512, reps=reps)
print(json.dumps({'S2 ms': r['kernel_ms'], 'frac': r['frac'], 'always ms': r['every_word_stored']['kernel_ms'], 'always frac': r['every_word_stored']['frac'],
                  'first ms': r['first_touch']['kernel_ms'], 'first frac': r['first_touch']['frac'], 'first always ms': r['first_touch']['every_word_stored']['kernel_ms'],
                  'exact ms': r['per_voxel_walk_everywhere']['kernel_ms'], 'call ms': r['whole_call_ms'], 'boxes': [r['boxes'][k] for k in ('free', 'nothing_to_write', 'per_voxel_walk')]}))
