#!/bin/bash
# GPU box, repository root: every complex kernel with the SLP vectorizer off (no v_pk_*_f32), per file, against the plain -O3 build —
# the variants built by profiles/tools/build_slp_variants.sh, alternating on one box.  Per variant: bench.py track (frames/s, S2 integrate
# kernel, raycast alone, ICP iteration periods, stage times), the Hessian / loss probe and the Gauss-Newton probe.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
OUT=gpurun_out/r04_ab_slp.txt; : > $OUT
cp x-slam_amd/libxslam_hip.so /tmp/libxslam_hip.product.so
VARS=${VARS:-"base tsdf raycast icp map all"}
for round in 1 2; do
for v in $VARS; do
  cp x-slam_amd/variants/libxslam_hip.$v.so x-slam_amd/libxslam_hip.so
  echo "== round $round variant $v" >> $OUT
  timeout -k 10 240 python3 bench.py --workload track --no-cpu-baseline --steps 100 2>gpurun_out/ab_slp.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d['icp_us_per_iteration']; s2=d['roofline_s2']; r=d['raycast']; st=d['stages_ms']
print('fps', d['value'], d['repetitions_fps'], 'sustained', d.get('sustained',{}).get('frames_per_s'))
print('  S2 integrate ms', s2['kernel_ms'], 'always-store ms', s2['every_word_stored']['kernel_ms'], '| S1 integrate kernel ms', d['roofline']['kernel_ms'])
print('  raycast alone ms', r['ms_per_frame_alone'], 'every step', r['every_step']['ms_per_frame_alone'], '| icp us/iter', i['level0'], i['level1'], i['level2'])
print('  stages', {k:v for k,v in st.items() if k!='note'})
print('  bilinear fps', d.get('bilinear',{}).get('frames_per_s'), 'integrate ms', d.get('bilinear',{}).get('integrate_kernel_ms'))
" >> $OUT || { echo "bench failed for $v" >> $OUT; tail -5 gpurun_out/ab_slp.err >> $OUT; cp /tmp/libxslam_hip.product.so x-slam_amd/libxslam_hip.so; exit 1; }
  if [ $round = 1 ]; then
    timeout -k 10 120 python3 profiles/tools/probe_hess.py 2>/dev/null | head -1 >> $OUT || { cp /tmp/libxslam_hip.product.so x-slam_amd/libxslam_hip.so; exit 1; }
    timeout -k 10 120 python3 profiles/tools/probe_gn.py 512 2>/dev/null | head -1 >> $OUT || { cp /tmp/libxslam_hip.product.so x-slam_amd/libxslam_hip.so; exit 1; }
  fi
  tail -8 $OUT
done
done
cp /tmp/libxslam_hip.product.so x-slam_amd/libxslam_hip.so
