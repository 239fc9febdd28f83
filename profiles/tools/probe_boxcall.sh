#!/bin/bash
# GPU box: S1 integrate kernel time with the classification call compiled in / out (XS_PROBE_NO_BOXCALL), exact mode both
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
export XS_INTEGRATE_NO_TILES=1
for d in "" "-DXS_PROBE_NO_BOXCALL"; do
  touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc HIPFLAGS="$F $d" > /dev/null 2>&1
  echo "== build [$d] exact mode"
  timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 || exit 1
done
unset XS_INTEGRATE_NO_TILES
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
echo "== product build, tiles on, under rocprofv3"
rm -rf gpurun_out/pb; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pb -- python3 profiles/tools/probe_integrate.py > gpurun_out/pb.log 2>&1; tail -2 gpurun_out/pb.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pb/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
