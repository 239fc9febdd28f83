#!/bin/bash
# GPU box, repository root: the integrate kernel with and without the brick classification from the depth tiles (XS_INTEGRATE_NO_TILES=1
# walks every voxel): S1 at 512^3 (24 tracked frames, kernel time from the dispatch's own event pair) and the S2 probe
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
OUT=gpurun_out/ab_tiles.txt; : > $OUT
for rep in 1 2; do
for v in tiles exact; do
  if [ $v = exact ]; then export XS_INTEGRATE_NO_TILES=1; else unset XS_INTEGRATE_NO_TILES; fi
  echo "== $v (round $rep)" >> $OUT
  timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 >> $OUT || exit 1
  timeout -k 10 120 python3 profiles/tools/probe_s2.py 20 2>/dev/null | tail -1 >> $OUT || exit 1
done; done
cat $OUT
