#!/bin/bash
# GPU box: the integrate kernel of scene S1 (probe_edge.py, own-pose and classified-ahead classes) for planes per streaming request group
# (XS_FREE_CHUNK, compile time) x walk bricks taken in parts (XS_WALK_SPLIT_CAP, environment of an XS_EXPERIMENTS build: 0 = never).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for chunk in ${CHUNKS:-1 2 4}; do
  touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc EXTRAFLAGS="-DXS_EXPERIMENTS -DXS_FREE_CHUNK=$chunk $EXTRA" > /dev/null 2>&1 || exit 1
  for cap in ${CAPS:-0 4096}; do
    echo "== XS_FREE_CHUNK=$chunk XS_WALK_SPLIT_CAP=$cap =="
    PROBE_QUICK=1 XS_WALK_SPLIT_CAP=$cap timeout -k 10 200 python3 profiles/tools/probe_edge.py ${SIZES:-512 1024} 2>&1 | grep "^n "
  done
done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
