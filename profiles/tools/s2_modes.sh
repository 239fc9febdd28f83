#!/bin/bash
# GPU box, repository root:  bash profiles/tools/s2_modes.sh [out_dir] [processes]
# Scene S2's integrate launch in its regimes (profiles/tools/probe_s2_modes.py) in several fresh processes one after another — round 4 saw the
# first processes of a box read 0.175 ms and later ones 0.215 — with the card's clocks / power / temperatures sampled meanwhile, then one
# counter pass each (kernel trace only) for the address-translation counters of the first-touch and every-word regimes, back to back and
# with the read sweep in front of the launch.
OUT=${1:-gpurun_out/s2_modes}
N=${2:-5}
mkdir -p $OUT
export TMPDIR=/tmp
for i in $(seq 1 $N); do
  timeout -k 10 120 python3 profiles/tools/probe_s2_modes.py process_$i > $OUT/process_$i.jsonl 2> $OUT/process_$i.err || exit 1
done
for C in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_SERIALIZATION_STALL"; do
  T=$(echo $C | tr ' ' '+')
  for R in first_touch every_word; do
    S2_ONLY=$R timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${R}_$T -- python3 profiles/tools/probe_s2_modes.py pmc > $OUT/pmc_${R}_$T.log 2>&1 || exit 1
  done
done
python3 profiles/tools/s2_modes_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
