#!/bin/bash
# Build container (or GPU box), repository root: libxslam_hip.so variants with the SLP vectorizer off per file, for profiles/tools/ab_slp.sh.
# Outputs x-slam_amd/variants/libxslam_hip.<variant>.so (git-ignored, travel with gpurun) and a static table on stdout:
# v_pk_* instructions, VGPRs and scratch of the kernels each file holds.
cd "$(dirname "$0")/../.." || exit 1
C=x-slam_amd/csrc; V=x-slam_amd/variants; T=${TMPDIR:-/tmp}/slp_variants; mkdir -p $V $T
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
FILES="xs_tsdf xs_raycast xs_icp xs_map"
for f in $(cd $C && ls *.hip | sed 's/\.hip//'); do
  ( /opt/rocm/bin/hipcc $F -c $C/$f.hip -o $T/$f.base.o ) &
  case " $FILES " in *" $f "*) ( /opt/rocm/bin/hipcc $F -fno-slp-vectorize -c $C/$f.hip -o $T/$f.noslp.o ) & ;; esac
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.5; done
done; wait
link() {  # variant name, list of files built without SLP
  local objs=""
  for f in $(cd $C && ls *.hip | sed 's/\.hip//'); do
    case " $2 " in *" $f "*) objs="$objs $T/$f.noslp.o" ;; *) objs="$objs $T/$f.base.o" ;; esac
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libxslam_hip.$1.so $objs
}
link base ""; link tsdf "xs_tsdf"; link raycast "xs_raycast"; link icp "xs_icp"; link map "xs_map"; link all "$FILES"
# static figures from the device assembly
for f in $FILES; do for v in base noslp; do
  X=""; [ $v = noslp ] && X="-fno-slp-vectorize"
  ( /opt/rocm/bin/hipcc $F $X -S --cuda-device-only -o $T/$f.$v.s $C/$f.hip 2>/dev/null ) &
done; done; wait
python3 - $T <<'PY'
import re, sys, os
T = sys.argv[1]
print(f"{'file':11s} {'build':6s} {'v_pk_*':>7s} {'VALU lines':>11s}   kernels: name vgpr scratch_bytes (base -> noslp where they differ)")
for f in ("xs_tsdf", "xs_raycast", "xs_icp", "xs_map"):
    meta = {}
    for v in ("base", "noslp"):
        s = open(os.path.join(T, f"{f}.{v}.s")).read()
        pk = len(re.findall(r"^\s*v_pk_", s, re.M)); valu = len(re.findall(r"^\s*v_", s, re.M))
        ks = {}
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", s):
            ks[m.group(1)] = (int(m.group(3)), int(m.group(2)))
        meta[v] = ks
        print(f"{f:11s} {v:6s} {pk:7d} {valu:11d}")
    for k in sorted(meta["base"]):
        b, n = meta["base"][k], meta["noslp"].get(k)
        if n is not None and b != n:
            print(f"    {k[:90]:90s} vgpr {b[0]:3d} -> {n[0]:3d}   scratch {b[1]:4d} -> {n[1]:4d}")
PY
