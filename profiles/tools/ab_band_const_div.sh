#!/bin/bash
# Round 6: the four IEEE divides by fx / fy of a band voxel (TsdfFusion.cu:144-145) through the exhaustively verified short
# division (xs_constdiv.hip) or not.  GPU box, repository root, library built with EXTRAFLAGS=-DXS_EXPERIMENTS.  The integrate kernel of scene S1
# alone at 1024^3 and 512^3 (profiles/tools/probe_edge.py: classes decided ahead + sign map = the pipeline's launch), alternating.
for mode in on off on off on off; do
  if [ $mode = off ]; then export XS_BAND_CONST_DIV_OFF=1; else unset XS_BAND_CONST_DIV_OFF; fi
  echo "== short division of the band's constant divides: $mode"
  PROBE_QUICK=1 python profiles/tools/probe_edge.py 1024 512 2>/dev/null | grep "ahead  +sign"
done
