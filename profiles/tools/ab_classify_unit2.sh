cd /root/repo
for u in 64 256 1 256; do
  echo "== XS_CLASSIFY_UNIT=$u"
  XS_CLASSIFY_UNIT=$u timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100
  XS_CLASSIFY_UNIT=$u timeout -k 10 120 python3 profiles/tools/probe_s2_r4.py 20 2>/dev/null | tail -1
  XS_CLASSIFY_UNIT=$u XS_PROBE_N=1024 timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100
done
