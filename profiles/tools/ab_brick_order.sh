cd /root/repo
timeout -k 10 600 python -m pytest tests/test_integrate_gpu.py tests/test_signmap_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for lim in 2000000000 0 2000000000; do echo "== XS_INTEGRATE_SORT_LIMIT=$lim";
 XS_INTEGRATE_SORT_LIMIT=$lim timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100
 XS_INTEGRATE_SORT_LIMIT=$lim timeout -k 10 120 python3 profiles/tools/probe_s2_r4.py 20 2>/dev/null | tail -1
 XS_INTEGRATE_SORT_LIMIT=$lim XS_PROBE_N=1024 timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100
done
bash profiles/tools/probe_wg_times.sh
