cd "$(dirname "$0")/../.."
for f in "--classify-slack 4" "--no-integrate-post" "--classify-slack 8" "--no-integrate-post" "--classify-slack 4" "--classify-slack 2"; do
  timeout -k 10 300 python3 bench.py --workload track --no-s2 --no-cpu-baseline $f 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f fps', d['value'], d['repetitions_fps'], 'integrate kernel ms', d['roofline']['kernel_ms'], 'sustained', d['sustained']['frames_per_s'])" || exit 1
done
