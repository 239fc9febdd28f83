for i in 1 2 3; do
python bench.py --no-s2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('low-prio aux', d['value'], d['stages_ms'])"
XS_AUX_FLAT_PRIORITY=1 python bench.py --no-s2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flat        ', d['value'], d['stages_ms'])"
done
