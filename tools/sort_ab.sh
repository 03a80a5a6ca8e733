for v in classic0 classic1 classic2 classic3 onesweep; do
  echo "== $v"
  FDGPU_SORT=$v timeout 300 python bench.py --structures 30000 --steps 3 --warmup 1 --no-query --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['roofline']['stages_ms'])"
done
