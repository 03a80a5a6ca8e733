for v in ${VARIANTS:-classic18 classic19 classic21}; do
  echo "== $v"
  FDGPU_SORT=$v timeout 300 python bench.py --steps 3 --warmup 1 --no-query --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['roofline']['stages_ms'])"
done
