cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "det" 2>&1 | tail -4
for v in 0 1 0 1 0 1; do
  PT_BENCH_DEFER_SCORES=$v timeout 300 python bench.py --stages det --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('det defer=$v', round(d['value']), d['config']['boxes_per_page'])"
done
