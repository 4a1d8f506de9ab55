cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_det.py tests/test_gpu_pipeline.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -3
for v in 1 2 3; do
  timeout 300 python bench.py --stages det --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('det', round(d['value']))"
done
