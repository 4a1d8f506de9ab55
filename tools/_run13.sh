cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
timeout 600 python -m pytest tests/test_gpu_det.py -m gpu -x -q 2>&1 | tail -3
for v in 1 2; do
timeout 300 python bench.py --stages det --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('det', d['value'])"
done
PT_BENCH_PROF=1 PT_PROF_VERBOSE=1 timeout 300 python bench.py --stages det --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>&1 | grep pt_prof > gpurun_out/r3w/det_layers.txt
cat gpurun_out/r3w/det_layers.txt
