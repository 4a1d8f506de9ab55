cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
timeout 900 python -m pytest tests/test_gpu_det.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3w/pytest_bin0.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "det" 2>&1 | tail -15 >> gpurun_out/r3w/pytest_bin0.txt
for v in 0 1; do
  PT_DB_FUSE_BIN0=$v timeout 300 python bench.py --stages det --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r3w/det_bin$v.json 2> gpurun_out/r3w/det_bin$v.err
done
cat gpurun_out/r3w/pytest_bin0.txt
for v in 0 1; do python -c "import json,sys; d=json.loads(open('gpurun_out/r3w/det_bin$v.json').read().strip().splitlines()[-1]); print($v, d['value'])"; done
