set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
timeout 600 python -m pytest tests/test_gpu_det.py -m gpu -x -q -k "ws64 or conv_variants" 2>&1 | tail -15 > gpurun_out/r3w/pytest_ws64.txt
for v in 0 2; do
  PT_CONV_WS64=$v timeout 120 python tools/conv_bench.py 64 240 240 64 64 3 1 50 >> gpurun_out/r3w/conv_bench.txt 2>&1
  PT_CONV_WS64=$v timeout 120 python tools/conv_bench.py 44 256 256 64 64 3 1 50 >> gpurun_out/r3w/conv_bench.txt 2>&1
done
for v in 0 1; do
  PT_CONV_WS64=$v timeout 300 python bench.py --stages det --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r3w/det_ws$v.json 2> gpurun_out/r3w/det_ws$v.err
done
cat gpurun_out/r3w/pytest_ws64.txt gpurun_out/r3w/conv_bench.txt
for v in 0 1; do python -c "import json,sys; d=json.loads(open('gpurun_out/r3w/det_ws$v.json').read().strip().splitlines()[-1]); print($v, d['value'], d.get('roofline'))"; done
