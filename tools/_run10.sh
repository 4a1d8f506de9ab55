cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
timeout 900 python -m pytest tests/test_gpu_det.py -m gpu -x -q 2>&1 | tail -5
PT_CONV_WS64=2 python tools/conv_bench.py 64 240 240 64 64 3 1 300 2>/dev/null | sed 's/^/ws2 : /'
PT_CONV_WS64=0 python tools/conv_bench.py 64 240 240 64 64 3 1 300 2>/dev/null | sed 's/^/v3h : /'
PT_CONV_WS64=2 python tools/conv_bench.py 44 256 256 64 64 3 1 300 2>/dev/null | sed 's/^/ws2 : /'
for v in 1 2; do
timeout 300 python bench.py --stages det --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2> gpurun_out/r3w/det_q.err | tail -1 > gpurun_out/r3w/det_q_$v.json
python -c "import json,sys; d=json.loads(open('gpurun_out/r3w/det_q_$v.json').read().strip().splitlines()[-1]); print($v, d['value'])"
done
