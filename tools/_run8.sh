cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
timeout 900 python -m pytest tests/test_gpu_det.py -m gpu -x -q 2>&1 | tail -5
for v in 0 1; do
PT_STEM_POOL_WS=$v timeout 300 python bench.py --stages det --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2> gpurun_out/r3w/det_st.err | tail -1 > gpurun_out/r3w/det_st_$v.json
python -c "import json,sys; d=json.loads(open('gpurun_out/r3w/det_st_$v.json').read().strip().splitlines()[-1]); print($v, d['value'])"
done
PT_BENCH_PROF=1 PT_PROF_VERBOSE=1 timeout 300 python bench.py --stages det --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>&1 | grep pt_prof > gpurun_out/r3w/det_layers.txt
cat gpurun_out/r3w/det_layers.txt
