cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_tsr.py tests/test_gpu_fullsize.py -m gpu -x -q -k "tsr or lore or Lore" 2>&1 | tail -3
for v in 0 1; do
  PT_STEM_THIN_X3=$v PT_BENCH_PROF=1 PT_PROF_VERBOSE=1 timeout 600 python bench.py --stages tsr --precision bf16x3 --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs 2> /tmp/tsr_$v.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tsr x3 thin=$v', round(d['value'],1), round(d['ms_per_step'],1))"
  grep -E "stem7x7" /tmp/tsr_$v.err
done
