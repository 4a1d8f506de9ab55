cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_tsr.py tests/test_gpu_fullsize.py -m gpu -x -q -k "tsr or lore or Lore or dcn" 2>&1 | tail -3
for v in 64 128 64 128; do
  PT_DCN_NB=$v timeout 600 python bench.py --stages tsr --precision bf16x3 --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tsr x3 nb=$v', round(d['value'],1), round(d['ms_per_step'],1))"
done
