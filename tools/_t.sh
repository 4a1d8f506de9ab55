cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rec.py -m gpu -x -q 2>&1 | tail -3
PT_LSTM_MI=2 timeout 900 python -m pytest tests/test_gpu_rec.py -m gpu -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "rec or crnn" 2>&1 | tail -3
for v in 2 3; do
  PT_LSTM_MI=$v PT_BENCH_PROF=1 PT_PROF_VERBOSE=1 timeout 600 python bench.py --stages rec --precision bf16x3 --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs 2> /tmp/rec_$v.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rec x3 MI=$v', round(d['value'],1), round(d['ms_per_step'],1))"
  grep -E "lstm" /tmp/rec_$v.err
done
