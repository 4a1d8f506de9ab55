cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
o=gpurun_out/r3w/ws_batch.txt; : > $o
for B in 4 8 16 32 64; do
  it=$((19200 / B))
  PT_CONV_WS64=2 python tools/conv_bench.py $B 240 240 64 64 3 1 $it 2>/dev/null | sed 's/^/ws  : /' >> $o
  PT_CONV_WS64=0 python tools/conv_bench.py $B 240 240 64 64 3 1 $it 2>/dev/null | sed 's/^/v3h : /' >> $o
done
for B in 8 32; do
  it=$((9600 / B))
  python tools/conv_bench.py $B 240 240 256 64 3 1 $it 2>/dev/null | sed 's/^/bin0: /' >> $o
  python tools/conv_bench.py $B 120 120 128 128 3 1 $((it*4)) 2>/dev/null | sed 's/^/l2  : /' >> $o
done
cat $o
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r3w/four_ws1.json 2> gpurun_out/r3w/four_ws1.err
python -c "import json; d=json.loads(open('gpurun_out/r3w/four_ws1.json').read().strip().splitlines()[-1]); print('four stages', d['value'], d['ms_per_step'])"
