cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
for v in 0 1; do
PT_CLS_X3_REFINE=$v PT_BENCH_PROF=1 PT_PROF_VERBOSE=1 timeout 600 python bench.py --stages rec --precision bf16x3 --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs 2> gpurun_out/r3w/rec_x3_$v.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rec x3 refine=$v', d['value'], d['ms_per_step'])"
grep -E "classifier|argmax|512->7680" gpurun_out/r3w/rec_x3_$v.err
done
