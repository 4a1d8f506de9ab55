cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
for rep in 1 2; do
for lib in "" $R/tools/scratch/lib_wsold.so; do
  tag=new; [ -n "$lib" ] && tag=old
  PT_LIB_PATH=$lib PT_CONV_WS64=2 python tools/conv_bench.py 64 240 240 64 64 3 1 300 2>/dev/null | sed "s/^/$tag       : /"
  PT_LIB_PATH=$lib PT_BENCH_RES=1 PT_CONV_WS64=2 python tools/conv_bench.py 64 240 240 64 64 3 1 300 2>/dev/null | sed "s/^/$tag +res  : /"
done
done
for lib in "" $R/tools/scratch/lib_wsold.so; do
  tag=new; [ -n "$lib" ] && tag=old
  PT_LIB_PATH=$lib timeout 300 python bench.py --stages det --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['value'])"
done
