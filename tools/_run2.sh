cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
R=$GRAFT_REPO_ROOT
o=gpurun_out/r3w/ws_abl.txt; : > $o
PT_CONV_WS64=0 python tools/conv_bench.py 64 240 240 64 64 3 1 300 2>/dev/null | sed 's/^/v3h : /' >> $o
PT_CONV_WS64=2 python tools/conv_bench.py 64 240 240 64 64 3 1 300 2>/dev/null | sed 's/^/ws  : /' >> $o
for v in 1 4 8 16 12 28; do
  PT_CONV_WS64=2 PT_LIB_PATH=$R/tools/scratch/lib_abl$v.so python tools/conv_bench.py 64 240 240 64 64 3 1 300 2>/dev/null | sed "s/^/abl$v: /" >> $o
done
cat $o
