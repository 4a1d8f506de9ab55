cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3w
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3w/recx3 -- python $R/bench.py --stages rec --precision bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/r3w/recx3.log 2>&1
KS=$(find $R/gpurun_out/r3w/recx3 -name "*kernel_stats.csv" | head -1); cp $KS $R/gpurun_out/r3w/recx3_kernel_stats.csv; rm -rf $R/gpurun_out/r3w/recx3
head -14 $R/gpurun_out/r3w/recx3_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
