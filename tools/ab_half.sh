#!/bin/bash
# A/B of the 4-wave, two-per-CU, phase-staggered variant of the 16-channel-slice conv kernel.  bash tools/ab_half.sh
export PYTHONPATH=$PWD
mkdir -p gpurun_out/ab_half
for shape in "32 240 240 64 64" "32 120 120 128 128" "32 60 60 256 256" "32 30 30 512 512" "32 240 240 256 64" "32 120 120 256 64" \
             "44 256 256 64 64" "44 128 128 128 128" "44 64 64 256 256" "44 32 32 512 512" "44 256 256 64 256"; do
  for cfg in "0 0" "1 0" "1 50" "1 100" "1 150"; do
    set -- $cfg
    PT_CONV_HALF=$1 PT_CONV_STAGGER=$2 python tools/conv_bench.py $shape 3 1 40 2>&1 | tail -1 | sed "s/\$/ half=$1 stagger=$2/"
  done
done | tee gpurun_out/ab_half/ab.txt
PT_CONV_HALF=1 PT_LIB_PATH=$PWD/tools/scratch/lib_timing.so python tools/conv_bench.py 32 240 240 64 64 3 1 1 2>&1 | grep "^T" | sort -k16 -n | head -60 > gpurun_out/ab_half/timing_l1.txt
