#!/bin/bash
# kernel breakdown of the ConvNextViT recogniser: tools/cvit_prof.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
python $R/tools/cvit_bench.py --lines 2048 --steps 5 > $O/cvit_bench.txt 2>&1
python $R/tools/cvit_bench.py --lines 2048 --steps 3 --x3 >> $O/cvit_bench.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o cvit -- python $R/tools/cvit_bench.py --lines 2048 --steps 3 > $O/prof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/cvit_kernel_stats.csv
rm -rf $O/prof
grep convnext $O/cvit_bench.txt
head -9 $O/cvit_kernel_stats.csv | cut -c1-150
