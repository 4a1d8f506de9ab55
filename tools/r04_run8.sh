#!/bin/bash
# round 4, eighth GPU pass: dcn_mfma_kernel -- op parity, Lore parity, A/B against dcn_fused64_kernel (tsr-only + four stages), kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r04k}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dcn_op.py -x -q -s -m gpu > $O/pytest_dcn.txt 2>&1; tail -4 $O/pytest_dcn.txt
timeout 900 python -m pytest tests/test_gpu_tsr.py tests/test_gpu_fullsize.py -x -q -s -m gpu -k "lore or tsr or dcn" > $O/pytest_lore.txt 2>&1; tail -4 $O/pytest_lore.txt
cd /tmp && export TMPDIR=/tmp
for v in 1 0 1 0; do
  PT_DCN_MFMA=$v timeout 300 python $R/bench.py --stages tsr --no-cpu-baseline --no-extra-legs --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tsr only PT_DCN_MFMA=$v', round(d['value'],1), 'pages/s')"
done | tee $O/ab_tsr.txt
for v in 1 0; do
  rm -rf /tmp/prof_$v
  PT_DCN_MFMA=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $R/bench.py --stages tsr --no-cpu-baseline --no-extra-legs --steps 4 --warmup 2 > $O/bench_tsr_prof_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_tsr_mfma$v.csv && grep "dcn_" $f | cut -c1-70,150-400
done
for v in 1 0 1 0; do
  PT_DCN_MFMA=$v timeout 400 python $R/bench.py --no-cpu-baseline --no-extra-legs --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('four stages PT_DCN_MFMA=$v', round(d['value'],1), 'pages/s')"
done | tee $O/ab_four.txt
