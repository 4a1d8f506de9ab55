#!/bin/bash
# round 4, ninth GPU pass: dcn_mfma_kernel v2 -- parity, crash check, per-layer times against dcn_fused64_kernel, tsr-only A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r04m}
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_dcn_op.py -x -q -m gpu > $O/pytest_dcn.txt 2>&1; tail -2 $O/pytest_dcn.txt
for i in 1 2 3; do PT_CONV_VARIANT=0 timeout 300 python tools/scratch/t2.py 2>&1 | tail -1 | cut -c1-60; done
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/prof_$v
  PT_DCN_MFMA=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$v -- python $R/bench.py --stages tsr --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 > $O/bench_tsr_prof_$v.log 2>&1
  python $R/tools/dcn_by_layer.py /tmp/prof_$v | tee $O/dcn_by_layer_mfma$v.txt
done
for v in 1 0 1 0; do
  PT_DCN_MFMA=$v timeout 300 python $R/bench.py --stages tsr --no-cpu-baseline --no-extra-legs --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tsr only PT_DCN_MFMA=$v', round(d['value'],1), 'pages/s')"
done | tee $O/ab_tsr.txt
