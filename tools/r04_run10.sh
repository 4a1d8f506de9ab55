#!/bin/bash
# round 4, tenth GPU pass: pipelined dcn_fused64_kernel (PIPE) -- bit equality with the un-pipelined kernel, parity, per-layer times, tsr-only A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r04p}
mkdir -p $O
cd $R
export PYTHONPATH=$R
PT_DCN_MFMA=0 PT_DCN_PIPE=1 python tools/scratch/t3.py /tmp/p1.npz && PT_DCN_MFMA=0 PT_DCN_PIPE=0 python tools/scratch/t3.py /tmp/p0.npz && python -c "
import numpy as np
a,b=np.load('/tmp/p1.npz'),np.load('/tmp/p0.npz')
print('PIPE vs no PIPE head maps identical:', all(np.array_equal(a[k],b[k]) for k in a.files))" | tee $O/pipe_equal.txt
PT_DCN_MFMA=0 timeout 900 python -m pytest tests/test_gpu_dcn_op.py -x -q -m gpu > $O/pytest_dcn.txt 2>&1; tail -2 $O/pytest_dcn.txt
cd /tmp && export TMPDIR=/tmp
for v in "PT_DCN_MFMA=0 PT_DCN_PIPE=1" "PT_DCN_MFMA=0 PT_DCN_PIPE=0"; do
  rm -rf /tmp/prof_v
  env $v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_v -- python $R/bench.py --stages tsr --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 > $O/bench_tsr_prof.log 2>&1
  echo "== $v"; python $R/tools/dcn_by_layer.py /tmp/prof_v | tee "$O/dcn_by_layer_$(echo $v | tr ' =' '__').txt"
done
for v in "PT_DCN_MFMA=0 PT_DCN_PIPE=1" "PT_DCN_MFMA=0 PT_DCN_PIPE=0" "PT_DCN_MFMA=1" "PT_DCN_MFMA=0 PT_DCN_PIPE=1" "PT_DCN_MFMA=0 PT_DCN_PIPE=0" "PT_DCN_MFMA=1"; do
  env $v timeout 300 python $R/bench.py --stages tsr --no-cpu-baseline --no-extra-legs --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tsr only $v', round(d['value'],1), 'pages/s')"
done | tee $O/ab_tsr.txt
