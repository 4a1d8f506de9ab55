#!/bin/bash
# round 4, second GPU pass: DCN op parity after the XCD-aware tile order, A/B against the identity order, e2e agreement test, bench with the new legs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dcn_op.py tests/test_gpu_tsr.py -x -q -m gpu > $O/pytest_dcn.txt 2>&1; tail -2 $O/pytest_dcn.txt
bash tools/dcn_abl.sh noxcd > $O/dcn_xcd_ab.txt 2>&1; cat $O/dcn_xcd_ab.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -s -m gpu > $O/pytest_e2e.txt 2>&1; tail -2 $O/pytest_e2e.txt; grep "E2E AGREEMENT" $O/pytest_e2e.txt
cd /tmp
timeout 900 python $R/bench.py --steps 10 --warmup 3 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-200 $O/bench.json; tail -3 $O/bench.err
