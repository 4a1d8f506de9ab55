#!/bin/bash
# round 4, fifth GPU pass: fp8 MFMA micro-benchmark, the whole GPU test suite, default bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
timeout 300 tools/scratch/fp8_gemm_bench > $O/fp8_gemm_bench.txt 2>&1; cat $O/fp8_gemm_bench.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
cd /tmp
timeout 1500 python $R/bench.py --steps 10 --warmup 3 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-200 $O/bench.json; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r04g/bench.json'))
db=d['roofline']['det_backbone']
print('value', d['value'], 'det frac', db['frac'], 'net_only', db.get('net_only'), 'x3', d['tolerance_mode']['pages_per_s'], 'host_pages', d.get('host_pages',{}).get('ratio_to_value'))
for k,v in d['roofline'].get('by_class',{}).get('top_labels',{}).items(): print(k, v)
PY
