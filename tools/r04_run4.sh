#!/bin/bash
# round 4, fourth GPU pass: thin-chain equality test, A/B of PT_DLA_CHAIN and PT_CONV1_ROWS in the four-stage step, default bench with the new legs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_tsr.py -x -q -m gpu > $O/pytest_tsr.txt 2>&1; tail -4 $O/pytest_tsr.txt
cd /tmp
for v in "1 1" "0 1" "1 0" "1 1" "0 1" "1 0"; do
  set -- $v
  PT_DLA_CHAIN=$1 PT_CONV1_ROWS=$2 timeout 400 python $R/bench.py --no-cpu-baseline --no-extra-legs --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('four stages PT_DLA_CHAIN=$1 PT_CONV1_ROWS=$2', round(d['value'],1), 'pages/s')"
done | tee $O/ab.txt
timeout 1500 python $R/bench.py --steps 10 --warmup 3 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-200 $O/bench.json; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r04f/bench.json'))
print('value', d['value'], 'det frac', d['roofline']['det_backbone']['frac'], 'x3', d['tolerance_mode']['pages_per_s'], 'host_pages', d.get('host_pages',{}).get('ratio_to_value'))
print(json.dumps(d['mtl_tabnet'].get('bf16'), indent=0))
bc = d['roofline'].get('by_class',{})
print({k: v for k, v in bc.items() if k != 'classes'})
for k,v in bc.get('classes',{}).items(): print(k, v)
PY
