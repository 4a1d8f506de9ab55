#!/bin/bash
# round 4, third GPU pass: new parity tests (1x1 row GEMM, MtlTabNet at 500/150 + table signal, e2e agreement), det-only A/B of the row-GEMM laterals,
# the default bench with the new legs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_det.py tests/test_gpu_mtl.py tests/test_gpu_e2e.py -x -q -s -m gpu > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
grep "E2E AGREEMENT\|mtl decoders at max\|cells x 151\|table-signal" $O/pytest_a.txt
cd /tmp
for v in 1 0 1 0; do
  PT_CONV1_ROWS=$v timeout 300 python $R/bench.py --stages det --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('det-only PT_CONV1_ROWS=$v', round(d['value'],1), 'pages/s')"
done
PT_PROF_VERBOSE=1 timeout 300 python $R/bench.py --stages det --no-cpu-baseline --no-extra-legs --steps 10 --warmup 3 2>$O/det_layers.err >/dev/null; grep "pt_prof" $O/det_layers.err | sort > $O/det_layers.txt; grep "conv1x1" $O/det_layers.txt
timeout 1200 python $R/bench.py --steps 10 --warmup 3 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-200 $O/bench.json; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r04e/bench.json'))
print('value', d['value'], 'det frac', d['roofline']['det_backbone']['frac'], 'x3', d['tolerance_mode']['pages_per_s'], 'host_pages', d.get('host_pages',{}).get('ratio_to_value'))
print(json.dumps(d['mtl_tabnet'].get('bf16'), indent=0))
for k,v in d['roofline'].get('by_class',{}).get('classes',{}).items(): print(k, v)
PY
