#!/bin/bash
# round 4, sixth GPU pass: thin chain with prefetch (equality test + A/B), default bench with the onnx_recogniser leg
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_tsr.py -x -q -m gpu -k "thin_chain or lore_net" > $O/pytest_tsr.txt 2>&1; tail -3 $O/pytest_tsr.txt
cd /tmp
for v in 1 0 1 0; do
  PT_DLA_CHAIN=$v timeout 400 python $R/bench.py --no-cpu-baseline --no-extra-legs --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('four stages PT_DLA_CHAIN=$v', round(d['value'],1), 'pages/s')"
done | tee $O/ab.txt
timeout 1500 python $R/bench.py --steps 10 --warmup 3 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-200 $O/bench.json; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r04h/bench.json'))
print('value', d['value'], 'x3', d['tolerance_mode']['pages_per_s'])
print(json.dumps(d.get('onnx_recogniser'), indent=0))
bc=d['roofline']['by_class']['classes']
for k in list(bc)[:8]: print(k, bc[k])
PY
