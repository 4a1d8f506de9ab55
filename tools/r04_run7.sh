#!/bin/bash
# round 4, seventh GPU pass: static-width PP-OCR task fix + guarded legs -> the default bench line end to end
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_onnx_seq.py tests/test_gpu_onnx_exec.py -x -q -m gpu > $O/pytest_onnx.txt 2>&1; tail -3 $O/pytest_onnx.txt
cd /tmp
timeout 1800 python $R/bench.py --steps 10 --warmup 3 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-200 $O/bench.json; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r04i/bench.json'))
print('value', d['value'], 'x3', d['tolerance_mode']['pages_per_s'])
print(json.dumps(d.get('onnx_recogniser'), indent=0))
print({k:(v.get('tables_per_s') if isinstance(v,dict) else v) for k,v in d.get('mtl_tabnet',{}).items() if k in ('bf16','bf16_kv8','bf16x3')})
PY
