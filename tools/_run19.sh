cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
timeout 1200 python -m pytest tests/test_gpu_rec.py tests/test_gpu_fullsize.py -m gpu -x -q -k "rec or crnn" 2>&1 | tail -4
bash tools/_run17.sh > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3w/recx3_kernel_stats.csv')))
for r in rows:
    if any(k in r['Name'] for k in ('cand','argmax','wnorm')):
        print(f"{int(r['TotalDurationNs'])/1e6:9.2f} ms {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:60]}")
PY
