cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
timeout 600 python tools/_drift.py 2>/dev/null > gpurun_out/r3w/drift.txt
cat gpurun_out/r3w/drift.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -5
