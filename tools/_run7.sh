cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r3w/pytest_gpu.txt
cat gpurun_out/r3w/pytest_gpu.txt
