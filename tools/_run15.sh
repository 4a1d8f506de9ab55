cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03c > gpurun_out/r03c_profile.log 2>&1
bash tools/profile_x3.sh r03c >> gpurun_out/r03c_profile.log 2>&1
cd $GRAFT_REPO_ROOT
PT_BENCH_PROF=1 PT_PROF_VERBOSE=1 timeout 300 python bench.py --stages det --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>&1 | grep pt_prof > gpurun_out/r03c/det_layers.txt
ls gpurun_out/r03c; cut -c1-600 gpurun_out/r03c/bench_full.json
