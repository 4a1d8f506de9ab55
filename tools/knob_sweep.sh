#!/bin/bash
# tools/knob_sweep.sh: A/B of the engine's environment knobs on the four-stage bench (GPU box).  Every variant is
# bracketed by a default run, so drift of the box (power / temperature) shows up as drift of the baseline.
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {  # label, env assignments...
  local label=$1; shift
  local v=$(env "$@" python $R/bench.py --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['roofline']['achieved'],1))")
  echo "$label: $v"
}
for spec in "det_mb16 PT_DET_MICROBATCH=16" "det_mb64 PT_DET_MICROBATCH=64" "tsr_mb40 PT_TSR_MICROBATCH=40" "tsr_mb20 PT_TSR_MICROBATCH=20" \
            "rec_mb4096 PT_REC_MICROBATCH=4096" "rec_mb2560 PT_REC_MICROBATCH=2560" "ngroup0 PT_N_GROUP=0" "convvar1 PT_CONV_VARIANT=1" \
            "convvar0 PT_CONV_VARIANT=0" "dcn256 PT_DCN_THREADS=256" "lstm_mi2 PT_LSTM_MI=2" "gemm1x1 PT_GEMM1X1=1"; do
  set -- $spec
  run "base      " X=1
  run "$1" "$2"
done
run "base      " X=1
