#!/bin/bash
# tools/knob_sweep2.sh: every fast path switched off in turn against the defaults (same bracketing as knob_sweep.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {
  local label=$1; shift
  local v=$(env "$@" python $R/bench.py --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['roofline']['achieved'],1))")
  echo "$label: $v"
}
for spec in "dcn_nb64 PT_DCN_NB=64" "stem_thin0 PT_STEM_THIN=0" "conv0_mfma0 PT_CONV0_MFMA=0" "pool_fused0 PT_POOL_FUSED=0" "cls_fused0 PT_CLS_FUSED=0" \
            "lstm_cluster0 PT_LSTM_CLUSTER=0" "tsr_fused0 PT_TSR_FUSED=0" "dwconvt2_0 PT_DWCONVT2=0" "dcn_fused0 PT_DCN_FUSED=0"; do
  set -- $spec
  run "base      " X=1
  run "$1" "$2"
done
run "base      " X=1
