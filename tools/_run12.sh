cd $GRAFT_REPO_ROOT
for mb in 64 32 16 64 32; do
  PT_DET_MICROBATCH=$mb timeout 300 python bench.py --stages det --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mb $mb', d['value'])"
done
