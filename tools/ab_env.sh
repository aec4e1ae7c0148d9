# A/B of environment knobs on the bench's minibatch group and iteration time:  tools/ab_env.sh "LG_X=1" "LG_X=0" ...
cd $GRAFT_REPO_ROOT
for kv in "$@"; do
  for rep in 1 2; do
    env $kv python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_alt --no_other --sustained 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$kv', 'iter_ms', j['ms_per_step'], 'rollout', j['config']['rollout_ms'], 'update', j['config']['update_ms'], 'mb_ms', j['roofline']['ms_per_minibatch'])"
  done
done
