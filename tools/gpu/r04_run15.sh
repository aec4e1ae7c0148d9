set -x
cd $GRAFT_REPO_ROOT
B="--steps 10 --warmup 3 --no_cpu_baseline --no_alt --no_other --sustained 0"
for rep in 1 2 3; do for v in 0 1; do
  LG_GATHER_SIDE=$v timeout -k 10 200 python bench.py $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('LG_GATHER_SIDE=$v rep $rep  iter_ms', d['ms_per_step'], 'rollout', d['config']['rollout_ms'], 'update', d['config']['update_ms'], 'mb_ms', d['roofline']['ms_per_minibatch'], 'frac', d['roofline']['frac'])" >> gpurun_out/r04_ab_gather.txt
done; done
cat gpurun_out/r04_ab_gather.txt
LG_GATHER_SIDE=1 timeout -k 10 600 python -m pytest tests/test_hip_ppo.py tests/test_hip_runner.py -m gpu -x -q 2>&1 | tail -3 | cut -c1-200
