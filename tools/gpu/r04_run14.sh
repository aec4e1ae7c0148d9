set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests_f.log 2>&1; tail -3 gpurun_out/r04_gpu_tests_f.log | cut -c1-200
B="--steps 10 --warmup 3 --no_cpu_baseline --no_alt --no_other --sustained 0"
for rep in 1 2 3; do for v in 0 1; do
  LG_HEAD_FIN_SIDE=$v timeout -k 10 200 python bench.py $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('LG_HEAD_FIN_SIDE=$v rep $rep  iter_ms', d['ms_per_step'], 'rollout', d['config']['rollout_ms'], 'update', d['config']['update_ms'], 'mb_ms', d['roofline']['ms_per_minibatch'], 'frac', d['roofline']['frac'])" >> gpurun_out/r04_ab_headfin.txt
done; done
cat gpurun_out/r04_ab_headfin.txt
