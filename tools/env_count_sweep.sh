# env-steps/s against the env count on one GPU (anymal_c_flat, [512,256,128]):  bash tools/env_count_sweep.sh > gpurun_out/rNN_env_count_sweep.txt
cd $GRAFT_REPO_ROOT
echo "python bench.py --num_envs N --steps 5 --warmup 2 --no_cpu_baseline --no_alt --no_other --sustained 0   (anymal_c_flat, [512,256,128], one MI355X, one box)"
echo "  envs/GPU  env-steps/s  ms/iter  rollout ms  update ms  GEMM-group frac of 416.7 TF  k_substeps us/launch"
for n in 1024 2048 4096 8192 16384 32768; do
  python bench.py --num_envs $n --steps 5 --warmup 2 --no_cpu_baseline --no_alt --no_other --sustained 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print(f\"  {$n:7d}  {j['value']:12,.0f}  {j['ms_per_step']:7.2f}  {j['config']['rollout_ms']:8.2f}  {j['config']['update_ms']:9.2f}  {j['roofline']['frac']:10.3f}  {j['roofline_env_step']['us_per_launch']:24.1f}\")"
done
