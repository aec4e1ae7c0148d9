cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_n1.json 2> gpurun_out/r04_bench_n1.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_n1.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['rollout_ms'], d['config']['update_ms'], d['roofline']['frac'], d['roofline']['ms_per_minibatch'], d['roofline_env_step']['us_per_launch'])
print('sustained', d['sustained']['value'], d['sustained']['physics_fault_resets'], 'rough', d['other_configs'][0]['value'], 'cassie', d['other_configs'][1]['value'], 'alt', d['alt_reference_policy_dims']['value'], 'cpu', d['cpu_baseline']['value'])
PY
