set -x
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r02_bench_n1.json 2> gpurun_out/r02_bench_n1.err
cat gpurun_out/r02_bench_n1.json
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02_gpu_tests.log 2>&1; tail -2 gpurun_out/r02_gpu_tests.log
python tools/bench_configs.py anymal_c_rough cassie anymal_c_flat_trajectory 2>&1 | grep -v "amdgpu.ids\|Setting seed" > gpurun_out/r02_other_configs.txt; cat gpurun_out/r02_other_configs.txt
python tools/env_step_time.py 2>&1 | grep -v "amdgpu.ids\|Setting seed" > gpurun_out/r02_env_step_time.txt; cat gpurun_out/r02_env_step_time.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_alt --no_other --sustained 0 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_alt --no_other --sustained 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_alt --no_other --sustained 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_pmc_phys -- python3 $GRAFT_REPO_ROOT/tools/physics_prof.py > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/r02_*
