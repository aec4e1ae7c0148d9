# One round's measurements on the GPU box:  bash tools/collect_profiles.sh r04   (then tools/digest_profiles.py r03 [r03_anymal_c_rough])
TAG=${1:-r04}
set -x
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
cat gpurun_out/${TAG}_bench_n1.json
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gpu_tests.log 2>&1; tail -2 gpurun_out/${TAG}_gpu_tests.log
python tools/env_step_time.py 2>&1 | grep -v "amdgpu.ids\|Setting seed" > gpurun_out/${TAG}_env_step_time.txt; cat gpurun_out/${TAG}_env_step_time.txt
# phase clocks need the stamped build (make prof in legged_gym_dev_amd/csrc)
LG_HIP_LIB=$GRAFT_REPO_ROOT/legged_gym_dev_amd/lib/liblegged_hip_prof.so python tools/post_step_phases.py 2>&1 | grep cycles > gpurun_out/${TAG}_post_step_phases.txt; cat gpurun_out/${TAG}_post_step_phases.txt
LG_CLOCK_JSON=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_substeps_clock.json LG_HIP_LIB=$GRAFT_REPO_ROOT/legged_gym_dev_amd/lib/liblegged_hip_prof.so python tools/substeps_sections.py anymal_c_flat 2>&1 | grep -v "amdgpu.ids\|Setting seed" > gpurun_out/${TAG}_substeps_sections.txt; cat gpurun_out/${TAG}_substeps_sections.txt
cd /tmp && export TMPDIR=/tmp
B="--no_cpu_baseline --no_alt --no_other --sustained 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_phys -- python3 $GRAFT_REPO_ROOT/tools/physics_prof.py > /dev/null 2>&1
# BASELINE configs[2]: the same three passes on the rough-terrain task (235 observations, height scan, 10 x 20 tile terrain)
R="--task anymal_c_rough"
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_anymal_c_rough_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $R $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_anymal_c_rough_pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 $R $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_anymal_c_rough_pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 $R $B > /dev/null 2>&1
# BASELINE configs[4]: the biped on the same terrain (169 observations, 2 x 6 kinematic tree, PD law)
R="--task cassie"
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_cassie_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $R $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_cassie_pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 $R $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_cassie_pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 $R $B > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
# keep what travels back small: the counter / trace CSVs only
find gpurun_out/${TAG}_*prof gpurun_out/${TAG}_*pmc_* -type f ! -name "*.csv" -delete 2>/dev/null
du -sh gpurun_out/${TAG}_* | tail -20
