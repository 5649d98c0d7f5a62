set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/r01h_bench.json 2> gpurun_out/r01h_bench.err
tail -c 600 gpurun_out/r01h_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01h_stats -o r01h -- python3 bench.py --no-cpu-baseline > gpurun_out/r01h_prof_bench.json 2>gpurun_out/r01h_prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r01h_pmc_$c -o p -- python3 tools/run_rollout.py 4096 2000 > gpurun_out/r01h_pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d gpurun_out/r01h_pmc_mix -o p -- python3 tools/run_rollout.py 4096 20000 > gpurun_out/r01h_pmc_mix.log 2>&1
find gpurun_out/r01h_* -name "*.csv" | head -20
