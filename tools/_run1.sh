set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_1.log 2>&1; echo "tests rc=$?" ; tail -5 gpurun_out/gpu_tests_1.log
python bench.py > gpurun_out/bench_1.json 2> gpurun_out/bench_1.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/bench_1.json
bash tools/profile_r03a.sh > gpurun_out/prof_r03a.log 2>&1; echo "prof rc=$?"; tail -20 gpurun_out/prof_r03a.log
