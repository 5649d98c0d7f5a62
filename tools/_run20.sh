set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "q_slab or q_network or dqn" > gpurun_out/gpu_tests_20.log 2>&1; echo "tests rc=$?" ; tail -3 gpurun_out/gpu_tests_20.log
bash tools/profile.sh dqn > gpurun_out/prof_20.log 2>&1; tail -1 gpurun_out/prof/dqn/config3.txt
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/prof/dqn/stats/p_kernel_stats.csv')):
    if float(r["Percentage"]) > 0.5: print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
