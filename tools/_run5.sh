set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_auto.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "auto or rule or slab or graph or smoke" --durations=5 > gpurun_out/gpu_tests_5.log 2>&1; echo "tests rc=$?" ; tail -12 gpurun_out/gpu_tests_5.log
python tools/stamp_auto.py 16384 > gpurun_out/stamp_auto_r03b.txt 2>&1; cat gpurun_out/stamp_auto_r03b.txt
python examples/config4_rule_opponent.py --tables 65536 --iters 100 > gpurun_out/cfg4_r03b.txt 2>&1; tail -3 gpurun_out/cfg4_r03b.txt
python tools/_exp_slab_occ.py > gpurun_out/exp_slab_occ.txt 2>&1; cat gpurun_out/exp_slab_occ.txt
