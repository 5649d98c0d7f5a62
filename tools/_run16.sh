set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "policy or graph or tiny or slab_api" > gpurun_out/gpu_tests_16.log 2>&1; echo "tests rc=$?" ; tail -3 gpurun_out/gpu_tests_16.log
bash tools/profile.sh rollout > gpurun_out/prof_16.log 2>&1; echo "rollout pass rc=$?"
