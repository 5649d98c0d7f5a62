mkdir -p gpurun_out
python tools/stamp_auto.py 65536 20 0b101 > gpurun_out/stamp_auto_r03d.txt 2>&1; cat gpurun_out/stamp_auto_r03d.txt
python tools/stamp_auto.py 16384 > gpurun_out/stamp_auto_r03e.txt 2>&1; tail -8 gpurun_out/stamp_auto_r03e.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "moves or legal or slab_api_chunks or golden or sweep" > gpurun_out/gpu_tests_7.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests_7.log
python - <<'PY'
import json, sys, importlib, torch
sys.path.insert(0, '.')
import bench
pkg = importlib.import_module("doudizhu-rl_amd")
print(json.dumps({k: v for k, v in bench.stress_leg(pkg, torch, torch.device("cuda:0")).items() if k != "workload"})[:700])
PY
