set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_auto.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "auto or rule or q_slab or q_network or env_view or smoke" --durations=5 > gpurun_out/gpu_tests_6.log 2>&1; echo "tests rc=$?" ; tail -12 gpurun_out/gpu_tests_6.log
python examples/config4_rule_opponent.py --tables 65536 --iters 100 > gpurun_out/cfg4_r03c.txt 2>&1; tail -2 gpurun_out/cfg4_r03c.txt
python examples/config4_rule_opponent.py --tables 4096 --iters 200 > gpurun_out/cfg4_r03c_4096.txt 2>&1; tail -2 gpurun_out/cfg4_r03c_4096.txt
python tools/stamp_auto.py 16384 > gpurun_out/stamp_auto_r03c.txt 2>&1; head -12 gpurun_out/stamp_auto_r03c.txt
python bench.py --no-cpu-baseline > gpurun_out/bench_6.json 2> gpurun_out/bench_6.err; echo "bench rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_6.json').read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"])
for k,v in j["configs"].items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(str,dict))})
PY
