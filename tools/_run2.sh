set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_2.log 2>&1; echo "tests rc=$?" ; tail -15 gpurun_out/gpu_tests_2.log
python bench.py > gpurun_out/bench_2.json 2> gpurun_out/bench_2.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/bench_2.err; python - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_2.json').read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"])
for k,v in j["configs"].items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(str,dict))})
print(j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"], j["cpu_baseline"]["threads_sweep"], j["cpu_baseline"]["cgroup_cpu_quota"])
PY
python examples/config3_dqn_inference.py --iters 10 > gpurun_out/cfg3.log 2>&1; tail -2 gpurun_out/cfg3.log
