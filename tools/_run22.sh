set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_22.log 2>&1; echo "tests rc=$?" ; tail -3 gpurun_out/gpu_tests_22.log
python __graft_entry__.py smoke 2>&1 | tail -1
bash tools/profile.sh bench dqn auto > gpurun_out/prof_22.log 2>&1; echo "profile rc=$?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/prof/bench/bench.json').read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"], j["roofline"]["issue"]["frac"], j["roofline"]["traffic"])
for k,v in j["configs"].items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(str,dict))})
c=j["cpu_baseline"]; print(c["value"], c["cores"], c["single_core_value"])
PY
tail -1 gpurun_out/prof/dqn/config3.txt; cat gpurun_out/prof/auto/config4_*.txt | grep tables
