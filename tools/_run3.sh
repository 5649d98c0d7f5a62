set -o pipefail
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp_probe tools/dpp_probe.hip 2>/dev/null && /tmp/dpp_probe > gpurun_out/dpp_probe.txt 2>&1; echo "dpp rc=$?"; cat gpurun_out/dpp_probe.txt
python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/gpu_tests_3.log 2>&1; echo "tests rc=$?" ; tail -25 gpurun_out/gpu_tests_3.log
python bench.py --no-cpu-baseline > gpurun_out/bench_3.json 2> gpurun_out/bench_3.err; echo "bench rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_3.json').read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"])
for k,v in j["configs"].items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(str,dict))})
print(json.dumps(j["configs"]["stress_plane_rich_leads"])[:900])
PY
python tools/stamp_auto.py 16384 > gpurun_out/stamp_auto_r03a.txt 2>&1; cat gpurun_out/stamp_auto_r03a.txt
python tools/stamp_slab.py 65536,4096 random > gpurun_out/stamp_slab_r03a.txt 2>&1; cat gpurun_out/stamp_slab_r03a.txt
