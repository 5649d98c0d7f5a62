"""Turn the raw outputs of tools/profile_round.sh (gpurun_out/<tag>_*) into the committed summaries:
profiles/<tag>_bench.json, <tag>_bench_kernel_stats.csv, <tag>_pmc_mix_counter_collection.csv and
profiles/pmc_traffic.json (what bench.py reads for roofline.traffic / roofline.issue).
  python tools/collect_profile.py r01f"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def counters(d):
    rows = list(csv.DictReader(open(os.path.join(G, d, "p_counter_collection.csv"))))
    disp = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        if "k_rollout" in r["Kernel_Name"]:
            disp[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    big = max(disp, key=lambda k: max(disp[k].values()))  # the long launch
    trace = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
             for r in csv.DictReader(open(os.path.join(G, d, "p_kernel_trace.csv")))}
    return dict(disp[big]), trace[big]


fetch, _ = counters(f"{tag}_pmc_FETCH_SIZE")
write, _ = counters(f"{tag}_pmc_WRITE_SIZE")
mix, ns = counters(f"{tag}_pmc_mix")
steps, steps_mix = 4096 * 2000, 4096 * 20000
fk, wk = fetch["FETCH_SIZE"], write["WRITE_SIZE"]
out = {"k_rollout": {
    "hbm_bytes_per_env_step": (2 * fk + wk) * 1024 / steps,
    "fetch_bytes_per_env_step": 2 * fk * 1024 / steps, "write_bytes_per_env_step": wk * 1024 / steps,
    "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "env_steps": steps,
    "workload": f"tools/run_rollout.py 4096 2000 (4096 tables, 2000 in-launch iterations, seed 0), {tag} kernel",
    "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
              "(tools/profile_round.sh); bytes = KB*1024, FETCH doubled (gfx950 counts 128-B requests at 64 B, "
              "MI355X_MICROARCH.md HBM section); WRITE_SIZE taken as is (16-B-per-lane stores). Every iteration "
              "overwrites the same state rows / list slab of its table, so the write-back L2 merges them: HBM sees "
              "far fewer bytes than the kernel stores",
    "valu": {
        "SQ_INSTS_VALU_per_env_step": mix["SQ_INSTS_VALU"] / steps_mix,
        "SQ_INSTS_SALU_per_env_step": mix["SQ_INSTS_SALU"] / steps_mix,
        "SQ_INSTS_BRANCH_per_env_step": mix["SQ_INSTS_BRANCH"] / steps_mix,
        "SQ_INSTS_LDS_per_env_step": mix["SQ_INSTS_LDS"] / steps_mix,
        "SQ_ACTIVE_INST_VALU_quads_per_env_step": mix["SQ_ACTIVE_INST_VALU"] / steps_mix,
        "GRBM_GUI_ACTIVE": mix["GRBM_GUI_ACTIVE"], "xcds": 8, "launch_ns": ns,
        "clock_GHz": mix["GRBM_GUI_ACTIVE"] / 8 / ns,
        "env_steps_per_s_in_this_launch": steps_mix / (ns * 1e-9),
        "workload": f"tools/run_rollout.py 4096 20000 (one launch, 81.92 M env steps), {tag} kernel; rocprofv3 "
                    "--kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH "
                    "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES",
        "note": "a wave64 VALU instruction occupies its SIMD16 for 4 cycles: peak = 256 CUs x 4 SIMDs x clock / 4 "
                "wave-instructions/s"}}}
json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(G, f"{tag}_bench.json"), os.path.join(P, f"{tag}_bench.json"))
shutil.copy(os.path.join(G, f"{tag}_stats", f"{tag}_kernel_stats.csv"), os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(G, f"{tag}_pmc_mix", "p_counter_collection.csv"),
            os.path.join(P, f"{tag}_pmc_mix_counter_collection.csv"))
b = json.load(open(os.path.join(P, f"{tag}_bench.json")))
v = out["k_rollout"]["valu"]
print("bench value %.4g steps/s, %.3f us/iter, hbm frac %.3f" % (b["value"], b["roofline"]["us_per_iteration"], b["roofline"]["frac"]))
print("per step: VALU %.1f SALU %.1f branch %.1f LDS %.1f; clock %.3f GHz; %.4g steps/s in the PMC launch; HBM %.2f B/step" % (
    v["SQ_INSTS_VALU_per_env_step"], v["SQ_INSTS_SALU_per_env_step"], v["SQ_INSTS_BRANCH_per_env_step"],
    v["SQ_INSTS_LDS_per_env_step"], v["clock_GHz"], v["env_steps_per_s_in_this_launch"], out["k_rollout"]["hbm_bytes_per_env_step"]))
peak = 256 * 4 * v["clock_GHz"] / 4
print("VALU issue: %.1f of %.1f G wave-instr/s = %.3f" % (v["SQ_INSTS_VALU_per_env_step"] * v["env_steps_per_s_in_this_launch"] / 1e9, peak,
      v["SQ_INSTS_VALU_per_env_step"] * v["env_steps_per_s_in_this_launch"] / 1e9 / peak))
