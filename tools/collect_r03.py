"""Turn the raw rocprofv3 output of tools/profile.sh (gpurun_out/prof/<pass>/) into the round-3 summaries under profiles/:
  r03_slab_pmc.json + r03_slab_kernel_stats.csv    pass `slab`  (k_slab<0,true> = step_slab(RANDOM), k_slab<4,true> = fused policy step)
  r03_auto_pmc.json + r03_config4_*                pass `auto`  (k_auto2 in the config-4 loop)
  r03_bench.json + r03_bench_kernel_stats.csv      pass `bench`
  r03_config3_kernel_stats.csv + r03_config3.txt   pass `dqn`
  r03_dpp_probe.txt / r03_stamps.txt               passes `probe` / `stamps`
Counters are per-dispatch sums over the whole chip as rocprofv3 reports them; per-table-step figures divide the mean over
the middle 80 % of the dispatches by the table count."""
import collections
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "prof")
P = os.path.join(ROOT, "profiles")


def counters(d, match):
    acc = collections.defaultdict(list)
    path = os.path.join(G, d, "p_counter_collection.csv")
    if not os.path.exists(path):
        return {}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if match in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for v in per.values():
        for k, x in v.items():
            acc[k].append(x)
    out = {}
    for k, x in acc.items():
        x = sorted(x)
        x = x[len(x) // 10: len(x) - len(x) // 10] or x
        out[k] = sum(x) / len(x)
    return out


def stats_rows(d, match):
    path = os.path.join(G, d, "p_kernel_stats.csv")
    return [r for r in csv.DictReader(open(path)) if match in r["Name"]] if os.path.exists(path) else []


def copy(src, dst):
    if os.path.exists(os.path.join(G, src)):
        shutil.copyfile(os.path.join(G, src), os.path.join(P, dst))


# ---- slab
slab = {"method": "tools/profile.sh slab: rocprofv3 --kernel-trace --stats, then two --pmc passes (instructions / waits; LDS / issue) "
                  "over tools/slab_modes_probe.py (BatchedEnv with ids, as bench.py's legs; 200 warm-up rollout iterations, then "
                  "the launches counted)", "kernels": {}}
rows_csv = []
for T in (65536, 4096):
    for mode, kern in (("random", "k_slab<0, true>"), ("fused", "k_slab<4, true>")):
        c = {}
        c.update(counters(f"slab/{mode}_{T}_p1", "k_slab"))
        c.update(counters(f"slab/{mode}_{T}_p2", "k_slab"))
        st = stats_rows(f"slab/{mode}_{T}_stats", "k_slab")
        if not c and not st:
            continue
        e = {"tables": T, "mode": mode, "kernel": kern}
        if st:
            e["launches"] = int(st[0]["Calls"]); e["avg_us"] = float(st[0]["AverageNs"]) / 1e3
            rows_csv.append({"tables": T, "mode": mode, **st[0]})
        for k, v in c.items():
            e[k + "_per_table_step"] = v / T
        if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
            e["wait_any_share_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        if st and "SQ_INSTS_VALU" in c:
            e["valu_G_wave_instr_per_s"] = c["SQ_INSTS_VALU"] / (float(st[0]["AverageNs"]) * 1e-9) / 1e9
            e["salu_branch_G_per_s"] = (c.get("SQ_INSTS_SALU", 0) + c.get("SQ_INSTS_BRANCH", 0)) / (float(st[0]["AverageNs"]) * 1e-9) / 1e9
        slab["kernels"][f"{mode}_{T}"] = e
if slab["kernels"]:
    json.dump(slab, open(os.path.join(P, "r03_slab_pmc.json"), "w"), indent=1)
    with open(os.path.join(P, "r03_slab_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows_csv[0].keys()))
        w.writeheader()
        w.writerows(rows_csv)

# ---- auto
c = {}
c.update(counters("auto/p1", "k_auto2"))
c.update(counters("auto/p2", "k_auto2"))
st = stats_rows("auto/stats", "k_auto2")
if c or st:
    T = 65536
    e = {"method": "tools/profile.sh auto: examples/config4_rule_opponent.py --tables 65536 --iters 40 (farmers = rule agent, lord = "
                   "engine RNG); per-decision figures divide by the ~2/3 of the tables whose actor is a farmer",
         "tables": T}
    if st:
        e["launches"] = int(st[0]["Calls"]); e["avg_us"] = float(st[0]["AverageNs"]) / 1e3
    for k, v in c.items():
        e[k + "_per_launch"] = v
        e[k + "_per_decision"] = v / (T * 2 / 3)
    if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
        e["wait_any_share_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    json.dump(e, open(os.path.join(P, "r03_auto_pmc.json"), "w"), indent=1)
copy("auto/stats/p_kernel_stats.csv", "r03_config4_kernel_stats.csv")
for f in ("config4_random_65536.txt", "config4_net_65536.txt", "config4_random_4096.txt"):
    copy("auto/" + f, "r03_" + f)
# ---- rollout: HBM counters + instruction counters of k_rollout -> pmc_traffic.json (what bench.py reads)
def rollout_counters(d):
    path = os.path.join(G, d, "p_counter_collection.csv")
    if not os.path.exists(path):
        return None, None
    disp = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if "k_rollout" in r["Kernel_Name"]:
            disp[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    big = max(disp, key=lambda k: max(disp[k].values()))  # the long launch
    trace = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
             for r in csv.DictReader(open(os.path.join(G, d, "p_kernel_trace.csv")))}
    return dict(disp[big]), trace[big]


fetch, _ = rollout_counters("rollout/pmc_FETCH_SIZE")
write, _ = rollout_counters("rollout/pmc_WRITE_SIZE")
mix, ns = rollout_counters("rollout/pmc_mix")
if fetch and write and mix:
    old = json.load(open(os.path.join(P, "pmc_traffic.json")))
    steps, steps_mix = 4096 * 2000, 4096 * 20000
    fk, wk = fetch["FETCH_SIZE"], write["WRITE_SIZE"]
    out = {"k_rollout": {
        "hbm_bytes_per_env_step": (2 * fk + wk) * 1024 / steps,
        "fetch_bytes_per_env_step": 2 * fk * 1024 / steps, "write_bytes_per_env_step": wk * 1024 / steps,
        "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "env_steps": steps,
        "workload": "tools/run_rollout.py 4096 2000 (4096 tables, 2000 in-launch iterations, seed 0), round-3 kernel",
        "method": "tools/profile.sh rollout: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; "
                  "bytes = KB*1024, FETCH doubled (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); "
                  "WRITE_SIZE taken as is (16-B-per-lane stores).  Every iteration overwrites the same state rows / list slab "
                  "of its table, so the write-back L2 merges them: HBM sees far fewer bytes than the kernel stores",
        "valu": {
            "SQ_INSTS_VALU_per_env_step": mix["SQ_INSTS_VALU"] / steps_mix,
            "SQ_INSTS_SALU_per_env_step": mix["SQ_INSTS_SALU"] / steps_mix,
            "SQ_INSTS_BRANCH_per_env_step": mix["SQ_INSTS_BRANCH"] / steps_mix,
            "SQ_INSTS_LDS_per_env_step": mix["SQ_INSTS_LDS"] / steps_mix,
            "SQ_WAIT_ANY_share_of_wave_cycles": mix["SQ_WAIT_ANY"] / mix["SQ_WAVE_CYCLES"],
            "GRBM_GUI_ACTIVE": mix["GRBM_GUI_ACTIVE"], "xcds": 8, "launch_ns": ns,
            "clock_GHz": mix["GRBM_GUI_ACTIVE"] / 8 / ns,
            "env_steps_per_s_in_this_launch": steps_mix / (ns * 1e-9),
            "workload": "tools/run_rollout.py 4096 20000 (one launch, 81.92 M env steps), round-3 kernel"},
        "valu_mix": old["k_rollout"].get("valu_mix")}}
    json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    v = out["k_rollout"]["valu"]
    print("k_rollout per step: VALU %.1f SALU %.1f branch %.1f LDS %.1f; clock %.3f GHz; %.4g steps/s; HBM %.2f B/step; waiting %.0f %%" % (
        v["SQ_INSTS_VALU_per_env_step"], v["SQ_INSTS_SALU_per_env_step"], v["SQ_INSTS_BRANCH_per_env_step"],
        v["SQ_INSTS_LDS_per_env_step"], v["clock_GHz"], v["env_steps_per_s_in_this_launch"],
        out["k_rollout"]["hbm_bytes_per_env_step"], 100 * v["SQ_WAIT_ANY_share_of_wave_cycles"]))

# ---- bench / dqn / probe / stamps
copy("bench/bench.json", "r03_bench.json")
copy("bench/bench_stats/p_kernel_stats.csv", "r03_bench_kernel_stats.csv")
copy("dqn/stats/p_kernel_stats.csv", "r03_config3_kernel_stats.csv")
copy("dqn/config3.txt", "r03_config3.txt")
copy("probe/dpp_probe.txt", "r03_dpp_probe.txt")
copy("probe/valu_issue_probe.txt", "r03_valu_issue_probe.txt")
for a_, b_ in (("stamps/stamp_slab.txt", "r03_stamps_slab.txt"), ("stamps/stamp_auto.txt", "r03_stamps_auto.txt")):
    copy(a_, b_)
print("profiles/ updated:", sorted(f for f in os.listdir(P) if f.startswith("r03")))
