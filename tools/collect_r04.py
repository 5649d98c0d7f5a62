"""Turn the raw rocprofv3 output of tools/profile.sh (gpurun_out/prof/<pass>/) into the round-4 summaries under profiles/
(+ profiles/pmc_kernels.json, the per-kernel instruction counts bench.py replays in its issue-roofline blocks, and
r04_dqn_pmc.json: MFMA busy cycles / clock / HBM bytes of every kernel of the configs[2] loop):
  r04_slab_pmc.json + r04_slab_kernel_stats.csv    pass `slab`  (k_slab<0,true> = step_slab(RANDOM), k_slab<4,true> = fused policy step)
  r04_auto_pmc.json + r04_config4_*                pass `auto`  (k_auto2 in the config-4 loop)
  r04_bench.json + r04_bench_kernel_stats.csv      pass `bench`
  r04_config3_kernel_stats.csv + r04_config3.txt   pass `dqn`
  r04_dpp_probe.txt / r04_stamps.txt               passes `probe` / `stamps`
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
        e = {"tables": T, "mode": mode, "kernel": kern[:-1] + (", true>" if T <= 4096 else ", false>")}   # <MODE, IDS, COOP>
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
    json.dump(slab, open(os.path.join(P, "r04_slab_pmc.json"), "w"), indent=1)
    with open(os.path.join(P, "r04_slab_kernel_stats.csv"), "w", newline="") as f:
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
    json.dump(e, open(os.path.join(P, "r04_auto_pmc.json"), "w"), indent=1)
copy("auto/stats/p_kernel_stats.csv", "r04_config4_kernel_stats.csv")
for f in ("config4_random_65536.txt", "config4_net_65536.txt", "config4_random_4096.txt"):
    copy("auto/" + f, "r04_" + f)
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


def rollout_entry(prefix, T, it_hbm, it_mix, old_mix):
    fetch, _ = rollout_counters(f"rollout/{prefix}pmc_FETCH_SIZE")
    write, _ = rollout_counters(f"rollout/{prefix}pmc_WRITE_SIZE")
    mix, ns = rollout_counters(f"rollout/{prefix}pmc_mix")
    if not (fetch and write and mix):
        return None
    steps, steps_mix = T * it_hbm, T * it_mix
    fk, wk = fetch["FETCH_SIZE"], write["WRITE_SIZE"]
    return {
        "hbm_bytes_per_env_step": (2 * fk + wk) * 1024 / steps,
        "fetch_bytes_per_env_step": 2 * fk * 1024 / steps, "write_bytes_per_env_step": wk * 1024 / steps,
        "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "env_steps": steps,
        "workload": f"tools/run_rollout.py {T} {it_hbm} ({T} tables, {it_hbm} in-launch iterations, seed 0), round-4 kernel",
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
            "workload": f"tools/run_rollout.py {T} {it_mix} (one launch, {steps_mix / 1e6:.2f} M env steps), round-4 kernel"},
        "valu_mix": old_mix}


old = json.load(open(os.path.join(P, "pmc_traffic.json")))
fresh = {}
for key, prefix, T, ih, im in (("k_rollout", "", 4096, 2000, 20000), ("k_rollout_65536", "big_", 65536, 500, 2000)):
    e = rollout_entry(prefix, T, ih, im, old["k_rollout"].get("valu_mix"))
    if e:
        old[key] = e
        fresh[key] = e
        v = e["valu"]
        print("%s per step: VALU %.1f SALU %.1f branch %.1f LDS %.1f; clock %.3f GHz; %.4g steps/s; HBM %.2f B/step; waiting %.0f %%" % (
            key, v["SQ_INSTS_VALU_per_env_step"], v["SQ_INSTS_SALU_per_env_step"], v["SQ_INSTS_BRANCH_per_env_step"],
            v["SQ_INSTS_LDS_per_env_step"], v["clock_GHz"], v["env_steps_per_s_in_this_launch"],
            e["hbm_bytes_per_env_step"], 100 * v["SQ_WAIT_ANY_share_of_wave_cycles"]))
if fresh:
    json.dump(old, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)

# ---- dqn: per kernel of the configs[2] loop -- duration, MFMA pipe busy share, clock, HBM bytes
def per_kernel(d, names):
    path = os.path.join(G, d, "p_counter_collection.csv")
    if not os.path.exists(path):
        return {}
    trace = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
             for r in csv.DictReader(open(os.path.join(G, d, "p_kernel_trace.csv")))}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    kern = {}
    for r in csv.DictReader(open(path)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        kern[r["Dispatch_Id"]] = r["Kernel_Name"]
    out = {}
    for key, match in names.items():
        lo, hi = 0, float("inf")
        if isinstance(match, tuple):      # (substring, shortest, longest duration in ns): two uses of one library kernel
            match, lo, hi = match
        ids = [i for i, k in kern.items() if match in k and lo <= trace.get(i, 0) < hi]
        if not ids:
            continue
        ids = sorted(ids, key=lambda i: trace.get(i, 0))
        ids = ids[len(ids) // 10: len(ids) - len(ids) // 10] or ids
        e = {"launches": len(ids), "avg_us": sum(trace[i] for i in ids) / len(ids) / 1e3}
        for c in per[ids[0]]:
            e[c] = sum(per[i][c] for i in ids) / len(ids)
        out[key] = e
    return out


DQN_KERNELS = {"k_fc1<false> (dense, K = 3840)": "k_fc1<false>", "k_fc1<true> (needed rows)": "k_fc1<true>", "k_q_feat_needed<6>": "k_q_feat_needed",
               "k_q_slab_needed": "k_q_slab_needed", "k_q_need_mask": "k_q_need_mask", "k_q_need_scan": "k_q_need_scan",
               "k_q_need_assign": "k_q_need_assign", "k_slab<4,true>": "k_slab<4, true>",
               "dense GEMM H0 += y0 x Wd (hipBLASLt)": ("Cijk_", 400000, float("inf")), "table term (hipBLASLt)": ("Cijk_", 0, 400000)}
dq = per_kernel("dqn/p_mfma", DQN_KERNELS)
dq.update({k + " [--gemm mfma run]": v for k, v in per_kernel("dqn/p_mfma_k_fc1", {"k_fc1<false> (dense, K = 3840)": "k_fc1<false>"}).items()})
fe, wr = per_kernel("dqn/pmc_FETCH_SIZE", DQN_KERNELS), per_kernel("dqn/pmc_WRITE_SIZE", DQN_KERNELS)
if dq:
    for k, e in dq.items():
        ns = e["avg_us"] * 1e3
        if "GRBM_GUI_ACTIVE" in e:
            e["clock_GHz"] = e["GRBM_GUI_ACTIVE"] / 8 / ns
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
            # busy cycles summed over the chip's 1024 SIMD matrix pipes / (cycles of the launch x 1024)
            e["mfma_pipe_busy_share"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8 * 1024)
        if "SQ_WAIT_ANY" in e and "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]:
            e["wait_any_share_of_wave_cycles"] = e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]
        if k in fe and k in wr:
            e["hbm_fetch_MB"] = 2 * fe[k]["FETCH_SIZE"] * 1024 / 1e6       # FETCH doubled: gfx950 counts 128-B requests at 64 B
            e["hbm_write_MB"] = wr[k]["WRITE_SIZE"] * 1024 / 1e6
    json.dump({"method": "tools/profile.sh dqn: rocprofv3 --kernel-trace --pmc over examples/config3_dqn_inference.py (65,536 tables, "
                         "needed-rows form); middle 80 % of each kernel's dispatches; counters are sums over the chip; HBM bytes "
                         "from separate FETCH_SIZE / WRITE_SIZE passes (KB x 1024, FETCH doubled per the gfx950 note)",
               "kernels": dq}, open(os.path.join(P, "r04_dqn_pmc.json"), "w"), indent=1)
    for k, e in dq.items():
        print(f"{k:34s} {e['avg_us']:9.1f} us  clock {e.get('clock_GHz', 0):.2f} GHz  MFMA busy {e.get('mfma_pipe_busy_share', 0):.3f}  "
              f"wait {e.get('wait_any_share_of_wave_cycles', 0):.2f}  HBM {e.get('hbm_fetch_MB', 0):.0f} + {e.get('hbm_write_MB', 0):.0f} MB")

# ---- pmc_kernels.json: what bench.py's issue blocks replay (kept entries are refreshed, others stay)
pk_path = os.path.join(P, "pmc_kernels.json")
pk = json.load(open(pk_path)) if os.path.exists(pk_path) else {}
for key in fresh:   # (only when this run held a fresh `rollout` pass)
  t4 = fresh[key]["valu"]
  pk.setdefault(key, dict(pk["k_rollout"]))
  pk[key]["tables"] = 4096 if key == "k_rollout" else 65536
  pk[key]["mix_key"] = "k_rollout<false,false,16 waves>" if key == "k_rollout" else "k_rollout<false,false>"   # (4096 tables: 16-wave blocks)
  pk[key]["waves_per_simd"] = 4 if key == "k_rollout" else 6
  pk[key].update({"valu_per_unit": t4["SQ_INSTS_VALU_per_env_step"], "salu_per_unit": t4["SQ_INSTS_SALU_per_env_step"],
                        "branch_per_unit": t4["SQ_INSTS_BRANCH_per_env_step"], "wait_any_share": t4["SQ_WAIT_ANY_share_of_wave_cycles"],
                        "clock_GHz": t4["clock_GHz"], "source": "profiles/pmc_traffic.json (pass `rollout`: " + t4["workload"] + ")"})
if slab["kernels"]:
    for key, name in (("random_65536", "k_slab_random_65536"), ("random_4096", "k_slab_random_4096"), ("fused_65536", "k_slab_fused_65536")):
        k = slab["kernels"].get(key)
        if k and "SQ_INSTS_VALU_per_table_step" in k:
            pk[name].update({"valu_per_unit": k["SQ_INSTS_VALU_per_table_step"], "salu_per_unit": k["SQ_INSTS_SALU_per_table_step"],
                             "branch_per_unit": k["SQ_INSTS_BRANCH_per_table_step"], "wait_any_share": k["wait_any_share_of_wave_cycles"],
                             "source": "profiles/r04_slab_pmc.json (pass `slab`)"})
            if key == "random_4096":
                pk[name]["mix_key"] = "k_slab<0,true,coop>"   # one table per wave: k_slab<0, true, true>
ap = os.path.join(P, "r04_auto_pmc.json")
if os.path.exists(ap):
    a4 = json.load(open(ap))
    if "SQ_INSTS_VALU_per_launch" in a4:
        pk["k_auto2_65536"].update({"valu_per_unit": a4["SQ_INSTS_VALU_per_launch"], "salu_per_unit": a4["SQ_INSTS_SALU_per_launch"],
                                    "branch_per_unit": a4["SQ_INSTS_BRANCH_per_launch"], "wait_any_share": a4["wait_any_share_of_wave_cycles"],
                                    "clock_GHz": min(2.4, a4["GRBM_GUI_ACTIVE_per_launch"] / 8 / (a4["avg_us"] * 1e3)),
                                    "source": "profiles/r04_auto_pmc.json (pass `auto`)"})
json.dump(pk, open(pk_path, "w"), indent=1)

# ---- bench / dqn / probe / stamps
copy("bench/bench.json", "r04_bench.json")
copy("bench/bench_stats/p_kernel_stats.csv", "r04_bench_kernel_stats.csv")
copy("dqn/stats/p_kernel_stats.csv", "r04_config3_kernel_stats.csv")
copy("dqn/config3.txt", "r04_config3.txt")
copy("probe/dpp_probe.txt", "r04_dpp_probe.txt")
copy("probe/valu_issue_probe.txt", "r04_valu_issue_probe.txt")
for a_, b_ in (("stamps/stamp_slab.txt", "r04_stamps_slab.txt"), ("stamps/stamp_auto.txt", "r04_stamps_auto.txt")):
    copy(a_, b_)
print("profiles/ updated:", sorted(f for f in os.listdir(P) if f.startswith("r04")))
