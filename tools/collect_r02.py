"""Turn the raw rocprofv3 outputs of tools/profile_r02a.sh / profile_r02c.sh (gpurun_out/r02a, r02c) into the committed
round-2 summaries under profiles/:
  r02_secondary_kernels.json   k_observe<v> / k_mask / k_moves: kernel-trace durations, FETCH_SIZE / WRITE_SIZE per launch
  r02_slab_kernel_stats.csv    rocprofv3 --stats of the slab API loop (k_slab) at 65,536 and 4096 tables
  r02_slab_pmc.json            instruction counters per table-step of k_slab (and k_rollout beside it)
  pmc_traffic.json             + valu_mix of k_rollout (static opcode classes of the code object, /tmp/st/mix.json)
"""
import collections
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def trace(d):
    return list(csv.DictReader(open(os.path.join(G, d, "p_kernel_trace.csv"))))


def counters(d):
    disp = collections.defaultdict(lambda: collections.defaultdict(float))
    names = {}
    for r in csv.DictReader(open(os.path.join(G, d, "p_counter_collection.csv"))):
        disp[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["Grid_Size"]))
    return disp, names


def short(n):
    for k in ("k_observe<0>", "k_observe<1>", "k_observe<2>", "k_observe<3>", "k_mask", "k_moves<false, false>",
              "k_moves<true, false>", "k_moves<true, true>", "k_slab", "k_rollout", "k_auto2", "k_auto", "k_select"):
        if k in n:
            return k
    return None


# ---- secondary kernels
sec = {"method": "tools/profile_r02a.sh: rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate "
                 "passes over tools/observe_probe.py / tools/get_moves_probe.py (program directly after --); bytes = KB * 1024, "
                 "FETCH doubled (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section)",
       "kernels": {}}
PLANES = {"k_observe<0>": 4, "k_observe<1>": 7, "k_observe<2>": 9, "k_observe<3>": 6}
dur = collections.defaultdict(list)
for r in trace("r02a/observe_stats"):
    k = short(r["Kernel_Name"])
    if k and (k.startswith("k_observe") or k == "k_mask"):
        dur[(k, int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
pmc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    disp, names = counters(f"r02a/observe_pmc_{c}")
    acc = collections.defaultdict(list)
    for d, v in disp.items():
        k = short(names[d][0])
        if k:
            acc[k].append(v[c])
    pmc[c] = {k: sum(v) / len(v) for k, v in acc.items()}
for (k, grid), v in sorted(dur.items()):
    T = grid // 64 if k == "k_mask" else None
    P_ = PLANES.get(k)
    if P_:
        T = grid // (P_ * 15) if grid % (P_ * 15) == 0 else None
    v = sorted(v)[: max(1, len(v) - 3)]  # drop the warm-up launches (the slowest)
    avg = sum(v) / len(v)
    e = {"grid": grid, "launches": len(v), "avg_ns": avg}
    if k.startswith("k_observe"):
        T = round(grid / (P_ * 15))
        byts = T * (240 * P_ + 176)
        e.update({"tables": T, "algorithmic_bytes": byts, "algorithmic_GBps": byts / avg})
    else:
        T = grid // 64 * 1  # one wave per table x tpw; bytes from the table count of the probe
    if grid >= 524288 * 4 * 15 or (k == "k_mask" and avg > 150000):
        if k in pmc["WRITE_SIZE"]:
            e["hbm_write_bytes_per_launch"] = pmc["WRITE_SIZE"][k] * 1024
            e["hbm_fetch_bytes_per_launch"] = 2 * pmc["FETCH_SIZE"].get(k, 0) * 1024
    sec["kernels"].setdefault(k, []).append(e)
# k_mask / k_moves by duration groups of the probe's sizes
mk = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace("r02a/observe_stats") if "k_mask" in r["Kernel_Name"])
big = [x for x in mk if x > 150000]
small = [x for x in mk if x <= 150000]
sec["kernels"]["k_mask"] = [
    {"tables": 65536, "launches": len(small), "avg_ns": sum(small) / len(small), "algorithmic_bytes": 65536 * (1696 + 176),
     "algorithmic_GBps": 65536 * (1696 + 176) / (sum(small) / len(small))},
    {"tables": 524288, "launches": len(big), "avg_ns": sum(big) / len(big), "algorithmic_bytes": 524288 * (1696 + 176),
     "algorithmic_GBps": 524288 * (1696 + 176) / (sum(big) / len(big)),
     "hbm_write_bytes_per_launch": pmc["WRITE_SIZE"].get("k_mask", 0) * 1024,
     "hbm_fetch_bytes_per_launch": 2 * pmc["FETCH_SIZE"].get("k_mask", 0) * 1024}]
mv = collections.defaultdict(list)
for r in trace("r02a/moves_stats"):
    k = short(r["Kernel_Name"])
    if k and k.startswith("k_moves"):
        mv[(k, int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
sec["kernels"]["k_moves"] = [{"kernel": k, "grid": g, "launches": len(v), "avg_ns": sum(v) / len(v)} for (k, g), v in sorted(mv.items())]
sec["kernels"]["k_moves_note"] = ("ddz_get_moves = k_moves<false,false> (sizes + block scan) + k_moves<true,*> (CSR write) per call; "
                                  "probe sizes 4096 / 65,536 / 524,288 queries, 6.3 moves per query (tools/get_moves_probe.py)")
json.dump(sec, open(os.path.join(P, "r02_secondary_kernels.json"), "w"), indent=1)

# ---- slab kernel
with open(os.path.join(P, "r02_slab_kernel_stats.csv"), "w") as f:
    for d, label in (("r02c/slab_stats", "tools/slab_loop.py 65536 200"), ("r02c/slab_stats_4096", "tools/slab_loop.py 4096 400")):
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 {label}\n")
        f.writelines(open(os.path.join(G, d, "p_kernel_stats.csv")).readlines()[:4])
out = {}
for w, kern, steps in (("slab", "k_slab", 65536), ("rollout", "k_rollout", 4096 * 20000)):
    e = {}
    for p_ in ("p1", "p2"):
        disp, names = counters(f"r02c/{w}_{p_}")
        tr = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace(f"r02c/{w}_{p_}")}
        ks = [d for d in disp if kern in names[d][0]]
        if w == "slab":
            ks = sorted(ks, key=int)[20:]
        else:
            ks = [max(ks, key=lambda d: tr[d])]
        acc = collections.defaultdict(float)
        for d in ks:
            for c, v in disp[d].items():
                acc[c] += v
        ns = sum(tr[d] for d in ks)
        e.update({f"{c}_per_table_step": v / (steps * len(ks)) for c, v in acc.items() if c != "GRBM_GUI_ACTIVE"})
        e["avg_launch_ns"] = ns / len(ks)
    out[kern] = e
out["note"] = ("tools/profile_r02c.sh; k_slab: step_slab(CHOICE) at 65,536 tables (180 launches after warm-up); k_rollout: one "
               "20,000-iteration launch at 4096 tables.  SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_ACTIVE_* count quad-cycles.  k_table<F_STEP|F_SLAB> "
               "(round 1) was 208 VALU + 315 SALU + 47 branch per table-step")
json.dump(out, open(os.path.join(P, "r02_slab_pmc.json"), "w"), indent=1)

# ---- valu mix into pmc_traffic.json
pt = json.load(open(os.path.join(P, "pmc_traffic.json")))
mix = json.load(open("/tmp/st/mix.json"))["k_rollout<false,false>"]
pt["k_rollout"]["valu_mix"] = {
    "share_half_rate": round(mix["half"], 3), "share_readlane": round(mix["readlane"], 3), "share_full_rate": round(mix["full"], 3),
    "source": "static opcode histogram of the k_rollout<false,false> code object (1,310 VALU instructions): half-rate class = 64-bit "
              "shifts, v_mul_lo/hi, v_mbcnt, v_cmp*, carry chains (v_add_co/v_addc_co/...), v_bfe/v_alignbit; readlane class = "
              "v_readlane/v_readfirstlane; the hardware has no per-class dynamic counter (SQ_INSTS_VALU_INT32 / _INT64: 36.4 / 6.6 of "
              "113.3 per env step, profiles/r02_slab_pmc.json)"}
json.dump(pt, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(sec["kernels"], indent=1)[:3000])
