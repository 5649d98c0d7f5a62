#!/usr/bin/env python3
"""Headline benchmark: env steps/s of the batched lock-step engine (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
A "step" is one lock-step iteration of the hot path over all tables of this rank: legal-move enumeration into the
per-table list + random-policy action application with auto-reset (game.py:169-181 with envi.py:79-116).
  N = 1: BASELINE.json configs[1], 4096 tables on one MI355X, random policy, legal-move list only.
  N > 1: 65,536 tables per GPU (configs[4] = 524,288 tables at N = 8), weak scaling: every GPU owns a range of the
         global table ids and the timed region is the same at every N (tables are independent: no data-path
         collective).  The path's one exchange -- packed trajectories gathered to rank 0 over RCCL -- is measured
         right after the headline region (config.exchange, env_steps_per_s_with_gather).
The K-iteration launch is repeated until at least 50 ms have been timed, whatever K is (a single short launch
would measure launch latency); `value` and `ms_per_step` are means over all timed iterations.  Rank 0 prints ONE
JSON line; at N = 1 it also carries short measured legs of the other single-GPU configs (`configs`).
"""
import argparse
import importlib
import json
import math
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MIN_TIMED_S = 0.05      # every timed region covers at least this much GPU time


def cpu_baseline(tables, budget_s):
    """Oracle (CPU port of the same rules/env) on the host cores of this box: one thread, then the tables split over
    EVERY core this process may use (tables are independent), median of three samples."""
    from oracle import oracle
    affinity = max(1, len(os.sched_getaffinity(0)))
    env = oracle.OracleEnv(tables, seed=0)
    env.reset()
    t0 = time.perf_counter()
    env.rollout_random(3)
    per_iter = (time.perf_counter() - t0) / 3
    n1 = max(3, min(2000, int(0.25 * budget_s / max(per_iter, 1e-6))))
    t0 = time.perf_counter()
    plies1, _, _ = env.rollout_random(n1)
    dt1 = time.perf_counter() - t0
    # how many threads the box really gives this process: the affinity mask can be wider than the container's CPU
    # quota (a 1-GPU share of a 256-thread host), so the thread count is MEASURED -- a short sample at each candidate
    # count up to the affinity, the fastest one is used
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    cand = sorted({c for c in (1, 8, 16, 32, 64, 128, affinity, int(quota) if quota else affinity) if 1 <= c <= affinity})
    sweep = {}
    oracle.rollout_random_mt(env, 20, cand[-1])   # warm: thread start-up, the oracle's tables paged in on every core
    for c in cand[1:] or cand:
        n = max(3, int(0.05 * budget_s * min(c, 32) * (plies1 / dt1) / max(tables, 1)))   # ~0.6 s if it scales to 32
        t0 = time.perf_counter()
        p, _, _ = oracle.rollout_random_mt(env, n, c)
        sweep[c] = p / (time.perf_counter() - t0)
    cores = max(sweep, key=sweep.get)
    nm = max(3, min(20000, int(0.1 * budget_s * sweep[cores] / max(tables, 1))))
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        pliesm, _, _ = oracle.rollout_random_mt(env, nm, cores)
        rates.append(pliesm / (time.perf_counter() - t0))
    rates.sort()
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": rates[1], "unit": "env steps/s", "cores": cores, "cores_available": os.cpu_count(),
            "affinity": affinity, "cgroup_cpu_quota": quota, "threads_sweep": {str(k): v for k, v in sweep.items()},
            "kind": "port", "cpu_model": model, "single_core_value": plies1 / dt1, "samples": rates,
            "python_rules_floor": {"value": 36.0, "unit": "env steps/s", "cores": 1,
                                   "provenance": "SURVEY.md 6: the reference's own rules in Python (card.py action space + "
                                                 "get_mask, ~32 ms per mask) measured in the build container, one core; the "
                                                 "reference does not travel to the GPU box, so this is a recorded figure"},
            "sample": f"median of 3 x {nm} lock-step iterations x {tables} tables over {cores} threads (the fastest of "
                      f"{sorted(sweep)} threads tried; affinity {affinity}, {os.cpu_count()} logical CPUs on the box, "
                      f"cgroup quota {quota}), and {n1} iterations on 1 thread in {dt1:.1f} s; "
                      "oracle/ddz_oracle.c, dense 13,527-row scan per state"}


def plane_rich_hands(n, seed):
    """SURVEY 8(d) stress set: 20-card hands built around a run of 2-5 consecutive triples / bombs starting at 3..10, the
    rest of the 20 cards drawn uniformly from the remaining deck -- the construction of tests/test_gpu_parity.py
    _plane_rich_hands, vectorised.  int8 [n,16] count rows."""
    import numpy as np
    rng = np.random.default_rng(seed)
    start, run = rng.integers(0, 8, n), rng.integers(2, 6, n)
    h = np.zeros((n, 15), np.int64)
    r = np.arange(15)[None, :]
    inrun = (r >= start[:, None]) & (r < (start + run)[:, None])
    h[inrun] = rng.choice([3, 3, 3, 4], size=int(inrun.sum()))
    card_rank = np.concatenate([np.repeat(np.arange(13), 4), [13, 14]])          # 54 cards
    card_copy = np.concatenate([np.tile(np.arange(4), 13), [0, 0]])
    used = card_copy[None, :] < h[:, card_rank]                                   # the run's cards are taken
    key = rng.random((n, 54))
    key[used] = 2.0                                                               # never drawn again
    order = np.argsort(key, axis=1)
    left = 20 - h.sum(1)
    take = np.arange(54)[None, :] < left[:, None]
    rows = np.repeat(np.arange(n), 54).reshape(n, 54)
    np.add.at(h, (rows[take], card_rank[order][take]), 1)
    out = np.zeros((n, 16), np.int8)
    out[:, :15] = h
    assert (h.sum(1) == 20).all() and (h[:, :13] <= 4).all() and (h[:, 13:] <= 1).all()
    return out


def stress_leg(pkg, torch, dev):
    """SURVEY 8(d) 'stress set': all-lead queries with 20-card plane-rich hands (lists of several hundred moves: the
    compaction worst case) through the stateless r.get_moves entry points, slab (one launch) and packed CSR."""
    n = 65536
    hands = torch.from_numpy(plane_rich_hands(n, 12345)).to(dev)
    lasts = torch.zeros_like(hands)
    sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
    out = pkg.get_moves_slab(hands, lasts, want_ids=True)
    sync()
    counts = out[0]
    mean_a, max_a = float(counts.float().mean().item()), int(counts.max().item())
    dt, reps = timed_loop(lambda: pkg.get_moves_slab(hands, lasts, want_ids=True, out=out), sync)
    us_slab = dt / reps * 1e6
    out2 = pkg.get_moves_slab(hands, lasts, want_ids=False)
    dt, reps = timed_loop(lambda: pkg.get_moves_slab(hands, lasts, want_ids=False, out=out2), sync)
    us_slab_noids = dt / reps * 1e6
    cap = int(counts.sum().item())
    dt, reps = timed_loop(lambda: pkg.get_moves(hands, lasts, want_ids=True, row_capacity=cap), sync)
    us_csr = dt / reps * 1e6
    assert int(out[3].item()) == 0
    b_q = 32 + 4 + 16 * mean_a               # hand + last in, list size, 16-byte rows out (SURVEY 8d per-unit bytes)
    b_q_ids = b_q + 4 * mean_a
    return {"queries": n, "mean_legal_moves": mean_a, "max_legal_moves": max_a,
            "get_moves_slab": {"us_per_call": us_slab, "queries_per_s": n / us_slab * 1e6, "rows_per_s": n * mean_a / us_slab * 1e6,
                               "algorithmic_GBps": n * b_q_ids / us_slab / 1e3, "hbm_frac": n * b_q_ids / us_slab / 1e3 / HBM_PEAK_GBPS},
            "get_moves_slab_no_ids": {"us_per_call": us_slab_noids, "queries_per_s": n / us_slab_noids * 1e6,
                                      "algorithmic_GBps": n * b_q / us_slab_noids / 1e3,
                                      "hbm_frac": n * b_q / us_slab_noids / 1e3 / HBM_PEAK_GBPS},
            "get_moves_csr": {"us_per_call": us_csr, "queries_per_s": n / us_csr * 1e6, "rows_per_s": n * mean_a / us_csr * 1e6,
                              "algorithmic_GBps": n * b_q_ids / us_csr / 1e3, "hbm_frac": n * b_q_ids / us_csr / 1e3 / HBM_PEAK_GBPS,
                              "note": "two passes (sizes + scan, then write) and one host sync to trim the result"},
            "workload": "65,536 lead queries, 20-card hands around a run of 2-5 consecutive triples / bombs (plane_rich_hands, "
                        "seed 12345); bytes per query = 36 + 16 A (+ 4 A ids)"}


def timed_loop(fn, sync, min_s=MIN_TIMED_S, max_reps=1 << 16):
    """Call fn() repeatedly between two syncs until >= min_s have been timed; returns (seconds, calls)."""
    fn()
    sync()
    t0 = time.perf_counter()
    fn()
    sync()
    est = max(time.perf_counter() - t0, 1e-6)
    reps = int(min(max_reps, max(1, math.ceil(min_s / est))))
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return time.perf_counter() - t0, reps


def other_config_legs(pkg, torch, dev):
    """Short measured legs of the other single-GPU configurations of BASELINE.json (numbers, not prose)."""
    out = {}
    T = 65536
    sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
    env = pkg.BatchedEnv(T, seed=0, device=dev)
    env.reset()
    env.rollout_random(200)
    # (1) the random-policy rollout of the headline at 65,536 tables
    k = 500
    dt, reps = timed_loop(lambda: env.rollout_random(k), sync)
    out["tables_65536_random_rollout"] = {"env_steps_per_s": T * k * reps / dt, "iterations": k * reps,
                                          "us_per_iteration": dt / (k * reps) * 1e6}
    # (1b) configs[4]'s per-rank half of the trajectory exchange (what every rank does before the gather over xGMI):
    # the same rollout writing a 32-byte record per ply and table, then ddz_pack_trajectory to 8-byte records
    kx = 100
    traj = torch.zeros((kx, T, pkg.TRAJ_BYTES), dtype=torch.uint8, device=dev)
    packed = [None]

    def traj_iter():
        env.rollout_random(kx, traj=traj)
        packed[0] = pkg.pack_trajectory(traj)

    dt, reps = timed_loop(traj_iter, sync)
    dtp, repp = timed_loop(lambda: pkg.pack_trajectory(traj), sync)
    out["tables_65536_traj_write_pack"] = {
        "env_steps_per_s": T * kx * reps / dt, "us_per_iteration": dt / (kx * reps) * 1e6, "iterations": kx * reps,
        "pack_us_per_iteration": dtp / (kx * repp) * 1e6, "packed_bytes_per_iteration": T * pkg.TRAJ_PACKED_BYTES,
        "packed_GBps_to_send": T * pkg.TRAJ_PACKED_BYTES * kx * reps / dt / 1e9,
        "loop": "rollout_random(traj = 32-byte records) + pack_trajectory (8-byte records): the per-rank cost of "
                "config 5 before the RCCL gather"}
    del traj, packed
    # (2) configs[2]'s environment side: the loop a policy drives through the slab API -- face, selection over
    # per-action values, apply + next lists; one launch each (game.py:95-104, dqn.py:50-71).  The Q values are random
    # numbers standing in for the network's output (the network itself is out of scope: SURVEY 2 #8).
    env.legal_slab()
    face = torch.empty((T, 6, 15, 4), dtype=torch.float32, device=dev)
    q = torch.rand((T, env.slab_stride), dtype=torch.float32, device=dev)
    choice = torch.empty(T, dtype=torch.int32, device=dev)

    def policy_iter():
        env.observe(3, out=face)
        env.select_slab(q, out=choice)
        env.step_slab(choice, pkg.STEP_CHOICE, auto_reset=True)

    dt, reps = timed_loop(lambda: [policy_iter() for _ in range(20)], sync)
    out["tables_65536_policy_loop_slab"] = {"env_steps_per_s": T * 20 * reps / dt, "iterations": 20 * reps,
                                            "us_per_iteration": dt / (20 * reps) * 1e6,
                                            "loop": "observe(EnvCooperationSimplify) + select_slab(random q) + step_slab(CHOICE)"}
    # the same iteration in ONE launch: arg-max over q, apply, new lists and the new `face` (ddz_policy_step_slab)
    dt, reps = timed_loop(lambda: [env.policy_step_slab(q, 0.0, face_variant=3, face_out=face) for _ in range(20)], sync)
    out["tables_65536_policy_loop_fused"] = {"env_steps_per_s": T * 20 * reps / dt, "iterations": 20 * reps,
                                             "us_per_iteration": dt / (20 * reps) * 1e6,
                                             "loop": "policy_step_slab(random q, face = EnvCooperationSimplify): one launch"}
    # configs[2] as SURVEY 8(d) defines it: EnvCooperationSimplify planes + NetCooperationSimplify (net.py:137-150)
    # randomly initialised (torch.manual_seed(0)), eval mode, greedy arg-max per table over its legal list -- the
    # network IN the loop (dqn_glue.PolicyLoop: per-rank GEMMs over the rows the actors' hands allow -> ddz_q_slab_packed ->
    # ddz_policy_step_slab)
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    torch.manual_seed(0)
    net = glue.QNet(6).to(dev).eval()
    loop = glue.PolicyLoop(env, net, face_variant=3, epsilon=0.0)
    loop.run(3)
    s0 = env.stats()
    dt, reps = timed_loop(lambda: loop.run(5), sync, min_s=0.3, max_reps=64)
    s1 = env.stats()
    dtn, repn = timed_loop(lambda: [loop.q_values() for _ in range(5)], sync, min_s=0.15, max_reps=64)
    dtt, rept = timed_loop(lambda: [loop.fq.tables_packed(loop.face, env.actor_hands()) for _ in range(5)], sync,
                           min_s=0.15, max_reps=64)
    dte, repe = timed_loop(lambda: [env.policy_step_slab(loop.q, 0.0, face_variant=3, face_out=loop.face) for _ in range(5)],
                           sync, min_s=0.05, max_reps=64)
    rows_eval = s1["legal_rows"] - s0["legal_rows"]
    out["tables_65536_dqn_inference"] = {
        "env_steps_per_s": T * 5 * reps / dt, "us_per_iteration": dt / (5 * reps) * 1e6, "iterations": 5 * reps,
        "us_net": dtn / (5 * repn) * 1e6, "us_net_tables_gemms": dtt / (5 * rept) * 1e6,
        "us_net_rows_q_slab": (dtn / (5 * repn) - dtt / (5 * rept)) * 1e6, "us_env": dte / (5 * repe) * 1e6,
        "q_evals_per_s": rows_eval / dt, "mean_legal_moves": rows_eval / max(1, s1["plies"] - s0["plies"]),
        "dtype_net": "f32", "net": "NetCooperationSimplify-shaped QNet(6 + 1 planes), torch.manual_seed(0), eval()",
        "loop": "FactorisedQ.tables_packed(face, actors' hands) [ddz_q_features_packed + one torch GEMM per rank over the "
                "(rank, count, table) rows a legal move can use] -> ddz_q_slab_packed -> ddz_policy_step_slab(greedy, face "
                "= EnvCooperationSimplify): every legal action of every table gets its Q value each iteration; one "
                "128-byte device -> host copy per iteration (the GEMM shapes)"}
    hc = env.actor_hands().clamp(max=4)
    hc[:, 13:].clamp_(max=1)
    rows_needed = 15 * T + int(hc.sum())
    out["tables_65536_dqn_inference"]["packed_rows_per_table"] = rows_needed / T
    loop.fq.batched_gemm = True    # one batched fc1 GEMM over segments padded to the longest instead of fifteen exact ones
    out["tables_65536_dqn_inference"]["batched_gemm_padding_share"] = 1.0 - rows_needed / loop.fq.pack(env.actor_hands())[1][15]
    loop.run(2)
    dt, reps = timed_loop(lambda: loop.run(5), sync, min_s=0.2, max_reps=64)
    out["tables_65536_dqn_inference"]["batched_gemm_env_steps_per_s"] = T * 5 * reps / dt
    del loop
    # the same with fixed shapes (all 69 (rank, count) rows of every table, nothing on the host)
    loop = glue.PolicyLoop(env, net, face_variant=3, epsilon=0.0, packed=False)
    loop.run(2)
    dt, reps = timed_loop(lambda: loop.run(5), sync, min_s=0.3, max_reps=64)
    out["tables_65536_dqn_inference"]["fixed_shapes_env_steps_per_s"] = T * 5 * reps / dt
    del loop, net
    # the stepping launch alone (uniformly random legal moves drawn in the kernel: every selection is in its list), and
    # the same with the new lists packed to CSR every iteration (what a ragged NN forward over all legal moves consumes)
    dt, reps = timed_loop(lambda: [env.step_slab(None, pkg.STEP_RANDOM, auto_reset=True) for _ in range(20)], sync)
    out["tables_65536_step_slab_only"] = {"env_steps_per_s": T * 20 * reps / dt, "us_per_iteration": dt / (20 * reps) * 1e6,
                                          "loop": "step_slab(RANDOM)"}

    def csr_iter():
        env.step_slab(None, pkg.STEP_RANDOM, auto_reset=True)
        env.slab_to_csr(rows_per_table=128)

    dt, reps = timed_loop(lambda: [csr_iter() for _ in range(20)], sync)
    out["tables_65536_step_slab_csr_lists"] = {"env_steps_per_s": T * 20 * reps / dt, "us_per_iteration": dt / (20 * reps) * 1e6,
                                               "loop": "step_slab(RANDOM) + slab_to_csr: offsets / rows / ids as ddz_legal writes them"}
    env.csr_rows = env.csr_ids = None
    env._csr_cap = 0
    # (3) configs[3]: farmers played by the rule-based opponent (Env.step_auto), the lord by the random policy
    def auto_iter():
        ids = env.auto_choose(0b101)
        env.step_slab(ids, pkg.STEP_IDS, auto_reset=True)

    env.reset()           # whole episodes against the rule agent: start from fresh deals and play past the first
    env.legal_slab()      # games (a rule-agent game lasts ~25 plies) before episodes are counted
    for _ in range(40):
        auto_iter()
    s0 = env.stats()
    dt, reps = timed_loop(lambda: [auto_iter() for _ in range(5)], sync, min_s=0.2, max_reps=40)
    s1 = env.stats()
    eps = max(1, s1["episodes"] - s0["episodes"])
    lord, farm = s1["lord_wins"] - s0["lord_wins"], (s1["up_wins"] - s0["up_wins"]) + (s1["down_wins"] - s0["down_wins"])
    out["tables_65536_rule_opponent"] = {
        "env_steps_per_s": T * 5 * reps / dt, "us_per_iteration": dt / (5 * reps) * 1e6,
        "mean_episode_return": {"lord": 100.0 * (lord - farm) / eps, "up": 50.0 * (farm - lord) / eps,
                                "down": 50.0 * (farm - lord) / eps},
        "episodes": eps, "loop": "auto_choose(farmers) + step_slab(IDS), lord = engine RNG; reward_dict of game.py:13-14"}
    assert env.status() == 0
    del env, q, face
    return out


def issue_roofline(steps_timed, dur_launch):
    """The bound this integer path really runs into: vector-ALU issue.  Instructions per env step and the clock come
    from the committed PMC passes (profiles/pmc_traffic.json), the cost of an instruction from the measured table of
    tools/valu_issue_probe.hip (profiles/r02_valu_issue_probe.json), the rate is live."""
    try:
        prof = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json"))).get("k_rollout", {})
        v = prof["valu"]
        probe = json.load(open(os.path.join(REPO, "profiles", "r02_valu_issue_probe.json")))
    except Exception:
        return None
    cyc = {}
    for r in probe["rows"]:
        if r["waves_per_simd"] == 4:  # k_rollout runs four waves per SIMD at 4096 tables
            cyc[r["kind"]] = r["cycles_per_instr_per_simd"]
    full = (cyc["v_add_u32"] + cyc["v_and_b32/v_or_b32"]) / 2            # plain 32-bit VALU
    half = (cyc["v_lshlrev_b64"] + cyc["v_add_co_u32+v_addc_co_u32"] + cyc["v_sub_co_u32+v_subb_co_u32"]
            + cyc["v_mul_lo_u32"] + cyc["v_mbcnt_lo+v_mbcnt_hi"]) / 5   # 64-bit shifts, carry chains, mul, mbcnt, compares
    lane = cyc["v_readlane_b32"]                                          # v_readlane / v_readfirstlane
    mix = prof.get("valu_mix") or {"share_half_rate": 0.5, "share_readlane": 0.0, "source": "assumed"}
    sh, sr = mix["share_half_rate"], mix.get("share_readlane", 0.0)
    cpi = (1 - sh - sr) * full + sh * half + sr * lane
    simds = 256 * 4
    peak = simds * v["clock_GHz"] * 1e9 / cpi
    ach = v["SQ_INSTS_VALU_per_env_step"] * steps_timed / dur_launch
    return {"bound": "valu-issue", "achieved": ach / 1e9, "peak": peak / 1e9, "unit": "G wave-instr/s", "frac": ach / peak,
            "valu_insts_per_env_step": v["SQ_INSTS_VALU_per_env_step"], "cycles_per_instr_full_rate": full,
            "cycles_per_instr_half_rate": half, "cycles_per_instr_readlane": lane, "share_half_rate": sh,
            "share_readlane": sr, "mix_weighted_cycles_per_instr": cpi, "share_source": mix.get("source", "")[:160],
            "peak_if_all_full_rate": simds * v["clock_GHz"] / full, "peak_if_all_half_rate": simds * v["clock_GHz"] / half,
            "note": "cycles per wave64 instruction per SIMD measured by tools/valu_issue_probe.hip at 4 waves/SIMD "
                    "(profiles/r02_valu_issue_probe.json): ~2.4 for plain 32-bit ops (the guide's 2-cycle SIMD-32 figure), "
                    "~4.5 for 64-bit shifts / carry chains / v_mul_lo / v_mbcnt / compares, ~6.3 for v_readlane; the peak "
                    "weights them by the kernel's opcode mix"}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes -- one per GPU,
    `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` -- before this process has imported
    torch or touched a GPU (never an exec of a process that has initialised the device), relay rank 0's JSON line
    (stdout) and everybody's stderr, and return the worst exit status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without WORLD_SIZE: launching {' '.join(cmd)}", file=sys.stderr, flush=True)
    p = subprocess.Popen(cmd, env=env)  # stdout / stderr are inherited: the children's lines appear as they come
    try:
        rc = p.wait()
    except KeyboardInterrupt:
        p.terminate()
        rc = p.wait()
    return rc if rc >= 0 else 128 - rc  # a child killed by signal s -> 128 + s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--tables", type=int, default=0,
                    help="tables per GPU; default 4096 at N = 1 (configs[1]), 65536 at N > 1 (configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--no-configs", action="store_true", help="N = 1: skip the short legs of the other configs")
    ap.add_argument("--exchange-steps", type=int, default=200, help="N > 1: iterations of the trajectory-gather leg")
    ap.add_argument("--no-exchange", action="store_true", help="N > 1: skip the trajectory-gather leg")
    ap.add_argument("--allow-exchange-failure", action="store_true",
                    help="N > 1: report a failing trajectory-gather leg in the JSON and exit 0 (default: exit 3)")
    ap.add_argument("--strict-exchange", action="store_true", help="(default behaviour now; kept for old command lines)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on ONE GPU (all ranks on cuda:0, gloo, host-staged gather): control-flow rehearsal only")
    a = ap.parse_args()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))   # `python bench.py --gpus N` as typed: this process becomes the launcher

    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("doudizhu-rl_amd")
    ddist = importlib.import_module("doudizhu-rl_amd.dist")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if not a.rehearse and local >= torch.cuda.device_count():   # (device_count() does not initialise the GPU)
        raise SystemExit(f"rank {rank}: local rank {local} has no GPU ({torch.cuda.device_count()} visible); "
                         "--rehearse runs all ranks on cuda:0")
    dev = torch.device("cuda", 0 if a.rehearse else local)
    torch.cuda.set_device(dev)
    if world > 1:
        if a.rehearse:
            dist.init_process_group("gloo")
        else:
            import datetime
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=5))

    T = a.tables if a.tables > 0 else (4096 if world == 1 else 65536)
    total_tables = T * world
    _, base = ddist.shard_tables(total_tables, rank, world)
    env = pkg.BatchedEnv(T, seed=0, device=dev, table_id_base=base, want_ids=False)
    env.reset()
    K, W = a.steps, a.warmup

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- headline: the K steps, repeated until >= 50 ms are timed (same repeat count on every rank).  The engine runs
    # any number of lock-step iterations per launch; a launch carries at least 1000 of them (a multiple of K), so that a
    # small --steps does not turn the figure into a launch-latency measurement (a launch costs ~25 us of ramp-up)
    KL = K if K >= 1000 else K * math.ceil(1000 / K)   # iterations per launch
    env.rollout_random(W)
    barrier()
    t0 = time.perf_counter()
    env.rollout_random(KL)
    torch.cuda.synchronize(dev)
    est = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(est, op=dist.ReduceOp.MIN)
    RL = int(min(1 << 16, max(1, math.ceil(MIN_TIMED_S / max(float(est.item()), 1e-6)))))   # launches timed
    R = RL * (KL // K)                                                                      # repeats of the K steps
    s0 = env.stats()  # cumulative counters so far (sync)
    barrier()
    t0 = time.perf_counter()
    for _ in range(RL):
        env.rollout_random(KL)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    s1 = env.stats()
    st = {k: s1[k] - s0[k] for k in s1}
    status = env.status()
    assert st["plies"] == T * K * R and status == 0, (st, status)

    # ---- the path's one exchange, measured beside the headline, never inside it: every ply also writes
    # its 32-byte trajectory record and the batch goes to the learner (rank 0) over RCCL in two half-batches; the
    # gather of the first half overlaps the rollout of the second one.
    exchange = None
    if world > 1 and not a.no_exchange:
        try:
            KX = max(2, a.exchange_steps)
            half = KX // 2
            shard = [T] * world
            stage = (lambda x: x.cpu()) if a.rehearse else (lambda x: x)
            traj_a = torch.zeros((half, T, pkg.TRAJ_BYTES), dtype=torch.uint8, device=dev)
            traj_b = torch.zeros((KX - half, T, pkg.TRAJ_BYTES), dtype=torch.uint8, device=dev)
            pack = pkg.pack_trajectory  # 32-byte records -> 8 bytes before they cross xGMI
            ddist.gather_trajectories(stage(pack(traj_a[:1].contiguous())), dst=0, shard_sizes=shard)  # warm the collective
            barrier()
            tx = time.perf_counter()
            env.rollout_random(half, traj=traj_a)
            if a.rehearse:
                torch.cuda.synchronize(dev)
            pending = ddist.gather_trajectories(stage(pack(traj_a)), dst=0, async_op=True, shard_sizes=shard)
            env.rollout_random(KX - half, traj=traj_b)
            if a.rehearse:
                torch.cuda.synchronize(dev)
            tail = ddist.gather_trajectories(stage(pack(traj_b)), dst=0, async_op=True, shard_sizes=shard)
            ga, gb = pending.result(), tail.result()
            barrier()
            dtx = time.perf_counter() - tx
            tmx = torch.tensor([dtx], dtype=torch.float64, device=dev)
            dist.all_reduce(tmx, op=dist.ReduceOp.MAX)
            dtx = float(tmx.item())
            digest = None
            if rank == 0:
                assert ga.shape == (half, total_tables, pkg.TRAJ_PACKED_BYTES) and gb.shape == (KX - half, total_tables, pkg.TRAJ_PACKED_BYTES)
                rec = ddist.unpack_trajectory(gb[-1].to(dev))
                assert int(rec["ply"].max()) < 200 and int(rec["role"].max()) <= 2
                assert int(rec["id"][rec["flags"] == 0].max()) < 13527
                x = torch.cat([ga, gb]).to(dev).contiguous().view(torch.int64).view(-1)
                # digest of every gathered record in (iteration, global table) order: the rehearsal test recomputes it
                # from a single-process rollout of the union of the shards
                digest = int((x * 31 + (x >> 13) + torch.arange(x.numel(), device=dev) * x).sum().item())
            exchange = {"steps": KX, "iterations_before": W + KL * (RL + 1), "env_steps_per_s_with_gather": total_tables * KX / dtx,
                        "bytes_to_rank0": (world - 1) * KX * T * pkg.TRAJ_PACKED_BYTES, "seconds": dtx, "digest": digest,
                        "note": "trajectory records (32 B per ply per table) written, packed to 8 B and gathered to rank 0, pipelined "
                                "in two half-batches; measured after the headline region"}
            del traj_a, traj_b, ga, gb
        except Exception as ex:  # the headline above stands on its own; report loudly, do not lose the line
            exchange = {"error": repr(ex)[:300]}
            print(f"[bench] rank {rank}: trajectory-gather leg FAILED: {ex!r}", file=sys.stderr, flush=True)
    mean_a = st["legal_rows"] / max(1, st["plies"])

    # ---- duration of the dominant kernel: all K iterations run inside ONE k_rollout launch; two HIP events around
    # every launch on the launching stream, repeated until >= 50 ms of kernel time are summed
    sa = env.stats()
    ms_total, n_launch = 0.0, 0
    while ms_total < MIN_TIMED_S * 1e3 and n_launch < (1 << 16):
        ms_total += env.rollout_random_timed(KL)
        n_launch += 1
    sb = env.stats()
    steps_timed = sb["plies"] - sa["plies"]
    mean_a2 = (sb["legal_rows"] - sa["legal_rows"]) / max(1, steps_timed)
    # algorithmic bytes per env step, SURVEY.md 8(d): 128 state read + 128 state write + 4 list size/offset + 16*A
    # legal rows (DESIGN.md 3 states what the kernel really moves)
    b_step = 260 + 16 * mean_a2
    dur_launch = ms_total * 1e-3 / n_launch           # mean duration of one K-iteration launch
    steps_per_launch = steps_timed / n_launch
    b_launch = b_step * steps_per_launch
    dominant = "k_rollout"
    ach = b_launch / dur_launch / 1e9
    traffic = None
    try:
        prof = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json"))).get(dominant, {})
        per_step = prof.get("hbm_bytes_per_env_step")
        traffic = per_step * steps_per_launch if per_step is not None else None
    except Exception:
        traffic = None
    issue = issue_roofline(steps_per_launch, dur_launch)

    # ---- the same loop with packed CSR lists (one launch per iteration): a number, not the headline
    env.rollout_random_csr(20)
    torch.cuda.synchronize(dev)
    dtc, repc = timed_loop(lambda: env.rollout_random_csr(50), lambda: torch.cuda.synchronize(dev))
    csr_rate = T * 50 * repc / dtc

    if rank == 0:
        out = {
            "metric": "env steps/sec (batched tables)", "value": total_tables * K * R / dt,
            "unit": "env steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / (K * R) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "timed_steps": K * R, "repeats_of_the_steps_launch": R, "iterations_per_launch": KL, "launches_timed": RL,
            "timed_seconds": dt,
            "config": {"workload": f"{T} tables per GPU, random policy (engine RNG), legal-move list only (no NN), "
                                   f"auto-reset; BASELINE.json {'configs[1]' if world == 1 else 'configs[4] (65,536 tables per GPU)'}",
                       "tables_per_gpu": T, "total_tables": total_tables,
                       "mean_legal_moves": round(mean_a, 3), "episodes": st["episodes"],
                       "list_layout": "slab (fixed-stride segment per table)",
                       "csr_env_steps_per_s": csr_rate,
                       "exchange": exchange},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "kernel": dominant,
                         "launch_us": dur_launch * 1e6, "launches_timed": n_launch, "env_steps_per_launch": steps_per_launch,
                         "us_per_iteration": dur_launch * 1e6 / KL, "iterations_per_launch": KL,
                         "algorithmic_bytes_per_env_step": b_step,
                         "algorithmic_bytes_per_launch": b_launch,
                         "note": "achieved = ALGORITHMIC bytes (SURVEY 8d) / launch time: a store rate into the write-back "
                                 "L2, not HBM utilisation -- every iteration overwrites the same state rows / list slab, so "
                                 "the HBM counters (traffic) see ~2 B per env step.  The path is instruction-issue bound: "
                                 "see issue",
                         "issue": issue},
        }
        if exchange and "env_steps_per_s_with_gather" in exchange:
            out["env_steps_per_s_with_gather"] = exchange["env_steps_per_s_with_gather"]
        if world == 1 and not a.no_configs:
            try:
                out["configs"] = other_config_legs(pkg, torch, dev)
            except Exception as ex:
                out["configs"] = {"error": repr(ex)[:300]}
                print(f"[bench] config legs FAILED: {ex!r}", file=sys.stderr, flush=True)
            try:
                out["configs"]["stress_plane_rich_leads"] = stress_leg(pkg, torch, dev)
            except Exception as ex:
                out["configs"]["stress_plane_rich_leads"] = {"error": repr(ex)[:300]}
                print(f"[bench] stress leg FAILED: {ex!r}", file=sys.stderr, flush=True)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T, a.cpu_budget)
        print(json.dumps(out), flush=True)
    failed = bool(exchange and "error" in exchange)
    if world > 1:
        dist.destroy_process_group()
    if failed and not a.allow_exchange_failure:
        sys.exit(3)   # loud: a broken gather must not look like a green run


if __name__ == "__main__":
    main()
