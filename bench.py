#!/usr/bin/env python3
"""Headline benchmark: env steps/s of the batched lock-step engine (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
A "step" is one lock-step iteration of the hot path over all tables of this rank: legal-move enumeration into the
per-table list + random-policy action application with auto-reset (game.py:169-181 with envi.py:79-116).
  Every N: 65,536 tables per GPU -- the largest single-GPU table count of BASELINE.json (configs[2], configs[3]; configs[4]
         = 524,288 tables at N = 8), random policy, legal-move list only (the workload of configs[1]).  Weak scaling: every
         GPU owns a range of the global table ids and the timed region is the same at every N (tables are independent: no
         data-path collective); the N = 1 line is the N > 1 line's per-GPU workload.  The path's one exchange -- packed
         trajectories gathered to rank 0 over RCCL -- is measured right after the headline region (config.exchange,
         env_steps_per_s_with_gather, the ranks' own rates and the gather on its own).
  N = 1 also carries measured legs of the other single-GPU configurations (`configs`), each with the roofline that
         bounds its dominant kernel: configs[1] as written (4096 tables), the policy-driven stepping launch, configs[2]
         with the Q-network in the loop, configs[3] against the rule agent, the CSR layout beside the slab layout.
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
F32_MATRIX_PEAK_TFLOPS = 157.3  # same guide: v_mfma_f32_32x32x2_f32, exact f32, 64 FLOP/clk/SIMD (= the f32 vector peak)
T_FULL = 65536          # tables per GPU: the largest single-GPU configuration of BASELINE.json
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


def rollout_leg(pkg, torch, dev, T, pmc_key):
    """The headline loop at another table count (configs[1] as BASELINE.json writes it: 4096 tables): rate by wall clock
    between syncs, k_rollout's launch duration by HIP events, the roofline block, and the packed-CSR form of the loop."""
    sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
    env = pkg.BatchedEnv(T, seed=0, device=dev, want_ids=False)
    env.reset()
    env.rollout_random(200)
    k = 1000
    dt, reps = timed_loop(lambda: env.rollout_random(k), sync)
    sa = env.stats()
    ms, n = 0.0, 0
    while ms < MIN_TIMED_S * 1e3 and n < 4096:
        ms += env.rollout_random_timed(k)
        n += 1
    sb = env.stats()
    steps = sb["plies"] - sa["plies"]
    mean_a = (sb["legal_rows"] - sa["legal_rows"]) / max(1, steps)
    dur = ms * 1e-3 / n
    out = {"env_steps_per_s": T * k * reps / dt, "us_per_iteration": dt / (k * reps) * 1e6, "iterations": k * reps,
           "mean_legal_moves": mean_a,
           "roofline": hbm_block("k_rollout", (260 + 16 * mean_a) * steps / n, dur, launches_timed=n,
                                 env_steps_per_launch=steps / n, algorithmic_bytes_per_env_step=260 + 16 * mean_a,
                                 issue=issue_block(pmc_key, steps / n, dur))}
    env.rollout_random_csr(64)
    nc = 1024 if T <= 8192 else 256
    dtc, repc = timed_loop(lambda: env.rollout_random_csr(nc), sync)
    out["csr_env_steps_per_s"] = T * nc * repc / dtc
    out["csr_us_per_iteration"] = dtc / (nc * repc) * 1e6
    out["csr_staging_GiB"] = env._staging.numel() / 2**30
    env.rollout_random_csr(20, batch=0)
    dtc, repc = timed_loop(lambda: env.rollout_random_csr(50, batch=0), sync)
    out["csr_launch_per_iteration_env_steps_per_s"] = T * 50 * repc / dtc
    out["csr_note"] = ("packed CSR lists every iteration: batches of iterations staged by one rollout launch + two compaction "
                       "launches (ddz_rollout_random_csr_staged); csr_launch_per_iteration: round 3's form")
    assert env.status() == 0
    return out


def other_config_legs(pkg, torch, dev, errors):
    """Measured legs of the other single-GPU configurations of BASELINE.json (numbers, not prose), each with the roofline
    of its dominant kernel.  A leg that raises lands in `errors` (the run then exits non-zero) and the others still run."""
    out = {}
    T = T_FULL
    sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731

    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as ex:  # noqa: BLE001
            errors.append({"leg": name, "error": repr(ex)[:300]})
            out[name] = {"error": repr(ex)[:300]}
            print(f"[bench] leg {name} FAILED: {ex!r}", file=sys.stderr, flush=True)

    # (0) configs[1] exactly as BASELINE.json writes it: 4096 tables, random policy, legal-move list only
    leg("tables_4096_random_rollout", lambda: rollout_leg(pkg, torch, dev, 4096, "k_rollout"))

    # (0b) the launch a policy drives at that size: step_slab(RANDOM) at 4096 tables = one table per wavefront -- a launch that
    # lasts as long as its slowest list (k_slab<0,true,true>: deals by the tables' waves, plane-rich leads written by the block)
    def step_slab_4096_leg():
        e4 = pkg.BatchedEnv(4096, seed=0, device=dev)
        e4.reset()
        e4.rollout_random(200)
        e4.legal_slab()
        s0 = e4.stats()
        dt, reps = timed_loop(lambda: [e4.step_slab(None, pkg.STEP_RANDOM, auto_reset=True) for _ in range(50)], sync)
        s1 = e4.stats()
        a = (s1["legal_rows"] - s0["legal_rows"]) / max(1, s1["plies"] - s0["plies"])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2000):
            e4.step_slab(None, pkg.STEP_RANDOM, auto_reset=True)
        e1.record()
        sync()
        per = e0.elapsed_time(e1) * 1e-3 / 2000      # HIP events around 2000 back-to-back launches on the launching stream
        b = 260 + 20 * a
        assert e4.status() == 0
        return {"env_steps_per_s": 4096 * 50 * reps / dt, "us_per_iteration": dt / (50 * reps) * 1e6, "us_per_launch_events": per * 1e6,
                "mean_legal_moves": a,
                "roofline": hbm_block("k_slab<0,true,true>", b * 4096, per, algorithmic_bytes_per_env_step=b,
                                      issue=issue_block("k_slab_random_4096", 4096, per)),
                "loop": "step_slab(RANDOM), 4096 tables: latency-bound (256 blocks of 16 waves, one table per wave)"}

    leg("tables_4096_step_slab_only", step_slab_4096_leg)
    env = pkg.BatchedEnv(T, seed=0, device=dev)
    env.reset()
    env.rollout_random(200)

    # (1) configs[4]'s per-rank half of the trajectory exchange (what every rank does before the gather over xGMI):
    # the rollout writing a 32-byte record per ply and table, then ddz_pack_trajectory to 8-byte records
    def traj_leg():
        kx = 100
        traj = torch.zeros((kx, T, pkg.TRAJ_BYTES), dtype=torch.uint8, device=dev)

        def traj_iter():
            env.rollout_random(kx, traj=traj)
            pkg.pack_trajectory(traj)

        dt, reps = timed_loop(traj_iter, sync)
        dtp, repp = timed_loop(lambda: pkg.pack_trajectory(traj), sync)
        nrec = kx * T
        return {"env_steps_per_s": T * kx * reps / dt, "us_per_iteration": dt / (kx * reps) * 1e6, "iterations": kx * reps,
                "pack_us_per_iteration": dtp / (kx * repp) * 1e6, "packed_bytes_per_iteration": T * pkg.TRAJ_PACKED_BYTES,
                "packed_GBps_to_send": T * pkg.TRAJ_PACKED_BYTES * kx * reps / dt / 1e9,
                "roofline": hbm_block("k_pack_traj", nrec * (pkg.TRAJ_BYTES + pkg.TRAJ_PACKED_BYTES), dtp / repp,
                                      algorithmic_bytes_per_record=pkg.TRAJ_BYTES + pkg.TRAJ_PACKED_BYTES),
                "loop": "rollout_random(traj = 32-byte records) + pack_trajectory (8-byte records): the per-rank cost of "
                        "config 5 before the RCCL gather"}

    leg("tables_65536_traj_write_pack", traj_leg)
    # (2) configs[2]'s environment side: the loop a policy drives through the slab API -- face, selection over
    # per-action values, apply + next lists; one launch each (game.py:95-104, dqn.py:50-71).  The Q values are random
    # numbers standing in for the network's output (the network in the loop: tables_65536_dqn_inference).
    env.legal_slab()
    face = torch.empty((T, 6, 15, 4), dtype=torch.float32, device=dev)
    q = torch.rand((T, env.slab_stride), dtype=torch.float32, device=dev)
    choice = torch.empty(T, dtype=torch.int32, device=dev)

    def mean_moves(fn):
        s0 = env.stats()
        r = fn()
        s1 = env.stats()
        return r, (s1["legal_rows"] - s0["legal_rows"]) / max(1, s1["plies"] - s0["plies"])

    def policy_slab_leg():
        def policy_iter():
            env.observe(3, out=face)
            env.select_slab(q, out=choice)
            env.step_slab(choice, pkg.STEP_CHOICE, auto_reset=True)

        (dt, reps), a = mean_moves(lambda: timed_loop(lambda: [policy_iter() for _ in range(20)], sync))
        # bytes per table-step of the three launches: state 176 read x 3 + 176 written, face 1,440 written, q 4 A read,
        # lists 20 A (+ 4) written, choice 4 + 4
        b = 4 * 176 + 1440 + 24 * a + 12
        return {"env_steps_per_s": T * 20 * reps / dt, "iterations": 20 * reps, "us_per_iteration": dt / (20 * reps) * 1e6,
                "mean_legal_moves": a, "roofline": hbm_block("k_observe<3> + k_select + k_slab<1,true>", b * T, dt / (20 * reps),
                                                            algorithmic_bytes_per_table_step=b),
                "loop": "observe(EnvCooperationSimplify) + select_slab(random q) + step_slab(CHOICE)"}

    leg("tables_65536_policy_loop_slab", policy_slab_leg)

    def policy_fused_leg():
        # the same iteration in ONE launch: arg-max over q, apply, new lists and the new `face` (ddz_policy_step_slab)
        (dt, reps), a = mean_moves(lambda: timed_loop(
            lambda: [env.policy_step_slab(q, 0.0, face_variant=3, face_out=face) for _ in range(20)], sync))
        b = 2 * 176 + 1440 + 24 * a + 8
        return {"env_steps_per_s": T * 20 * reps / dt, "iterations": 20 * reps, "us_per_iteration": dt / (20 * reps) * 1e6,
                "mean_legal_moves": a,
                "roofline": hbm_block("k_slab<4,true>", b * T, dt / (20 * reps), algorithmic_bytes_per_table_step=b,
                                      issue=issue_block("k_slab_fused_65536", T, dt / (20 * reps))),
                "loop": "policy_step_slab(random q, face = EnvCooperationSimplify): one launch"}

    leg("tables_65536_policy_loop_fused", policy_fused_leg)
    leg("tables_65536_dqn_inference", lambda: dqn_leg(pkg, torch, dev, env))

    # the stepping launch alone (uniformly random legal moves drawn in the kernel: every selection is in its list), and
    # the same with the new lists packed to CSR every iteration (what a ragged NN forward over all legal moves consumes)
    def step_slab_leg():
        (dt, reps), a = mean_moves(lambda: timed_loop(
            lambda: [env.step_slab(None, pkg.STEP_RANDOM, auto_reset=True) for _ in range(20)], sync))
        b = 260 + 20 * a       # 128 + 128 + 4 + 16 A, + 4 A for the ids this handle also writes
        per = dt / (20 * reps)
        return {"env_steps_per_s": T * 20 * reps / dt, "us_per_iteration": per * 1e6, "mean_legal_moves": a,
                "roofline": hbm_block("k_slab<0,true>", b * T, per, algorithmic_bytes_per_env_step=b,
                                      issue=issue_block("k_slab_random_65536", T, per)),
                "loop": "step_slab(RANDOM)"}

    leg("tables_65536_step_slab_only", step_slab_leg)

    def step_csr_leg():
        def csr_iter():
            env.step_slab(None, pkg.STEP_RANDOM, auto_reset=True)
            env.slab_to_csr(rows_per_table=128)

        (dt, reps), a = mean_moves(lambda: timed_loop(lambda: [csr_iter() for _ in range(20)], sync))
        dtc, repc = timed_loop(lambda: [env.slab_to_csr(rows_per_table=128) for _ in range(20)], sync)
        env.csr_rows = env.csr_ids = None
        env._csr_cap = 0
        return {"env_steps_per_s": T * 20 * reps / dt, "us_per_iteration": dt / (20 * reps) * 1e6, "mean_legal_moves": a,
                "slab_to_csr_us": dtc / (20 * repc) * 1e6,
                "roofline": hbm_block("k_csr_scan + k_csr_copy", (8 + 40 * a) * T, dtc / (20 * repc),
                                      algorithmic_bytes_per_table=8 + 40 * a,
                                      note="the compaction pass alone: list sizes in, offsets out, 20-byte rows + ids read and written"),
                "loop": "step_slab(RANDOM) + slab_to_csr: offsets / rows / ids as ddz_legal writes them"}

    leg("tables_65536_step_slab_csr_lists", step_csr_leg)

    # (3) configs[3]: farmers played by the rule-based opponent (Env.step_auto), the lord by the random policy
    def rule_leg():
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
        dta, repa = timed_loop(lambda: env.auto_choose(0b101), sync, min_s=0.05, max_reps=200)   # the decisions alone
        eps = max(1, s1["episodes"] - s0["episodes"])
        lord, farm = s1["lord_wins"] - s0["lord_wins"], (s1["up_wins"] - s0["up_wins"]) + (s1["down_wins"] - s0["down_wins"])
        role = env.role
        n_dec = int(((role == 0) | (role == 2)).sum().item())
        return {"env_steps_per_s": T * 5 * reps / dt, "us_per_iteration": dt / (5 * reps) * 1e6,
                "us_auto_choose": dta / repa * 1e6, "decisions_per_launch": n_dec,
                "mean_episode_return": {"lord": 100.0 * (lord - farm) / eps, "up": 50.0 * (farm - lord) / eps,
                                        "down": 50.0 * (farm - lord) / eps},
                "episodes": eps,
                "roofline": hbm_block("k_auto2<true> (+ k_auto_order)", (176 + 4) * T, dta / repa,
                                      algorithmic_bytes_per_table=180,
                                      note="a search kernel: 176-byte state in, one id out -- the HBM roof says nothing here; "
                                           "the bound is instruction issue (issue) and, per launch, its single longest decision",
                                      issue=issue_block("k_auto2_65536", 1, dta / repa)),
                "loop": "auto_choose(farmers) + step_slab(IDS), lord = engine RNG; reward_dict of game.py:13-14"}

    leg("tables_65536_rule_opponent", rule_leg)
    if env.status() != 0:
        errors.append({"leg": "configs", "error": f"device status {env.status()}"})
    del env, q, face
    return out


def dqn_leg(pkg, torch, dev, env):
    """configs[2] as SURVEY 8(d) defines it: EnvCooperationSimplify planes + NetCooperationSimplify (net.py:137-150)
    randomly initialised (torch.manual_seed(0)), eval mode, greedy arg-max per table over its legal list -- the network
    IN the loop (dqn_glue.PolicyLoop).  Per-stage times by HIP events on the launching stream, a roofline per stage and
    end to end."""
    T = env.T
    sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    torch.manual_seed(0)
    net = glue.QNet(6).to(dev).eval()
    loop = glue.PolicyLoop(env, net, face_variant=3, epsilon=0.0)
    loop.run(3)
    s0 = env.stats()
    dt, reps = timed_loop(lambda: loop.run(5), sync, min_s=0.3, max_reps=64)
    s1 = env.stats()
    per_iter = dt / (5 * reps)
    rows_eval = s1["legal_rows"] - s0["legal_rows"]
    res = {"env_steps_per_s": T / per_iter, "us_per_iteration": per_iter * 1e6, "iterations": 5 * reps,
           "q_evals_per_s": rows_eval / dt, "mean_legal_moves": rows_eval / max(1, s1["plies"] - s0["plies"]),
           "dtype_net": "f32", "net": "NetCooperationSimplify-shaped QNet(6 + 1 planes), torch.manual_seed(0), eval()",
           "loop": loop.describe()}
    # per stage: HIP events around every stage of 10 iterations on the launching stream
    stages = loop.profile(10)
    res["stages_us"] = {k: v["us"] for k, v in stages.items()}
    blocks = {}
    flop_total = 0.0
    for name, st in stages.items():
        if st.get("flop"):
            blocks[name] = mfma_block(st["kernel"], st["flop"], st["us"] * 1e-6, note=st.get("note", ""))
            flop_total += st["flop"]
        elif st.get("bytes"):
            blocks[name] = hbm_block(st["kernel"], st["bytes"], st["us"] * 1e-6, note=st.get("note", ""))
    triples = stages.get("shared_need", {}).get("needed_triples")     # rows of the per-table form of D
    dense_flop = 2.0 * T * (6 * 60 + 15 * 256) * 256 + (2.0 * triples * 256 * 256 if triples else
                                                         sum(st["flop"] for k, st in stages.items() if k == "fc1_rows"))
    res["roofline"] = mfma_block("the whole iteration (all launches)", flop_total, per_iter,
                                 note="GEMM FLOP EXECUTED in one iteration / the iteration's wall time.  With shared rows (one row "
                                      "per distinct (rank, face column) of the batch) the count-0 term needs ~20 x fewer FLOP than its "
                                      "dense form (dense_form_flop: [T, 3840] x [3840, 256] over every table), so the matrix share of "
                                      "the iteration is small and the fraction of the MFMA peak says little: the iteration is bound by "
                                      "its HBM / latency stages (stages: per-kernel blocks)",
                                 stages=blocks, executed_flop=flop_total, dense_form_flop=dense_flop,
                                 dense_form_equivalent_TFLOPs=dense_flop / per_iter / 1e12)
    try:   # the same loop replayed from ONE hipGraph of 5 iterations (no host work between its ~150 launches)
        g = loop.capture(5)
        dtg, repg = timed_loop(g.replay, sync, min_s=0.3, max_reps=64)
        res["graph_replay"] = {"env_steps_per_s": T * 5 * repg / dtg, "us_per_iteration": dtg / (5 * repg) * 1e6,
                               "note": "PolicyLoop.capture(5): five iterations as one hipGraph, replayed; the headline of this leg is the "
                                       "eager loop"}
        del g
    except Exception as e:  # noqa: BLE001
        res["graph_replay"] = {"error": repr(e)}
    if hasattr(loop, "variants"):
        res["variants"] = loop.variants(timed_loop, sync)
    del loop, net
    return res


_PROFILE_CACHE = {}


def _profile(name):
    if name not in _PROFILE_CACHE:
        try:
            _PROFILE_CACHE[name] = json.load(open(os.path.join(REPO, "profiles", name)))
        except Exception:
            _PROFILE_CACHE[name] = None
    return _PROFILE_CACHE[name]


def hbm_block(kernel, bytes_per_launch, seconds_per_launch, **extra):
    """roofline object against the HBM roof: ALGORITHMIC bytes of one launch (SURVEY 8d per-unit figure x units) / the
    launch's measured duration."""
    ach = bytes_per_launch / seconds_per_launch / 1e9
    out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
           "kernel": kernel, "launch_us": seconds_per_launch * 1e6, "algorithmic_bytes_per_launch": bytes_per_launch}
    out.update(extra)
    return out


def mfma_block(kernel, flop_per_launch, seconds_per_launch, **extra):
    """roofline object against the dense fp32 matrix roof (v_mfma_f32_32x32x2_f32: 157.3 TFLOP/s, exact f32)."""
    ach = flop_per_launch / seconds_per_launch / 1e12
    out = {"bound": "mfma", "achieved": ach, "peak": F32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / F32_MATRIX_PEAK_TFLOPS,
           "kernel": kernel, "launch_us": seconds_per_launch * 1e6, "flop_per_launch": flop_per_launch, "dtype": "f32"}
    out.update(extra)
    return out


def issue_block(key, units_per_launch, seconds_per_launch):
    """The bound this integer path really runs into: vector-ALU issue.  LIVE: the launch duration (and so the rate).
    REPLAYED from committed profiles, each named in `sources`: the kernel's VALU instructions per unit and its clock
    (profiles/pmc_kernels.json: rocprofv3 --pmc passes), its static opcode mix (profiles/r04_opcode_mix.json:
    tools/opcode_mix.py over the code object) and the cost of an instruction of each class (profiles/
    r02_valu_issue_probe.json: tools/valu_issue_probe.hip)."""
    pk, mixes, probe = _profile("pmc_kernels.json"), _profile("r04_opcode_mix.json"), _profile("r02_valu_issue_probe.json")
    if not pk or key not in pk or not mixes or not probe:
        return None
    k = pk[key]
    mix = mixes["kernels"].get(k["mix_key"])
    if mix is None:
        return None
    wps = 4 if k.get("waves_per_simd", 4) >= 4 else 2
    cyc = {r["kind"]: r["cycles_per_instr_per_simd"] for r in probe["rows"] if r["waves_per_simd"] == wps}
    full = (cyc["v_add_u32"] + cyc["v_and_b32/v_or_b32"]) / 2            # plain 32-bit VALU
    half = (cyc["v_lshlrev_b64"] + cyc["v_add_co_u32+v_addc_co_u32"] + cyc["v_sub_co_u32+v_subb_co_u32"]
            + cyc["v_mul_lo_u32"] + cyc["v_mbcnt_lo+v_mbcnt_hi"]) / 5   # 64-bit shifts, carry chains, mul, mbcnt, compares
    lane = cyc["v_readlane_b32"]                                          # v_readlane / v_readfirstlane
    sh, sr = mix["share_half_rate"], mix["share_readlane"]
    cpi = (1 - sh - sr) * full + sh * half + sr * lane
    simds = 256 * 4
    peak = simds * k["clock_GHz"] * 1e9 / cpi
    ach = k["valu_per_unit"] * units_per_launch / seconds_per_launch
    return {"bound": "valu-issue", "achieved": ach / 1e9, "peak": peak / 1e9, "unit": "G wave-instr/s", "frac": ach / peak,
            "kernel": k["mix_key"], "launch_us": seconds_per_launch * 1e6, "unit_of_work": k["unit"],
            "valu_insts_per_unit": k["valu_per_unit"], "salu_insts_per_unit": k.get("salu_per_unit"),
            "branch_insts_per_unit": k.get("branch_per_unit"), "wait_any_share_of_wave_cycles": k.get("wait_any_share"),
            "clock_GHz": k["clock_GHz"], "cycles_per_instr_full_rate": full, "cycles_per_instr_half_rate": half,
            "cycles_per_instr_readlane": lane, "share_half_rate": sh, "share_readlane": sr,
            "mix_weighted_cycles_per_instr": cpi, "peak_if_all_full_rate": simds * k["clock_GHz"] / full,
            "peak_if_all_half_rate": simds * k["clock_GHz"] / half,
            "live": ["launch_us", "achieved", "frac"],
            "sources": {"valu_insts_per_unit, salu / branch / wait shares, clock_GHz": k["source"] + " via profiles/pmc_kernels.json",
                        "share_half_rate, share_readlane": "profiles/r04_opcode_mix.json (static opcode histogram of the code object)",
                        "cycles_per_instr_*": f"profiles/r02_valu_issue_probe.json ({wps} waves per SIMD)"},
            "note": "replayed inputs (see sources) x the live launch duration; cycles per wave64 instruction per SIMD: ~2.4 "
                    "plain 32-bit, ~4.5 for 64-bit shifts / carry chains / v_mul_lo / v_mbcnt / compares, ~6.3 v_readlane"}


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
                    help="tables per GPU; default 65536 at every N (the largest single-GPU configuration; configs[4] at N = 8)")
    ap.add_argument("--allow-leg-failure", action="store_true",
                    help="N = 1: report a failing leg in the JSON (`errors`) and exit 0 (default: exit 4)")
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

    T = a.tables if a.tables > 0 else T_FULL
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
    per_rank = [T * K * R / dt]
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        mine = tmax.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                       # every rank's own clock over the same K x R iterations
        per_rank = [T * K * R / float(x.item()) for x in every]
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
            # the gather on its own (nothing overlapping it): the second half-batch's packed records once more
            pk_b = stage(pack(traj_b))
            barrier()
            tg = time.perf_counter()
            gg = ddist.gather_trajectories(pk_b, dst=0, shard_sizes=shard)
            barrier()
            dtg = time.perf_counter() - tg
            tmg = torch.tensor([dtg], dtype=torch.float64, device=dev)
            dist.all_reduce(tmg, op=dist.ReduceOp.MAX)
            dtg = float(tmg.item())
            del gg, pk_b
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
                        "gather_only": {"seconds": dtg, "bytes_into_rank0": (world - 1) * (KX - half) * T * pkg.TRAJ_PACKED_BYTES,
                                        "GBps_into_rank0": (world - 1) * (KX - half) * T * pkg.TRAJ_PACKED_BYTES / dtg / 1e9,
                                        "collective": "gather (dst = rank 0) of uint8 [iterations, tables, 8], "
                                                      + ("gloo, host-staged (rehearsal)" if a.rehearse else "RCCL over xGMI")},
                        "note": "trajectory records (32 B per ply per table) written, packed to 8 B and gathered to rank 0, pipelined "
                                "in two half-batches; measured after the headline region"}
            del traj_a, traj_b, ga, gb
        except Exception as ex:  # the headline above stands on its own; report loudly, do not lose the line
            exchange = {"error": repr(ex)[:300]}
            print(f"[bench] rank {rank}: trajectory-gather leg FAILED: {ex!r}", file=sys.stderr, flush=True)
    mean_a = st["legal_rows"] / max(1, st["plies"])
    leg_errors = []

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
    pmc_key = "k_rollout_65536" if T >= 32768 and "k_rollout_65536" in (_profile("pmc_traffic.json") or {}) else "k_rollout"
    traffic, traffic_source = None, None
    prof = (_profile("pmc_traffic.json") or {}).get(pmc_key, {})
    if prof.get("hbm_bytes_per_env_step") is not None:
        traffic = prof["hbm_bytes_per_env_step"] * steps_per_launch
        traffic_source = ("REPLAYED, not measured in this run: profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                          "separate passes, FETCH doubled per the gfx950 note; " + str(prof.get("workload", "")) + ") x this run's env steps per launch")
    issue = issue_block(pmc_key if pmc_key in (_profile("pmc_kernels.json") or {}) else "k_rollout", steps_per_launch, dur_launch)
    ach = b_launch / dur_launch / 1e9

    # ---- the same loop with packed CSR lists (one launch per iteration): a number, not the headline
    env.rollout_random_csr(64)
    torch.cuda.synchronize(dev)
    dtc, repc = timed_loop(lambda: env.rollout_random_csr(256), lambda: torch.cuda.synchronize(dev))
    csr_rate = T * 256 * repc / dtc
    csr_us = dtc / (256 * repc) * 1e6
    env.rollout_random_csr(10, batch=0)
    dtc, repc = timed_loop(lambda: env.rollout_random_csr(20, batch=0), lambda: torch.cuda.synchronize(dev))
    csr_rate_per_launch = T * 20 * repc / dtc

    if rank == 0:
        out = {
            "metric": "env steps/sec (batched tables)", "value": total_tables * K * R / dt,
            "unit": "env steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / (K * R) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "timed_steps": K * R, "repeats_of_the_steps_launch": R, "iterations_per_launch": KL, "launches_timed": RL,
            "timed_seconds": dt,
            "config": {"workload": f"{T} tables per GPU, random policy (engine RNG), legal-move list only (no NN), auto-reset: "
                                   "the workload of BASELINE.json configs[1] at the largest single-GPU table count "
                                   "(65,536 per GPU: configs[2] / configs[3]; configs[4] = 8 x 65,536), the same per-GPU "
                                   "workload at every N",
                       "tables_per_gpu": T, "total_tables": total_tables,
                       "mean_legal_moves": round(mean_a, 3), "episodes": st["episodes"],
                       "list_layout": "slab (fixed-stride segment per table)",
                       "csr_env_steps_per_s": csr_rate, "csr_us_per_iteration": csr_us,
                       "csr_staging_GiB": env._staging.numel() / 2**30,
                       "csr_launch_per_iteration_env_steps_per_s": csr_rate_per_launch,
                       "csr_note": "the same loop with packed CSR lists (offsets / rows as ddz_legal writes them: SURVEY 8(d) "
                                   "config 2 'outputs = CSR legal list only') beside the slab layout of `value`: batches of "
                                   "iterations staged by one rollout launch, compacted by two more "
                                   "(ddz_rollout_random_csr_staged); csr_launch_per_iteration: round 3's form",
                       "per_rank_env_steps_per_s": per_rank,
                       "exchange": exchange},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source, "kernel": dominant,
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
            del env                                    # (the legs make their own environments)
            try:
                out["configs"] = other_config_legs(pkg, torch, dev, leg_errors)
            except Exception as ex:  # noqa: BLE001
                out["configs"] = {"error": repr(ex)[:300]}
                leg_errors.append({"leg": "configs", "error": repr(ex)[:300]})
                print(f"[bench] config legs FAILED: {ex!r}", file=sys.stderr, flush=True)
            try:
                out["configs"]["stress_plane_rich_leads"] = stress_leg(pkg, torch, dev)
            except Exception as ex:  # noqa: BLE001
                out["configs"]["stress_plane_rich_leads"] = {"error": repr(ex)[:300]}
                leg_errors.append({"leg": "stress_plane_rich_leads", "error": repr(ex)[:300]})
                print(f"[bench] stress leg FAILED: {ex!r}", file=sys.stderr, flush=True)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T, a.cpu_budget)   # the same workload, a bounded sample
        out["errors"] = leg_errors                      # legs that raised: a non-empty list is a non-zero exit
        print(json.dumps(out), flush=True)
    failed = bool(exchange and "error" in exchange)
    if world > 1:
        dist.destroy_process_group()
    if failed and not a.allow_exchange_failure:
        sys.exit(3)   # loud: a broken gather must not look like a green run
    if leg_errors and not a.allow_leg_failure:
        sys.exit(4)   # ... and neither must a broken leg (the configs[2] / configs[3] / stress figures come from them)


if __name__ == "__main__":
    main()
