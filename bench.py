#!/usr/bin/env python3
"""Headline benchmark: env steps/s of the batched lock-step engine (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
A "step" is one lock-step iteration of the hot path over all tables of this rank:
legal-move enumeration into the per-table list + random-policy action application with
auto-reset (configs[1]: 4096 tables per MI355X, random policy, legal-move list only).
Weak scaling: every GPU runs 4096 tables of the global id range and the timed region is the
same at every N (tables are independent: no data-path collective).  For N > 1 the path's one
exchange -- packed trajectories gathered to rank 0 over RCCL -- is measured right after the
headline region and reported in config.exchange.  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def cpu_baseline(tables, budget_s):
    """Oracle (CPU port of the same rules/env) on the host cores of this box: one thread, then the
    tables split over all cores this process may use (tables are independent)."""
    from oracle import oracle
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    env = oracle.OracleEnv(tables, seed=0)
    env.reset()
    t0 = time.perf_counter()
    env.rollout_random(3)
    per_iter = (time.perf_counter() - t0) / 3
    n1 = max(3, min(2000, int(0.4 * budget_s / max(per_iter, 1e-6))))
    t0 = time.perf_counter()
    plies1, _, _ = env.rollout_random(n1)
    dt1 = time.perf_counter() - t0
    nm = max(3, min(20000, int(0.6 * budget_s * cores / max(per_iter, 1e-6))))
    t0 = time.perf_counter()
    pliesm, _, _ = oracle.rollout_random_mt(env, nm, cores)
    dtm = time.perf_counter() - t0
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": pliesm / dtm, "unit": "env steps/s", "cores": cores, "kind": "port", "cpu_model": model,
            "single_core_value": plies1 / dt1,
            "sample": f"{nm} lock-step iterations x {tables} tables over {cores} threads in {dtm:.1f} s "
                      f"(and {n1} iterations on 1 thread in {dt1:.1f} s); oracle/ddz_oracle.c, dense "
                      "13,527-row scan per state"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--tables", type=int, default=4096, help="tables per GPU (configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--exchange-steps", type=int, default=500, help="N > 1: iterations of the trajectory-gather leg")
    ap.add_argument("--no-exchange", action="store_true", help="N > 1: skip the trajectory-gather leg")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on ONE GPU (all ranks on cuda:0, gloo, host-staged gather): control-flow rehearsal only")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("doudizhu-rl_amd")
    ddist = importlib.import_module("doudizhu-rl_amd.dist")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = torch.device("cuda", 0 if a.rehearse else local)
    torch.cuda.set_device(dev)
    if world > 1:
        if a.rehearse:
            dist.init_process_group("gloo")
        else:
            import datetime
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=5))

    T = a.tables
    total_tables = T * world
    _, base = ddist.shard_tables(total_tables, rank, world)
    env = pkg.BatchedEnv(T, seed=0, device=dev, table_id_base=base, want_ids=False)
    env.reset()
    K, W = a.steps, a.warmup

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # The timed region is the same at every N: K lock-step iterations of this rank's 4096 tables.  Tables
    # are independent, so there is no data-path collective while stepping (weak scaling).
    env.rollout_random(W)
    s0 = env.stats()  # cumulative counters so far (sync)
    barrier()
    t0 = time.perf_counter()
    env.rollout_random(K)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    s1 = env.stats()
    st = {k: s1[k] - s0[k] for k in s1}
    status = env.status()
    assert st["plies"] == T * K and status == 0, (st, status)

    # The path's one exchange, measured beside the headline, never inside it: every ply also
    # writes its 32-byte trajectory record and the batch goes to the learner (rank 0) over RCCL in two
    # half-batches; the gather of the first half overlaps the rollout of the second one.
    exchange = None
    if world > 1 and not a.no_exchange:
        try:
            KX = max(2, min(K, a.exchange_steps))
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
            if rank == 0:
                assert ga.shape == (half, total_tables, pkg.TRAJ_PACKED_BYTES) and gb.shape == (KX - half, total_tables, pkg.TRAJ_PACKED_BYTES)
                rec = ddist.unpack_trajectory(gb[-1].to(dev))
                assert int(rec["ply"].max()) < 200 and int(rec["role"].max()) <= 2 and int(rec["id"].max()) < 13527
            exchange = {"steps": KX, "env_steps_per_s_with_gather": total_tables * KX / dtx,
                        "bytes_to_rank0": (world - 1) * KX * T * pkg.TRAJ_PACKED_BYTES, "seconds": dtx,
                        "note": "trajectory records (32 B per ply per table) written, packed to 8 B and gathered to rank 0, pipelined "
                                "in two half-batches; measured after the headline region"}
            del traj_a, traj_b, ga, gb
            s1 = env.stats()
        except Exception as ex:  # the headline above stands on its own; report, do not lose the line
            exchange = {"error": repr(ex)[:300]}
            s1 = env.stats()
    mean_a = st["legal_rows"] / max(1, st["plies"])

    # duration of the dominant kernel: all iterations run inside ONE k_rollout launch; two HIP
    # events around that launch on the launching stream (same workload, K iterations)
    n_timed = K
    ms_launch = env.rollout_random_timed(n_timed)
    s2 = env.stats()
    steps_timed = s2["plies"] - s1["plies"]
    mean_a2 = (s2["legal_rows"] - s1["legal_rows"]) / max(1, steps_timed)
    # algorithmic bytes per env step, SURVEY.md 8(d): 128 state read + 128 state write + 4 list
    # size/offset + 16*A legal rows (DESIGN.md 3 states what the kernel really moves)
    b_step = 260 + 16 * mean_a2
    b_launch = b_step * steps_timed
    dur_launch = ms_launch * 1e-3
    dominant = "k_rollout"
    ach = b_launch / dur_launch / 1e9
    traffic = None
    issue = None
    tfile = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile):
        try:
            prof = json.load(open(tfile)).get(dominant, {})
            per_step = prof.get("hbm_bytes_per_env_step")
            traffic = per_step * steps_timed if per_step is not None else None
            # the bound this integer path really runs into: vector-ALU issue slots.  Instructions
            # per env step and the clock come from the committed PMC pass, the rate is live
            v = prof.get("valu")
            if v:
                peak = 256 * 4 * v["clock_GHz"] * 1e9 / 4  # wave-instructions/s, 4 cycles each per SIMD16
                ach_i = v["SQ_INSTS_VALU_per_env_step"] * steps_timed / dur_launch
                issue = {"bound": "valu-issue", "achieved": ach_i / 1e9, "peak": peak / 1e9,
                         "unit": "G wave-instr/s", "frac": ach_i / peak,
                         "valu_insts_per_env_step": v["SQ_INSTS_VALU_per_env_step"]}
        except Exception:
            traffic = None

    # the same loop with packed CSR lists (one launch per iteration): reported, not the headline
    n_csr = min(K, 500)
    env.rollout_random_csr(20)
    torch.cuda.synchronize(dev)
    tc = time.perf_counter()
    env.rollout_random_csr(n_csr)
    torch.cuda.synchronize(dev)
    csr_rate = T * n_csr / (time.perf_counter() - tc)

    if rank == 0:
        out = {
            "metric": "env steps/sec (batched tables)", "value": total_tables * K / dt,
            "unit": "env steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{T} tables per GPU, random policy (engine RNG), legal-move list "
                                   "only (no NN), auto-reset; BASELINE.json configs[1]",
                       "tables_per_gpu": T, "total_tables": total_tables,
                       "mean_legal_moves": round(mean_a, 3), "episodes": st["episodes"],
                       "list_layout": "slab (fixed-stride segment per table); packed-CSR variant of the "
                                      "same loop: %.4g env steps/s per GPU" % csr_rate,
                       "exchange": exchange},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "kernel": dominant,
                         "launch_us": dur_launch * 1e6, "env_steps_per_launch": steps_timed,
                         "us_per_iteration": dur_launch * 1e6 / n_timed,
                         "algorithmic_bytes_per_env_step": b_step,
                         "algorithmic_bytes_per_launch": b_launch, "issue": issue},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T, a.cpu_budget)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
