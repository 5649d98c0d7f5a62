"""CPU, world_size 2 (gloo): the sharding + end-of-batch trajectory gather of
doudizhu-rl_amd/dist.py.  Per-rank trajectories come from the oracle env (the GPU engine
is bit-identical to it, tests/test_gpu_parity.py), so this also checks that sharding by
global table id reproduces the single-process run exactly."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ITERS = 40
TOTAL = 37  # ragged on purpose: 19 + 18 tables


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rollout(n, base, seed):
    from oracle import oracle
    env = oracle.OracleEnv(n, seed=seed, gid_base=base)
    env.reset()
    out = np.zeros((ITERS, n, 32), np.uint8)
    for it in range(ITERS):
        env.legal()
        _, _, _, traj = env.step(oracle.STEP_RANDOM, auto_reset=True, want_traj=True)
        out[it] = traj
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ddist = importlib.import_module("doudizhu-rl_amd.dist")
        n, base = ddist.shard_tables(TOTAL, rank, world)
        local = torch.from_numpy(_rollout(n, base, seed=99))
        full = ddist.gather_trajectories(local)
        # gather to the learner rank, asynchronously, two half-batches in flight (bench.py's use)
        half = ITERS // 2
        h1 = ddist.gather_trajectories(local[:half].contiguous(), dst=0, async_op=True)
        h2 = ddist.gather_trajectories(local[half:].contiguous(), dst=0, async_op=True)
        a, b = h1.result(), h2.result()
        to0 = None if a is None else torch.cat([a, b], dim=0).numpy()
        q.put((rank, n, base, full.numpy(), to0))
    finally:
        dist.destroy_process_group()


def test_shard_tables_partition():
    ddist = importlib.import_module("doudizhu-rl_amd.dist")
    for total, world in [(37, 2), (4096, 8), (5, 8), (524288, 8)]:
        parts = [ddist.shard_tables(total, r, world) for r in range(world)]
        assert sum(n for n, _ in parts) == total
        assert parts[0][1] == 0
        for (n0, b0), (n1, b1) in zip(parts, parts[1:]):
            assert b1 == b0 + n0 and n0 >= n1 >= n0 - 1
    assert ddist.shard_tables(524288, 3, 8) == (65536, 3 * 65536)   # BASELINE configs[4]
    with pytest.raises(ValueError):
        ddist.shard_tables(8, 8, 8)


def test_gather_trajectories_world2_matches_single_process():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(19, 0), (18, 19)]
    single = _rollout(TOTAL, 0, seed=99)
    for rank, _, _, full, to0 in res:              # every rank holds the full batch, global id order
        assert full.shape == (ITERS, TOTAL, 32)
        assert np.array_equal(full, single)
        if rank == 0:
            assert np.array_equal(to0, single)     # dst=0: only the learner rank receives
        else:
            assert to0 is None


BIG_BASE = (1 << 32) + 12345   # global table ids beyond 32 bits: the RNG counter takes both halves


def _worker8(rank, world, port, q, total, base0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ddist = importlib.import_module("doudizhu-rl_amd.dist")
        n, base = ddist.shard_tables(total, rank, world)
        local = torch.from_numpy(_rollout(n, base0 + base, seed=7))
        to0 = ddist.gather_trajectories(local, dst=0)                       # the bench's exchange: gather to the learner
        sizes = [ddist.shard_tables(total, r, world)[0] for r in range(world)]
        to0b = ddist.gather_trajectories(local, dst=0, shard_sizes=sizes)   # ... with the sizes known (no size exchange)
        allr = ddist.gather_trajectories(local)                             # all_gather form
        ok = (to0 is None) == (rank != 0) and (to0b is None) == (rank != 0)
        if rank == 0:
            ok = ok and torch.equal(to0, to0b) and torch.equal(to0, allr)
        x = allr.contiguous().view(torch.int64).view(-1)
        digest = int((x * 31 + (x >> 13) + torch.arange(x.numel()) * x).sum().item())     # bench.py's digest
        q.put((rank, n, base, ok, digest, allr.numpy() if rank == 0 else None))
    finally:
        dist.destroy_process_group()


def test_gather_trajectories_world8_ragged_ids_beyond_32_bits():
    """configs[4]'s id arithmetic at eight ranks (gloo, CPU): 8 ragged shards (61 tables -> 8,8,8,8,8,7,7,7) of a batch
    whose global table ids start beyond 2^32; gather to rank 0 (with and without known sizes) and all_gather give the
    single-process rollout of the union, same digest on every rank.  (524,288 tables -> 65,536 per rank, bases r * 65,536:
    test_shard_tables_partition.)"""
    world, total = 8, 61
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q, total, BIG_BASE)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [8, 8, 8, 8, 8, 7, 7, 7] and [r[2] for r in res] == [0, 8, 16, 24, 32, 40, 47, 54]
    assert all(r[3] for r in res)
    single = _rollout(total, BIG_BASE, seed=7)
    assert np.array_equal(res[0][5], single)
    x = torch.from_numpy(single).contiguous().view(torch.int64).view(-1)
    want = int((x * 31 + (x >> 13) + torch.arange(x.numel()) * x).sum().item())
    assert all(r[4] == want for r in res)
    assert not np.array_equal(single, _rollout(total, 12345, seed=7))       # the high half of the id reaches the RNG


def test_unpack_trajectory_fields():
    ddist = importlib.import_module("doudizhu-rl_amd.dist")
    traj = torch.from_numpy(_rollout(8, 0, seed=5))
    f = ddist.unpack_trajectory(traj)
    assert f["row"].shape == (ITERS, 8, 16) and f["row"].dtype == torch.int8
    assert set(f["role"].unique().tolist()) <= {0, 1, 2}
    assert (f["role"][0] == 1).all()                       # lord leads (game.py:173)
    assert set(f["reward"].unique().tolist()) <= {-1, 0, 1}
    assert ((f["reward"] != 0) == (f["done"] == 1)).all()
    assert (f["choice"] >= 0).all() and (f["choice"] < f["n_legal"]).all()
    assert (f["ply"][0] == 0).all() and (f["episode"][0] == 0).all()
    assert (f["row"][..., 15] >= 0).all() and (f["row"][..., 15] <= 14).all()   # category byte


def test_gather_without_process_group_is_identity():
    ddist = importlib.import_module("doudizhu-rl_amd.dist")
    t = torch.zeros((3, 4, 32), dtype=torch.uint8)
    assert ddist.gather_trajectories(t) is t
    with pytest.raises(ValueError):
        ddist.gather_trajectories(torch.zeros((3, 4, 31), dtype=torch.uint8))


def test_bench_self_spawn_is_loud_without_gpus():
    """`python bench.py --gpus 2` as the driver types it (no launcher, no WORLD_SIZE): bench.py starts its ranks itself
    (spawn_ranks -> torch.distributed.run children).  In this container there is no GPU, so the ranks must fail -- and
    the launcher must pass that on as a non-zero exit and print no JSON line (never a green-looking run)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the rehearsal test in test_gpu_parity.py covers the launcher")
    env_ = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--rehearse", "--tables", "64",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env_)
    assert p.returncode != 0
    assert "torch.distributed.run" in p.stderr and "--nproc-per-node 2" in p.stderr   # the launcher announced itself
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
