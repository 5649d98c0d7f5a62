"""Diagnostic: throughput of the stateless r.get_moves drop-in (ddz_get_moves) on random (hand, last) pairs."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
if len(sys.argv) > 1:   # another engine build (A/B: tools/lib_ab_probe.py does the stepping kernels)
    importlib.import_module("doudizhu-rl_amd._lib").use_library(os.path.abspath(sys.argv[1]))
pkg = importlib.import_module("doudizhu-rl_amd")
rng = np.random.default_rng(0)
deck = np.repeat(np.arange(15), [4] * 13 + [1, 1])
rows = pkg.action_table("cuda:0").cpu().numpy()
for n in (4096, 65536, 524288):
    hands = np.zeros((n, 16), np.int8)
    base = np.stack([np.bincount(rng.choice(deck, int(rng.integers(1, 21)), replace=False), minlength=15) for _ in range(4096)])
    hands[:, :15] = base[rng.integers(0, 4096, n)]
    lasts = np.zeros((n, 16), np.int8)
    follow = rng.random(n) < 0.76
    lasts[follow, :15] = rows[rng.integers(1, 13527, int(follow.sum())), :15]
    h, l = torch.from_numpy(hands).cuda(), torch.from_numpy(lasts).cuda()
    cap = n * 64
    for _ in range(2):
        off, r, _ = pkg.get_moves(h, l, want_ids=False, row_capacity=cap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        off, r, _ = pkg.get_moves(h, l, want_ids=False, row_capacity=cap)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n:7d}: {dt * 1e6:9.1f} us per call  {n / dt / 1e6:8.1f} M queries/s  {int(off[-1]) / n:.2f} moves per query", flush=True)

    out = pkg.get_moves_slab(h, l, want_ids=False)
    for _ in range(2):
        pkg.get_moves_slab(h, l, want_ids=False, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        pkg.get_moves_slab(h, l, want_ids=False, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n:7d}: {dt * 1e6:9.1f} us per call  {n / dt / 1e6:8.1f} M queries/s  (slab layout, one launch, no host sync)", flush=True)
