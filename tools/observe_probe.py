"""Diagnostic: duration / achieved HBM rate of the state-encoding kernel k_observe (HIP events)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
pkg = importlib.import_module("doudizhu-rl_amd")
for T in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["65536", "524288"])]:
    env = pkg.BatchedEnv(T, seed=0, want_ids=False)
    env.reset()
    env.rollout_random(40)
    for variant in range(4):
        P = pkg.FACE_PLANES[variant]
        out = torch.empty((T, P, 15, 4), dtype=torch.float32, device="cuda")
        for _ in range(3):
            env.observe(variant, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            env.observe(variant, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        byts = T * (P * 240 + 176)
        print(f"T={T:7d} variant={variant} P={P}: {us:8.1f} us  {byts / us / 1e3:7.1f} GB/s  ({byts / 1e6:.1f} MB per call)", flush=True)
    del env

# the dense legal-move mask (k_mask): 424 u32 per table
for T in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["65536", "524288"])]:
    env = pkg.BatchedEnv(T, seed=0, want_ids=False)
    env.reset()
    env.rollout_random(40)
    out = torch.empty((T, 424), dtype=torch.int32, device="cuda")
    for _ in range(3):
        env.legal_mask(out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        env.legal_mask(out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    byts = T * (424 * 4 + 176)
    print(f"T={T:7d} legal_mask: {us:8.1f} us  {byts / us / 1e3:7.1f} GB/s  ({byts / 1e6:.1f} MB per call)  {T / us:8.1f} M tables/s", flush=True)
    del env
