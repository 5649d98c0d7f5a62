"""Diagnostic: -DDDZ_STAMP build, where do k_auto2's waves spend their cycles per decision (staging+sort / frontier / search)."""
import ctypes as C
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
csrc = os.path.join(ROOT, "doudizhu-rl_amd", "csrc")
out = os.path.join(ROOT, "build_variants")
os.makedirs(out, exist_ok=True)
lib = os.path.join(out, "stamp.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DDDZ_STAMP=1",
                       "-o", lib, os.path.join(csrc, "ddz_engine.hip")])
import numpy as np  # noqa: E402
import torch  # noqa: E402

importlib.import_module("doudizhu-rl_amd._lib").use_library(lib)
pkg = importlib.import_module("doudizhu-rl_amd")
raw = C.CDLL(lib)
raw.ddz_debug_set_stamps.argtypes = [C.c_void_p]
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
env = pkg.BatchedEnv(T, seed=0)
env.reset()
if os.environ.get("DDZ_STAMP_HEAVY"):   # the heaviest states of fixture G8h on the first N tables instead (tools/team_probe.py)
    from team_probe import heavy_states
    st_, _, _ = heavy_states(pkg, T, int(os.environ["DDZ_STAMP_HEAVY"]))
    env.state_import(torch.from_numpy(st_).view(-1))
else:
    env.legal_slab()
    for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):   # mid-game states of rule farmers against a random lord
        env.step_auto(0b101, slab=True)
if len(sys.argv) > 4:  # teams off
    env.lib.ddz_debug_set_auto_teams(env._h, int(sys.argv[4]))
ROLES = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0b111
buf = torch.zeros((T + 1, 16), dtype=torch.int64, device="cuda")
assert raw.ddz_debug_set_stamps(C.c_void_p(buf.data_ptr())) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ids = env.auto_choose(ROLES)
e1.record()
torch.cuda.synchronize()
assert raw.ddz_debug_set_stamps(None) == 0
dbg = buf[T].cpu().numpy()
buf = buf[:T]
print(f"  teams: opened {dbg[0]}, helper stints {dbg[1]}, items put {dbg[2]} in {dbg[6]} rounds, taken {dbg[3]} in {dbg[7]} rounds; nodes walked by helpers {dbg[4]}, by owners of teams {dbg[5]}")
if dbg[8]:
    print(f"  members' lanes at the hand-over checks: active {dbg[8]}, of which no open level {dbg[9] / dbg[8]:.0%}, below a pending pair "
          f"option {dbg[10] / dbg[8]:.0%}, below the key levels {dbg[11] / dbg[8]:.0%}, can give {dbg[12] / dbg[8]:.0%}")
s = buf.cpu().numpy().astype(np.float64)
s = s[(ids.cpu().numpy() >= 0)]
passes = s[:, 3].astype(np.int64)
s[:, 3] = passes & 0xFFFF
print(f"  frontier passes per decision: register pass {((passes >> 16) & 0xFF).mean():.2f}, general pass {((passes >> 24) & 0xFF).mean():.2f}")
print(f"T={T}: {len(s)} decisions, launch {e0.elapsed_time(e1) * 1e3:.0f} us; cycles per decision (s_memtime)")
for k, nm in enumerate(["staging + sort", "frontier passes", "search"]):
    print(f"  {nm:18s} mean {s[:, k].mean():9.0f}  p50 {np.percentile(s[:, k], 50):9.0f}  p99 {np.percentile(s[:, k], 99):9.0f}  max {s[:, k].max():9.0f}")
print(f"    staging + sort = enumeration {s[:, 13].mean():.0f} + sort {s[:, 14].mean():.0f} + bounds {s[:, 15].mean():.0f} + greedy "
      f"{(s[:, 0] - s[:, 13] - s[:, 14] - s[:, 15]).mean():.0f}")
tot = s[:, :3].sum(1)
print(f"  total              mean {tot.mean():9.0f}  p99 {np.percentile(tot, 99):9.0f}  max {tot.max():9.0f}   sum/1024 waves {tot.sum() / 1024:.0f}")
print(f"  frontier items mean {s[:, 3].mean():.0f} max {s[:, 3].max():.0f}; nodes mean {s[:, 4].mean():.0f} max {s[:, 4].max():.0f}; candidates mean {s[:, 5].mean():.0f} max {s[:, 5].max():.0f}")
print(f"  search loop: trips mean {s[:, 6].mean():.0f} max {s[:, 6].max():.0f}; cycles per trip {s[:, 2].sum() / s[:, 6].sum():.0f}; active lanes per trip {s[:, 7].sum() / s[:, 6].sum():.1f}")
names = ["take items / donations", "step (scan, descend)"]
for k, nm in enumerate(names):
    print(f"    per trip: {nm:24s} {s[:, 8 + k].sum() / s[:, 6].sum():7.0f} cycles")
# timeline: who decided what when (s_memtime is one clock for the whole chip)
wave, t_start, t_end = s[:, 10].astype(np.int64), s[:, 11], s[:, 12]
t0 = t_start.min()
span = t_end.max() - t0
nw = int(wave.max()) + 1
last_end = np.zeros(nw); first_start = np.full(nw, np.inf); busy = np.zeros(nw); ndec = np.zeros(nw)
for w, a_, b_ in zip(wave, t_start, t_end):
    last_end[w] = max(last_end[w], b_ - t0); first_start[w] = min(first_start[w], a_ - t0); busy[w] += b_ - a_; ndec[w] += 1
print(f"  timeline: span {span:.0f} cycles over {nw} waves; per wave: decisions {ndec.mean():.1f}, busy {busy.mean():.0f} "
      f"({busy.mean() / span:.0%} of the span), first start {first_start.mean():.0f}, last end mean {last_end.mean():.0f} min {last_end.min():.0f}")
own_span = np.array([last_end[w] - first_start[w] for w in range(nw) if ndec[w] > 0])
print(f"  per wave (its own clock): first start to last end mean {own_span.mean():.0f} max {own_span.max():.0f} cycles; busy {busy[ndec > 0].mean():.0f} "
      f"= {busy[ndec > 0].mean() / own_span.max():.0%} of the longest wave -- the rest is the launch's tail")
order_ = np.argsort(t_start)
dur = (t_end - t_start)[order_]
q = len(dur) // 10
print("  mean decision cycles by start-time decile:", [int(dur[i * q:(i + 1) * q].mean()) for i in range(10)])
heavy = np.argsort(-(t_end - t_start))[:5]
print("  heaviest decisions: (start, duration) =", [(int(t_start[h] - t0), int(t_end[h] - t_start[h])) for h in heavy])
gaps = []
for w in range(0, nw, 37):
    idx = np.nonzero(wave == w)[0]
    idx = idx[np.argsort(t_start[idx])]
    gaps += list(t_start[idx][1:] - t_end[idx][:-1])
if gaps:
    print(f"  gap between a wave's decisions (ticket + order + state loads): mean {np.mean(gaps):.0f} p90 {np.percentile(gaps, 90):.0f} cycles")
heavy = np.argsort(-tot)[:5]
for h in heavy:
    print("   heavy:", [int(x) for x in s[h, :8]])
