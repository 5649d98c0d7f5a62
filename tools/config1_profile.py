"""Where the N = 1 `Env` view spends a ply (host side): cProfile of the loop of tools/config1_probe.py."""
import cProfile, importlib, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
envi = importlib.import_module("doudizhu-rl_amd.envi")
e = envi.EnvCooperationSimplify(seed=0)
e.reset(); e.prepare()


def loop(n):
    for _ in range(n):
        f = e.face
        a = e.valid_actions()
        _, done, _ = e.step_random()
        if done:
            e.reset(); e.prepare()


loop(200)
torch.cuda.synchronize()
t0 = time.perf_counter(); loop(2000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{2000 / dt:.0f} steps/s")
pr = cProfile.Profile(); pr.enable(); loop(2000); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
