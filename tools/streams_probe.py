#!/usr/bin/env python3
"""Probe: the tables of a policy-driven loop split into S sub-batches (global table ids kept: table_id_base), each on its
own HIP stream -- does the tail of one sub-batch's launch (k_auto2: its single longest decision; k_slab at 4096 tables: its
slowest block) overlap with the other sub-batches' launches?

  python tools/streams_probe.py [--tables 65536] [--streams 1,2,4,8] [--leg auto|slab|both]

Prints us per lock-step iteration over ALL tables and env steps/s per S.  The union of the sub-batches is the same set of
games as the single environment (RNG keyed by the global table id; tests/test_gpu_fullsize.py)."""
import argparse
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tables", type=int, default=65536)
    ap.add_argument("--streams", default="1,2,4,8")
    ap.add_argument("--leg", default="both")
    ap.add_argument("--iters", type=int, default=60)
    a = ap.parse_args()
    pkg = importlib.import_module("doudizhu-rl_amd")
    dev = torch.device("cuda:0")
    T = a.tables
    for leg in (("auto", "slab") if a.leg == "both" else (a.leg,)):
        for S in [int(x) for x in a.streams.split(",")]:
            if T % S:
                continue
            Ts = T // S
            streams = [torch.cuda.Stream(dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
            envs = []
            for s in range(S):
                with torch.cuda.stream(streams[s]):
                    e = pkg.BatchedEnv(Ts, seed=0, device=dev, table_id_base=s * Ts)
                    e.reset()
                    e.legal_slab()
                    envs.append(e)

            def iteration():
                for s in range(S):
                    with torch.cuda.stream(streams[s]):
                        e = envs[s]
                        if leg == "auto":
                            ids = e.auto_choose(0b101)
                            e.step_slab(ids, pkg.STEP_IDS, auto_reset=True)
                        else:
                            e.step_slab(None, pkg.STEP_RANDOM, auto_reset=True)

            for _ in range(40):
                iteration()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(a.iters):
                iteration()
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            st = [e.status() for e in envs]
            print(f"{leg:5s} T={T} S={S}: {dt / a.iters * 1e6:9.1f} us per iteration (host issue {t_host / a.iters * 1e6:7.1f}), "
                  f"{T * a.iters / dt / 1e6:9.1f} M env steps/s, status {max(st)}", flush=True)
            del envs


if __name__ == "__main__":
    main()
