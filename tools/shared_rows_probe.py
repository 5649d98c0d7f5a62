#!/usr/bin/env python3
"""Prototype: the count-0 term H0 of the needed-rows Q forward WITHOUT the K = 3840 dense GEMM.

H0[t] = tab[t] + sum_r Y0[t, r, :] x fc1[r]  and  Y0[t, r, :] depends only on the face COLUMN of rank r -- for
EnvCooperationSimplify (envi.py:201-217) on (hand_r, taken_r, b1_r, b2_r, n1, n2) (the prob planes: DESIGN 4).  Across 65,536
tables the 983,040 (table, rank) columns take < 10 % distinct values per rank, so: one row per DISTINCT (rank, column) --
first layer + rows GEMM (G = Y x fc1[rank], k_fc1 with rank segments) over ~90 k rows instead of a 65,536 x 3840 x 256
product -- and H0[t] = tab[t] + sum_r G[row(t, r)] (a gather-sum of fifteen 1-KB rows per table).

This probe builds that from torch ops + the existing kernels, checks H0 against FactorisedQ.needed and times the stages.
  python tools/shared_rows_probe.py [--tables 65536] [--iters 40]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KR = 625 * 441          # keys per rank: 5^4 columns x 21 x 21 (n1, n2)
KEYS = 15 * KR


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tables", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=40)
    a = ap.parse_args()
    pkg = importlib.import_module("doudizhu-rl_amd")
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    E = importlib.import_module("doudizhu-rl_amd.engine")
    dev = torch.device("cuda:0")
    T = a.tables
    env = pkg.BatchedEnv(T, seed=0, device=dev)
    env.reset()
    env.legal_slab()
    torch.manual_seed(0)
    net = glue.QNet(6).to(dev).eval()
    loop = glue.PolicyLoop(env, net, face_variant=3, epsilon=0.0)
    loop.run(a.iters)                     # mid-game states under the greedy network
    fq, face, P = loop.fq, loop.face, 6
    ref = fq.needed(env, face)
    h0_ref = ref.h0.clone()
    torch.cuda.synchronize()

    FC = 128
    cap = 15 * FC * 86                    # 165,120 rows: a multiple of the GEMM tile and of 15 (fake tables of 15 columns)
    Tf = cap // 15
    ar = torch.arange(T, device=dev)
    rr = torch.arange(15, device=dev)[None, :]
    inst = torch.arange(T * 15, device=dev, dtype=torch.int32) + 1
    neg = torch.full((Tf, 64), -1, dtype=torch.int32, device=dev)
    y0f = torch.zeros((Tf, 15 * 256), dtype=torch.float32, device=dev)
    dyd = torch.zeros((FC, 256), dtype=torch.float32, device=dev)
    G = torch.zeros((cap + FC, 256), dtype=torch.float32, device=dev)
    rc0 = torch.zeros(cap + FC, dtype=torch.uint8, device=dev)
    z0 = torch.zeros((15, 5, 256), dtype=torch.float32, device=dev)
    h0 = torch.zeros((T, 256), dtype=torch.float32, device=dev)
    seg = torch.zeros(40, dtype=torch.int32, device=dev)
    st = {}

    def keys():
        s = env.state.view(T, 11, 16)
        role = s[:, 10, 0].long()
        hand = s[ar, role, :15].long()
        taken = s[:, 9, :15].long()
        b1 = s[ar, 6 + (role + 2) % 3, :15].long()
        b2 = s[ar, 6 + (role + 1) % 3, :15].long()
        n1 = s[ar, (role + 1) % 3, 15].long()
        n2 = s[ar, (role + 2) % 3, 15].long()
        st["key"] = ((((rr * 5 + hand) * 5 + taken) * 5 + b1) * 5 + b2) * 441 + (n1 * 21 + n2)[:, None]

    def dedupe():
        flat = st["key"].view(-1)
        rep = torch.zeros(KEYS, dtype=torch.int32, device=dev).scatter_reduce_(0, flat, inst, "amax")
        present = rep > 0
        pv = present.view(15, KR)
        cnt = pv.sum(1)
        segrows = (cnt + FC - 1) // FC * FC
        start = torch.cumsum(segrows, 0) - segrows
        rowid = start[:, None] + torch.cumsum(pv, 1) - 1
        slot = torch.where(pv, rowid, torch.full_like(rowid, cap)).view(-1)            # absent keys -> the dummy row
        st["rows"] = slot[flat].view(T, 15)
        st["row2inst"] = torch.zeros(cap + 1, dtype=torch.int64, device=dev).scatter_(0, slot.clamp(max=cap), (rep - 1).clamp(min=0).long())
        seg[0:15] = start.to(torch.int32)
        seg[15] = segrows.sum().to(torch.int32)
        seg[16:31] = (start // FC).to(torch.int32)
        seg[31] = (segrows.sum() // FC).to(torch.int32)
        seg[32] = cnt.sum().to(torch.int32)
        st["n"] = cnt.sum()

    def first_layer():
        cols = face.permute(0, 2, 1, 3).reshape(T * 15, P, 4)[st["row2inst"][:cap]]    # [cap, P, 4]
        fake = cols.view(Tf, 15, P, 4).permute(0, 2, 1, 3).contiguous()
        E.q_features_needed(fake, fq.Wf, fq.bias_f, fq.A, neg, y0f, dyd)

    def gemm():
        E.q_fc1_rows(y0f.view(cap, 256), seg, rc0, fq.W2, z0, G[:cap])

    def table_term():
        torch.addmm(fq.base, face.view(T, P * 60), fq.Mz_f, out=h0)

    def gather():
        h0.add_(torch.nn.functional.embedding_bag(st["rows"], G, mode="sum"))

    stages = (("keys", keys), ("dedupe", dedupe), ("first_layer", first_layer), ("rows_gemm", gemm), ("table_term", table_term),
              ("gather_sum", gather))
    for _, f in stages:
        f()
    torch.cuda.synchronize()
    n = int(st["n"])
    err = (h0 - h0_ref).abs().max().item()
    print(f"T={T}: {T * 15} (table, rank) columns, {n} distinct = {n / (T * 15):.3f}; rows incl. padding {int(seg[15])} of cap {cap}; "
          f"max |H0 - H0_ref| = {err:.3e} (|H0_ref| max {h0_ref.abs().max().item():.3f})", flush=True)
    tot = 0.0
    for name, f in stages:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(); f(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        tot += best
        print(f"  {name:12s} {best:8.1f} us", flush=True)
    print(f"  total        {tot:8.1f} us   (the path it replaces: first layer y0 part + dense GEMM 860 us)")
    # the first layer of the product with only the needed rows left would be the other saving; time the existing stages
    w = fq._ws[("needed", face.device, T)]
    for name, f in (("features(old)", lambda: E.q_features_needed(face, fq.Wf, fq.bias_f, fq.A, w["row_index"], w["y0"], w["dy"])),
                    ("dense(old)", lambda: w["h0"].addmm_(w["y0"], fq.Wd))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(); f(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        print(f"  {name:14s} {best:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
