#!/usr/bin/env python3
"""Static opcode histogram of the product kernels' gfx950 code objects -> profiles/<tag>_opcode_mix.json.

bench.py's issue roofline weights the measured VALU issue costs (tools/valu_issue_probe.hip: ~2.3 cycles per plain 32-bit
op, ~4.3 for the half-rate class, ~6.1 for v_readlane) by a kernel's opcode mix.  The hardware has no per-class dynamic
counter, so the mix is the static one of the kernel's code object, taken here (build container, no GPU needed):
    llvm-objdump --offloading -> llvm-objdump -d -> per kernel: VALU instructions by class, SALU, branches, LDS, VMEM
half-rate class: 64-bit shifts / ashr / lshr, v_mul_lo/hi, v_mbcnt, v_cmp* / v_cmpx*, carry chains (v_add_co, v_addc_co,
v_sub_co, v_subb_co, v_subrev_co...), v_bfe, v_alignbit, v_mad_u64/i64, f64 arithmetic; readlane class: v_readlane,
v_readfirstlane, v_writelane; everything else VALU = full rate.
  python tools/opcode_mix.py [tag]        (default tag r04)"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "doudizhu-rl_amd", "csrc", "libddz_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNELS = {  # short name -> substring of the mangled kernel name
    # k_rollout<IDS, TRAJ, STAGED, waves per block>: the two-argument keys are the 12-wave blocks of the headline (65,536 tables)
    "k_rollout<false,false>": "9k_rolloutILb0ELb0ELb0ELi12EE", "k_rollout<false,true>": "9k_rolloutILb0ELb1ELb0ELi12EE",
    "k_rollout<false,false,16 waves>": "9k_rolloutILb0ELb0ELb0ELi16EE",
    # k_slab<MODE, IDS, COOP>: the many-tables-per-wave form (COOP = false: every batch above 4096 tables) keeps its two-argument key
    "k_slab<0,true>": "6k_slabILi0ELb1ELb0EE", "k_slab<1,true>": "6k_slabILi1ELb1ELb0EE", "k_slab<3,true>": "6k_slabILi3ELb1ELb0EE",
    "k_slab<4,true>": "6k_slabILi4ELb1ELb0EE", "k_slab<0,true,coop>": "6k_slabILi0ELb1ELb1EE", "k_auto2<true>": "7k_auto2ILb1EE", "k_table<3,0,false>": "7k_tableILi3ELi0ELb0EE",
    "k_moves_slab<true>": "12k_moves_slabILb1EE", "k_mask": "6k_maskE",
}
HALF = re.compile(r"^v_(lshlrev_b64|lshrrev_b64|ashrrev_i64|mul_lo_|mul_hi_|mbcnt|cmp|cmpx|add_co|addc_co|sub_co|subb_co|"
                  r"subrev_co|subbrev_co|bfe_|alignbit|alignbyte|mad_u64|mad_i64|.*_f64)")
LANE = re.compile(r"^v_(readlane|readfirstlane|writelane)")


def disassemble():
    tmp = tempfile.mkdtemp(prefix="ddz_co_")
    import glob
    import shutil
    so = os.path.join(tmp, "lib.so")
    shutil.copyfile(LIB, so)   # (llvm-objdump --offloading writes the bundles next to its input: keep them out of the tree)
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], stdout=subprocess.DEVNULL)
    co = glob.glob(so + ".*gfx950*")[0]
    text = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True)
    shutil.rmtree(tmp)
    return text


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    text = disassemble()
    out = {"method": __doc__.split("\n\n")[1].replace("\n", " "), "kernels": {}}
    cur = None
    hist = collections.defaultdict(collections.Counter)
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = next((k for k, sub in KERNELS.items() if sub in m.group(1)), None)
            continue
        if cur is None:
            continue
        tok = line.strip().split()
        if not tok or "//" not in line:
            continue
        hist[cur][tok[0]] += 1
    for k, h in hist.items():
        valu = {op: n for op, n in h.items() if op.startswith("v_") and not op.startswith("v_mfma")}
        nv = sum(valu.values())
        half = sum(n for op, n in valu.items() if HALF.match(op))
        lane = sum(n for op, n in valu.items() if LANE.match(op))
        salu = sum(n for op, n in h.items() if op.startswith("s_") and not op.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_load", "s_buffer_load", "s_endpgm", "s_barrier", "s_sleep")))
        br = sum(n for op, n in h.items() if op.startswith(("s_cbranch", "s_branch")))
        out["kernels"][k] = {"valu": nv, "share_half_rate": round(half / nv, 4), "share_readlane": round(lane / nv, 4),
                             "share_full_rate": round(1 - (half + lane) / nv, 4), "salu": salu, "branch": br,
                             "lds": sum(n for op, n in h.items() if op.startswith("ds_")),
                             "vmem": sum(n for op, n in h.items() if op.startswith(("global_", "buffer_", "flat_"))),
                             "top_valu": dict(collections.Counter(valu).most_common(12))}
    path = os.path.join(ROOT, "profiles", f"{tag}_opcode_mix.json")
    json.dump(out, open(path, "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:24s} VALU {v['valu']:6d}  half {v['share_half_rate']:.3f}  readlane {v['share_readlane']:.3f}  SALU {v['salu']:6d}  branch {v['branch']:5d}")
    print("wrote", path)


if __name__ == "__main__":
    main()
