# timing-only experiments on k_slab (results are wrong for callers that use the removed outputs): what do the trajectory
# record and the per-table outputs cost?
NOTRAJ = [
 ('''    if (a.traj && lane < 2) a.traj[2 * t + lane] = sel4(lane == 0, tr0, tr1);
    if (changed && lane < DDZ_NFIELDS) trow[lane] = R;  // one coalesced 176-byte store
    TACC(3);''',
  '''    if (changed && lane < DDZ_NFIELDS) trow[lane] = R;  // one coalesced 176-byte store
    TACC(3);'''),
]
NOOUT = NOTRAJ + [
 ('''    if (lane == i) out_l = (o_done & 0xFF) | ((o_reward & 0xFF) << 8) | ((o_illegal & 0xFF) << 16);
''', ''),
]
VARIANTS = {"base": [], "notraj": NOTRAJ, "noout": NOOUT}
