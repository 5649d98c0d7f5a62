# experiments on k_slab: (a) state rows of the wave's tables staged in LDS by the prologue (no global load, hence no
# vmcnt wait behind the stores, inside the table loop); (b) done / reward / illegal / counts collected per lane and
# stored once per wave after the loop
LDS_STATE = [
 ('''  __shared__ uint4 s_face[MODE == STEP_Q ? WPB : 1][12];
  const int lane = threadIdx.x & 63;''',
  '''  __shared__ uint4 s_face[MODE == STEP_Q ? WPB : 1][12];
  __shared__ uint4 s_state[WPB][16 * DDZ_NFIELDS];  // the state rows of (up to) 16 tables of the wave
  const int lane = threadIdx.x & 63;'''),
 ('''  uint4 Rnext = make_uint4(0, 0, 0, 0);
  if (ntab > 0 && lane < DDZ_NFIELDS) Rnext = ((const uint4*)(a.state + t0 * STATE_ROW_BYTES))[lane];
  int cnt_l = 0;          // lane i: size of the current list of table t0 + i''',
  '''  uint4 Rnext = make_uint4(0, 0, 0, 0);
  const bool staged = a.tpw <= 16;  // wave-uniform
  uint4 st0 = make_uint4(0, 0, 0, 0), st1 = st0, st2 = st0;
  if (staged) {  // the wave's tables are 176 * ntab contiguous bytes: three coalesced rounds of 16-byte loads
    const uint4* src = (const uint4*)(a.state + t0 * STATE_ROW_BYTES);
    const int nrow = ntab * DDZ_NFIELDS;
    if (lane < nrow) st0 = src[lane];
    if (lane + 64 < nrow) st1 = src[lane + 64];
    if (lane + 128 < nrow) st2 = src[lane + 128];
  } else if (ntab > 0 && lane < DDZ_NFIELDS) {
    Rnext = ((const uint4*)(a.state + t0 * STATE_ROW_BYTES))[lane];
  }
  int cnt_l = 0;          // lane i: size of the current list of table t0 + i'''),
 ('''  hot_fill<TB>(hot);
  __syncthreads();
  TACC(0);
  // ... and packs / classifies it (lane-parallel: the scalar unit is the bottleneck of this kernel)''',
  '''  if (staged) {
    const int nrow = ntab * DDZ_NFIELDS;
    if (lane < nrow) s_state[wv][lane] = st0;
    if (lane + 64 < nrow) s_state[wv][lane + 64] = st1;
    if (lane + 128 < nrow) s_state[wv][lane + 128] = st2;
  }
  hot_fill<TB>(hot);
  __syncthreads();
  TACC(0);
  // ... and packs / classifies it (lane-parallel: the scalar unit is the bottleneck of this kernel)'''),
 ('''    uint4 R = Rnext;  // lane f < 11 holds row f of the table
    if (i + 1 < ntab && lane < DDZ_NFIELDS) Rnext = ((const uint4*)(a.state + (t + 1) * STATE_ROW_BYTES))[lane];
    // ---- decode, lane-parallel: every lane packs and classifies its own row''',
  '''    uint4 R = Rnext;  // lane f < 11 holds row f of the table
    if (staged) {
      R = lane < DDZ_NFIELDS ? s_state[wv][i * DDZ_NFIELDS + lane] : make_uint4(0, 0, 0, 0);
    } else if (i + 1 < ntab && lane < DDZ_NFIELDS) {
      Rnext = ((const uint4*)(a.state + (t + 1) * STATE_ROW_BYTES))[lane];
    }
    // ---- decode, lane-parallel: every lane packs and classifies its own row'''),
]
BATCH_OUT = [
 ('''  int s_ply = 0, s_eps = 0, s_lord = 0, s_up = 0;
  int64_t s_rows = 0;
  for (int i = 0; i < ntab; ++i) {
    const int64_t t = t0 + i;
    uint4* trow = (uint4*)(a.state + t * STATE_ROW_BYTES);
    uint4 R = Rnext;  // lane f < 11 holds row f of the table''',
  '''  int s_ply = 0, s_eps = 0, s_lord = 0, s_up = 0;
  int64_t s_rows = 0;
  uint32_t out_l = 0;  // lane i: done | reward << 8 | illegal << 16 of table t0 + i; n_l: the size of its new list
  int n_l = 0;
  for (int i = 0; i < ntab; ++i) {
    const int64_t t = t0 + i;
    uint4* trow = (uint4*)(a.state + t * STATE_ROW_BYTES);
    uint4 R = Rnext;  // lane f < 11 holds row f of the table'''),
 ('''    if (lane == 0) {
      if (a.done) a.done[t] = (uint8_t)o_done;
      if (a.reward) a.reward[t] = (int8_t)o_reward;
      if (a.illegal) a.illegal[t] = (uint8_t)o_illegal;
    }
    if (a.traj && lane < 2) a.traj[2 * t + lane] = sel4(lane == 0, tr0, tr1);
    if (changed && lane < DDZ_NFIELDS) trow[lane] = R;  // one coalesced 176-byte store
    TACC(3);''',
  '''    if (lane == i) out_l = (o_done & 0xFF) | ((o_reward & 0xFF) << 8) | ((o_illegal & 0xFF) << 16);
    if (a.traj && lane < 2) a.traj[2 * t + lane] = sel4(lane == 0, tr0, tr1);
    if (changed && lane < DDZ_NFIELDS) trow[lane] = R;  // one coalesced 176-byte store
    TACC(3);'''),
 ('''    if (lane == 0) a.counts[t] = n;
    s_rows += n;
    if (MODE == STEP_Q && a.face) {''',
  '''    if (lane == i) n_l = n;
    s_rows += n;
    if (MODE == STEP_Q && a.face) {'''),
 ('''#ifdef DDZ_STAMP
  if (g_stamps && lane == 0 && ntab > 0) {
    tacc[5] = ntab;
    for (int q = 0; q < 8; ++q) g_stamps[16 * t0 + q] = tacc[q];
  }
#endif
  if (ntab > 0 && lane == 0) {  // each wave owns its statistics slot (as in k_rollout)
    int64_t* ws = a.wave_stats + 4 * wave;
    ws[0] += s_ply; ws[1] += s_eps; ws[2] += (int64_t)s_lord | ((int64_t)s_up << 32); ws[3] += s_rows;
  }
}''',
  '''#ifdef DDZ_STAMP
  if (g_stamps && lane == 0 && ntab > 0) {
    tacc[5] = ntab;
    for (int q = 0; q < 8; ++q) g_stamps[16 * t0 + q] = tacc[q];
  }
#endif
  if (lane < ntab) {  // the per-table outputs of the wave's tables: consecutive addresses, one store each
    a.counts[t0 + lane] = n_l;
    if (a.done) a.done[t0 + lane] = (uint8_t)out_l;
    if (a.reward) a.reward[t0 + lane] = (int8_t)(out_l >> 8);
    if (a.illegal) a.illegal[t0 + lane] = (uint8_t)(out_l >> 16);
  }
  if (ntab > 0 && lane == 0) {  // each wave owns its statistics slot (as in k_rollout)
    int64_t* ws = a.wave_stats + 4 * wave;
    ws[0] += s_ply; ws[1] += s_eps; ws[2] += (int64_t)s_lord | ((int64_t)s_up << 32); ws[3] += s_rows;
  }
}'''),
]
VARIANTS = {"base": [], "lds_state": LDS_STATE, "batch_out": BATCH_OUT, "both": LDS_STATE + BATCH_OUT}
