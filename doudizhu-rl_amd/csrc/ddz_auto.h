// ddz_auto.h -- k_auto: the rule-based opponent, i.e. what Env.step_auto (envi.py:72-77) /
// rule_based/rule_play.py:16-23 reach through RuleBasedModel.choose (rule_based/utils/rule_based_model.py:43-101):
// decompose the hand into combinations of actions (Decomposer.get_combinations, rule_based/utils/decomposer.py:17-76),
// score every combination with cards_value (rule_based/utils/evaluator.py:10-47) and the round penalty, play the
// best move.  The two native functions under the decomposer (env.get_combinations_recursive / _nosplit) are absent
// from the reference; "decomposer spec v1" (oracle/ddz_auto_oracle.c) defines them:
//   <= 10 cards: every multiset of actions that sums to the hand   (lowest remaining rank first, ascending id)
//   >  10 cards: every exact cover of the hand's thermometer slots by the augmented action space
//                (card.py:534-549) = each rank is touched by ONE regular action, surplus cards leave as solo
//                singles / one solo pair (lowest uncovered slot first, ascending row index).
// Included by ddz_engine.hip after plan_scan / HotTab.
//
// Mapping: one wavefront per table (tpw consecutive tables per wave).  Per table:
//   1. the candidate rows = every action that fits the hand: the lead enumeration of plan_scan, staged in the
//      wave's LDS list (nib | category, id, value | len) exactly as k_rollout stages it;
//   2. per candidate: cards_value x 2 (integers: all values are multiples of 0.5) and the fine_mask bit
//      (decomposer.py:39-46: the move beats `last`), lane-parallel; a counting sort by the LOWEST rank of the row
//      (15 buckets, order inside a bucket = ascending id) -- a node of the search only ever needs the bucket of
//      the lowest rank still to cover;
//   3. depth-first search with an explicit stack in LDS.  Control is wave-uniform (scalar); the 64 lanes test 64
//      candidates of the bucket against the remaining hand per round (SWAR subset test, __ballot -> child mask).
//      The running (sum of values, best eligible move) travel down the stack, so a finished combination costs one
//      short double-precision evaluation (rule_based_model.py:60-86, same operation order, no FMA contraction).
// Depth-first order = the order the spec lists the combinations in, so "first maximum wins" needs no tie-break keys.
#pragma once

constexpr int AUTO_DEPTH = 24;      // a combination has at most 20 actions (20 singles)
constexpr int AUTO_NONE = 127;      // "no eligible move on the path yet"

struct AutoArgs {
  const uint8_t* state;     // STATE form: [T][11][16]
  const uint4* hands;       // QUERY form: [n][16] rows
  const uint4* lasts;       //             [n][16] rows (all-zero = lead)
  const uint32_t* info;     //             [n] u8 x 4: cards left of role 0, 1, 2 (envi.py:23), acting role
  int64_t T;
  int tpw;
  int auto_roles;           // STATE form: bit r = role r is played by the rule agent
  int32_t* ids;             // out: canonical action id, -1 = not a rule agent's turn / frozen table, DDZ_AUTO_INVALID = bad query
  int64_t* stats;           // optional: [T][2] {combinations, search nodes}
  int32_t* status;
  uint32_t* ticket;         // k_auto2: this launch's own zeroed ticket word -- tables are handed out through it
  const int32_t* order;     // k_auto2, STATE form: the tables to decide, heaviest hands first (k_auto_order), or null
  const uint32_t* order_hdr;  // ... and its header (AutoOrder: [0] queue length, [2] positions drawn one by one; device words)
  int teams;                // k_auto2: waves that find the queue empty help the searches still running in their block
  int team_first;           // k_auto2: the predicted-heaviest decisions (one per block) are searched by their whole block from the start
  double rp[24];            // round_penalty by min_oppo_cards (rule_based_model.py:57), computed on the host
};

// cards_value x 2 of an action (evaluator.py:10-47); char2val = rank index + 3, so "maxCard - 10" = index - 7
__device__ __forceinline__ int auto_val2(uint64_t nib, int cat, int val, int len) {
  const int v = val - 7;
  switch (cat) {
    case EMPTY: return 0;                                             // :20-21
    case SINGLE: return 2 * v;                                        // :22-23
    case DOUBLE: return v > 0 ? 3 * v : 2 * v;                        // :24-25  positive + 50 %
    case TRIPLE: return v > 0 ? 4 * v : 2 * v;                        // :26-27  positive + 100 %
    case QUADRIC: return 18;                                          // :28-29
    case THREE_ONE: case THREE_TWO: return v > 0 ? 3 * v : 2 * v;     // :30-33
    case SINGLE_LINE: case DOUBLE_LINE: case TRIPLE_LINE: {           // :34-35  a[-1] = highest card of the chain
      const int h = val + len - 1 - 7;
      return h > 0 ? h : 0;
    }
    case THREE_ONE_LINE: case THREE_TWO_LINE: {                       // :36-47  a[-1] = the HIGHEST KICKER (card.py:117,129)
      const uint64_t mainm = (((1ull << (4 * len)) - 1ull) << (4 * val));
      const uint32_t kick = ge_mask(nib & ~mainm, 1);
      const int hi = 31 - __builtin_clz(kick | 1u);
      int r = hi - 7 > 0 ? hi - 7 : 0;
      const int w = cat == THREE_ONE_LINE ? 2 : 3;                    // + (k - 10) per single, + 1.5 (k - 10) per pair
      for (uint32_t m = kick >> 8; m; m &= m - 1) r += w * (__builtin_ctz(m) + 1);
      return r;
    }
    case BIGBANG: return 24;                                          // :48-49
    default: return 2 * v;                                            // :50-51  four-with-two
  }
}

// CardGroup.bigger_than(action, last) (card.py:307-325) for a non-pass action against a non-empty `last`
__device__ __forceinline__ bool auto_beats(int cat, int val, int len, const Follow& f) {
  if (f.lc == BIGBANG) return false;
  if (cat == BIGBANG) return true;
  if (f.lc == QUADRIC) return cat == QUADRIC && val > f.lv;
  return cat == QUADRIC || (cat == f.lc && len == f.ll && val > f.lv);
}

struct AutoBest {
  double value;
  int move;  // -1 = None
};

// one finished combination (rule_based_model.py:60-86): n actions on the path (+ the leading pass when following)
// The score is rule_based_model.py's f64 arithmetic operation by operation: the product small_num * rp is ROUNDED before
// it is subtracted (two roundings, as Python does them).  HIP's __dmul_rn / __dsub_rn are plain operators, and under
// hipcc's default -ffp-contract=fast the backend fuses them into one fma (one rounding) wherever it likes -- a
// `#pragma clang fp contract(off)` does not stop it.  It did so at one inlined site and not at another when a neighbouring
// statement changed: the greedy descent's score then sat one ulp above the same combination's score in the search and
// the exact branch and bound pruned every combination (round 3, caught by tests/test_gpu_auto.py).  auto_rounded() is
// the barrier: the product exists, rounded, in a register before anything consumes it (a2_hopeless likewise).
__device__ __forceinline__ double auto_rounded(double x) {
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ void auto_leaf(AutoBest& b, int sum2, int cvmin, int idmin, int n, bool follow, bool pass_ok,
                                          double rp) {
  const int L = n + (follow ? 1 : 0);
  const int small_num = (L - 1) - (L >= 14 ? 1 : 0);  // :63-66: positions 1..L-1 except j == 13 (action_space[13] = '2')
  double total = (double)sum2 * 0.5;                   // :62 (exact)
  total = total - auto_rounded((double)small_num * rp);  // :67
  if (follow && pass_ok && total > b.value) {          // :70-74 (position 0 of every combination)
    b.value = total;
    b.move = 0;
  }
  if (cvmin != AUTO_NONE) {                            // :76 some action of the combination may be played
    if (n == 1) {                                      // :78-81 the hand goes in one move
      b.value = __builtin_inf();
      b.move = idmin;
    }
    // :82-86 over the positions: the largest move_value belongs to the smallest cards_value, first position on ties
    const double mv = auto_rounded(total - (double)cvmin * 0.5) + rp;  // (cvmin / 2 is exact)
    if (mv > b.value) {
      b.value = mv;
      b.move = idmin;
    }
  }
}

struct AutoStack {  // per wave; every field is written and read wave-uniformly
  uint64_t a[AUTO_DEPTH];     // remaining hand (<= 10 cards) / untouched ranks U (> 10 cards)
  uint64_t b[AUTO_DEPTH];     // pending surplus P of the touched ranks (> 10 cards)
  uint64_t mask[AUTO_DEPTH];  // children still to visit in the current 64-candidate window
  uint32_t pos[AUTO_DEPTH];   // window start | bucket end << 16 (positions in the sorted order)
  uint32_t acc[AUTO_DEPTH];   // (sum2 + 512) | (cvmin & 0xFF) << 16 | stage << 24
  uint32_t idm[AUTO_DEPTH];   // idmin
};

template <bool STATE>
__global__ __launch_bounds__(TB, 4) void k_auto(AutoArgs a) {
  __shared__ HotTabT<false> hot;
  __shared__ uint64_t s_stage[WPB][STAGE_CAP];
  __shared__ uint16_t s_svl[WPB][STAGE_CAP];
  __shared__ uint16_t s_sid[WPB][STAGE_CAP];
  __shared__ uint16_t s_ord[WPB][STAGE_CAP];
  __shared__ AutoStack s_stack[WPB];
  const int lane = threadIdx.x & 63;
  const int wv = (int)rfl(threadIdx.x >> 6);
  const int64_t t0 = ((int64_t)blockIdx.x * WPB + wv) * a.tpw;
  const int ntab = t0 < a.T ? (int)(a.T - t0 < a.tpw ? a.T - t0 : a.tpw) : 0;
  hot_fill<TB>(hot);
  __syncthreads();
  uint64_t* stage = s_stage[wv];
  uint16_t* svl = s_svl[wv];
  uint16_t* sid = s_sid[wv];
  uint16_t* ord = s_ord[wv];
  AutoStack& S = s_stack[wv];
  constexpr uint64_t H8 = 0x8888888888888888ull, NIBM = 0x0FFFFFFFFFFFFFFFull;
  for (int i = 0; i < ntab; ++i) {
    const int64_t t = t0 + i;
    // ---- the query: hand, combo to beat, cards left, acting role
    uint64_t hand;
    uint32_t linfo;
    int role, left0, left1, left2;
    bool active, invalid = false;
    if (STATE) {
      uint4 R = make_uint4(0, 0, 0, 0);
      if (lane < DDZ_NFIELDS) R = ((const uint4*)(a.state + t * STATE_ROW_BYTES))[lane];
      const uint64_t P = pack_row(R);
      const uint32_t mx = rl(R.x, DDZ_F_META), my = rl(R.y, DDZ_F_META);
      role = mx & 0xFF;
      active = ((my >> 16) & 0xFF) && !((mx >> 8) & 0xFF) && role <= 2 && ((a.auto_roles >> role) & 1);
      if (role > 2) role = 0;
      const int rm1 = role == 0 ? 2 : role - 1, rp1 = role == 2 ? 0 : role + 1;
      hand = rl64(P, DDZ_F_HAND0 + role);
      linfo = last_info(rl64(P, DDZ_F_RECENT0 + rm1), (int)(rl(R.w, DDZ_F_RECENT0 + rm1) >> 24),
                        rl64(P, DDZ_F_RECENT0 + rp1), (int)(rl(R.w, DDZ_F_RECENT0 + rp1) >> 24));
      left0 = (int)(rl(R.w, DDZ_F_HAND0) >> 24); left1 = (int)(rl(R.w, DDZ_F_HAND0 + 1) >> 24);
      left2 = (int)(rl(R.w, DDZ_F_HAND0 + 2) >> 24);
    } else {
      const uint4 hr = a.hands[t], lr = a.lasts[t];
      hand = pack_row(make_uint4(rfl(hr.x), rfl(hr.y), rfl(hr.z), rfl(hr.w)));
      linfo = classify(pack_row(make_uint4(rfl(lr.x), rfl(lr.y), rfl(lr.z), rfl(lr.w))));
      const uint32_t q = rfl(a.info[t]);
      left0 = q & 0xFF; left1 = (q >> 8) & 0xFF; left2 = (q >> 16) & 0xFF; role = (int)(q >> 24);
      active = true;
      if (linfo == INFO_INVALID || ge_mask(hand, 5) || (hand >> 60) || role > 2) {  // no combo of the action space
        if (lane == 0 && a.status) atomicOr(a.status, 4);
        active = false; invalid = true;
      }
    }
    if (active && nib_sum(hand) > 20) {  // no player ever holds more than 20 cards: the search is sized for that
      if (lane == 0 && a.status) atomicOr(a.status, 4);
      active = false; invalid = true;
    }
    if (!active || hand == 0) {
      if (lane == 0) {
        // -1 = not a rule agent's turn (DDZ_STEP_IDS: engine RNG); an invalid query is NOT that: DDZ_AUTO_INVALID is no
        // action id, so DDZ_STEP_IDS flags the table illegal instead of silently playing a random move
        a.ids[t] = invalid ? DDZ_AUTO_INVALID : -1;
        if (a.stats) { a.stats[2 * t] = 0; a.stats[2 * t + 1] = 0; }
      }
      continue;
    }
    const Follow f = follow_of(linfo);
    const bool follow = !f.lead;
    // rule_based_model.py:56-57 (the role test is the reference's own: role 0 looks at lord and down, the others at up)
    int min_opp = role == 0 ? (left1 < left2 ? left1 : left2) : left0;
    if (min_opp > 23) min_opp = 23;
    const double rp = a.rp[min_opp];
    const bool pass_ok = min_opp > 4;
    const int ncards = nib_sum(hand);
    const bool nosplit = ncards > 10;  // decomposer.py:18

    // ---- 1. candidates: every action that fits the hand (decomposer.py:19-28 valid_row_idx / :50-55 valid)
    __builtin_amdgcn_wave_barrier();
    int n;
    {
      const Out o{nullptr, nullptr, 0, 0, stage, svl, sid};
      Pick pk{-1, 0, 0, 0, 0};
      n = plan_scan<EM_STAGE, true>(hand, mk_info(EMPTY, 0, 1), hot, lane, o, pk);
    }
    __builtin_amdgcn_wave_barrier();
    if (n > STAGE_CAP) {  // cannot happen for a <= 20-card hand
      if (lane == 0) { if (a.status) atomicOr(a.status, 2); a.ids[t] = DDZ_AUTO_INVALID; }
      continue;
    }
    // ---- 2. per candidate: cards_value x 2, fine_mask bit, lowest rank; counting sort by lowest rank
    int cnt_lane = 0;  // lane r: number of candidates whose lowest rank is r
    for (int j0 = 0; j0 < n; j0 += 64) {
      const int j = j0 + lane;
      int lr = 16;
      if (j < n) {
        const uint64_t e = stage[j];
        const uint64_t nib = e & NIBM;
        const int cat = (int)(e >> 60), vl = svl[j], val = vl & 0xFF, len = vl >> 8;
        const int v2 = auto_val2(nib, cat, val, len);
        const bool el = !follow || auto_beats(cat, val, len, f);
        svl[j] = (uint16_t)((v2 & 0xFF) | (el ? 0x100 : 0));
        lr = __builtin_ctzll(nib) >> 2;
        // the rule agent works on card.py's 13,527 rows in every build: a joker-kicker extra gets a row that never fits
        if (DDZ_NATIVE_JOKER_KICKERS && sid[j] >= DDZ_NUM_ACTIONS) { stage[j] = NIBM | (e & ~NIBM); lr = 0; }
      }
#pragma unroll
      for (int r = 0; r < 15; ++r) {
        const int c = __popcll(__ballot(lr == r));
        if (lane == r) cnt_lane += c;
      }
    }
    // exclusive prefix over the 15 buckets: lane r <- first position of bucket r (lane 15 = n)
    int start_lane = wave_incl_scan(lane < 15 ? cnt_lane : 0, lane) - (lane < 15 ? cnt_lane : 0);
    int run_lane = start_lane;
    __builtin_amdgcn_wave_barrier();
    for (int j0 = 0; j0 < n; j0 += 64) {
      const int j = j0 + lane;
      const int lr = j < n ? (__builtin_ctzll(stage[j] & NIBM) >> 2) : 16;
#pragma unroll
      for (int r = 0; r < 15; ++r) {
        const uint64_t m = __ballot(lr == r);
        if (m) {  // wave-uniform
          const int base = (int)rl((uint32_t)run_lane, r);
          const int pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          if (lr == r) ord[base + pre] = (uint16_t)j;
          if (lane == r) run_lane += __popcll(m);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // fine_mask of the augmented solo rows (card.py:537-547 -> clamp_action_idx: single / pair of the rank)
    const uint32_t esingle = !follow ? M15 : (f.lc == SINGLE ? gt_mask(f.lv) : 0u);
    const uint32_t epair = !follow ? M13 : (f.lc == DOUBLE ? (gt_mask(f.lv) & M13) : 0u);

    // ---- 3. depth-first search
    AutoBest best{-__builtin_inf(), -1};
    int64_t ncombs = 0, nodes = 1;
    int d = 0;                       // level = number of actions on the path
    uint64_t A = hand, B = 0;        // this node: remaining hand / (U, P)
    uint64_t mask = 0;
    int cpos = 0, hi = 0, stage_k = 0;  // stage_k: 255 = regular node, else the next solo option of a pending node
    int sum2 = 0, cvmin = AUTO_NONE, idmin = 0;
    // set (cpos, hi, stage_k, mask) of a node from its (A, B); from_pos >= 0: first allowed position (<= 10 cards:
    // while the lowest rank stays the same the row index may not decrease)
    auto open_node = [&](int from_pos) {
      const int ul = A ? (__builtin_ctzll(A) >> 2) : 16, pl = B ? (__builtin_ctzll(B) >> 2) : 16;
      if (pl < ul) {  // the lowest uncovered slot is a surplus card of a touched rank
        stage_k = 0; mask = 0; cpos = pl; hi = 0;
        return;
      }
      stage_k = 255;
      const int bs = (int)rl((uint32_t)start_lane, ul), be = (int)rl((uint32_t)start_lane, ul + 1);
      cpos = from_pos >= 0 ? from_pos : bs;
      hi = be;
      mask = 0;
      if (cpos < hi) {
        const int p = cpos + lane;
        bool ok = false;
        if (p < hi) {
          const uint64_t nib = stage[ord[p]] & NIBM;
          ok = (((A | H8) - nib) & H8) == H8;
        }
        mask = __ballot(ok);
      }
    };
    open_node(-1);
    for (;;) {
      // ---- next child of the current node
      bool have = false;
      uint64_t cnib = 0;
      int cid = 0, cv2 = 0, cpos_child = -1;
      bool cel = false, solo = false;
      if (stage_k != 255) {
        const int p = cpos;  // the rank whose surplus leaves
        const int pend = (int)((B >> (4 * p)) & 15);
        if (stage_k == 0) {  // augmented single of the slot (index 13527 + 13 (k - 1) + p: before the pair)
          stage_k = 1; have = true; solo = true;
          cnib = 1ull << (4 * p); cid = 1 + p; cv2 = 2 * (p - 7); cel = (esingle >> p) & 1u;
        } else if (stage_k == 1 && pend == 2 && ((hand >> (4 * p)) & 15) == 4) {  // augmented pair of slots 2, 3
          stage_k = 2; have = true; solo = true;
          cnib = 2ull << (4 * p); cid = 16 + p; cv2 = p - 7 > 0 ? 3 * (p - 7) : 2 * (p - 7); cel = (epair >> p) & 1u;
        }
      } else {
        while (mask == 0 && cpos + 64 < hi) {  // next window of the bucket
          cpos += 64;
          const int p = cpos + lane;
          bool ok = false;
          if (p < hi) {
            const uint64_t nib = stage[ord[p]] & NIBM;
            ok = (((A | H8) - nib) & H8) == H8;
          }
          mask = __ballot(ok);
        }
        if (mask) {
          const int bit = __builtin_ctzll(mask);
          mask &= mask - 1;
          const int p = cpos + bit;
          const int j = ord[p];
          const uint64_t e = stage[j];
          const int q = svl[j];
          cnib = (uint64_t)rfl((uint32_t)e) | ((uint64_t)(rfl((uint32_t)(e >> 32)) & 0x0FFFFFFFu) << 32);
          cid = (int)rfl((uint32_t)sid[j]);
          cv2 = (int)(int8_t)(rfl((uint32_t)q) & 0xFF);
          cel = (rfl((uint32_t)q) >> 8) & 1u;
          cpos_child = p;
          have = true;
        }
      }
      if (!have) {  // node exhausted: back to the parent
        if (d == 0) break;
        --d;
        A = S.a[d]; B = S.b[d]; mask = S.mask[d];
        const uint32_t pp = S.pos[d], ac = S.acc[d];
        cpos = (int)(pp & 0xFFFF); hi = (int)(pp >> 16);
        sum2 = (int)(ac & 0xFFFF) - 512; cvmin = (int)(int8_t)((ac >> 16) & 0xFF); stage_k = (int)(ac >> 24);
        idmin = (int)S.idm[d];
        A = (uint64_t)rfl((uint32_t)A) | ((uint64_t)rfl((uint32_t)(A >> 32)) << 32);
        B = (uint64_t)rfl((uint32_t)B) | ((uint64_t)rfl((uint32_t)(B >> 32)) << 32);
        mask = (uint64_t)rfl((uint32_t)mask) | ((uint64_t)rfl((uint32_t)(mask >> 32)) << 32);
        cpos = (int)rfl((uint32_t)cpos); hi = (int)rfl((uint32_t)hi); sum2 = (int)rfl((uint32_t)sum2);
        cvmin = (int)rfl((uint32_t)cvmin); stage_k = (int)rfl((uint32_t)stage_k); idmin = (int)rfl((uint32_t)idmin);
        continue;
      }
      nodes += 1;
      // ---- the child's state and running sums
      uint64_t A2, B2;
      if (solo) {
        A2 = A; B2 = B - cnib;
      } else if (nosplit) {
        uint64_t tm = cnib | (cnib >> 1);
        tm |= tm >> 2;
        const uint64_t rm = (tm & ONES) * 15ull;  // 0xF on every rank the action touches
        A2 = A & ~rm;
        B2 = B + (A & rm) - cnib;                 // what the action leaves of those ranks
      } else {
        A2 = A - cnib; B2 = 0;
      }
      const int sum2c = sum2 + cv2;
      int cvminc = cvmin, idminc = idmin;
      if (cel && (cvmin == AUTO_NONE || cv2 < cvmin)) { cvminc = cv2; idminc = cid; }
      if ((A2 | B2) == 0) {  // the combination is complete
        ncombs += 1;
        auto_leaf(best, sum2c, cvminc, idminc, d + 1, follow, pass_ok, rp);
        continue;
      }
      if (d + 1 >= AUTO_DEPTH) {  // cannot happen: at most 20 actions
        if (lane == 0 && a.status) atomicOr(a.status, 16);
        continue;
      }
      // ---- descend
      if (lane == 0) {
        S.a[d] = A; S.b[d] = B; S.mask[d] = mask;
        S.pos[d] = (uint32_t)cpos | ((uint32_t)hi << 16);
        S.acc[d] = (uint32_t)((sum2 + 512) & 0xFFFF) | ((uint32_t)(cvmin & 0xFF) << 16) | ((uint32_t)stage_k << 24);
        S.idm[d] = (uint32_t)idmin;
      }
      const int same_rank_from = (!nosplit && !solo && A2 && (__builtin_ctzll(A2) >> 2) == (__builtin_ctzll(A) >> 2)) ? cpos_child : -1;
      ++d;
      A = A2; B = B2; sum2 = sum2c; cvmin = cvminc; idmin = idminc;
      open_node(same_rank_from);
    }
    if (lane == 0) {
      a.ids[t] = best.move < 0 ? 0 : best.move;  // rule_based_model.py:87-89
      if (a.stats) { a.stats[2 * t] = ncombs; a.stats[2 * t + 1] = nodes; }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// test hook: auto_leaf on its own (tests/test_gpu_auto.py checks it bit for bit against the two-rounding f64 arithmetic of
// rule_based_model.py:60-86 in numpy: the contract the exact branch and bound stands on).  in[i] = {sum2, cvmin, n, follow |
// pass_ok << 1}
__global__ __launch_bounds__(BLOCK) void k_debug_leaf(const int4* __restrict__ in, const double* __restrict__ rp, int64_t n,
                                                      double* __restrict__ value, int32_t* __restrict__ move) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const int4 q = in[i];
  AutoBest b{-__builtin_inf(), -1};
  auto_leaf(b, q.x, q.y, 7, q.z, q.w & 1, (q.w >> 1) & 1, rp[i]);
  value[i] = b.value;
  move[i] = b.move;
}

// test hook / fixture G7 on the device: cards_value x 2 of every action id
__global__ __launch_bounds__(BLOCK) void k_cards_value(int8_t* __restrict__ out) {
  const int id = (int)(blockIdx.x * BLOCK + threadIdx.x);
  if (id >= DDZ_NUM_ACTIONS) return;
  const uint4 m = g_tab[2 * id + 1];
  const uint64_t nib = (uint64_t)m.x | ((uint64_t)m.y << 32);
  out[id] = (int8_t)auto_val2(nib, (int)((m.z >> 16) & 0xFF), (int)(m.z & 0xFF), (int)((m.z >> 8) & 0xFF));
}
