// ddz_build_table.h -- the STRUCTURAL enumerator: legal moves of a hand generated per category
// in closed form from rank masks (chains by slot table, kicker sets by combinatorial
// unranking), ids computed from the layout of rule_based/utils/card.py:34-159.  At run time it
// is used once per device, by k_build_table, on the full deck: that yields every row of the
// action space in id order and fills the record table the scanning kernels read.  (It was the
// first run-time enumerator; the pruned dense scan over that table replaced it)
// Included by ddz_engine.hip after Out / Pick / rl().
#pragma once

// ------------------------------------------------------------------------------------
// chain slot table: lane -> (start, len) of the lane-th single / double / triple line in
// canonical order (start-major, len ascending; card.py:86-105).  byte k of entry `lane`:
// start | len << 4 for k = 0 single (36 slots), 1 double (52), 2 triple (45); 0xFF = none.
struct LineLut {
  uint32_t v[64];
};
constexpr LineLut make_line_lut() {
  LineLut t{};
  for (int i = 0; i < 64; ++i) t.v[i] = 0xFFFFFFFFu;
  const int lo[3] = {5, 3, 2}, hi[3] = {12, 10, 6};
  for (int k = 0; k < 3; ++k) {
    int slot = 0;
    for (int s = 0; s < 12; ++s)
      for (int L = lo[k]; L <= hi[k] && s + L <= 12; ++L) {
        t.v[slot] = (t.v[slot] & ~(0xFFu << (8 * k))) | ((uint32_t)(s | (L << 4)) << (8 * k));
        ++slot;
      }
  }
  return t;
}
__constant__ LineLut c_line_lut = make_line_lut();

template <bool IDS, bool PICK>
__device__ __forceinline__ int emit(bool legal, uint64_t nib, int cat, int id, const Out& o, int n, Pick& pk) {
  const uint64_t b = __ballot(legal);
  const int pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
  const uint4 row = unpack_row(nib, (uint32_t)cat);
  if (legal) {
    const int64_t pos = o.base + n + pre;
    if (pos < o.cap) {
      o.rows[pos] = row;
      if (IDS) o.ids[pos] = id;
    }
  }
  const int k = __popcll(b);
  if (PICK) {
    const int w = pk.want - n;
    if (w >= 0 && w < k) {  // wave-uniform
      const int src = __builtin_ctzll(__ballot(legal && pre == w));
      pk.r0 = rl(row.x, src); pk.r1 = rl(row.y, src); pk.r2 = rl(row.z, src); pk.r3 = rl(row.w, src);
    }
  }
  return n + k;
}

__device__ __forceinline__ int binom_sel(int a, int b) {  // a uniform, b per lane (-1..4)
  const int c2 = a * (a - 1) / 2, c3 = c2 * (a - 2) / 3, c4 = c3 * (a - 3) / 4;
  int r = b == 0 ? 1 : b == 1 ? a : b == 2 ? c2 : b == 3 ? c3 : b == 4 ? c4 : 0;
  return a < b ? 0 : r;
}

// every L-subset K of the ranks in `av`, in lexicographic order, as
// row = mainnib + mult * K; id = idbase + lex rank of K among the L-subsets of `rm`
// (itertools.combinations(remains, L), card.py:115,128,141,152).
template <bool IDS, bool PICK>
__device__ int emit_combos(uint32_t av, uint32_t rm, int L, uint64_t mainnib, int mult, int cat, int idbase,
                           bool skipj, int lane, const Out& o, int n, Pick& pk) {
  const int m = __builtin_popcount(av);
  const int total = binom(m, L);
  for (int i0 = 0; i0 < total; i0 += 64) {
    const int i = i0 + lane;
    const bool act = i < total;
    int rem = act ? i : 0, need = L, lex = 0;
    uint32_t K = 0;
    int avail_left = m, rem_left = __builtin_popcount(rm);
    for (uint32_t rr = rm; rr; rr &= rr - 1) {  // wave-uniform loop over the remains
      const int r = __builtin_ctz(rr);
      --rem_left;
      const int skip_full = binom_sel(rem_left, need - 1);
      if ((av >> r) & 1u) {
        --avail_left;
        const int c = binom_sel(avail_left, need - 1);
        const bool take = need > 0 && rem < c;
        if (take) {
          K |= 1u << r;
          --need;
        } else if (need > 0) {
          rem -= c;
          lex += skip_full;
        }
      } else if (need > 0) {
        lex += skip_full;
      }
    }
    const bool legal = act && !(skipj && L == 2 && K == JOKERS);  // card.py:116, :142
    n = emit<IDS, PICK>(legal, mainnib + (uint64_t)mult * spread15(K), cat, idbase + lex, o, n, pk);
  }
  return n;
}

// The enumerator for one table, executed by one wavefront.  `hand`, `info` are
// wave-uniform.  Returns the number of rows (uniform).  Emission order == ascending
// canonical id == index order of card.py:get_action_space().
template <bool IDS, bool PICK>
__device__ int enumerate_table(uint64_t hand, uint32_t info, uint32_t lut, int lane, const Out& o, Pick& pk) {
  if (hand == 0 || (info & (QF_FROZEN | QF_BADLAST))) return 0;
  const Follow f = follow_of(info);
  // rank masks by ballot: lane r < 15 looks at rank r
  const int cnt = lane < 15 ? (int)((hand >> (4 * (lane & 15))) & 15) : 0;
  const uint32_t m1 = (uint32_t)__ballot(cnt >= 1) & M15;
  const uint32_t m2 = (uint32_t)__ballot(cnt >= 2) & M13;
  const uint32_t m3 = (uint32_t)__ballot(cnt >= 3) & M13;
  const uint32_t m4 = (uint32_t)__ballot(cnt >= 4) & M13;
  int n = 0;
  if (!f.lead && f.lc == BIGBANG) return emit<IDS, PICK>(lane == 0, 0, EMPTY, 0, o, n, pk);  // card.py:312-313

  {  // ids 0..54: pass, singles, pairs, triples, bombs -- one lane each
    const int g = lane == 0 ? 0 : lane < 16 ? 1 : lane < 29 ? 2 : lane < 42 ? 3 : lane < 55 ? 4 : 5;
    const int r = lane - (g == 0 ? 0 : g == 1 ? 1 : g == 2 ? 16 : g == 3 ? 29 : 42);
    const uint32_t mk = g == 1   ? (m1 & value_gate(f, SINGLE))
                        : g == 2 ? (m2 & value_gate(f, DOUBLE))
                        : g == 3 ? (m3 & value_gate(f, TRIPLE))
                        : g == 4 ? (m4 & value_gate(f, QUADRIC))
                                 : 0u;
    const bool legal = g == 0 ? !f.lead : ((mk >> (r & 15)) & 1u);
    n = emit<IDS, PICK>(legal, (uint64_t)(g & 7) << (4 * (r & 15)), g, lane, o, n, pk);
  }
  if (f.lead || f.lc == THREE_ONE) {  // card.py:69-73, four mains per round
    const uint32_t mains = m3 & value_gate(f, THREE_ONE);
    for (int it = 0; it < 4; ++it) {
      if (((mains >> (4 * it)) & 15u) == 0) continue;
      const int q = lane / 15, k = lane - 15 * q, main = 4 * it + q;
      const bool legal = lane < 60 && main < 13 && ((mains >> main) & 1u) && k != main && ((m1 >> k) & 1u);
      n = emit<IDS, PICK>(legal, (3ull << (4 * (main & 15))) + (1ull << (4 * k)), THREE_ONE,
                          ID_THREE_ONE + main * 14 + (k < main ? k : k - 1), o, n, pk);
    }
  }
  if (f.lead || f.lc == THREE_TWO) {  // card.py:78-82
    const uint32_t mains = m3 & value_gate(f, THREE_TWO);
    for (int it = 0; it < 4; ++it) {
      if (((mains >> (4 * it)) & 15u) == 0) continue;
      const int q = lane / 13, k = lane - 13 * q, main = 4 * it + q;
      const bool legal = lane < 52 && main < 13 && ((mains >> main) & 1u) && k != main && ((m2 >> k) & 1u);
      n = emit<IDS, PICK>(legal, (3ull << (4 * (main & 15))) + (2ull << (4 * k)), THREE_TWO,
                          ID_THREE_TWO + main * 12 + (k < main ? k : k - 1), o, n, pk);
    }
  }
  // chains, one slot per lane (card.py:86-105)
  auto chain_round = [&](uint32_t mask, int mult, int cat, int idbase, int nslots, int byte, int minlen) {
    const uint32_t mm = mask & M12;
    if (f.lead ? run_starts(mm, minlen) == 0 : f.lc != cat) return;
    const int e = (lut >> (8 * byte)) & 0xFF, s = e & 15, L = e >> 4;
    const uint32_t full = (1u << L) - 1u;
    const bool legal = lane < nslots && ((mm >> s) & full) == full && (f.lead || (L == f.ll && s > f.lv));
    const uint64_t nib = (((uint64_t)mult * ONES) & ((1ull << (4 * L)) - 1ull)) << (4 * s);
    n = emit<IDS, PICK>(legal, nib, cat, idbase + lane, o, n, pk);
  };
  chain_round(m1, 1, SINGLE_LINE, ID_SINGLE_LINE, 36, 0, 5);
  chain_round(m2, 2, DOUBLE_LINE, ID_DOUBLE_LINE, 52, 1, 3);
  chain_round(m3, 3, TRIPLE_LINE, ID_TRIPLE_LINE, 45, 2, 2);
  // planes with kickers (card.py:110-129): walk (start, len) in canonical order
  auto planes = [&](int cat, uint32_t kick, uint32_t ranks, int hi, int mult, bool skipj, int idb) {
    if (!(f.lead || f.lc == cat) || (m3 & (m3 >> 1) & M12) == 0) return;
    for (int s = 0; s <= 10; ++s)
      for (int L = 2; L <= hi && s + L <= 12; ++L) {
        const uint32_t run = ((1u << L) - 1u) << s;
        const int R = __builtin_popcount(ranks) - L;
        if ((m3 & run) == run && (f.lead || (L == f.ll && s > f.lv)))
          n = emit_combos<IDS, PICK>(kick & ranks & ~run, ranks & ~run, L,
                                     ((3ull * ONES) & ((1ull << (4 * L)) - 1ull)) << (4 * s), mult, cat, idb,
                                     skipj, lane, o, n, pk);
        idb += binom(R, L) - ((skipj && L == 2) ? 1 : 0);
      }
  };
  planes(THREE_ONE_LINE, m1, M15, 5, 1, true, ID_THREE_ONE_LINE);
  planes(THREE_TWO_LINE, m2, M13, 4, 2, false, ID_THREE_TWO_LINE);
  // rocket (card.py:134); beats everything (card.py:314-315)
  if ((m1 & JOKERS) == JOKERS)
    n = emit<IDS, PICK>(lane == 0, (1ull << 52) | (1ull << 56), BIGBANG, ID_BIGBANG, o, n, pk);
  // four with two kickers (card.py:139-153)
  auto fours = [&](int cat, uint32_t kick, uint32_t ranks, int mult, bool skipj, int idbase, int per) {
    if (!(f.lead || f.lc == cat)) return;
    for (uint32_t qm = m4 & value_gate(f, cat); qm; qm &= qm - 1) {
      const int q = __builtin_ctz(qm);
      n = emit_combos<IDS, PICK>(kick & ranks & ~(1u << q), ranks & ~(1u << q), 2, 4ull << (4 * q), mult, cat,
                                 idbase + q * per, skipj, lane, o, n, pk);
    }
  };
  fours(FOUR_TAKE_ONE, m1, M15, 1, true, ID_FOUR_TAKE_ONE, 90);
  fours(FOUR_TAKE_TWO, m2, M13, 2, false, ID_FOUR_TAKE_TWO, 66);
  return n;
}

