/*
 * ddz_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See ddz_oracle.h for the parity status of every function.
 *
 * Style: deliberately the *dense* formulation the reference's Python uses
 * (scan all 13,527 rows, subset test, bigger_than) so that it is an
 * independent check of the structural enumerator in the HIP kernels.
 * Citations are relative to /root/reference.
 */
#include "ddz_oracle.h"
#include <stdlib.h>
#include <string.h>

#ifndef DDZO_NATIVE_JOKER_KICKERS
#define DDZO_NATIVE_JOKER_KICKERS 0
#endif
/* Optional rule-set extension (libddz_oracle_jk.so, default OFF): the 13 "quad + both jokers" and 11
 * "two consecutive triples + both jokers" vectors that card.py:116,142 exclude but the reference's
 * native get_moves is known to emit -- server/mcts/get_moves.py:22-34 builds exactly these 24
 * vectors to filter them out of its result.  They get the ids 13527..13539 (by quad rank) and
 * 13540..13550 (by start rank), i.e. they come last in every list.  Unverifiable beyond that.    */
#define NA_BASE DDZO_NUM_ACTIONS
#define NA (DDZO_NUM_ACTIONS + 24 * DDZO_NATIVE_JOKER_KICKERS)
#define NR DDZO_NUM_RANKS

static int8_t g_rows[NA][DDZO_ROW]; /* counts + category in byte 15 */
static uint8_t g_info[NA][4];       /* cat, value, len, ncards */
static uint64_t g_pk[NA];           /* nibble-packed counts: rank i at bits 4i */
static int32_t g_sorted[NA];        /* ids sorted by g_pk, for lookup */
static int g_ready = 0;
static int g_n = 0;

static uint64_t pack15(const int8_t* c) {
  uint64_t x = 0;
  for (int i = 0; i < NR; ++i) x |= (uint64_t)(c[i] & 15) << (4 * i);
  return x;
}

static void push(const int* cnt, int cat, int value, int len) {
  int n = 0;
  for (int i = 0; i < NR; ++i) {
    g_rows[g_n][i] = (int8_t)cnt[i];
    n += cnt[i];
  }
  g_rows[g_n][15] = (int8_t)cat;
  g_info[g_n][0] = (uint8_t)cat;
  g_info[g_n][1] = (uint8_t)value;
  g_info[g_n][2] = (uint8_t)len;
  g_info[g_n][3] = (uint8_t)n;
  g_pk[g_n] = pack15(g_rows[g_n]);
  ++g_n;
}

/* itertools.combinations(remains, k) in lexicographic order (card.py:115,128,141,152) */
static void combos(const int* remains, int nrem, int k, const int* base, int mult, int cat,
                   int value, int len, int skip_joker_pair) {
  int idx[8];
  if (k > nrem) return;
  for (int i = 0; i < k; ++i) idx[i] = i;
  for (;;) {
    int cnt[NR];
    memcpy(cnt, base, sizeof(cnt));
    int has13 = 0, has14 = 0;
    for (int i = 0; i < k; ++i) {
      int r = remains[idx[i]];
      cnt[r] += mult;
      has13 |= r == 13;
      has14 |= r == 14;
    }
    /* card.py:116 / :142: "not ('*' in extra and '$' in extra and len(extra) == 2)" */
    if (!(skip_joker_pair && k == 2 && has13 && has14)) push(cnt, cat, value, len);
    int i = k - 1;
    while (i >= 0 && idx[i] == nrem - k + i) --i;
    if (i < 0) break;
    ++idx[i];
    for (int j = i + 1; j < k; ++j) idx[j] = idx[j - 1] + 1;
  }
}

static int cmp_pk(const void* a, const void* b) {
  uint64_t x = g_pk[*(const int32_t*)a], y = g_pk[*(const int32_t*)b];
  return x < y ? -1 : x > y;
}

static int imin(int a, int b) { return a < b ? a : b; }

/* get_action_space, card.py:34-159 -- same order, so index == canonical id */
void ddzo_init(void) {
  if (g_ready) return;
  g_n = 0;
  int z[NR] = {0}, c[NR];
  push(z, DDZO_EMPTY, 0, 1); /* card.py:35, analyze :374-375 */
  for (int r = 0; r < 15; ++r) { /* :40-41 */
    memcpy(c, z, sizeof c); c[r] = 1; push(c, DDZO_SINGLE, r, 1);
  }
  for (int m = 2; m <= 4; ++m) /* :46-48, :54-56, :62-64 */
    for (int r = 0; r < 13; ++r) {
      memcpy(c, z, sizeof c); c[r] = m;
      push(c, m == 2 ? DDZO_DOUBLE : m == 3 ? DDZO_TRIPLE : DDZO_QUADRIC, r, 1);
    }
  for (int main = 0; main < 13; ++main) /* 3+1  :69-73 */
    for (int e = 0; e < 15; ++e)
      if (e != main) {
        memcpy(c, z, sizeof c); c[main] = 3; c[e] += 1; push(c, DDZO_THREE_ONE, main, 1);
      }
  for (int main = 0; main < 13; ++main) /* 3+2  :78-82 */
    for (int e = 0; e < 13; ++e)
      if (e != main) {
        memcpy(c, z, sizeof c); c[main] = 3; c[e] += 2; push(c, DDZO_THREE_TWO, main, 1);
      }
  /* chains: start in range(to_value('3'), to_value('2')) = 0..11,
   * end in range(start+minlen, min(start+maxlen+1, to_value('*')=13))        */
  for (int s = 0; s < 12; ++s) /* single line :86-89 (len 5..12) */
    for (int e = s + 5; e < 13; ++e) {
      memcpy(c, z, sizeof c);
      for (int r = s; r < e; ++r) c[r] = 1;
      push(c, DDZO_SINGLE_LINE, s, e - s);
    }
  for (int s = 0; s < 12; ++s) /* double line :94-97 (3..10 pairs) */
    for (int e = s + 3; e < imin(s + 20 / 2 + 1, 13); ++e) {
      memcpy(c, z, sizeof c);
      for (int r = s; r < e; ++r) c[r] = 2;
      push(c, DDZO_DOUBLE_LINE, s, e - s);
    }
  for (int s = 0; s < 12; ++s) /* triple line :102-105 (2..6) */
    for (int e = s + 2; e < imin(s + 20 / 3 + 1, 13); ++e) {
      memcpy(c, z, sizeof c);
      for (int r = s; r < e; ++r) c[r] = 3;
      push(c, DDZO_TRIPLE_LINE, s, e - s);
    }
  for (int s = 0; s < 12; ++s) /* 3+1 line :110-117 (2..5 triples) */
    for (int e = s + 2; e < imin(s + 20 / 4 + 1, 13); ++e) {
      int rem[NR], nrem = 0;
      memcpy(c, z, sizeof c);
      for (int r = s; r < e; ++r) c[r] = 3;
      for (int r = 0; r < 15; ++r)
        if (r < s || r >= e) rem[nrem++] = r;
      combos(rem, nrem, e - s, c, 1, DDZO_THREE_ONE_LINE, s, e - s, 1);
    }
  for (int s = 0; s < 12; ++s) /* 3+2 line :122-129 (2..4 triples) */
    for (int e = s + 2; e < imin(s + 20 / 5 + 1, 13); ++e) {
      int rem[NR], nrem = 0;
      memcpy(c, z, sizeof c);
      for (int r = s; r < e; ++r) c[r] = 3;
      for (int r = 0; r < 13; ++r)
        if (r < s || r >= e) rem[nrem++] = r;
      combos(rem, nrem, e - s, c, 2, DDZO_THREE_TWO_LINE, s, e - s, 0);
    }
  memcpy(c, z, sizeof c); /* rocket :134, analyze :381 value 100 */
  c[13] = c[14] = 1;
  push(c, DDZO_BIGBANG, 100, 1);
  for (int main = 0; main < 13; ++main) { /* 4+1+1 :139-143 */
    int rem[NR], nrem = 0;
    memcpy(c, z, sizeof c); c[main] = 4;
    for (int r = 0; r < 15; ++r)
      if (r != main) rem[nrem++] = r;
    combos(rem, nrem, 2, c, 1, DDZO_FOUR_TAKE_ONE, main, 1, 1);
  }
  for (int main = 0; main < 13; ++main) { /* 4+2+2 :148-153 */
    int rem[NR], nrem = 0;
    memcpy(c, z, sizeof c); c[main] = 4;
    for (int r = 0; r < 13; ++r)
      if (r != main) rem[nrem++] = r;
    combos(rem, nrem, 2, c, 2, DDZO_FOUR_TAKE_TWO, main, 1, 0);
  }
  if (g_n != NA_BASE) abort();
#if DDZO_NATIVE_JOKER_KICKERS
  for (int main = 0; main < 13; ++main) { /* sidaihuojian, server/mcts/get_moves.py:22-27 */
    memcpy(c, z, sizeof c); c[main] = 4; c[13] = c[14] = 1;
    push(c, DDZO_FOUR_TAKE_ONE, main, 1);
  }
  for (int s = 0; s < 11; ++s) { /* sandaihuojian, server/mcts/get_moves.py:29-34 */
    memcpy(c, z, sizeof c); c[s] = c[s + 1] = 3; c[13] = c[14] = 1;
    push(c, DDZO_THREE_ONE_LINE, s, 2);
  }
#endif
  if (g_n != NA) abort();
  for (int i = 0; i < NA; ++i) g_sorted[i] = i;
  qsort(g_sorted, NA, sizeof(int32_t), cmp_pk);
  g_ready = 1;
}

int ddzo_num_actions(void) { return NA; }

void ddzo_action_table(int8_t* rows, uint8_t* info) {
  ddzo_init();
  if (rows) memcpy(rows, g_rows, sizeof g_rows);
  if (info) memcpy(info, g_info, sizeof g_info);
}

/* CardGroup.bigger_than, card.py:307-325 (self = a, g = b) */
int ddzo_beats(int a, int b) {
  ddzo_init();
  int ta = g_info[a][0], tb = g_info[b][0];
  if (ta == DDZO_EMPTY) return tb != DDZO_EMPTY;   /* :308-309 */
  if (tb == DDZO_EMPTY) return 1;                  /* :310-311 */
  if (tb == DDZO_BIGBANG) return 0;                /* :312-313 */
  if (ta == DDZO_BIGBANG) return 1;                /* :314-315 */
  if (tb == DDZO_QUADRIC)                          /* :316-320 */
    return ta == DDZO_QUADRIC && g_info[a][1] > g_info[b][1];
  return ta == DDZO_QUADRIC ||                     /* :321-325 */
         (ta == tb && g_info[a][2] == g_info[b][2] && g_info[a][1] > g_info[b][1]);
}

int ddzo_lookup(const int8_t* c15) {
  ddzo_init();
  for (int i = 0; i < NR; ++i)
    if (c15[i] < 0 || c15[i] > 4) return -1;
  uint64_t key = pack15(c15);
  int lo = 0, hi = NA - 1;
  while (lo <= hi) {
    int mid = (lo + hi) >> 1;
    uint64_t v = g_pk[g_sorted[mid]];
    if (v == key) return g_sorted[mid];
    if (v < key) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

/* per-nibble a <= b, nibbles in 0..7 (counter_subset, utils.py:16-22) */
static int subset_pk(uint64_t a, uint64_t b) {
  const uint64_t H = 0x8888888888888888ull;
  return (((b | H) - a) & H) == H;
}

/* get_mask, rule_based/utils/utils.py:45-63 */
static int legal_pk(uint64_t hand, int last_id, int32_t* ids, int8_t* rows, int64_t cap) {
  int n = 0;
  if (hand == 0) return 0; /* :48-49 "if not cards: return mask" */
  for (int j = 0; j < NA; ++j) {
    if (!subset_pk(g_pk[j], hand)) continue;           /* :50-52 */
    if (last_id <= 0) { if (j == 0) continue; }         /* :53-55 lead: mask[0] = 0 */
    else if (j > 0 && !ddzo_beats(j, last_id)) continue; /* :56-60 */
    if (n < cap) {
      if (ids) ids[n] = j;
      if (rows) memcpy(rows + (int64_t)n * DDZO_ROW, g_rows[j], DDZO_ROW);
    }
    ++n;
  }
  return n;
}

int ddzo_legal(const int8_t* hand15, const int8_t* last15, int32_t* ids, int cap) {
  ddzo_init();
  int last_id = 0;
  if (last15) {
    last_id = ddzo_lookup(last15);
    if (last_id < 0) return -1;
  }
  for (int i = 0; i < NR; ++i)
    if (hand15[i] < 0 || hand15[i] > 7) return -1;
  return legal_pk(pack15(hand15), last_id, ids, NULL, cap);
}

/* ---- Philox4x32-10 (Salmon et al., SC'11) ------------------------------- */
void ddzo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ---- state helpers ------------------------------------------------------ */
static inline uint8_t* fld(uint8_t* s, int64_t T, int f, int64_t t) {
  (void)T;
  return s + (t * DDZO_NFIELDS + f) * DDZO_ROW;
}
static inline const uint8_t* cfld(const uint8_t* s, int64_t T, int f, int64_t t) {
  (void)T;
  return s + (t * DDZO_NFIELDS + f) * DDZO_ROW;
}
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline void wr32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }
static inline uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline void wr16(uint8_t* p, uint16_t v) { memcpy(p, &v, 2); }

/* deal spec v2 (the reference's shuffle lives in the absent native `env`,
 * envi.py:10-13 / game.py:171 prepare()): card k = 0..53 (rank k/4 for
 * k < 52, 13 = BJ, 14 = CJ) draws the 32-bit key philox(gid, episode,
 * 1<<16 | k/4)[k%4]; the cards are ranked by (key, k) and the 17 smallest
 * go to role 0 (up), the next 20 to role 1 (lord), the last 17 to role 2
 * (down) (envi.py:23 left = [17, 20, 17]): a uniform random deal, and a
 * lane-parallel one on the GPU.                                             */
static void deal(uint8_t* s, int64_t T, int64_t t, uint64_t seed, uint64_t gid, uint32_t episode) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint8_t h[3][DDZO_ROW];
  memset(h, 0, sizeof h);
  uint32_t draws[56];
  for (uint32_t b = 0; b < 14; ++b) {
    uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), episode, (1u << 16) | b};
    ddzo_philox4x32_10(ctr, key, draws + 4 * b);
  }
  for (int k = 0; k < 54; ++k) {
    int pos = 0;
    for (int j = 0; j < 54; ++j)
      pos += draws[j] < draws[k] || (draws[j] == draws[k] && j < k);
    int role = pos < 17 ? 0 : pos < 37 ? 1 : 2;
    int rank = k < 52 ? k / 4 : k - 39;
    ++h[role][rank];
  }
  h[0][15] = 17; h[1][15] = 20; h[2][15] = 17;
  for (int r = 0; r < 3; ++r) {
    memcpy(fld(s, T, DDZO_F_HAND0 + r, t), h[r], DDZO_ROW);
    memset(fld(s, T, DDZO_F_HIST0 + r, t), 0, DDZO_ROW);   /* envi.py:34 */
    memset(fld(s, T, DDZO_F_RECENT0 + r, t), 0, DDZO_ROW); /* envi.py:35 */
  }
  memset(fld(s, T, DDZO_F_TAKEN, t), 0, DDZO_ROW);         /* envi.py:32 */
  uint8_t* m = fld(s, T, DDZO_F_META, t);
  memset(m, 0, DDZO_ROW);
  m[DDZO_M_ROLE] = 1; /* lord moves first, game.py:173 */
  m[DDZO_M_WINNER] = 0xFF;
  m[DDZO_M_DEALT] = 1;
  wr32(m + DDZO_M_EPISODE, episode);
}

void ddzo_env_reset(uint8_t* s, int64_t T, uint64_t seed, uint64_t gid_base, const uint8_t* mask) {
  ddzo_init();
  for (int64_t t = 0; t < T; ++t) {
    if (mask && !mask[t]) continue;
    const uint8_t* m = cfld(s, T, DDZO_F_META, t);
    uint32_t ep = m[DDZO_M_DEALT] ? rd32(m + DDZO_M_EPISODE) + 1 : 0;
    deal(s, T, t, seed, gid_base + (uint64_t)t, ep);
  }
}

/* the combo the actor must beat: previous player's handout, else the one
 * before, else lead (envi.py:103-109).  Returns the action id (0 = lead).   */
static int last_id_of(const uint8_t* s, int64_t T, int64_t t, int role) {
  const uint8_t* b1 = cfld(s, T, DDZO_F_RECENT0 + (role + 2) % 3, t);
  const uint8_t* b2 = cfld(s, T, DDZO_F_RECENT0 + (role + 1) % 3, t);
  int id = ddzo_lookup((const int8_t*)b1);
  if (id <= 0) id = ddzo_lookup((const int8_t*)b2);
  return id < 0 ? 0 : id;
}

int64_t ddzo_env_legal(const uint8_t* s, int64_t T, int32_t* offsets, int8_t* rows, int32_t* ids,
                       int64_t cap) {
  ddzo_init();
  int64_t total = 0;
  for (int64_t t = 0; t < T; ++t) {
    const uint8_t* m = cfld(s, T, DDZO_F_META, t);
    offsets[t] = (int32_t)total;
    if (m[DDZO_M_DONE] || !m[DDZO_M_DEALT]) continue; /* frozen table: empty list */
    int role = m[DDZO_M_ROLE];
    uint64_t hand = pack15((const int8_t*)cfld(s, T, DDZO_F_HAND0 + role, t));
    int64_t room = cap > total ? cap - total : 0;
    total += legal_pk(hand, last_id_of(s, T, t, role), (ids && room) ? ids + total : NULL,
                      (rows && room) ? rows + total * DDZO_ROW : NULL, room);
  }
  offsets[T] = (int32_t)total;
  return total;
}

void ddzo_env_step(uint8_t* s, int64_t T, uint64_t seed, uint64_t gid_base, int mode,
                   const void* sel, const int32_t* offsets, const int8_t* rows, int auto_reset,
                   uint8_t* done, int8_t* reward, uint8_t* illegal, uint8_t* traj) {
  ddzo_init();
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (int64_t t = 0; t < T; ++t) {
    uint8_t* m = fld(s, T, DDZO_F_META, t);
    uint8_t* tr = traj ? traj + t * DDZO_TR_BYTES : NULL;
    int role = m[DDZO_M_ROLE];
    uint32_t ep = rd32(m + DDZO_M_EPISODE);
    uint16_t ply = rd16(m + DDZO_M_PLY);
    int32_t A = offsets[t + 1] - offsets[t];
    if (tr) {
      memset(tr, 0, DDZO_TR_BYTES);
      tr[DDZO_TR_ROLE] = (uint8_t)role;
      wr16(tr + DDZO_TR_NLEGAL, (uint16_t)A);
      wr16(tr + DDZO_TR_PLY, ply);
      wr32(tr + DDZO_TR_EPISODE, ep);
      wr32(tr + DDZO_TR_CHOICE, (uint32_t)-1);
    }
    if (m[DDZO_M_DONE] || !m[DDZO_M_DEALT] || A <= 0) { /* frozen */
      if (done) done[t] = m[DDZO_M_DONE];
      if (reward) reward[t] = 0;
      if (illegal) illegal[t] = 0;
      if (tr) { tr[DDZO_TR_DONE] = m[DDZO_M_DONE]; tr[DDZO_TR_FLAGS] = 2; }
      continue;
    }
    int32_t idx = -1;
    const int8_t* seg = rows + (int64_t)offsets[t] * DDZO_ROW;
    if (mode == DDZO_STEP_RANDOM || (mode == DDZO_STEP_IDS && ((const int32_t*)sel)[t] == -1)) {
      /* random.choice(actions), envi.py:83 -> engine RNG (spec v1) */
      uint32_t ctr[4] = {(uint32_t)(gid_base + t), (uint32_t)((gid_base + t) >> 32), ep,
                         (2u << 16) | ply};
      uint32_t out[4];
      ddzo_philox4x32_10(ctr, key, out);
      idx = (int32_t)(((uint64_t)out[0] * (uint32_t)A) >> 32);
    } else if (mode == DDZO_STEP_CHOICE) {
      idx = ((const int32_t*)sel)[t];
      if (idx < 0 || idx >= A) idx = -1;
    } else if (mode == DDZO_STEP_IDS) { /* canonical action ids; looked up in the legal list */
      int32_t id = ((const int32_t*)sel)[t];
      if (id >= 0 && id < NA)
        for (int32_t j = 0; j < A && idx < 0; ++j)
          if (memcmp(seg + (int64_t)j * DDZO_ROW, g_rows[id], NR) == 0) idx = j;
    } else {
      const int8_t* want = (const int8_t*)sel + t * DDZO_ROW;
      for (int32_t j = 0; j < A && idx < 0; ++j)
        if (memcmp(seg + (int64_t)j * DDZO_ROW, want, NR) == 0) idx = j;
    }
    if (idx < 0) { /* illegal action: table untouched, flagged */
      if (done) done[t] = 0;
      if (reward) reward[t] = 0;
      if (illegal) illegal[t] = 1;
      if (tr) tr[DDZO_TR_FLAGS] = 1;
      continue;
    }
    const int8_t* row = seg + (int64_t)idx * DDZO_ROW;
    uint8_t* hand = fld(s, T, DDZO_F_HAND0 + role, t);
    uint8_t* hist = fld(s, T, DDZO_F_HIST0 + role, t);
    uint8_t* taken = fld(s, T, DDZO_F_TAKEN, t);
    int n = 0;
    for (int i = 0; i < NR; ++i) { /* envi.py:38-43 _update + native removal */
      hand[i] -= row[i]; hist[i] += row[i]; taken[i] += row[i]; n += row[i];
    }
    hand[15] -= n;                                              /* left[role], envi.py:39 */
    memcpy(fld(s, T, DDZO_F_RECENT0 + role, t), row, DDZO_ROW); /* envi.py:43 */
    int won = hand[15] == 0;
    int8_t r = won ? (role == 1 ? -1 : 1) : 0; /* rule_play.py:14: -1 lord wins, +1 farmers */
    m[DDZO_M_ROLE] = (uint8_t)((role + 1) % 3); /* lord -> down -> up, game.py:173-181 */
    wr16(m + DDZO_M_PLY, (uint16_t)(ply + 1));
    m[DDZO_M_REWARD] = (uint8_t)r;
    if (won) { m[DDZO_M_DONE] = 1; m[DDZO_M_WINNER] = (uint8_t)role; }
    if (done) done[t] = (uint8_t)won;
    if (reward) reward[t] = r;
    if (illegal) illegal[t] = 0;
    if (tr) {
      memcpy(tr + DDZO_TR_ROW, row, DDZO_ROW);
      tr[DDZO_TR_DONE] = (uint8_t)won;
      tr[DDZO_TR_REWARD] = (uint8_t)r;
      wr32(tr + DDZO_TR_CHOICE, (uint32_t)idx);
    }
    if (won && auto_reset) deal(s, T, t, seed, gid_base + (uint64_t)t, ep + 1);
  }
}

int ddzo_planes(int variant) {
  static const int p[4] = {4, 7, 9, 6};
  return variant >= 0 && variant < 4 ? p[variant] : -1;
}

/* thermometer: slot j of rank i set iff count > j (envi.py:139-146) */
static void thermo(const uint8_t* c, float* out) {
  for (int i = 0; i < NR; ++i)
    for (int j = 0; j < 4; ++j) out[i * 4 + j] = c[i] > j ? 1.0f : 0.0f;
}

void ddzo_rows_to_onehot(const int8_t* rows, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i) thermo((const uint8_t*)rows + i * DDZO_ROW, out + i * 60);
}

/* face: known planes (envi.py:87-96,165-217) + two prob planes.  The prob
 * planes come from native get_state_prob() (envi.py:94) whose source is
 * absent; spec v1 (PARITY UNPINNED): with known = hand + taken
 * (server/core.py:26-33), n1 = left[(role+1)%3], n2 = left[(role+2)%3],
 *   p1[i][j] = [known_i <= j < total_i] * n1/(n1+n2),  p2 likewise with n2,
 * total_i = 4 (1 for the jokers), all in IEEE f32.                          */
void ddzo_env_observe(const uint8_t* s, int64_t T, int variant, float* out) {
  int P = ddzo_planes(variant);
  if (P < 0) return;
  for (int64_t t = 0; t < T; ++t) {
    const uint8_t* m = cfld(s, T, DDZO_F_META, t);
    int role = m[DDZO_M_ROLE];
    const uint8_t* hand = cfld(s, T, DDZO_F_HAND0 + role, t);
    const uint8_t* taken = cfld(s, T, DDZO_F_TAKEN, t);
    const uint8_t* h0 = cfld(s, T, DDZO_F_HIST0 + (role + 2) % 3, t);   /* (role-1)%3 */
    const uint8_t* h1 = cfld(s, T, DDZO_F_HIST0 + role, t);
    const uint8_t* h2 = cfld(s, T, DDZO_F_HIST0 + (role + 1) % 3, t);
    const uint8_t* b1 = cfld(s, T, DDZO_F_RECENT0 + (role + 2) % 3, t); /* (role-1)%3 */
    const uint8_t* b2 = cfld(s, T, DDZO_F_RECENT0 + (role + 1) % 3, t); /* (role-2)%3 */
    float* o = out + t * (int64_t)P * 60;
    int p = 0;
    thermo(hand, o + 60 * p++);
    thermo(taken, o + 60 * p++);
    if (variant == 1 || variant == 2) {
      thermo(h0, o + 60 * p++); thermo(h1, o + 60 * p++); thermo(h2, o + 60 * p++);
    }
    if (variant == 2 || variant == 3) {
      thermo(b1, o + 60 * p++); thermo(b2, o + 60 * p++);
    }
    int n1 = cfld(s, T, DDZO_F_HAND0 + (role + 1) % 3, t)[15];
    int n2 = cfld(s, T, DDZO_F_HAND0 + (role + 2) % 3, t)[15];
    float f1 = n1 + n2 > 0 ? (float)n1 / (float)(n1 + n2) : 0.0f;
    float f2 = n1 + n2 > 0 ? (float)n2 / (float)(n1 + n2) : 0.0f;
    float* p1 = o + 60 * p++;
    float* p2 = o + 60 * p++;
    for (int i = 0; i < NR; ++i) {
      int known = hand[i] + taken[i], total = i < 13 ? 4 : 1;
      for (int j = 0; j < 4; ++j) {
        int unseen = j >= known && j < total;
        p1[i * 4 + j] = unseen ? f1 : 0.0f;
        p2[i * 4 + j] = unseen ? f2 : 0.0f;
      }
    }
  }
}

/* get_state_prob_manual(known60, size1, size2) (server/core.py:26-33), prob planes spec v1: the two planes of
 * ddzo_env_observe from an explicit thermometer of the seen cards; out f32[2][15][4].                       */
void ddzo_state_prob(const uint8_t* known60, int n1, int n2, float* out) {
  float f1 = n1 + n2 > 0 ? (float)n1 / (float)(n1 + n2) : 0.0f;
  float f2 = n1 + n2 > 0 ? (float)n2 / (float)(n1 + n2) : 0.0f;
  for (int i = 0; i < NR; ++i) {
    int known = 0, total = i < 13 ? 4 : 1;
    for (int j = 0; j < 4; ++j) known += known60[i * 4 + j] != 0; /* onehot2arr: row sum (envi.py:148-157) */
    for (int j = 0; j < 4; ++j) {
      int unseen = j >= known && j < total;
      out[i * 4 + j] = unseen ? f1 : 0.0f;
      out[60 + i * 4 + j] = unseen ? f2 : 0.0f;
    }
  }
}

/* DQNFirst.greedy_action / e_greedy_action (dqn.py:50-71): first index of the maximum;
 * exploration by the engine RNG, domain 3 (spec v1): draw.x < floor(eps * 2^32) -> uniform */
void ddzo_select(const uint8_t* s, int64_t T, uint64_t seed, uint64_t gid_base, const float* q,
                 const int32_t* offsets, double epsilon, int32_t* choice) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint64_t thr = (uint64_t)(epsilon * 4294967296.0);
  for (int64_t t = 0; t < T; ++t) {
    int32_t off = offsets[t], A = offsets[t + 1] - off;
    if (A <= 0) { choice[t] = -1; continue; }
    int best = 0;
    for (int j = 1; j < A; ++j)
      if (q[off + j] > q[off + best]) best = j;
    if (thr) {
      const uint8_t* m = cfld(s, T, DDZO_F_META, t);
      uint64_t gid = gid_base + (uint64_t)t;
      uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), rd32(m + DDZO_M_EPISODE),
                         (3u << 16) | rd16(m + DDZO_M_PLY)};
      uint32_t out[4];
      ddzo_philox4x32_10(ctr, key, out);
      if ((uint64_t)out[0] < thr) best = (int)(((uint64_t)out[1] * (uint32_t)A) >> 32);
    }
    choice[t] = best;
  }
}

int64_t ddzo_rollout_random(uint8_t* s, int64_t T, uint64_t seed, uint64_t gid_base,
                            int64_t n_iters, int64_t* sum_legal, int64_t* episodes_done) {
  ddzo_init();
  int64_t cap = T * 512;
  int32_t* offsets = (int32_t*)malloc((size_t)(T + 1) * 4);
  int8_t* rows = (int8_t*)malloc((size_t)cap * DDZO_ROW);
  uint8_t* done = (uint8_t*)malloc((size_t)T);
  int64_t plies = 0, legal = 0, eps = 0;
  for (int64_t it = 0; it < n_iters; ++it) {
    legal += ddzo_env_legal(s, T, offsets, rows, NULL, cap);
    ddzo_env_step(s, T, seed, gid_base, DDZO_STEP_RANDOM, NULL, offsets, rows, 1, done, NULL,
                  NULL, NULL);
    for (int64_t t = 0; t < T; ++t) eps += done[t];
    plies += T;
  }
  free(offsets); free(rows); free(done);
  if (sum_legal) *sum_legal = legal;
  if (episodes_done) *episodes_done = eps;
  return plies;
}
