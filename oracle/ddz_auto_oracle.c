/*
 * ddz_auto_oracle.c -- CPU ORACLE of the rule-based opponent, test infrastructure, NOT product.
 *
 * What the reference has (paths relative to /root/reference):
 *   rule_based/utils/evaluator.py:10-47        cards_value[13527]            -> restated here, PINNED by fixture G7
 *   rule_based/utils/decomposer.py:17-76       Decomposer.get_combinations   -> Python glue restated here; PINNED by
 *   rule_based/utils/rule_based_model.py:43-101 RuleBasedModel.choose           fixture G8 *given* the two stand-ins below
 * What it does NOT have: env.get_combinations_recursive / env.get_combinations_nosplit (decomposer.py:10) live in
 * the absent native module, and so does the native Env.step_auto (envi.py:72-77).  They are DEFINED here
 * ("decomposer spec v1", PARITY UNPINNED):
 *
 *   get_combinations_recursive(mask[M][15], target[15])  (hands of <= 10 cards, decomposer.py:50-58)
 *       every multiset of rows whose count vectors sum exactly to `target`; a row may repeat (that is why the
 *       reference must not feed it the empty row: "will cause infinite loop", decomposer.py:56).  Order:
 *       depth-first, always covering the LOWEST remaining rank next, candidate rows in ascending index; while the
 *       lowest rank stays the same the row index may not decrease (each multiset once).  A combination lists its
 *       rows in selection order; the combinations come out in depth-first order.
 *   get_combinations_nosplit(mask[M][60], card_mask[60])  (hands of > 10 cards, decomposer.py:18-34)
 *       every exact cover of the hand's thermometer slots (Card.char2onehot60, card.py:184-192) by rows of the
 *       augmented action space (card.py:534-549): each slot covered exactly once.  Because an action's thermometer
 *       always starts at slot 0 of a rank, a rank can be touched by ONE regular action only -- groups are not split
 *       between actions, which is the function's name and the "known issue" the reference notes at decomposer.py:32 --
 *       and its surplus cards leave as the augmented solo singles / pair (card.py:537-547).  Order: Algorithm X with
 *       the lowest uncovered slot as the column, rows in ascending index; rows in selection order.
 *
 * Everything here is double arithmetic in exactly the reference's operation order (rule_based_model.py:54,63-69,83)
 * so that ties break the same way: values are compared with `>` and the first maximum in (combination, position)
 * order wins.
 */
#include "ddz_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define NAB DDZO_NUM_ACTIONS /* the rule agent works on card.py's 13,527 rows in every build */
#define NR DDZO_NUM_RANKS
#define NAUG (NAB + 13 * 3 + 13) /* card.py:534-535 augment_action_space */

static int8_t a_rows[NAB][DDZO_ROW];
static uint8_t a_info[NAB][4];
static int16_t a_val2[NAB];    /* cards_value * 2 */
static int32_t* a_by_rank[NR]; /* ids (>= 1) whose row contains the rank, ascending */
static int a_by_rank_n[NR];
static int a_ready = 0;

/* rule_based/utils/evaluator.py:10-47; char2val[c] = rank index + 3, so "maxCard - 10" = index - 7.
 * The card lists follow card.py:34-159: a[0] is the (first) main card, a[-1] the last card of the list. */
static double eval_row(int id) {
  const int8_t* c = a_rows[id];
  int cat = a_info[id][0], val = a_info[id][1], len = a_info[id][2];
  double v = 0;
  if (cat == DDZO_EMPTY) return 0;                        /* :20-21 */
  if (cat <= DDZO_TRIPLE) {                               /* :22-27 */
    v = val - 7;
    if (cat == DDZO_DOUBLE && v > 0) v *= 1.5;
    if (cat == DDZO_TRIPLE && v > 0) v *= 2;
    return v;
  }
  if (cat == DDZO_QUADRIC) return 9;                      /* :28-29 */
  if (cat <= DDZO_THREE_TWO) {                            /* :30-33 */
    v = val - 7;
    if (v > 0) v *= 1.5;
    return v;
  }
  if (cat <= DDZO_TRIPLE_LINE) {                          /* :34-35: a[-1] = highest card of the chain */
    v = ((val + len - 1) - 7) / 2.0;
    return v > 0 ? v : 0;
  }
  if (cat == DDZO_THREE_ONE_LINE || cat == DDZO_THREE_TWO_LINE) { /* :36-47 */
    /* a = sorted(main * 3) + kickers (card.py:117,129): a[-1] is the HIGHEST KICKER, not a main card */
    int hi = -1;
    for (int r = 0; r < NR; ++r)
      if (c[r] > 0 && (r < val || r >= val + len)) hi = r;
    v = (hi - 7) / 2.0;
    if (v < 0) v = 0;
    for (int r = 0; r < NR; ++r)
      if (c[r] > 0 && (r < val || r >= val + len) && r - 7 > 0)
        v += cat == DDZO_THREE_ONE_LINE ? (double)(r - 7) : 1.5 * (r - 7); /* :39-41 / :45-47 (each pair once) */
    return v;
  }
  if (cat == DDZO_BIGBANG) return 12;                     /* :48-49 */
  return val - 7;                                         /* :50-51 four-with-two */
}

static pthread_once_t a_once = PTHREAD_ONCE_INIT;
static void auto_init_once(void);
static void auto_init(void) { pthread_once(&a_once, auto_init_once); } /* the tests call the oracle from thread pools */
static void auto_init_once(void) {
  if (a_ready) return;
  ddzo_init();
  if (ddzo_num_actions() == NAB) {
    ddzo_action_table(&a_rows[0][0], &a_info[0][0]);
  } else { /* the joker-kicker build appends 24 rows: take the first 13,527 */
    int n = ddzo_num_actions();
    int8_t* r = (int8_t*)malloc((size_t)n * DDZO_ROW);
    uint8_t* f = (uint8_t*)malloc((size_t)n * 4);
    ddzo_action_table(r, f);
    memcpy(a_rows, r, sizeof a_rows);
    memcpy(a_info, f, sizeof a_info);
    free(r); free(f);
  }
  for (int id = 0; id < NAB; ++id) a_val2[id] = (int16_t)lrint(2.0 * eval_row(id));
  for (int r = 0; r < NR; ++r) {
    int n = 0;
    for (int id = 1; id < NAB; ++id) n += a_rows[id][r] > 0;
    a_by_rank[r] = (int32_t*)malloc((size_t)n * 4);
    a_by_rank_n[r] = 0;
    for (int id = 1; id < NAB; ++id)
      if (a_rows[id][r] > 0) a_by_rank[r][a_by_rank_n[r]++] = id;
  }
  a_ready = 1;
}

void ddzo_cards_value_x2(int16_t* out) {
  auto_init();
  memcpy(out, a_val2, sizeof a_val2);
}

/* ---- generic stand-ins for the two absent native functions (decomposer spec v1) -------------------------- */
typedef struct {
  int32_t* out;
  int64_t cap, used, ncombs;
  int32_t stack[64];
  int depth;
} Sink;

static void emit(Sink* s) {
  if (s->out && s->used + 1 + s->depth <= s->cap) {
    s->out[s->used] = s->depth;
    memcpy(s->out + s->used + 1, s->stack, (size_t)s->depth * 4);
  }
  s->used += 1 + s->depth;
  s->ncombs += 1;
}

static void rec_generic(const uint8_t* mask, int M, int* rem, int prev_rank, int prev_row, Sink* s) {
  int r = 0;
  while (r < NR && rem[r] == 0) ++r;
  if (r == NR) { emit(s); return; }
  if (s->depth >= 63) return;
  for (int i = (r == prev_rank ? prev_row : 0); i < M; ++i) {
    const uint8_t* row = mask + (size_t)i * NR;
    if (row[r] == 0) continue;
    int ok = 1;
    for (int k = 0; k < NR && ok; ++k) ok = row[k] <= rem[k];
    if (!ok) continue;
    for (int k = 0; k < NR; ++k) rem[k] -= row[k];
    s->stack[s->depth++] = i;
    rec_generic(mask, M, rem, r, i, s);
    --s->depth;
    for (int k = 0; k < NR; ++k) rem[k] += row[k];
  }
}

/* out: for each combination its length followed by its row indices; returns the int32 words needed */
int64_t ddzo_combinations_recursive(const uint8_t* mask, int M, const uint8_t* target, int32_t* out, int64_t cap,
                                    int64_t* ncombs) {
  int rem[NR];
  for (int k = 0; k < NR; ++k) rem[k] = target[k];
  Sink s = {out, cap, 0, 0, {0}, 0};
  rec_generic(mask, M, rem, -1, 0, &s);
  if (ncombs) *ncombs = s.ncombs;
  return s.used;
}

static void cover_generic(const uint8_t* mask, int M, uint8_t* need, Sink* s) {
  int c = 0;
  while (c < 60 && !need[c]) ++c;
  if (c == 60) { emit(s); return; }
  if (s->depth >= 63) return;
  for (int i = 0; i < M; ++i) {
    const uint8_t* row = mask + (size_t)i * 60;
    if (!row[c]) continue;
    int ok = 1;
    for (int k = 0; k < 60 && ok; ++k) ok = !row[k] || need[k];
    if (!ok) continue;
    for (int k = 0; k < 60; ++k) need[k] = (uint8_t)(need[k] && !row[k]);
    s->stack[s->depth++] = i;
    cover_generic(mask, M, need, s);
    --s->depth;
    for (int k = 0; k < 60; ++k) need[k] = (uint8_t)(need[k] || row[k]);
  }
}

int64_t ddzo_combinations_nosplit(const uint8_t* mask, int M, const uint8_t* card_mask, int32_t* out, int64_t cap,
                                  int64_t* ncombs) {
  uint8_t need[60];
  for (int k = 0; k < 60; ++k) need[k] = card_mask[k] != 0;
  Sink s = {out, cap, 0, 0, {0}, 0};
  cover_generic(mask, M, need, &s);
  if (ncombs) *ncombs = s.ncombs;
  return s.used;
}

/* ---- Decomposer.get_combinations + RuleBasedModel.choose on action ids ---------------------------------- */
typedef struct {
  int follow, last_id, pass_ok;
  double rp;                 /* round_penalty */
  const int32_t* cand[NR];   /* per rank: ids of the rows that fit the hand and contain the rank, ascending */
  int ncand[NR];
  int32_t comb[24];          /* the combination being built (clamped ids, rule_based_model.py sees these) */
  int n;                     /* its length, including the leading 0 when following (decomposer.py:34,60) */
  double max_value;
  int best_move;             /* -1 = None */
  int64_t ncombs, nodes;
} Choose;

/* one combination: rule_based_model.py:60-93 */
static void score(Choose* c) {
  c->ncombs += 1;
  double total = 0; /* :62  sum of cards_value (exact: multiples of 0.5) */
  for (int j = 0; j < c->n; ++j) total += a_val2[c->comb[j]] / 2.0;
  /* :63-66  the loop indexes action_space[j] by POSITION: positions 1..n-1 count unless action_space[j][0] == '2',
   * i.e. j == 13 (the single '2'); "R" and "B" never occur in card.py's alphabet */
  int small_num = 0;
  for (int j = 1; j < c->n; ++j) small_num += j != 13;
  total -= small_num * c->rp; /* :67 */
  for (int j = 0; j < c->n; ++j) {
    int x = c->comb[j];
    if (x == 0 && c->pass_ok) {                 /* :70-74 pass is scored only while min_oppo_cards > 4 */
      if (total > c->max_value) { c->max_value = total; c->best_move = 0; }
    } else if (x > 0 && (!c->follow || ddzo_beats(x, c->last_id))) { /* :76 fine_mask (decomposer.py:39-46,64-72) */
      if (c->n == 1 || (c->n == 2 && c->comb[0] == 0)) { /* :78-81 the whole hand goes in one move */
        c->max_value = INFINITY;
        c->best_move = c->comb[c->n - 1];
      }
      double mv = total - a_val2[x] / 2.0 + c->rp; /* :82 */
      if (mv > c->max_value) { c->max_value = mv; c->best_move = x; }
    }
  }
  if (c->best_move < 0) c->best_move = 0; /* :87-89 */
}

/* hands of <= 10 cards: multisets of the rows that fit the hand (decomposer.py:50-60) */
static void rec_ids(Choose* c, int* rem, int prev_rank, int prev_id) {
  c->nodes += 1;
  int r = 0;
  while (r < NR && rem[r] == 0) ++r;
  if (r == NR) { score(c); return; }
  const int32_t* cand = c->cand[r];
  for (int q = 0; q < c->ncand[r]; ++q) {
    int id = cand[q];
    if (r == prev_rank && id < prev_id) continue;
    const int8_t* row = a_rows[id];
    int ok = 1;
    for (int k = 0; k < NR && ok; ++k) ok = row[k] <= rem[k];
    if (!ok) continue;
    for (int k = 0; k < NR; ++k) rem[k] -= row[k];
    c->comb[c->n++] = id;
    rec_ids(c, rem, r, id);
    --c->n;
    for (int k = 0; k < NR; ++k) rem[k] += row[k];
  }
}

/* hands of > 10 cards: exact cover of the thermometer slots (decomposer.py:18-36).  cnt[r] = cards of rank r in the
 * hand, cov[r] = its slots [0, cov[r]) are covered.  Lowest uncovered slot = (r, cov[r]) of the lowest rank with
 * cov[r] < cnt[r]:
 *   slot 0   -> a regular action containing r whose ranks are all untouched and which fits the hand;
 *   slot k>0 -> the augmented single of slot k (index NAB + 13(k-1) + r -> clamp_action_idx: single of r,
 *               card.py:552-559), then -- k == 2 of a quad only -- the augmented pair of slots 2,3 (-> pair of r).  */
static void cover_ids(Choose* c, const int* cnt, int* cov) {
  c->nodes += 1;
  int r = 0;
  while (r < NR && cov[r] >= cnt[r]) ++r;
  if (r == NR) { score(c); return; }
  if (cov[r] == 0) {
    const int32_t* cand = c->cand[r];
    for (int q = 0; q < c->ncand[r]; ++q) {
      int id = cand[q];
      const int8_t* row = a_rows[id];
      int ok = 1;
      for (int k = 0; k < NR && ok; ++k) ok = row[k] == 0 || (cov[k] == 0 && row[k] <= cnt[k]);
      if (!ok) continue;
      for (int k = 0; k < NR; ++k) cov[k] += row[k];
      c->comb[c->n++] = id;
      cover_ids(c, cnt, cov);
      --c->n;
      for (int k = 0; k < NR; ++k) cov[k] -= row[k];
    }
  } else {
    cov[r] += 1;
    c->comb[c->n++] = 1 + r; /* single of rank r */
    cover_ids(c, cnt, cov);
    --c->n;
    cov[r] -= 1;
    if (cov[r] == 2 && cnt[r] == 4) {
      cov[r] = 4;
      c->comb[c->n++] = 16 + r; /* pair of rank r */
      cover_ids(c, cnt, cov);
      --c->n;
      cov[r] = 2;
    }
  }
}

/* RuleBasedModel.choose (rule_based_model.py:43-101) -> canonical action id.  hand15/last15: count vectors
 * (last all-zero or NULL = lead); left[3] = cards left of role 0 up / 1 lord / 2 down (envi.py:23); role = actor.
 * stats (may be NULL): {combinations, search nodes}.  Returns -1 on a bad `last`.                            */
int ddzo_auto_choose(const int8_t* hand15, const int8_t* last15, const int32_t* left, int role, int64_t* stats) {
  auto_init();
  Choose c;
  memset(&c, 0, sizeof c);
  c.last_id = last15 ? ddzo_lookup(last15) : 0;
  if (c.last_id < 0 || c.last_id >= NAB) return -1;
  c.follow = c.last_id != 0;
  /* :56-57 (the role test is the reference's: 0 = up compares against lord and down, everybody else against up) */
  int min_opp = role == 0 ? (left[1] < left[2] ? left[1] : left[2]) : left[0];
  c.rp = 15 - 12 * min_opp / 20.0;
  c.pass_ok = min_opp > 4;
  c.max_value = -INFINITY;
  c.best_move = -1;
  int cnt[NR], ncards = 0;
  for (int k = 0; k < NR; ++k) { cnt[k] = hand15[k]; ncards += cnt[k]; }
  if (c.follow) c.comb[c.n++] = 0; /* decomposer.py:34,60 */
  /* the rows that fit the hand (decomposer.py:19-28 valid_row_idx / :50-55 valid), split by the ranks they contain */
  int32_t* pool = NULL;
  {
    int nfit = 0, per[NR] = {0};
    static _Thread_local int32_t fit[NAB];
    for (int id = 1; id < NAB; ++id) {
      const int8_t* row = a_rows[id];
      int ok = 1;
      for (int k = 0; k < NR && ok; ++k) ok = row[k] <= cnt[k];
      if (!ok) continue;
      fit[nfit++] = id;
      for (int k = 0; k < NR; ++k) per[k] += row[k] > 0;
    }
    int tot = 0;
    for (int k = 0; k < NR; ++k) tot += per[k];
    pool = (int32_t*)malloc((size_t)(tot + 1) * 4);
    int32_t* w = pool;
    for (int k = 0; k < NR; ++k) {
      c.cand[k] = w;
      c.ncand[k] = 0;
      for (int q = 0; q < nfit; ++q)
        if (a_rows[fit[q]][k] > 0) w[c.ncand[k]++] = fit[q];
      w += c.ncand[k];
    }
  }
  if (ncards > 10) {
    int cov[NR] = {0};
    cover_ids(&c, cnt, cov);
  } else {
    rec_ids(&c, cnt, -1, 0);
  }
  free(pool);
  if (stats) { stats[0] = c.ncombs; stats[1] = c.nodes; }
  return c.best_move < 0 ? 0 : c.best_move;
}

/* the combinations themselves (clamped action ids), through the generic stand-ins on the reference's own matrices */
int64_t ddzo_auto_combinations(const int8_t* hand15, int follow, int32_t* out, int64_t cap, int64_t* ncombs) {
  /* the matrices Decomposer.get_combinations builds: rows that fit the hand (decomposer.py:19-28,50-55) */
  auto_init();
  int cnt[NR], ncards = 0;
  for (int k = 0; k < NR; ++k) { cnt[k] = hand15[k]; ncards += cnt[k]; }
  int64_t used = 0, nc = 0;
  if (ncards > 10) {
    uint8_t card_mask[60];
    for (int r = 0; r < NR; ++r)
      for (int j = 0; j < 4; ++j) card_mask[4 * r + j] = cnt[r] > j;
    uint8_t* mask = (uint8_t*)calloc((size_t)NAUG * 60, 1);
    int32_t* map = (int32_t*)malloc((size_t)NAUG * 4);
    int M = 0;
    for (int idx = 0; idx < NAUG; ++idx) {
      uint8_t row[60] = {0};
      int clamp = idx;
      if (idx < NAB) {
        for (int r = 0; r < NR; ++r)
          for (int j = 0; j < a_rows[idx][r]; ++j) row[4 * r + j] = 1;
      } else if (idx < NAB + 39) {
        int j = (idx - NAB) / 13, i = (idx - NAB) % 13; /* card.py:538-542 */
        row[4 * i + j + 1] = 1;
        clamp = i + 1;                                  /* card.py:558 */
      } else {
        int i = idx - NAB - 39;                         /* card.py:544-547 */
        row[4 * i + 2] = row[4 * i + 3] = 1;
        clamp = i + 16;                                 /* card.py:556 */
      }
      int ok = !(idx == 0 && !follow);                  /* decomposer.py:23-24 */
      for (int k = 0; k < 60 && ok; ++k) ok = !row[k] || card_mask[k];
      if (!ok) continue;
      memcpy(mask + (size_t)M * 60, row, 60);
      map[M++] = clamp;
    }
    used = ddzo_combinations_nosplit(mask, M, card_mask, out, cap, &nc);
    for (int64_t p = 0; out && p < used && p < cap;) {
      int n = out[p];
      for (int k = 1; k <= n && p + k < cap; ++k) out[p + k] = map[out[p + k]];
      p += 1 + n;
    }
    free(mask); free(map);
  } else {
    uint8_t* mask = (uint8_t*)malloc((size_t)NAB * NR);
    int32_t* map = (int32_t*)malloc((size_t)NAB * 4);
    uint8_t target[NR];
    int M = 0;
    for (int r = 0; r < NR; ++r) target[r] = (uint8_t)cnt[r];
    for (int id = 1; id < NAB; ++id) { /* row 0 has an all-zero mask: not `valid` (decomposer.py:54) */
      int ok = 1;
      for (int r = 0; r < NR && ok; ++r) ok = a_rows[id][r] <= cnt[r];
      if (!ok) continue;
      for (int r = 0; r < NR; ++r) mask[(size_t)M * NR + r] = (uint8_t)a_rows[id][r];
      map[M++] = id;
    }
    used = ddzo_combinations_recursive(mask, M, target, out, cap, &nc);
    for (int64_t p = 0; out && p < used && p < cap;) {
      int n = out[p];
      for (int k = 1; k <= n && p + k < cap; ++k) out[p + k] = map[out[p + k]];
      p += 1 + n;
    }
    free(mask); free(map);
  }
  if (ncombs) *ncombs = nc;
  return used;
}

/* Env.step_auto's choice for every table whose actor's role bit is set in auto_roles (bit r = role r is played by
 * the rule agent), -1 for the other tables and for frozen ones.  hand / last / left as the reference's choose()
 * reads them from the env (rule_based_model.py:17-33,56).                                                      */
void ddzo_env_auto_choose(const uint8_t* s, int64_t T, int auto_roles, int32_t* ids, int64_t* stats) {
  auto_init();
  int64_t acc[2] = {0, 0};
  for (int64_t t = 0; t < T; ++t) {
    const uint8_t* base = s + t * DDZO_NFIELDS * DDZO_ROW;
    const uint8_t* m = base + DDZO_F_META * DDZO_ROW;
    int role = m[DDZO_M_ROLE];
    ids[t] = -1;
    if (m[DDZO_M_DONE] || !m[DDZO_M_DEALT] || role > 2 || !((auto_roles >> role) & 1)) continue;
    const int8_t* hand = (const int8_t*)(base + (DDZO_F_HAND0 + role) * DDZO_ROW);
    const int8_t* b1 = (const int8_t*)(base + (DDZO_F_RECENT0 + (role + 2) % 3) * DDZO_ROW);
    const int8_t* b2 = (const int8_t*)(base + (DDZO_F_RECENT0 + (role + 1) % 3) * DDZO_ROW);
    int any1 = 0, any2 = 0;
    for (int k = 0; k < NR; ++k) { any1 |= b1[k]; any2 |= b2[k]; }
    const int8_t* last = any1 ? b1 : any2 ? b2 : NULL; /* envi.py:103-109 */
    int32_t left[3];
    for (int r = 0; r < 3; ++r) left[r] = base[(DDZO_F_HAND0 + r) * DDZO_ROW + 15];
    int64_t st[2];
    ids[t] = ddzo_auto_choose(hand, last, left, role, st);
    acc[0] += st[0]; acc[1] += st[1];
  }
  if (stats) { stats[0] = acc[0]; stats[1] = acc[1]; }
}
