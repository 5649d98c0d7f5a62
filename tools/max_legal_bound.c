/*
 * max_legal_bound.c -- one-off proof tool (not product): the largest legal-move list any hand of at most 20 cards can
 * have, by EXHAUSTIVE enumeration of every 20-card count vector (13 ranks 0..4, two jokers 0..1).
 *
 * Why 20 cards and leads suffice: counter_subset (rule_based/utils/utils.py:16-22) is monotone in the hand, so a hand
 * with fewer cards has no more legal leads than any 20-card superset of it (the deck has 54 cards: a superset exists);
 * a follow list is the pass plus a subset of the lead list that excludes at least every single of a different kind,
 * so it is never longer than the lead list.
 *
 * The count is closed-form per category from the masks "rank has >= k cards" -- the same structure the reference's
 * get_action_space enumerates (rule_based/utils/card.py:34-159).  tests/test_rules_bounds.py checks this closed form
 * against the oracle's dense scan (ddzo_legal) on random hands and pins the maximum printed here.
 *
 * -DDDZ_JK_RULES: the optional rule set with the 24 joker-kicker rows (quad + both jokers, two consecutive triples + both jokers:
 * server/mcts/get_moves.py:22-34) -- the two "no joker pair" exclusions (card.py:116,142) are dropped.
 *
 *   gcc -O2 -fopenmp [-DDDZ_JK_RULES] -o /tmp/max_legal_bound tools/max_legal_bound.c && /tmp/max_legal_bound
 */
#ifdef DDZ_JK_RULES
#define JK_EXCLUDED 0
#else
#define JK_EXCLUDED 1
#endif
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static int popc(uint32_t x) { return __builtin_popcount(x); }
static int c2(int n) { return n * (n - 1) / 2; }
static int binom(int n, int k) {
  if (k < 0 || n < k) return 0;
  int r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return r;
}

/* number of legal LEAD moves of the hand with count vector c[15] (card.py:34-159 categories, pass excluded) */
int count_leads(const int8_t* c) {
  uint32_t m[5] = {0, 0, 0, 0, 0};
  for (int r = 0; r < 15; ++r)
    for (int k = 1; k <= 4; ++k)
      if (c[r] >= k) m[k] |= 1u << r;
  const uint32_t M13 = 0x1FFF, M12 = 0x0FFF, JK = 0x6000;
  int n = popc(m[1]) + popc(m[2] & M13) + popc(m[3] & M13) + popc(m[4] & M13); /* singles, pairs, triples, bombs */
  for (int r = 0; r < 13; ++r)
    if (m[3] >> r & 1) {
      n += popc(m[1] & ~(1u << r));       /* 3+1: any other rank, jokers included (card.py:69-73) */
      n += popc(m[2] & M13 & ~(1u << r)); /* 3+2 (card.py:78-82) */
    }
  const int lo[4] = {0, 5, 3, 2}, hi[4] = {0, 12, 10, 6};
  for (int k = 1; k <= 3; ++k) /* chains of singles / pairs / triples over 3..A (card.py:86-105) */
    for (int s = 0; s < 12; ++s)
      for (int L = lo[k]; L <= hi[k] && s + L <= 12; ++L) {
        uint32_t run = ((1u << L) - 1) << s;
        if ((m[k] & M12 & run) == run) ++n;
      }
  for (int s = 0; s < 12; ++s) /* planes with kickers (card.py:110-129) */
    for (int L = 2; L <= 5 && s + L <= 12; ++L) {
      uint32_t run = ((1u << L) - 1) << s;
      if ((m[3] & run) != run) continue;
      uint32_t k1 = m[1] & ~run;
      n += binom(popc(k1), L) - (L == 2 && (k1 & JK) == JK ? JK_EXCLUDED : 0); /* card.py:116: no joker pair as the 2 kickers */
      if (L <= 4) n += binom(popc(m[2] & M13 & ~run), L);
    }
  if ((m[1] & JK) == JK) ++n; /* rocket */
  for (int r = 0; r < 13; ++r)
    if (m[4] >> r & 1) {
      uint32_t k1 = m[1] & ~(1u << r);
      n += c2(popc(k1)) - ((k1 & JK) == JK ? JK_EXCLUDED : 0); /* 4+1+1 (card.py:139-143) */
      n += c2(popc(m[2] & M13 & ~(1u << r)));         /* 4+2+2 (card.py:148-153) */
    }
  return n;
}

#ifndef MAXLEGAL_NO_MAIN
typedef struct { int best; int8_t hand[15]; long long visited; } Res;

static void rec(int8_t* c, int r, int left, Res* res) {
  if (r == 13) {
    for (int a = 0; a <= 1; ++a)
      for (int b = 0; b <= 1; ++b) {
        if (a + b != left) continue;
        c[13] = (int8_t)a; c[14] = (int8_t)b;
        int n = count_leads(c);
        res->visited++;
        if (n > res->best) { res->best = n; memcpy(res->hand, c, 15); }
      }
    return;
  }
  int room = 4 * (12 - r) + 2; /* cards the remaining ranks can still hold */
  for (int v = 0; v <= 4 && v <= left; ++v) {
    if (left - v > room) continue;
    c[r] = (int8_t)v;
    rec(c, r + 1, left - v, res);
  }
}

int main(void) {
  Res tot = {0, {0}, 0};
#pragma omp parallel for schedule(dynamic, 1)
  for (int task = 0; task < 125; ++task) {
    int8_t c[15] = {0};
    c[0] = (int8_t)(task % 5); c[1] = (int8_t)(task / 5 % 5); c[2] = (int8_t)(task / 25);
    Res res = {0, {0}, 0};
    rec(c, 3, 20 - c[0] - c[1] - c[2], &res);
#pragma omp critical
    {
      tot.visited += res.visited;
      if (res.best > tot.best) { tot.best = res.best; memcpy(tot.hand, res.hand, 15); }
    }
  }
  printf("20-card count vectors visited: %lld\nmax legal leads: %d\nhand:", tot.visited, tot.best);
  for (int r = 0; r < 15; ++r) printf(" %d", tot.hand[r]);
  printf("\n");
  return 0;
}
#endif
