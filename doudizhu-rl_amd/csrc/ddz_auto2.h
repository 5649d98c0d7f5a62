// ddz_auto2.h -- k_auto2: the rule-based opponent's search (see ddz_auto.h for what is computed and which reference
// files it restates) with the 64 LANES of the table's wavefront searching different subtrees.
//
// k_auto (ddz_auto.h) walks the decomposition tree with wave-uniform control: one node at a time, ~0.7 us per node, and
// a launch lasts as long as its single heaviest table (10-card hands reach 10^5 nodes: 12 ms).  Here, per table:
//   1. candidates as before (plan_scan lead enumeration -> LDS), then sorted IN PLACE by (lowest rank, id) through
//      registers: cn[pos] = nib, ci[pos] = id | value x 2 | fine_mask bit; bstart[r] = first position of rank r's bucket;
//   2. frontier: starting from the root, lane-parallel expansion passes replace nodes by their children IN PLACE
//      (children counts -> wave prefix sums -> positions), the nodes with the most cards left first, as long as the
//      list fits 448 items.  The list is always an ordered cut of the tree: item k's subtree precedes item k + 1's in
//      depth-first order; small trees end up as a list of finished combinations;
//   3. search: lanes take items (largest index first: the unexpanded, larger subtrees sit at the end) and walk their
//      subtree depth-first with a private stack (one packed u32 per level in LDS) -- one candidate test or one
//      descend / backtrack per loop trip, so that divergent lanes stay cheap;
//   4. every lane keeps (value, item index, move) of its best scored combination; inside an item the strict `>` of
//      rule_based_model.py keeps the first maximum, across items the SMALLER item index wins a tie: together exactly
//      "first maximum in depth-first order", the order the combinations are listed in by decomposer spec v1.
//   5. branch and bound (exact; a2_hopeless below): a greedy descent gives a first finished combination,
//      subtrees whose score bound is STRICTLY below a score already reached are skipped -- unless the caller asked for the
//      node / combination counts of the full enumeration.
// Tables are handed to the waves one by one through a per-launch ticket counter (a decision costs between 10^4 and 10^6
// cycles), for the handle's entry point and the stateless one alike.
// Included by ddz_engine.hip after ddz_auto.h.
#pragma once

// geometry (the -D overrides are tools/auto_occupancy_probe.py's: experiments, not product builds)
#ifndef DDZ_A2_WPB
#define DDZ_A2_WPB 8
#endif
#ifndef DDZ_A2_CAP
#define DDZ_A2_CAP 96
#endif
#ifndef DDZ_A2_BOX
#define DDZ_A2_BOX 128
#endif
#ifndef DDZ_A2_OCC
#define DDZ_A2_OCC 2
#endif
constexpr int A2_CAND = STAGE_CAP;  // candidates per table: the proven maximum of a <= 20-card hand
constexpr int A2_WPB = DDZ_A2_WPB;  // waves per block (17 KB of LDS per wave + the shared record table + the team: 156 KB)
constexpr int A2_TB = A2_WPB * 64;
constexpr int A2_CAP = DDZ_A2_CAP;  // frontier items per table (>= 64: the spare buffer is the mailbox of the donations)
constexpr int A2_KEYLEVELS = 7;   // order keys: 9-bit digits (child position + 1) of the first seven levels of the path
constexpr int A2_PASSES = 6;      // expansion passes at most
constexpr int A2_TARGET = 64;     // ... or until the list feeds 64 lanes (the take / mailbox code is written for exactly one item per lane)
constexpr int A2_DEPTH = 20;      // actions below an item's root (a combination has at most 20 actions)
constexpr int A2_NOFROM = 1023;
constexpr int A2_TICKET_SLOTS = 1024;  // ring of per-launch ticket words (ddz_engine.hip launch_auto)
__device__ uint32_t g_tickets[A2_TICKET_SLOTS];
__device__ int32_t g_device_status;  // status bits of the stateless entry points (ddz_device_status reads and clears it)
#ifndef A2_SCAN_ROUNDS
#define A2_SCAN_ROUNDS 1          // candidate-scan rounds (of four candidates) per search-loop trip
#endif

struct Auto2Wave {              // per wave
  uint64_t cn[A2_CAND];         // candidates: nibble-packed counts (during staging: nib | category << 60, unsorted)
  uint32_t ci[A2_CAND];         // id | (value x 2 & 0xFF) << 14 | fine_mask << 22   (during staging: two u16 arrays)
  uint64_t itA[2][A2_CAP];      // frontier items (double buffer): remaining hand / untouched ranks
  uint64_t itB[2][A2_CAP];      //   pending surplus of the touched ranks (> 10 cards)
  uint32_t itM[2][A2_CAP];      //   (sum2 + 512) | (cvmin & 0xFF) << 10 | actions so far << 18
  uint32_t itI[2][A2_CAP];      //   idmin | first allowed position << 14 (A2_NOFROM = bucket start)
  uint64_t itK[2][A2_CAP];      //   order key: the path of the node (and the cursor of a donated item) as 9-bit digits
  uint32_t stack[A2_DEPTH][64]; // per lane: code | (cvmin & 0xFF) << 10 | idmin << 18 of the level's parent
  int32_t hist[24];             // frontier pass: extra slots wanted by the items with c cards left
  uint16_t tcnt[A2_CAP];        //   children of item i
  uint8_t tcards[A2_CAP];       //   cards left of item i
  uint16_t bstart[16];
};

constexpr int A2_BOX = DDZ_A2_BOX; // items the team's box holds
// A search shared by the waves of a block (k_auto2 "teams"; one per block at a time -- a launch of 65,536 tables opens
// about ten): the decision of wave `owner`, opened to the block's waves that found the queue empty.  Everything but the
// peeks at `open`, `hungry`, `box_n`, `active` and `thr` is read and written under `lock`.
struct A2Team {
  uint32_t lock;
  uint32_t open;       // a search is shared (the context below is published)
  uint32_t owner;      // the wave whose decision it is: the candidates are in ITS Auto2Wave
  uint32_t active;     // members that hold work (own list / lanes in a subtree)
  uint32_t box_n;      // items in the box
  uint32_t hungry;     // members without work, waiting at the box
  uint32_t members;    // the owner + the helpers that joined (a helper's index = its result slot)
  uint32_t finished;   // helpers whose result is written
  uint32_t pad;
  uint32_t flags;      // nosplit | follow << 1 | pass_ok << 2 | prune << 3
  uint64_t thr_bits;   // the best score any member has reached, as the bit pattern of the double: read and written with
                       // 64-bit relaxed atomics only (a2_thr_load / a2_thr_store) -- pruning only, a stale value is a
                       // weaker bound, a torn or compiler-cached one could be a wrong one
  uint64_t hand, bsw0, bsw1, bsw2, sm0, sm1;
  double rp;
  uint32_t esingle, epair;
  uint64_t bA[A2_BOX], bB[A2_BOX], bK[A2_BOX];  // the box: subtrees given to the team (items as in Auto2Wave)
  uint32_t bM[A2_BOX], bI[A2_BOX];
  double rvalue[A2_WPB];  // per member: its best combination (step 4's triple) and its counts
  uint64_t rkey[A2_WPB];
  int32_t rmove[A2_WPB], rcombs[A2_WPB], rnodes[A2_WPB];
};
__device__ __forceinline__ uint32_t a2_peek(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ double a2_thr_load(const A2Team& t) {
  return __longlong_as_double((long long)__hip_atomic_load(&t.thr_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void a2_thr_store(A2Team& t, double v) {
  __hip_atomic_store(&t.thr_bits, (uint64_t)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the team's lock is taken by lane 0 for the whole wave (the other lanes wait at the reconvergence point).  Critical
// sections are a few hundred cycles and every waiting loop peeks before it locks, so the spin is short; the bound is a
// hang guard (status bit 3), never reached in a working launch.
__device__ __forceinline__ void a2_lock(uint32_t* l, int lane, int32_t* status) {
  if (lane == 0) {
    uint32_t spins = 0;
    while (__hip_atomic_exchange(l, 1u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 24)) { if (status) atomicOr(status, 8); break; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void a2_unlock(uint32_t* l, int lane) {
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store(l, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct A2Ctx {                  // wave-uniform facts of the query
  uint64_t hand;
  uint32_t esingle, epair;
  bool nosplit, follow, pass_ok;
  double rp;
};

__device__ __forceinline__ int a2_lowrank(uint64_t x) { return x ? (__builtin_ctzll(x) >> 2) : 16; }
__device__ __forceinline__ bool a2_fits(uint64_t nib, uint64_t a) {
  constexpr uint64_t H8 = 0x8888888888888888ull;
  return (((a | H8) - nib) & H8) == H8;
}
__device__ __forceinline__ uint64_t a2_rankmask(uint64_t nib) {  // 0xF on every rank the action touches
  uint64_t tm = nib | (nib >> 1);
  tm |= tm >> 2;
  tm &= ONES;
  return (tm << 4) - tm;  // x 15 without a 64-bit multiply
}
// child of a regular node after playing candidate `nib`
__device__ __forceinline__ void a2_child(const A2Ctx& q, uint64_t A, uint64_t B, uint64_t nib, uint64_t& A2, uint64_t& B2) {
  if (q.nosplit) {
    const uint64_t rm = a2_rankmask(nib);
    A2 = A & ~rm;
    B2 = B + (A & rm) - nib;  // what the action leaves of the ranks it touches
  } else {
    A2 = A - nib;
    B2 = 0;
  }
}
// >= 0: the lowest uncovered slot is a surplus card of that touched rank; -1: a regular node
__device__ __forceinline__ int a2_pend_rank(uint64_t A, uint64_t B) {
  const int ul = a2_lowrank(A), pl = a2_lowrank(B);
  return pl < ul ? pl : -1;
}
// the augmented pair of slots 2, 3 (card.py:544-547) is an option when two cards of a quad are pending
__device__ __forceinline__ bool a2_pair_option(uint64_t hand, uint64_t B, int pr) {
  return ((B >> (4 * pr)) & 15) == 2 && ((hand >> (4 * pr)) & 15) == 4;
}
// first position of rank r's bucket from three words of seven 9-bit entries (r = 15: the number of candidates).
// Masks, not ?: on the words: a select by a small index is lowered to an indexed SCRATCH array (measured: the search
// loop ran several times slower with one scratch access per node)
__device__ __forceinline__ int a2_bs(uint64_t w0, uint64_t w1, uint64_t w2, int r) {
  const uint64_t w = (w0 & (0ull - (uint64_t)(r < 7))) | (w1 & (0ull - (uint64_t)(r >= 7 && r < 14))) |
                     (w2 & (0ull - (uint64_t)(r >= 14)));
  const int k = r - 7 * (int)(r >= 7) - 7 * (int)(r >= 14);
  return (int)((w >> (9 * k)) & 511u);
}
// order keys: digit `dig` (1..511) of path level `lvl` (0 = the root's choice, most significant); deeper levels have none
__device__ __forceinline__ uint64_t a2_keydigit(int lvl, int dig) {
  return lvl < A2_KEYLEVELS ? (uint64_t)dig << (9 * (A2_KEYLEVELS - 1 - lvl)) : 0ull;
}
__device__ __forceinline__ int a2_code_digit(int code) { return code < 512 ? code + 1 : ((code - 512) & 1) + 1; }
__device__ __forceinline__ int a2_single_v2(int r) { return 2 * (r - 7); }
__device__ __forceinline__ int a2_pair_v2(int r) { return r - 7 > 0 ? 3 * (r - 7) : 2 * (r - 7); }

// Branch and bound (exact): can a node with `nact` actions worth sum2 on its path, cheapest eligible move cvmin and cards
// A + B still to cover reach a score >= thr?  An upper bound of rule_based_model.py:60-86 over every completion:
//   * the values still to come (x 2) sum to at most U2 = floor(sum over the cards of the best value per card any action
//     of the action space pays for that rank): 4.5 (a bomb) below K, the single's own value from K up (fixture G7);
//   * at least Lb more actions follow, Lb = ceil(cards / largest candidate of this decision whose lowest rank is not below
//     the lowest rank left), counted up to 4;
//   * no action still to come is worth less than the cheapest action whose lowest rank is the lowest rank left.
// Every operation below is monotone in these three and rounds where auto_leaf rounds (auto_rounded: no fused multiply-
// add across the product), so the f64 result bounds the f64 score of every completion; a node
// is skipped only when that bound is STRICTLY below a score some finished combination has reached: neither the maximum
// nor a tie for it is lost, whatever the order of the search.
__device__ __forceinline__ bool a2_hopeless(const A2Ctx& q, uint64_t A, uint64_t B, int sum2, int nact, int cvmin,
                                            uint64_t sm0, uint64_t sm1, double thr) {
  if (nact == 0) return false;  // the root: the one action that is the whole hand scores +inf (rule_based_model.py:78-81)
  const uint64_t R = A + B;  // (disjoint ranks)
  const int nR = nib_sum(R), lo = a2_lowrank(R);
  const uint32_t hw = (uint32_t)(R >> 40);  // ranks 10..14: K A 2 BJ CJ
  const int U4 = 9 * nR + 3 * (int)(hw & 15) + 7 * (int)((hw >> 4) & 15) + 11 * (int)((hw >> 8) & 15) +
                 15 * (int)((hw >> 12) & 15) + 19 * (int)((hw >> 16) & 15);
  const uint64_t lom = 0ull - (uint64_t)(lo < 8);
  const int sh = 8 * (lo & 7);
  int M = (int)((((sm0 & lom) | (sm1 & ~lom)) >> sh) & 0xFF);
  if (M < 2) M = 2;
  const int Lb = 1 + (nR > M ? 1 : 0) + (nR > 2 * M ? 1 : 0) + (nR > 3 * M ? 1 : 0);
  const int L = nact + Lb + (q.follow ? 1 : 0);
  const int small_num = (L - 1) - (L >= 14 ? 1 : 0);
  const double total = (double)(sum2 + (U4 >> 1)) * 0.5 - auto_rounded((double)small_num * q.rp);  // (roundings as auto_leaf's)
  // cheapest action (x 2) by lowest rank: -14 -12 -10 -8 -6 -4 -2 0 | 2 3 4 8 10 12 14 (non-decreasing)
  constexpr uint64_t VM0 = 0x00FEFCFAF8F6F4F2ull, VM1 = 0x7F0E0C0A08040302ull;
  const int m2 = (int)(int8_t)((((VM0 & lom) | (VM1 & ~lom)) >> sh) & 0xFF);
  const int cm = (cvmin == AUTO_NONE || m2 < cvmin) ? m2 : cvmin;
  double ub = auto_rounded(total - (double)cm * 0.5) + q.rp;
  if (q.follow && q.pass_ok && total > ub) ub = total;
  return ub < thr;
}

// k_auto_order: the tables whose actor is a rule agent, bucketed by the predicted size of the decision (largest first), for
// k_auto2's queue; every other table gets its -1 here.  Two launches over the tables (the bucket sizes have to be complete
// before a position can be given out): PASS 0 counts, PASS 1 places.  Inside a bucket the order is whatever the atomics
// yield -- a decision does not depend on when it is made.  Device-scope atomics are per block and class.
// The predictor (auto_class; a least-squares fit of log(cycles) of 195,000 measured decisions, R^2 0.85,
// profiles/r03_notes.md): every first or second card of a rank multiplies the cost by ~1.13, every third by ~1.6, every
// fourth by ~1.9 (triples, quads and planes are what gives a hand many ways to split).  With the four hand-size buckets
// of before, the launch's heaviest decisions started anywhere inside the 17-card bucket -- a third of the queue -- and
// the late ones were the tail; a list-scheduling replay of the measured costs puts this order within 1 % of an ideal
// longest-first order on three launches of four (64 buckets: coarser ones lose most of it).
constexpr int AO_CLASSES = 64;
constexpr int AO_BT = 256;
struct AutoOrder {  // one slot of the handle's ring (zeroed before PASS 0)
  uint32_t total;             // all buckets (k_auto2's queue length)
  uint32_t ticket;            // k_auto2's queue head
  uint32_t nsingle;           // hands of 13 + cards: about that many positions at the head of the queue are drawn one by one
  uint32_t pad[13];
  uint32_t cnt[AO_CLASSES];   // bucket sizes
  uint32_t fill[AO_CLASSES];  // PASS 1: positions given out per bucket
};
static_assert(sizeof(AutoOrder) == AO_HDR_BYTES, "slot header (ddz_engine.hip make_layout)");
// class 0 = the largest predicted decisions; `row` = the actor's hand row (15 counts, then the cards left)
__device__ __forceinline__ int auto_class(const uint8_t* row) {
  int s = 0;  // ~100 x the log of the cost, up to a constant
#pragma unroll
  for (int r = 0; r < 15; ++r) {
    const int c = row[r];
    s += c >= 4 ? 132 : c == 3 ? 70 : 12 * c;  // 12 + 12 + 46 + 62
  }
  const int b = s / 7;
  return AO_CLASSES - 1 - (b < AO_CLASSES ? b : AO_CLASSES - 1);
}
template <int PASS>
__global__ __launch_bounds__(AO_BT) void k_auto_order(const uint8_t* __restrict__ state, int64_t T, int auto_roles,
                                                      AutoOrder* __restrict__ slot, int32_t* __restrict__ order,
                                                      int32_t* __restrict__ ids, int64_t* __restrict__ stats) {
  __shared__ uint32_t s_cnt[AO_CLASSES], s_base[AO_CLASSES];
  if (threadIdx.x < AO_CLASSES) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * AO_BT + threadIdx.x;
  int cls = -1;
  bool single = false;
  if (t < T) {
    const uint8_t* row = state + t * STATE_ROW_BYTES;
    const uint4 meta = *(const uint4*)(row + DDZ_F_META * 16);
    const int role = meta.x & 0xFF;
    const bool active = ((meta.y >> 16) & 0xFF) && !((meta.x >> 8) & 0xFF) && role <= 2 && ((auto_roles >> role) & 1);
    if (active) {
      cls = auto_class(row + (DDZ_F_HAND0 + role) * 16);
      single = row[(DDZ_F_HAND0 + role) * 16 + 15] >= 13;  // cards left (envi.py:23)
    } else if (PASS == 0) {
      ids[t] = -1;  // not a rule agent's turn / frozen table
      if (stats) { stats[2 * t] = 0; stats[2 * t + 1] = 0; }
    }
  }
  uint32_t rank = 0;
  if (cls >= 0) rank = atomicAdd(&s_cnt[cls], 1u);
  __syncthreads();
  if (PASS == 0) {
    if (threadIdx.x < AO_CLASSES && s_cnt[threadIdx.x]) atomicAdd(&slot->cnt[threadIdx.x], s_cnt[threadIdx.x]);
    const uint64_t act_m = __ballot(cls >= 0), one_m = __ballot(single);  // per wave
    if ((threadIdx.x & 63) == 0 && act_m) {
      atomicAdd(&slot->total, (uint32_t)__popcll(act_m));
      if (one_m) atomicAdd(&slot->nsingle, (uint32_t)__popcll(one_m));
    }
  } else {
    if (threadIdx.x < AO_CLASSES) {
      uint32_t b = 0;
      for (int c = 0; c < (int)threadIdx.x; ++c) b += slot->cnt[c];  // (complete: written by the previous launch)
      s_base[threadIdx.x] = b + (s_cnt[threadIdx.x] ? atomicAdd(&slot->fill[threadIdx.x], s_cnt[threadIdx.x]) : 0u);
    }
    __syncthreads();
    if (cls >= 0) order[s_base[cls] + rank] = (int32_t)t;
  }
}

#ifdef DDZ_STAMP  // team counters behind the last table's stamps: teams opened, helper stints, items put / taken, nodes of helpers / owners
#define A2DBG(k, v) do { if (g_stamps && lane == 0) atomicAdd(&g_stamps[16 * a.T + (k)], (unsigned long long)(v)); } while (0)
#else
#define A2DBG(k, v) do { } while (0)
#endif
template <bool STATE>
__global__ __launch_bounds__(A2_TB, DDZ_A2_OCC) void k_auto2(AutoArgs a) {
  __shared__ HotTabT<false> hot;
  __shared__ Auto2Wave s_w[A2_WPB];
  __shared__ A2Team s_team;
  __shared__ uint32_t s_inflight, s_drained;  // waves that may still draw from the queue / that found it empty
  __shared__ uint32_t s_first_done;           // the block's team-first decision (below) is finished
  const int lane = threadIdx.x & 63;
  const int wv = (int)rfl(threadIdx.x >> 6);
  hot_fill<A2_TB>(hot);
  for (int i = threadIdx.x; i < (int)(sizeof(s_team) / 4); i += A2_TB) ((uint32_t*)&s_team)[i] = 0;
  if (threadIdx.x == 0) { s_inflight = A2_WPB; s_drained = 0; s_first_done = 0; }
  __syncthreads();
  Auto2Wave& W = s_w[wv];
  uint16_t* svl = (uint16_t*)W.ci;               // staging views of ci: value | len << 8 ...
  uint16_t* sid = (uint16_t*)W.ci + A2_CAND;      // ... and the canonical ids
  constexpr uint64_t NIBM = 0x0FFFFFFFFFFFFFFFull;
  // tables are handed out one by one (a decision costs between 10^4 and 10^6 cycles: a fixed share per wave would end the
  // launch with its unluckiest wave): *ticket = next table.  The word belongs to THIS launch alone (the host takes it from
  // a ring of device globals and zeroes it on the launch stream, ddz_engine.hip launch_auto): nothing to re-arm, nothing a
  // failed or concurrent launch can leave behind.  Every wave leaves when it draws a ticket >= T: the grid always drains.
  // Guided shares: a draw takes up to 8 consecutive tables while many are left and one at a time near the end (the size
  // follows from the previous draw's return value: no extra load).  Device-scope atomics on ONE address serialise at
  // ~15 ns each on this multi-XCD part (measured, profiles/r03_notes.md): one draw per table would keep that address
  // busy for 1.0 ms of a 1.5 ms launch at 65,536 tables.
  // Heaviest first: the size of a decision grows ~1.6 x per card of the hand (a 17-card hand: 2,500 search nodes on
  // average, 20,000 at most; a 10-card hand: 90) and the largest single decision of a launch lasts as long as a balanced
  // launch of everything else -- drawn late, it WAS the launch's tail.  With `order` (k_auto_order: the agent's tables
  // bucketed by hand size, largest hands first) the queue position k means table order[k], so the tail is made of the
  // cheapest decisions.
  const int64_t nwaves = (int64_t)gridDim.x * A2_WPB;
  const int64_t NQ = a.order ? (int64_t)rfl(a.order_hdr[0]) : a.T;  // length of the queue (AutoOrder::total)
  // ... whose first NH positions are (about) the hands of 13 + cards: one per draw (a share of eight of those could be a
  // millisecond)
  const int64_t NH = a.order ? (int64_t)rfl(a.order_hdr[2]) : 0;  // AutoOrder::nsingle
  // Team first (round 4): the KT predicted-heaviest decisions -- the head of the ordered queue, one per block -- are searched
  // by the WHOLE block from their first trip: wave 0 owns position blockIdx.x, opens its team as soon as a lane has a
  // level to give, and the block's other waves start as its helpers instead of drawing tickets (a decision of 10^4 nodes
  // alone on one wave lasted as long as everything else of the launch per wave: it was the launch, wherever it sat in the
  // queue).  The ticket then hands out the positions from KT on.
  const int64_t KT = (a.order && a.teams && a.team_first) ? (NH < (int64_t)gridDim.x ? NH : (int64_t)gridDim.x) : 0;
  bool first_owner = wv == 0 && (int64_t)blockIdx.x < KT;    // this wave owns the block's team-first decision
  bool first_helper = wv != 0 && (int64_t)blockIdx.x < KT;   // ... helps it until it is finished
  int64_t tnext = 0, tend = 0;  // queue positions in hand: [tnext, tend)
  if (first_owner) { tnext = (int64_t)blockIdx.x; tend = tnext + 1; }
  bool first_started = false;   // the owner is inside its team-first decision (whichever way it leaves it, the loop's top sees it)
  int64_t seen = 0;             // the queue's head as of this wave's last draw
  // Teams: one decision in a few hundred walks 10^4 nodes, and whenever a launch holds one, that search -- alone on its
  // wave, every other wave done -- WAS the rest of the launch (a third of the average launch).  A wave that finds the
  // queue empty therefore does not leave: it helps the searches still running in its block.  The owner of a search that
  // has lasted a few trips opens its team (context + candidates stay where they are, in the owner's LDS); members without
  // work wait at the team's box, members with work put unexplored siblings there (the same hand-over as between the
  // lanes of a wave: keys keep the depth-first order), and the owner keeps the best of all members under step 4's rule.
  // Exact in any order (a2_hopeless prunes strictly below a score already reached).  The search is over when no member
  // holds work and the box is empty; a wave leaves the kernel when the queue is empty and no wave of its block can
  // still open a team.
  bool drained = false;
  uint32_t polls = 0;           // hang guard of the waiting loops (status bit 3; never reached in a working launch)
  constexpr uint32_t A2_POLL_LIMIT = 1u << 24;
  for (;;) {
    if (first_started) {        // the block's team-first decision is finished: its helpers may go to the queue now
      first_started = false; first_owner = false;
      if (lane == 0) __hip_atomic_store(&s_first_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // (taking the next ticket early, to fetch its state rows while this table is decided, was slower: a wave inside a
    // 10^6-cycle decision then holds its next table hostage)
    if (!drained && !first_helper && tnext >= tend) {
      int64_t sz = (NQ - seen) / ((a.order ? 4 : 8) * nwaves);
      sz = (sz < 1 || seen < NH) ? 1 : sz > 8 ? 8 : sz;
      uint32_t tk = 0;
      if (lane == 0) tk = atomicAdd(a.ticket, (uint32_t)sz);
      seen = KT + (int64_t)rfl(tk);
      if (seen >= NQ) {
        if (!a.teams) break;
        drained = true;
        if (lane == 0) { atomicAdd(&s_drained, 1u); atomicSub(&s_inflight, 1u); }
      } else {
        tnext = seen;
        tend = seen + sz < NQ ? seen + sz : NQ;
      }
    }
    int own = wv, my_slot = 0;  // whose search this wave works on, and as which member
    if (drained || first_helper) {  // join a team of the block, or leave when none can open any more
      own = -1;
      // (peeks first: a wave that cannot join must not keep the lock busy -- seven such waves starved the owner)
      if (a2_peek(&s_team.open) && (a2_peek(&s_team.active) != 0 || a2_peek(&s_team.box_n) != 0) &&
          a2_peek(&s_team.members) < A2_WPB) {
        A2Team& TK = s_team;
        a2_lock(&TK.lock, lane, a.status);
        if (TK.open && (TK.active != 0 || TK.box_n != 0) && TK.members < A2_WPB) {
          my_slot = (int)rfl(TK.members);
          own = (int)rfl(TK.owner);
          if (lane == 0) { TK.members = my_slot + 1; TK.hungry += 1; }
        }
        a2_unlock(&TK.lock, lane);
      }
      if (own < 0) {
        if (first_helper) {     // no team to join (yet, or any more): on to the queue once the owner is through
          if (a2_peek(&s_first_done) != 0) { first_helper = false; polls = 0; continue; }
          if (++polls > A2_POLL_LIMIT) { if (lane == 0 && a.status) atomicOr(a.status, 8); first_helper = false; continue; }
          __builtin_amdgcn_s_sleep(4);
          continue;
        }
        if (a2_peek(&s_inflight) == 0) break;
        if (++polls > A2_POLL_LIMIT) { if (lane == 0 && a.status) atomicOr(a.status, 8); break; }
        __builtin_amdgcn_s_sleep(16);
        continue;
      }
    }
    const bool helper = own != wv;
    const int64_t t = helper ? 0 : a.order ? (int64_t)rfl((uint32_t)a.order[tnext]) : tnext;
    if (!helper) ++tnext;
    if (first_owner) first_started = true;
    // what the search below works with: the owner's own preparation (1, 1b, 2), or the context the team's owner published
    A2Ctx q;
    uint64_t bsw0 = 0, bsw1 = 0, bsw2 = 0;  // bucket starts of the sorted candidates (seven 9-bit entries per word)
    bool PRUNE = false;
    uint64_t sm0 = 0, sm1 = 0;  // per lowest rank r: the largest candidate whose lowest rank is >= r (bytes)
    double thr = -__builtin_inf();  // a score some finished combination reaches
    int nitems = 0, cur = 0;        // the wave's list of subtrees: items [0, nitems) of buffer `cur`
    int nodes_l = 0, combs_l = 0;
#ifdef DDZ_STAMP
    unsigned long long tq[6] = {0, 0, 0, 0, 0, 0}, tq_enum = 0, tq_sort = 0, tq_bounds = 0;
    int n_fastpass = 0, n_genpass = 0, nitems_final = 0;
#endif
    int n = 0;
    if (!helper) {
      // ---- the query: hand, combo to beat, cards left, acting role (as k_auto)
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
        const uint32_t qq = rfl(a.info[t]);
        left0 = qq & 0xFF; left1 = (qq >> 8) & 0xFF; left2 = (qq >> 16) & 0xFF; role = (int)(qq >> 24);
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
#ifdef DDZ_STAMP
      tq[0] = __builtin_amdgcn_s_memtime();
#endif
      const Follow f = follow_of(linfo);
      q.hand = hand;
      q.follow = !f.lead;
      // rule_based_model.py:56-57 (the role test is the reference's own: role 0 looks at lord and down, the others at up)
      int min_opp = role == 0 ? (left1 < left2 ? left1 : left2) : left0;
      if (min_opp > 23) min_opp = 23;
      q.rp = a.rp[min_opp];
      q.pass_ok = min_opp > 4;
      q.nosplit = nib_sum(hand) > 10;  // decomposer.py:18
      q.esingle = !q.follow ? M15 : (f.lc == SINGLE ? gt_mask(f.lv) : 0u);
      q.epair = !q.follow ? M13 : (f.lc == DOUBLE ? (gt_mask(f.lv) & M13) : 0u);

      // ---- 1. candidates: every action that fits the hand (decomposer.py:19-28 valid_row_idx / :50-55 valid)
      __builtin_amdgcn_wave_barrier();
      {
        const Out o{nullptr, nullptr, 0, 0, W.cn, svl, sid};
        Pick pk{-1, 0, 0, 0, 0};
        n = plan_scan<EM_STAGE, true>(hand, mk_info(EMPTY, 0, 1), hot, lane, o, pk);
      }
      __builtin_amdgcn_wave_barrier();
#ifdef DDZ_STAMP
      tq_enum = __builtin_amdgcn_s_memtime();
#endif
      if (n > A2_CAND) {  // cannot happen for a <= 20-card hand (tools/max_legal_bound.c)
        if (lane == 0) { if (a.status) atomicOr(a.status, 2); a.ids[t] = DDZ_AUTO_INVALID; }
        continue;
      }
      // per candidate (lane holds entries lane, lane + 64, ...: at most 8): value x 2, fine_mask, lowest rank, cards.
      // Counting sort by lowest rank: bucket sizes from an LDS histogram (one ds_add per round: the order of the adds does
      // not matter for a count), bucket starts from one DPP scan, positions inside a bucket from one ballot + mbcnt per rank
      // that is PRESENT (a hand has candidates on 5-8 lowest ranks, not 15) -- id order is kept inside a bucket.
      constexpr int PER = (A2_CAND + 63) / 64;
      uint64_t e_nib[PER];
      uint32_t e_ci[PER];
      int e_lr[PER], e_pos[PER], e_cards[PER];
      if (lane < 16) W.hist[lane] = 0;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int j = u * 64 + lane;
        e_lr[u] = 16; e_nib[u] = 0; e_ci[u] = 0; e_pos[u] = 0; e_cards[u] = 0;
        if (u * 64 < n && j < n) {  // (the first test is wave-uniform: whole rounds without candidates are skipped)
          const uint64_t e = W.cn[j];
          const uint64_t nib = e & NIBM;
          const int cat = (int)(e >> 60), vl = svl[j], val = vl & 0xFF, len = vl >> 8;
          const int v2 = auto_val2(nib, cat, val, len);
          const bool el = !q.follow || auto_beats(cat, val, len, f);
          // the rule agent works on card.py's 13,527 rows in every build: the joker-kicker extras of the other rule set
          // (ids >= 13527) get a row that never fits
          e_nib[u] = (DDZ_NATIVE_JOKER_KICKERS && sid[j] >= DDZ_NUM_ACTIONS) ? NIBM : nib;
          e_ci[u] = (uint32_t)sid[j] | ((uint32_t)(v2 & 0xFF) << 14) | (el ? 1u << 22 : 0u);
          e_lr[u] = __builtin_ctzll(nib) >> 2;
          e_cards[u] = nib_sum(nib);
          atomicAdd(&W.hist[e_lr[u]], 1);
        }
      }
      __builtin_amdgcn_wave_barrier();  // every entry is in registers: the arrays may be overwritten in sorted order
      const int cnt_lane = lane < 15 ? W.hist[lane] : 0;  // lane r: number of candidates whose lowest rank is r
      const int start_lane = wave_scan_add(cnt_lane) - cnt_lane;  // lane 15: n
      const uint32_t present = (uint32_t)__ballot(cnt_lane > 0);
      int run_lane = start_lane;
      if (lane < 16) W.bstart[lane] = (uint16_t)start_lane;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        if (u * 64 < n) {
          for (uint32_t pm = present; pm; pm &= pm - 1) {  // wave-uniform
            const int r = __builtin_ctz(pm);
            const uint64_t m = __ballot(e_lr[u] == r);
            if (m) {
              const int base = (int)rl((uint32_t)run_lane, r);
              const int pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
              if (e_lr[u] == r) e_pos[u] = base + pre;
              if (lane == r) run_lane += __popcll(m);
            }
          }
          if (e_lr[u] < 16) { W.cn[e_pos[u]] = e_nib[u]; W.ci[e_pos[u]] = e_ci[u]; }
        }
      }
      __builtin_amdgcn_wave_barrier();
      // the bucket starts once more as three wave-uniform words of seven 9-bit entries: a lookup by a lane's own rank is
      // a shift instead of an LDS round trip in the search loop
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint64_t v = (uint64_t)(rl((uint32_t)start_lane, r) & 511u);
        if (r < 7) bsw0 |= v << (9 * r);
        else if (r < 14) bsw1 |= v << (9 * (r - 7));
        else bsw2 |= v << (9 * (r - 14));
      }

#ifdef DDZ_STAMP
      tq_sort = __builtin_amdgcn_s_memtime();
      tq_bounds = tq_sort;
#endif
      // ---- 1b. branch and bound (off when the caller wants the exact node / combination counts of the full enumeration)
      PRUNE = a.stats == nullptr;
      if (PRUNE) {
        int mx = 0;  // lane r: the largest candidate of bucket r ...
        {
          const int lo_ = lane < 15 ? start_lane : 0, hi_ = lane < 15 ? start_lane + cnt_lane : 0;
          for (int pp = lo_; pp < hi_; ++pp) {
            const int c_ = nib_sum(W.cn[pp]);
            mx = c_ > mx ? c_ : mx;
          }
        }
#pragma unroll
        for (int r = 14; r >= 0; --r) {  // ... and of every bucket from r up
          const int v = (int)rl((uint32_t)mx, r);
          static_assert(STAGE_CAP <= 1023, "positions fit the 10 low bits of the greedy key");
          const int prev = r == 14 ? 0 : (int)((r + 1 < 8 ? sm0 >> (8 * (r + 1)) : sm1 >> (8 * (r + 1 - 8))) & 0xFF);
          const uint64_t m_ = (uint64_t)(v > prev ? v : prev);
          if (r < 8) sm0 |= m_ << (8 * r); else sm1 |= m_ << (8 * (r - 8));
        }
#ifdef DDZ_STAMP
        tq_bounds = __builtin_amdgcn_s_memtime();
#endif
        // a first finished combination: always the largest candidate that fits (few actions = a high score)
        uint64_t gA = hand, gB = 0;
        int gs = 0, gn = 0, gcv = AUTO_NONE, gid = 0, gfrom = A2_NOFROM;
        bool stuck = false;
        for (int step = 0; step < 24 && (gA | gB) != 0 && !stuck; ++step) {
          const int pr = a2_pend_rank(gA, gB);
          int v2, id;
          bool el;
          if (pr >= 0) {
            if (a2_pair_option(hand, gB, pr)) { gB -= 2ull << (4 * pr); v2 = a2_pair_v2(pr); el = (q.epair >> pr) & 1u; id = 16 + pr; }
            else { gB -= 1ull << (4 * pr); v2 = a2_single_v2(pr); el = (q.esingle >> pr) & 1u; id = 1 + pr; }
          } else {
            const int ul = a2_lowrank(gA);
            const int lo_ = gfrom != A2_NOFROM ? gfrom : a2_bs(bsw0, bsw1, bsw2, ul);
            int bk = -1;  // (cards << 10) | (1023 - position): the largest, on ties the first
            // the candidates are still in the registers of the sort: bucket ul = the entries whose lowest rank is ul
#pragma unroll
            for (int u = 0; u < PER; ++u) {
              if (u * 64 < n) {
                const int k_ = (e_cards[u] << 10) | (1023 - e_pos[u]);
                if (e_lr[u] == ul && e_pos[u] >= lo_ && a2_fits(e_nib[u], gA) && k_ > bk) bk = k_;
              }
            }
            bk = wave_max_i32(bk);  // (DPP: wave-uniform)
            if (bk < 0) { stuck = true; break; }  // (<= 10 cards: the row index may not decrease within a rank)
            const int ps = 1023 - (bk & 1023);
            const uint64_t nb = W.cn[ps];
            const uint32_t ci = W.ci[ps];
            uint64_t A2, B2;
            a2_child(q, gA, gB, nb, A2, B2);
            gfrom = (!q.nosplit && a2_lowrank(A2) == ul) ? ps : A2_NOFROM;
            gA = A2; gB = B2;
            v2 = (int)(int8_t)((ci >> 14) & 0xFF); el = (ci >> 22) & 1u; id = (int)(ci & 0x3FFF);
          }
          gs += v2; gn += 1;
          if (el && (gcv == AUTO_NONE || v2 < gcv)) { gcv = v2; gid = id; }
        }
        if (!stuck && (gA | gB) == 0) {
          AutoBest gb{-__builtin_inf(), -1};
          auto_leaf(gb, gs, gcv, gid, gn, q.follow, q.pass_ok, q.rp);
          if (gb.move >= 0) thr = gb.value;
        }
      }

#ifdef DDZ_STAMP
      tq[1] = __builtin_amdgcn_s_memtime();
#endif
      // ---- 2. frontier: an ordered cut of the tree.  Every pass replaces nodes by their children in place (order kept),
      // the nodes with the MOST cards left first (their subtrees are the largest), as long as the list fits.
      nitems = 1;
      nodes_l = lane == 0 ? 1 : 0;
      if (lane == 0) {
        W.itA[0][0] = hand; W.itB[0][0] = 0;
        W.itM[0][0] = 512u | ((uint32_t)(AUTO_NONE & 0xFF) << 10);
        W.itI[0][0] = (uint32_t)A2_NOFROM << 14;
        W.itK[0][0] = 0;
      }
      __builtin_amdgcn_wave_barrier();
      // enough subtrees to feed 64 lanes; the heavy trees (many candidates) get the whole list for balance
      const int target = A2_TARGET;
      int pass0 = 0;
      {  // the first pass has ONE item, the root, and every candidate of its bucket fits: one lane per child
        const int ul = a2_lowrank(hand);
        const int lo = a2_bs(bsw0, bsw1, bsw2, ul), hi = a2_bs(bsw0, bsw1, bsw2, ul + 1);
        if (hi - lo >= 1 && hi - lo <= A2_CAP && A2_PASSES > 0) {
          int wb = 0;
          for (int p0 = lo; p0 < hi; p0 += 64) {
            const int pp = p0 + lane;
            bool keep = false;
            uint64_t A2 = 0, B2 = 0;
            uint32_t ci = 0;
            int v2 = 0;
            bool el = false;
            if (pp < hi) {
              const uint64_t nib = W.cn[pp];
              ci = W.ci[pp];
              a2_child(q, hand, 0ull, nib, A2, B2);
              v2 = (int)(int8_t)((ci >> 14) & 0xFF);
              el = (ci >> 22) & 1u;
              keep = !(PRUNE && (A2 | B2) != 0 && a2_hopeless(q, A2, B2, v2, 1, el ? v2 : AUTO_NONE, sm0, sm1, thr));
            }
            const uint64_t km = __ballot(keep);
            if (keep) {
              const int w = wb + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
              const bool same = !q.nosplit && a2_lowrank(A2) == ul;
              W.itA[1][w] = A2; W.itB[1][w] = B2;
              W.itM[1][w] = (uint32_t)((v2 + 512) & 1023) | ((uint32_t)((el ? v2 : AUTO_NONE) & 0xFF) << 10) | (1u << 18);
              W.itI[1][w] = (uint32_t)(el ? (int)(ci & 0x3FFF) : 0) | ((uint32_t)(same ? pp : A2_NOFROM) << 14);
              W.itK[1][w] = a2_keydigit(0, pp + 1);
              ++nodes_l;
            }
            wb += __popcll(km);
          }
          nitems = wb;
          cur = 1;
          pass0 = 1;
          __builtin_amdgcn_wave_barrier();
        }
      }
      const int npass = A2_PASSES;
      for (int pass = pass0; pass < npass && nitems < target; ++pass) {
#ifndef DDZ_A2_NO_FASTPASS
        {  // the common pass: every item has its own lane (nitems < target = 64) and ALL children fit the list -- count,
           // scan and emit from registers, the fitting candidates remembered as a bit mask (no class histogram, no
           // tcnt / tcards round trip).  Same list as the general pass below with tau = 0.
          const bool mine = lane < nitems;
          uint64_t A = 0, B = 0, K = 0, fm = 0;
          uint32_t M = 0, I = 0;
          if (mine) { A = W.itA[cur][lane]; B = W.itB[cur][lane]; M = W.itM[cur][lane]; I = W.itI[cur][lane]; K = W.itK[cur][lane]; }
          const bool open = mine && (A | B) != 0;
          int cnt = mine ? 1 : 0, pr = -1, lo = 0, ul = 0;
          bool wide = false;
          if (open) {
            pr = a2_pend_rank(A, B);
            if (pr >= 0) {
              cnt = 1 + (a2_pair_option(q.hand, B, pr) ? 1 : 0);
            } else {
              ul = a2_lowrank(A);
              const int from = (int)((I >> 14) & 1023);
              lo = from != A2_NOFROM ? from : a2_bs(bsw0, bsw1, bsw2, ul);
              const int hi = a2_bs(bsw0, bsw1, bsw2, ul + 1);
              if (hi - lo > 64) {
                wide = true;
              } else {
                for (int p = lo; p < hi; ++p)
                  if (a2_fits(W.cn[p], A)) fm |= 1ull << (p - lo);
              }
              cnt = __popcll(fm);
            }
          }
          const int grow = wave_sum_i32(open ? cnt - 1 : 0);
          if (__ballot(wide) == 0 && nitems + grow <= A2_CAP) {
            if (__ballot(open) == 0) break;  // only finished combinations are left: the list in `cur` stands
            const int nxt = cur ^ 1;
            const int oin = wave_scan_add(cnt);
            int w = oin - cnt;
            if (mine && !open) {
              W.itA[nxt][w] = A; W.itB[nxt][w] = B; W.itM[nxt][w] = M; W.itI[nxt][w] = I; W.itK[nxt][w] = K;
            } else if (open) {
              const int sum2 = (int)(M & 1023) - 512, cvmin = (int)(int8_t)((M >> 10) & 0xFF), nact = (int)((M >> 18) & 31);
              const int idmin = (int)(I & 0x3FFF);
              auto put = [&](uint64_t A2, uint64_t B2, int v2, bool el, int id, int fromc, int dig) {
                int cv = cvmin, im = idmin;
                if (el && (cvmin == AUTO_NONE || v2 < cvmin)) { cv = v2; im = id; }
                W.itA[nxt][w] = A2; W.itB[nxt][w] = B2;
                W.itM[nxt][w] = (uint32_t)((sum2 + v2 + 512) & 1023) | ((uint32_t)(cv & 0xFF) << 10) | ((uint32_t)(nact + 1) << 18);
                W.itI[nxt][w] = (uint32_t)im | ((uint32_t)fromc << 14);
                W.itK[nxt][w] = K | a2_keydigit(nact, dig);
                ++w;
                ++nodes_l;
              };
              if (pr >= 0) {
                put(A, B - (1ull << (4 * pr)), a2_single_v2(pr), (q.esingle >> pr) & 1u, 1 + pr, A2_NOFROM, 1);
                if (cnt == 2) put(A, B - (2ull << (4 * pr)), a2_pair_v2(pr), (q.epair >> pr) & 1u, 16 + pr, A2_NOFROM, 2);
              } else {
                for (uint64_t m = fm; m; m &= m - 1) {
                  const int pp = lo + __builtin_ctzll(m);
                  const uint64_t nib = W.cn[pp];
                  const uint32_t ci = W.ci[pp];
                  uint64_t A2, B2;
                  a2_child(q, A, B, nib, A2, B2);
                  const bool same = !q.nosplit && a2_lowrank(A2) == ul;
                  put(A2, B2, (int)(int8_t)((ci >> 14) & 0xFF), (ci >> 22) & 1u, (int)(ci & 0x3FFF), same ? pp : A2_NOFROM, pp + 1);
                }
              }
            }
            __builtin_amdgcn_wave_barrier();
            nitems = (int)rl((uint32_t)oin, 63);
            cur = nxt;
#ifdef DDZ_STAMP
            ++n_fastpass;
#endif
            continue;
          }
        }
#endif
#ifdef DDZ_STAMP
        ++n_genpass;
#endif
        // (a) children count and cards left of every item; extra slots wanted per cards-left class
        if (lane < 24) W.hist[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        for (int i0 = 0; i0 < nitems; i0 += 64) {
          const int i = i0 + lane;
          if (i < nitems) {
            const uint64_t A = W.itA[cur][i], B = W.itB[cur][i];
            const uint32_t I = W.itI[cur][i];
            int cnt = 1, cards = 0;  // a finished combination stays as it is
            if ((A | B) != 0) {
              cards = nib_sum(A) + nib_sum(B);
              const int pr = a2_pend_rank(A, B);
              if (pr >= 0) {
                cnt = 1 + (a2_pair_option(q.hand, B, pr) ? 1 : 0);
              } else {
                const int ul = a2_lowrank(A), from = (int)((I >> 14) & 1023);
                const int lo = from != A2_NOFROM ? from : a2_bs(bsw0, bsw1, bsw2, ul), hi = a2_bs(bsw0, bsw1, bsw2, ul + 1);
                cnt = 0;
                for (int p = lo; p < hi; ++p) cnt += a2_fits(W.cn[p], A) ? 1 : 0;
              }
              atomicAdd(&W.hist[cards], cnt - 1);
            }
            W.tcnt[i] = (uint16_t)cnt;
            W.tcards[i] = (uint8_t)cards;
          }
        }
        __builtin_amdgcn_wave_barrier();
        // (b) classes above tau expand entirely, class tau as far as the budget goes (a prefix, in item order)
        const int budget = A2_CAP - nitems;
        int tau = 21, used = 0;
        {
          // lane c holds hist[c]; suffix sums over c = 20 .. 1; tau = the first class (from 20 down) whose suffix sum
          // exceeds the budget (0: everything fits), used = the sum of the classes above it
          const int hc = (lane >= 1 && lane <= 20) ? W.hist[lane] : 0;
          const int pre_ = wave_scan_add(hc);  // inclusive prefix sums (DPP)
          const int suf = (int)rl((uint32_t)pre_, 63) - pre_ + hc;  // inclusive suffix sum: sum of hist[lane .. 20]
          const uint32_t over = (uint32_t)__ballot(lane >= 1 && lane <= 20 && suf > budget);
          if (over) {
            tau = 31 - __builtin_clz(over);
            used = (int)rl((uint32_t)suf, tau) - (int)rl((uint32_t)hc, tau);
          } else {
            tau = 0;
            used = (int)rl((uint32_t)suf, 1);
          }
        }
        int mleft = budget - used;  // for the marginal class tau (when tau >= 1 and it did not fit entirely)
        const bool marginal_partial = tau >= 1 && used + W.hist[tau] > budget;
        // (c) emit
        const int nxt = cur ^ 1;
        int wbase = 0;
        bool any = false;
        for (int i0 = 0; i0 < nitems; i0 += 64) {
          const int i = i0 + lane;
          const bool mine = i < nitems;
          uint64_t A = 0, B = 0, K = 0;
          uint32_t M = 0, I = 0;
          int cnt = 0, cards = 0;
          if (mine) {
            A = W.itA[cur][i]; B = W.itB[cur][i]; M = W.itM[cur][i]; I = W.itI[cur][i]; K = W.itK[cur][i];
            cnt = W.tcnt[i]; cards = W.tcards[i];
          }
          const bool leaf = (A | B) == 0;
          bool expand = mine && !leaf && (marginal_partial ? cards > tau : cards >= tau);
          if (marginal_partial) {  // members of the marginal class, in order, while their growth still fits
            const bool cand = mine && !leaf && cards == tau;
            const int g = cand ? cnt - 1 : 0;
            const int gin = wave_scan_add(g);
            if (cand && gin <= mleft) expand = true;
            // the prefix property: stop at the first member that does not fit
            const uint64_t bad = __ballot(cand && gin > mleft);
            if (bad) {
              mleft = -1;  // nothing behind the first misfit expands any more (gin is non-decreasing: a prefix)
            } else {
              mleft -= (int)rl((uint32_t)gin, 63);
            }
          }
          const int oc = mine ? (expand ? cnt : 1) : 0;
          const int oin = wave_scan_add(oc);
          int w = wbase + oin - oc;
          wbase += (int)rl((uint32_t)oin, 63);
          any = any || __ballot(expand) != 0;
          if (mine) {
            if (!expand) {
              W.itA[nxt][w] = A; W.itB[nxt][w] = B; W.itM[nxt][w] = M; W.itI[nxt][w] = I; W.itK[nxt][w] = K;
            } else {
              const int sum2 = (int)(M & 1023) - 512, cvmin = (int)(int8_t)((M >> 10) & 0xFF), nact = (int)((M >> 18) & 31);
              const int idmin = (int)(I & 0x3FFF);
              auto put = [&](uint64_t A2, uint64_t B2, int v2, bool el, int id, int fromc, int dig) {
                int cv = cvmin, im = idmin;
                if (el && (cvmin == AUTO_NONE || v2 < cvmin)) { cv = v2; im = id; }
                W.itA[nxt][w] = A2; W.itB[nxt][w] = B2;
                W.itM[nxt][w] = (uint32_t)((sum2 + v2 + 512) & 1023) | ((uint32_t)(cv & 0xFF) << 10) | ((uint32_t)(nact + 1) << 18);
                W.itI[nxt][w] = (uint32_t)im | ((uint32_t)fromc << 14);
                W.itK[nxt][w] = K | a2_keydigit(nact, dig);
                ++w;
                ++nodes_l;
              };
              const int pr = a2_pend_rank(A, B);
              if (pr >= 0) {
                put(A, B - (1ull << (4 * pr)), a2_single_v2(pr), (q.esingle >> pr) & 1u, 1 + pr, A2_NOFROM, 1);
                if (a2_pair_option(q.hand, B, pr)) put(A, B - (2ull << (4 * pr)), a2_pair_v2(pr), (q.epair >> pr) & 1u, 16 + pr, A2_NOFROM, 2);
              } else {
                const int ul = a2_lowrank(A), from = (int)((I >> 14) & 1023);
                const int lo = from != A2_NOFROM ? from : a2_bs(bsw0, bsw1, bsw2, ul), hi = a2_bs(bsw0, bsw1, bsw2, ul + 1);
                for (int pp = lo; pp < hi; ++pp) {
                  const uint64_t nib = W.cn[pp];
                  if (!a2_fits(nib, A)) continue;
                  const uint32_t ci = W.ci[pp];
                  uint64_t A2, B2;
                  a2_child(q, A, B, nib, A2, B2);
                  const bool same = !q.nosplit && a2_lowrank(A2) == ul;  // <= 10 cards: the row index may not decrease
                  put(A2, B2, (int)(int8_t)((ci >> 14) & 0xFF), (ci >> 22) & 1u, (int)(ci & 0x3FFF), same ? pp : A2_NOFROM, pp + 1);
                }
              }
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
        if (!any) break;  // nothing fitted (or only finished combinations are left): the list in `cur` stands
        nitems = wbase;
        cur = nxt;
      }

#ifdef DDZ_STAMP
      tq[2] = __builtin_amdgcn_s_memtime();
      nitems_final = nitems;
#endif
    } else {
      const A2Team& TO = s_team;  // (published before `open` was set, constant while the team is open)
      q.hand = TO.hand; q.esingle = TO.esingle; q.epair = TO.epair; q.rp = TO.rp;
      const uint32_t fl = TO.flags;
      q.nosplit = fl & 1u; q.follow = (fl >> 1) & 1u; q.pass_ok = (fl >> 2) & 1u; PRUNE = (fl >> 3) & 1u;
      bsw0 = TO.bsw0; bsw1 = TO.bsw1; bsw2 = TO.bsw2; sm0 = TO.sm0; sm1 = TO.sm1;
      thr = a2_thr_load(TO);
    }
    // ---- 3. search: lanes take items from the end of the list and walk their subtrees; when the list is empty, idle
    // lanes take over ALL unexplored siblings of a busy lane's SHALLOWEST open level (every shallower level of that lane is
    // exhausted, so what it keeps -- the subtree of its current child at that level -- precedes the donated siblings in
    // depth-first order; the taker descends into the first of them and its other siblings can be taken over in turn).
    // Order keys are paths: seven 9-bit digits, digit l = (position of the child chosen at level l) + 1 (solo single 1,
    // solo pair 2), 0 beyond the item's root; a given-away item (node, cursor) carries the cursor's digit at the node's
    // level.  Depth-first order of the items = numeric order of the keys; combinations are scored under their item's key;
    // levels below the seventh cannot be given away.
    AutoBest best{-__builtin_inf(), -1};
    uint64_t best_key = ~0ull;
    int next_item = nitems;  // wave-uniform: items [0, next_item) are not taken yet
    bool act = false;
    uint64_t A = 0, B = 0, A0 = 0, B0 = 0, klo = 0;
    int d = 0, nact = 0, sum2 = 0, cvmin = AUTO_NONE, idmin = 0, sum0 = 0, n0 = 0;
    int p = 0, hi = 0, opt = 3, pr = 0;  // opt: 3 = regular node (cursor p < hi), 0 / 1 / 2 = next solo option of rank pr
    uint32_t more = 0;                   // bit l: level l (a regular node the lane descended from) has positions left
    uint32_t dead = 0;                   // bit l: the siblings of level l were given away: back there, the node is done
    uint32_t pendopen = 0;               // bit l: level l is a solo-single choice whose solo-pair alternative is still to come:
                                         //        nothing BELOW it may be given away (it would precede that alternative)
#ifdef DDZ_STAMP
    unsigned long long n_trips = 0, n_lane_trips = 0, tsec[6] = {0, 0, 0, 0, 0, 0};
#endif
    const int mbox = cur ^ 1;            // the other item buffer is the mailbox of the donations
    const Auto2Wave& WC = s_w[own];      // the sorted candidates (the owner's)
    A2Team& TM = s_team;
    bool in_team = helper;               // the owner: once it has opened its team
    bool waiting = helper;               // counted in TM.hungry (and not in TM.active)
    // the subtree a lane gives away: ALL unexplored siblings of its shallowest open level L, as an item
    auto give = [&](uint64_t& iA, uint64_t& iB, uint32_t& iM, uint32_t& iI, uint64_t& iK) {
      const int L = __builtin_ctz(more);
      uint64_t a_ = A0, b_ = B0;      // replay the path down to level L
      int s2 = sum0;
      // key of the node at level L: this item's key up to its root, then the digits of the path below the root
      uint64_t kk = n0 > 0 ? klo & ~((1ull << (9 * (A2_KEYLEVELS - n0))) - 1ull) : 0ull;
      for (int l = 0; l < L; ++l) {
        const int code = (int)(W.stack[l][lane] & 1023);
        kk |= a2_keydigit(n0 + l, a2_code_digit(code));
        if (code < 512) {
          uint64_t a2, b2;
          a2_child(q, a_, b_, WC.cn[code], a2, b2);
          a_ = a2; b_ = b2;
          s2 += (int)(int8_t)((WC.ci[code] >> 14) & 0xFF);
        } else {
          const int r_ = (code - 512) >> 1, oi = (code - 512) & 1;
          b_ -= (uint64_t)(oi + 1) << (4 * r_);
          s2 += oi ? a2_pair_v2(r_) : a2_single_v2(r_);
        }
      }
      const uint32_t e = W.stack[L][lane];
      const int posL = (int)(e & 1023);  // a regular level: the child being explored; the taker starts behind it
      iA = a_; iB = b_;
      iM = (uint32_t)((s2 + 512) & 1023) | (((e >> 10) & 0xFFu) << 10) | ((uint32_t)(n0 + L) << 18);
      iI = (e >> 18) | ((uint32_t)(posL + 1) << 14);
      iK = kk | a2_keydigit(n0 + L, posL + 2);  // digit of position posL + 1: everything from there on
      more &= ~(1u << L);
      dead |= 1u << L;
    };
    for (unsigned trip = 0;; ++trip) {
#ifdef DDZ_STAMP
      n_trips += 1; n_lane_trips += __popcll(__ballot(act));
      unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#define A2T(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tsec[k] += n_ - ts0; ts0 = n_; } while (0)
#else
#define A2T(k) do { } while (0)
#endif
      if (PRUNE && (trip & 3) == 0) {  // the best score any lane (any member) has reached so far
        const double bvw = wave_max_f64(best.move >= 0 ? best.value : -__builtin_inf());  // (values are never NaN)
        thr = bvw > thr ? bvw : thr;
        if (in_team) {
          const double tt = a2_thr_load(TM);
          if (tt > thr) thr = tt;
          else if (thr > tt && lane == 0) a2_thr_store(TM, thr);
        }
      }
      if (a.teams && (in_team ? (trip & 1) == 0 : (trip & 3) == 0 && (trip >= 8 || first_owner))) {
        const bool can_give_t = act && more != 0 && (more & (0u - more)) < ((pendopen & (0u - pendopen)) | 0x80000000u) &&
                                n0 + __builtin_ctz(more | 0x80000000u) < A2_KEYLEVELS;
        const uint64_t donors_t = __ballot(can_give_t);
#ifdef DDZ_STAMP
        if (in_team) {  // why lanes with work cannot give: no open level / below a pending pair option / below the key levels
          const bool blocked_pend = act && more != 0 && !((more & (0u - more)) < ((pendopen & (0u - pendopen)) | 0x80000000u));
          const bool blocked_key = act && more != 0 && !blocked_pend && !can_give_t;
          const int n_act = __popcll(__ballot(act)), n_none = __popcll(__ballot(act && more == 0));
          const int n_pend = __popcll(__ballot(blocked_pend)), n_key = __popcll(__ballot(blocked_key));
          A2DBG(8, n_act); A2DBG(9, n_none); A2DBG(10, n_pend); A2DBG(11, n_key); A2DBG(12, __popcll(donors_t));
        }
#endif
        if (!in_team && (donors_t || (first_owner && trip == 0)) && (a2_peek(&s_drained) != 0 || first_owner) && !a2_peek(&TM.open)) {
          // the owner opens the team: waves of the block have run out of queue, this search has lasted a while and has
          // subtrees to give (and no other search of the block is being shared)
          a2_lock(&TM.lock, lane, a.status);
          const bool mine = !rfl(TM.open);
          if (mine && lane == 0) {
            TM.owner = (uint32_t)wv; TM.active = 1; TM.box_n = 0; TM.hungry = 0; TM.members = 1; TM.finished = 0;
            TM.flags = (q.nosplit ? 1u : 0u) | (q.follow ? 2u : 0u) | (q.pass_ok ? 4u : 0u) | (PRUNE ? 8u : 0u);
            a2_thr_store(TM, thr); TM.hand = q.hand; TM.bsw0 = bsw0; TM.bsw1 = bsw1; TM.bsw2 = bsw2; TM.sm0 = sm0; TM.sm1 = sm1;
            TM.rp = q.rp; TM.esingle = q.esingle; TM.epair = q.epair;
            TM.open = 1;
          }
          if (mine && first_owner && trip == 0 && next_item > 8) {
            // a team from the start: the frontier goes to the box, the owner keeps an eighth of it
            int nb = next_item - (next_item + 7) / 8;
            if (nb > A2_BOX) nb = A2_BOX;
            const int s0 = next_item - nb;
            for (int i = lane; i < nb; i += 64) {
              TM.bA[i] = W.itA[cur][s0 + i]; TM.bB[i] = W.itB[cur][s0 + i]; TM.bM[i] = W.itM[cur][s0 + i];
              TM.bI[i] = W.itI[cur][s0 + i]; TM.bK[i] = W.itK[cur][s0 + i];
            }
            if (lane == 0) TM.box_n = (uint32_t)nb;
            next_item = s0;
          }
          a2_unlock(&TM.lock, lane);
          in_team = mine;
          A2DBG(0, mine ? 1 : 0);
        } else if (in_team && donors_t && a2_peek(&TM.hungry) != 0 && a2_peek(&TM.box_n) < A2_BOX) {
          // members wait at the box: the first donors fill it
          a2_lock(&TM.lock, lane, a.status);
          const int bn = (int)rfl(TM.box_n);
          const int room = A2_BOX - bn;
          const int jd = __builtin_amdgcn_mbcnt_hi((uint32_t)(donors_t >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)donors_t, 0u));
          if (can_give_t && jd < room) give(TM.bA[bn + jd], TM.bB[bn + jd], TM.bM[bn + jd], TM.bI[bn + jd], TM.bK[bn + jd]);
          const int nd = __popcll(donors_t);
          if (lane == 0) TM.box_n = (uint32_t)(bn + (nd < room ? nd : room));
          a2_unlock(&TM.lock, lane);
          A2DBG(2, nd < room ? nd : room);
          A2DBG(6, 1);
        }
      }
      bool want_score = false, want_open = false, want_back = false;
      int open_from = A2_NOFROM;
      uint64_t idle = __ballot(!act);
      bool took = false;
      uint32_t M = 0, I = 0;
      if (idle && next_item > 0) {          // idle lanes take the next items of the list
        const int r = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
        const int k = next_item - 1 - r;
        if (!act && k >= 0) {
          A = W.itA[cur][k]; B = W.itB[cur][k]; M = W.itM[cur][k]; I = W.itI[cur][k]; klo = W.itK[cur][k];
          took = true;
        }
        next_item -= __popcll(idle);
        if (next_item < 0) next_item = 0;
      } else if (idle) {
        const uint64_t busy = __ballot(act);
        if (busy == 0) {                    // no lane works and no item is left
          if (!in_team) break;
          // a member without work: take what the box holds, or wait there until every member is without work
          if (!waiting || a2_peek(&TM.box_n) != 0 || a2_peek(&TM.active) == 0) {
            a2_lock(&TM.lock, lane, a.status);
            const int bn = (int)rfl(TM.box_n);
            const bool over = bn == 0 && (int)rfl(TM.active) == (waiting ? 0 : 1);
            int nt = (bn + 7) >> 3;                          // a share of the box: other members wait there too
            nt = nt < 4 ? 4 : nt; nt = nt > 64 ? 64 : nt; nt = nt > bn ? bn : nt;
            const int b0 = bn - nt;                          // the last items put (one per lane at most)
            if (lane < nt) {
              W.itA[cur][lane] = TM.bA[b0 + lane]; W.itB[cur][lane] = TM.bB[b0 + lane]; W.itM[cur][lane] = TM.bM[b0 + lane];
              W.itI[cur][lane] = TM.bI[b0 + lane]; W.itK[cur][lane] = TM.bK[b0 + lane];
            }
            if (lane == 0) {
              if (bn > 0) {
                TM.box_n = (uint32_t)b0;
                if (waiting) { TM.active += 1; TM.hungry -= 1; }
              } else if (!waiting) {
                TM.active -= 1; TM.hungry += 1;
              }
            }
            a2_unlock(&TM.lock, lane);
            waiting = bn == 0;
            if (bn > 0) { A2DBG(3, nt); A2DBG(7, 1); next_item = nt; polls = 0; continue; }
            if (over) break;                // nobody holds work, nothing in the box: the search is complete
          }
          if (++polls > A2_POLL_LIMIT) { if (lane == 0 && a.status) atomicOr(a.status, 8); break; }
          __builtin_amdgcn_s_sleep(4);
          continue;
        }
        // a donation round costs about a trip: hold it when a quarter of the lanes idle, or every eighth trip
        const bool round = __popcll(idle) >= 16 || (trip & 7) == 0;
        const bool can_give = act && more != 0 && (more & (0u - more)) < ((pendopen & (0u - pendopen)) | 0x80000000u) &&
                              n0 + __builtin_ctz(more | 0x80000000u) < A2_KEYLEVELS;  // the given level must have a key digit
        const uint64_t donors = round ? __ballot(can_give) : 0ull;
        if (donors) {                       // idle lanes take over siblings of busy lanes
          const int nd = __popcll(donors), nt = __popcll(idle), np = nd < nt ? nd : nt;
          const int jd = __builtin_amdgcn_mbcnt_hi((uint32_t)(donors >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)donors, 0u));
          if (can_give && jd < np) give(W.itA[mbox][jd], W.itB[mbox][jd], W.itM[mbox][jd], W.itI[mbox][jd], W.itK[mbox][jd]);
          __builtin_amdgcn_wave_barrier();
          const int jt = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
          if (!act && jt < np) {
            A = W.itA[mbox][jt]; B = W.itB[mbox][jt]; M = W.itM[mbox][jt]; I = W.itI[mbox][jt];
            klo = W.itK[mbox][jt];
            took = true;
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      A2T(0);
      if (took) {  // a fresh item: its root is level 0 of this lane's stack
        sum2 = (int)(M & 1023) - 512; cvmin = (int)(int8_t)((M >> 10) & 0xFF); nact = (int)((M >> 18) & 31);
        idmin = (int)(I & 0x3FFF);
        d = 0; more = 0; dead = 0; pendopen = 0;
        A0 = A; B0 = B; sum0 = sum2; n0 = nact;
        if ((A | B) == 0) { want_score = true; }
        else if (PRUNE && a2_hopeless(q, A, B, sum2, nact, cvmin, sm0, sm1, thr)) { }  // nothing in this subtree can matter
        else { act = true; want_open = true; open_from = (int)((I >> 14) & 1023); }
      }
      if (act && !want_open) {  // one step: the next child of this node, or back to the parent
        bool desc = false;
        uint64_t A2 = 0, B2 = 0;
        int v2 = 0, id = 0, code = 0, fromc = A2_NOFROM;
        bool el = false;
        if (opt == 3) {
          // next candidate of the bucket that fits what is left: four independent LDS reads per round (a bucket holds
          // every action of the ORIGINAL hand with this lowest rank; deep in the tree few of them still fit)
          bool found = false;
          uint64_t fnib = 0;
          int rounds = A2_SCAN_ROUNDS;  // bounded per trip: a long bucket must not hold the other 63 lanes
          while (p < hi && rounds-- > 0) {
            const uint64_t n0_ = WC.cn[p], n1 = WC.cn[p + 1 < hi ? p + 1 : p], n2 = WC.cn[p + 2 < hi ? p + 2 : p],
                           n3 = WC.cn[p + 3 < hi ? p + 3 : p];
            const uint32_t fm = (a2_fits(n0_, A) ? 1u : 0u) | (p + 1 < hi && a2_fits(n1, A) ? 2u : 0u) |
                                (p + 2 < hi && a2_fits(n2, A) ? 4u : 0u) | (p + 3 < hi && a2_fits(n3, A) ? 8u : 0u);
            if (fm) {
              // the first fit of the four, selected with masks (a select by index would become a scratch array)
              const uint32_t lowbit = fm & (0u - fm);
              fnib = (n0_ & (0ull - (uint64_t)(lowbit == 1u))) | (n1 & (0ull - (uint64_t)(lowbit == 2u))) |
                     (n2 & (0ull - (uint64_t)(lowbit == 4u))) | (n3 & (0ull - (uint64_t)(lowbit == 8u)));
              p += __builtin_ctz(fm);
              found = true;
              break;
            }
            p += 4;
          }
          if (found) {
            const uint64_t nib = fnib;
            const uint32_t ci = WC.ci[p];
            a2_child(q, A, B, nib, A2, B2);
            const bool same = !q.nosplit && a2_lowrank(A2) == a2_lowrank(A);
            v2 = (int)(int8_t)((ci >> 14) & 0xFF); el = (ci >> 22) & 1u; id = (int)(ci & 0x3FFF);
            code = p; fromc = same ? p : A2_NOFROM;
            more = (more & ~(1u << d)) | ((p + 1 < hi ? 1u : 0u) << d);
            pendopen &= ~(1u << d);
            desc = true;
          } else if (p >= hi) {
            want_back = true;
          }
        } else if (opt == 0) {
          A2 = A; B2 = B - (1ull << (4 * pr));
          v2 = a2_single_v2(pr); el = (q.esingle >> pr) & 1u; id = 1 + pr; code = 512 + 2 * pr;
          more &= ~(1u << d);
          pendopen = (pendopen & ~(1u << d)) | ((a2_pair_option(q.hand, B, pr) ? 1u : 0u) << d);
          desc = true;
        } else if (opt == 1 && a2_pair_option(q.hand, B, pr)) {
          A2 = A; B2 = B - (2ull << (4 * pr));
          v2 = a2_pair_v2(pr); el = (q.epair >> pr) & 1u; id = 16 + pr; code = 512 + 2 * pr + 1;
          more &= ~(1u << d);
          pendopen &= ~(1u << d);
          desc = true;
        } else {
          want_back = true;
        }
        if (PRUNE && desc && (A2 | B2) != 0) {
          const int cvc = (el && (cvmin == AUTO_NONE || v2 < cvmin)) ? v2 : cvmin;
          if (a2_hopeless(q, A2, B2, sum2 + v2, nact + 1, cvc, sm0, sm1, thr)) {
            // as if the child had been explored: the cursor moves behind it, the lane stays at this node
            desc = false;
            more &= ~(1u << d); pendopen &= ~(1u << d);  // the lane is AT level d again, not below it
            if (opt == 3) p += 1; else opt += 1;
          }
        }
        if (desc) {
          W.stack[d][lane] = (uint32_t)code | ((uint32_t)(cvmin & 0xFF) << 10) | ((uint32_t)idmin << 18);
          ++d; ++nact; ++nodes_l;
          sum2 += v2;
          if (el && (cvmin == AUTO_NONE || v2 < cvmin)) { cvmin = v2; idmin = id; }
          A = A2; B = B2;
          if ((A | B) == 0) { want_score = true; want_back = true; }
          else { want_open = true; open_from = fromc; }
        }
      }
      A2T(1);
      if (want_score) {  // a finished combination
        AutoBest b2{-__builtin_inf(), -1};
        auto_leaf(b2, sum2, cvmin, idmin, nact, q.follow, q.pass_ok, q.rp);
        ++combs_l;
        // under one key: first maximum (strict >); otherwise the smaller key (earlier in depth-first order) wins a tie
        if (b2.move >= 0 && (best.move < 0 || b2.value > best.value || (b2.value == best.value && klo < best_key))) {
          best = b2;
          best_key = klo;
        }
      }
      A2T(2);
      if (want_open) {  // cursor of a fresh node
        const int r = a2_pend_rank(A, B);
        if (r >= 0) {
          opt = 0; pr = r;
        } else {
          opt = 3;
          const int ul = a2_lowrank(A);
          p = open_from != A2_NOFROM ? open_from : a2_bs(bsw0, bsw1, bsw2, ul);
          hi = a2_bs(bsw0, bsw1, bsw2, ul + 1);
        }
      }
      A2T(3);
      if (want_back && act) {  // undo the last action of the path: back at its node, behind that child
        if (d == 0) {
          act = false; more = 0; dead = 0; pendopen = 0;
        } else {
          --d; --nact;
          more &= (1u << d) - 1u;             // only levels the lane is BELOW can be given away
          const bool gone = (dead >> d) & 1u; // the rest of this level was given away
          dead &= (1u << d) - 1u;
          pendopen &= (1u << d) - 1u;
          const uint32_t e = W.stack[d][lane];
          const int code = (int)(e & 1023);
          cvmin = (int)(int8_t)((e >> 10) & 0xFF);
          idmin = (int)(e >> 18);
          if (code < 512) {
            const uint64_t nib = WC.cn[code];
            sum2 -= (int)(int8_t)((WC.ci[code] >> 14) & 0xFF);
            if (q.nosplit) {
              const uint64_t rm = a2_rankmask(nib);
              A = A | ((B & rm) + nib);
              B = B & ~rm;
            } else {
              A = A + nib;
            }
            opt = 3; p = code + 1;
            hi = gone ? p : a2_bs(bsw0, bsw1, bsw2, a2_lowrank(A) + 1);
          } else {
            pr = (code - 512) >> 1;
            const int oi = (code - 512) & 1;
            B += (uint64_t)(oi + 1) << (4 * pr);
            sum2 -= oi ? a2_pair_v2(pr) : a2_single_v2(pr);
            opt = oi + 1;
          }
        }
      }
      A2T(4);
    }
#ifdef DDZ_STAMP
    tq[3] = __builtin_amdgcn_s_memtime();
#endif
    // ---- 4. the wave's best: larger value, on ties the smaller order key
    // (three DPP reductions: the maximum value, then the smallest key among the lanes that hold it -- high word, low word)
    int bm = -1;
    double wbv;
    uint64_t wkey;
    {
      const bool have = best.move >= 0;
      wbv = wave_max_f64(have ? best.value : -__builtin_inf());
      const bool top = have && best.value == wbv;
      const uint32_t khi = wave_min_u32(top ? (uint32_t)(best_key >> 32) : 0xFFFFFFFFu);
      const bool top2 = top && (uint32_t)(best_key >> 32) == khi;
      const uint32_t klo_ = wave_min_u32(top2 ? (uint32_t)best_key : 0xFFFFFFFFu);
      const uint64_t win = __ballot(top2 && (uint32_t)best_key == klo_);
      if (win) bm = (int)rl((uint32_t)best.move, __builtin_ctzll(win));
      wkey = ((uint64_t)khi << 32) | klo_;
    }
    int nodes = wave_sum_i32(nodes_l), combs = wave_sum_i32(combs_l);
    if (helper) { A2DBG(1, 1); A2DBG(4, nodes); }
    else if (in_team) A2DBG(5, nodes);
    if (helper) {  // a member's result goes to its slot; the owner keeps the best
      if (lane == 0) {
        TM.rvalue[my_slot] = wbv; TM.rkey[my_slot] = wkey; TM.rmove[my_slot] = bm; TM.rcombs[my_slot] = combs;
        TM.rnodes[my_slot] = nodes;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(&TM.finished, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    if (in_team) {  // close the team, wait for the helpers' results
      a2_lock(&TM.lock, lane, a.status);
      const int nm = (int)rfl(TM.members);
      a2_unlock(&TM.lock, lane);
      while (__hip_atomic_load(&TM.finished, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u < (uint32_t)nm) {
        if (++polls > A2_POLL_LIMIT) { if (lane == 0 && a.status) atomicOr(a.status, 8); break; }
        __builtin_amdgcn_s_sleep(2);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      for (int m = 1; m < nm; ++m) {  // (wave-uniform: every lane reads the same slots)
        const double v = TM.rvalue[m];
        const uint64_t k = TM.rkey[m];
        const int mv = TM.rmove[m];
        combs += TM.rcombs[m]; nodes += TM.rnodes[m];
        if (mv >= 0 && (bm < 0 || v > wbv || (v == wbv && k < wkey))) { bm = mv; wbv = v; wkey = k; }
      }
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) __hip_atomic_store(&TM.open, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // the next owner's
    }
    if (lane == 0) {
      a.ids[t] = bm < 0 ? 0 : bm;  // rule_based_model.py:87-89
      if (a.stats) { a.stats[2 * t] = combs; a.stats[2 * t + 1] = nodes; }
    }

#ifdef DDZ_STAMP
    if (g_stamps && lane == 0 && !helper) {
      g_stamps[16 * t + 0] = tq[1] - tq[0]; g_stamps[16 * t + 1] = tq[2] - tq[1]; g_stamps[16 * t + 2] = tq[3] - tq[2];
      g_stamps[16 * t + 3] = (unsigned long long)nitems_final | ((unsigned long long)n_fastpass << 16) | ((unsigned long long)n_genpass << 24); g_stamps[16 * t + 4] = nodes; g_stamps[16 * t + 5] = n;
      g_stamps[16 * t + 6] = n_trips; g_stamps[16 * t + 7] = n_lane_trips;
      for (int k_ = 0; k_ < 2; ++k_) g_stamps[16 * t + 8 + k_] = tsec[k_];
      g_stamps[16 * t + 10] = (unsigned long long)(blockIdx.x * A2_WPB + wv);  // which wave decided it, when (timeline)
      g_stamps[16 * t + 11] = tq[0]; g_stamps[16 * t + 12] = tq[3];
      g_stamps[16 * t + 13] = tq_enum - tq[0]; g_stamps[16 * t + 14] = tq_sort - tq_enum; g_stamps[16 * t + 15] = tq_bounds - tq_sort;
    }
#endif
    __builtin_amdgcn_wave_barrier();
  }
}
