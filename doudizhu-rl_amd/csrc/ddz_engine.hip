// ddz_engine.hip -- gfx950 kernels + C ABI (include/ddz_env.h) of the batched Doudizhu engine.
//
// Kernel map (reference paths relative to /root/reference):
//   k_rollout THE DOMINANT KERNEL (ddz_rollout_random): one wavefront per table, all lock-step iterations of a
//             random-policy rollout (game.py:169-181 with envi.py:79-85) inside one launch, state in registers,
//             lists in fixed-stride slabs; arithmetic fast path for follows of singles / pairs / triples, planner
//             + LDS staging list for the rest (see the comment at the kernel).
//   k_table   ONE WAVEFRONT PER TABLE, one lock-step iteration per launch (the API a policy drives; CSR lists,
//             or slab lists with F_SLAB = apply + enumerate in the same launch):
//             lanes 0..10 load the table's 11 packed rows (176 contiguous bytes), then
//             [enumerate] the combo enumerator + follow filter (r.get_moves, envi.py:111;
//                 rules utils.py:45-63 get_mask, card.py:307-325 bigger_than) as a pruned
//                 dense scan: a scalar planner picks the id ranges of the action space
//                 (card.py:34-159 order) that can be legal for this hand; each lane tests
//                 one id per round against the per-action record table (SWAR nibble
//                 subset test + follow gate), __ballot + mbcnt compaction into the CSR
//                 row list, 16-byte coalesced row stores, ascending canonical id;
//             [step] pick (engine RNG, captured during the scan / index / row search),
//                 apply (envi.py:38-43 _update + native step_manual) as a lane-parallel
//                 byte-wise update of the 11 rows, terminal + reward (rule_play.py:14),
//                 auto-reset deal (native prepare(), spec v2: lane-parallel ranking);
//             [count] the same scan without stores on the NEW state -> list size, block
//                 scan of the sizes -> CSR bases of the next launch (double-buffered).
//             Table-level control flow is wave-uniform (scalar branches, no divergence).
//   k_moves   stateless r.get_moves(hand, last) queries, one wavefront per query.
//   k_build_table  fills the record table once per device with the structural enumerator
//             (rank masks -> rows/ids in closed form), itself pinned by the golden tests.
//   k_observe the `face` tensors (envi.py:87-96,165-217), k_onehot batch_arr2onehot (:139-146).
//   k_mask    get_mask (utils.py:45-63) of every table as a bit-packed dense mask; k_select segment arg-max /
//             epsilon-greedy (dqn.py:50-71); k_pack_traj 32-byte -> 8-byte trajectory records; k_export_table.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "../../include/ddz_env.h"
#include "ddz_device.h"

using namespace ddz;

namespace {

constexpr int BLOCK = 256;       // threads per block of the thread-per-item kernels
constexpr int WPB = 16;          // wavefronts per block of k_table and k_slab (one block per CU: one work list for 256 tables)
constexpr int TB = WPB * 64;     // threads per block of k_table
constexpr int STATE_ROW_BYTES = DDZ_NFIELDS * DDZ_ROW;  // 176

// k_table phases
constexpr int F_ENUM = 1, F_STEP = 2, F_RESET = 4, F_COUNT = 8, F_SLAB = 16;

// ------------------------------------------------------------------------------------
// scratch layout (caller-owned, zero-filled at create)
constexpr int AUTO_SLOTS = 4;  // ring of k_auto2 queue slots per handle (concurrent launches on several streams)
constexpr int AO_HDR_BYTES = 576;  // sizeof(AutoOrder), ddz_auto2.h
struct Layout {
  int64_t T, nblk;
  int64_t off_counts, off_local, off_blk_tot, off_blk_stats, off_status, off_auto, auto_slot_bytes, bytes;
};
inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
inline Layout make_layout(int64_t T) {
  Layout l;
  l.T = T;
  l.nblk = (T + 3) / 4;  // upper bound of blocks for any launch geometry (>= 4 tables per block)
  int64_t o = 0;
  l.off_counts = o;     o = align_up(o + 2 * T * 4, 256);          // double-buffered
  l.off_local = o;      o = align_up(o + 2 * T * 4, 256);
  l.off_blk_tot = o;    o = align_up(o + 2 * l.nblk * 4, 256);
  l.off_blk_stats = o;  o = align_up(o + T * 32, 256);           // one slot per wave (<= T)
  l.auto_slot_bytes = align_up(AO_HDR_BYTES + 4 * T, 256);        // AutoOrder header + order[T]
  l.off_auto = o;       o = align_up(o + AUTO_SLOTS * l.auto_slot_bytes, 256);
  l.off_status = o;     o = align_up(o + 64, 256);               // (the last 256 bytes: callers of the stateless entry
  l.bytes = o;                                                    //  points read the status word at bytes - 256)
  return l;
}
struct Scratch {  // device pointers into the scratch buffer
  int32_t* counts[2];    // size of each table's legal list (ping-pong)
  int32_t* local_off[2]; // exclusive scan of counts inside the table's block
  int32_t* blk_tot[2];   // sum of counts per block
  int64_t* blk_stats;    // [T][4] per block / per wave: plies, episodes, lord wins | up wins << 32, rows
  int32_t* status;       // [0] status bits
  int64_t* legal_rows;   // running total of rows produced
};
inline Scratch bind(void* scratch, const Layout& l) {
  uint8_t* p = (uint8_t*)scratch;
  Scratch s;
  for (int k = 0; k < 2; ++k) {
    s.counts[k] = (int32_t*)(p + l.off_counts) + k * l.T;
    s.local_off[k] = (int32_t*)(p + l.off_local) + k * l.T;
    s.blk_tot[k] = (int32_t*)(p + l.off_blk_tot) + k * l.nblk;
  }
  s.blk_stats = (int64_t*)(p + l.off_blk_stats);
  s.status = (int32_t*)(p + l.off_status);
  s.legal_rows = (int64_t*)(p + l.off_status + 16);
  return s;
}

// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rl(uint32_t v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, int l) {
  return (uint64_t)rl((uint32_t)v, l) | ((uint64_t)rl((uint32_t)(v >> 32), l) << 32);
}
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// ------------------------------------------------------------------------------------
// wave-cooperative emission: lanes holding a legal candidate append their row at
// base + n + (number of legal lanes below), in lane order.
struct Out {
  uint4* rows;
  int32_t* ids;
  int64_t base, cap;
  uint64_t* stage;      // EM_STAGE: per-wave LDS list of nib | category << 60, in id order,
  uint16_t* stage_vl;   //           value | len << 8 of each entry (card.py:327-335)
  uint16_t* stage_ids;  //           and (IDS) the canonical ids
  uint32_t* mask = nullptr;  // EM_MASK: per-wave LDS bit mask over the action space (bit id)
};
// what a scan does with the legal lanes
constexpr int EM_COUNT = 0;   // nothing (list size only)
constexpr int EM_WRITE = 1;   // rows (+ids) into the CSR list at base + running index
constexpr int EM_PICK = 2;    // EM_WRITE + capture the row with list index pk.want
constexpr int EM_STAGE = 3;   // nib + category (+id) into the wave's LDS staging list
constexpr int EM_MASK = 4;    // bit `id` of the wave's LDS mask (get_mask, utils.py:45-63)
constexpr int EM_SLAB = 5;    // rows (+ids) built from nib | category straight into the table's slab at base + running index
                              // (no record rows in LDS, no staging list, no flush pass: what k_slab's list phase uses)
// One list written by ALL the waves of a block (k_slab with one table per wave: such a launch waits for its slowest list,
// the lord's 20-card lead of a fresh game).  The plan of a list is wave-uniform SCALAR work -- sixteen waves repeating it
// would queue on the CU's one scalar unit -- so the table's own wave runs it once in this mode, leaving one 8-byte record per
// scan round (64 candidate ids) in LDS; the rounds are then executed from the records (team_round): by the wave itself when
// they are few, otherwise dealt out over the block's waves -- counted, scanned, written at their bases: byte for byte the
// list EM_SLAB writes (ascending canonical id).
constexpr int EM_TEAM_PLAN = 6;
constexpr int TEAM_ROUNDS = 256;  // >= the scan rounds of any lead of a <= 20-card hand (at most ~210: see team_round)
[[maybe_unused]] constexpr int ID_JK_FOUR = DDZ_NUM_ACTIONS;        // quad q + both jokers: ids 13527 + q
[[maybe_unused]] constexpr int ID_JK_PLANE = DDZ_NUM_ACTIONS + 13;  // triples s, s+1 + both jokers: ids 13540 + s
#ifndef DDZ_STAGE_CAP
#define DDZ_STAGE_CAP 500
#endif
constexpr int STAGE_CAP = DDZ_STAGE_CAP;  // >= the largest list of a <=20-card hand: 497, PROVEN by exhaustive enumeration (tools/max_legal_bound.c, tests/test_rules_bounds.py);
                                 // 500 keeps k_rollout's block at 53 KB of LDS = three blocks per CU
// the row with list index `want` is captured (wave-uniform) while it is emitted
struct Pick {
  int want;
  uint32_t r0, r1, r2, r3;
  // EM_TEAM_PLAN: the number of rounds so far, their records in LDS
  int rc = 0;
  uint2* desc = nullptr;
};

#include "ddz_build_table.h"

// (DPP row operations, ddz_device.h: six VALU steps instead of six ds_bpermute round trips; every lane must be active)
__device__ __forceinline__ int wave_sum(int v) { return wave_sum_i32(v); }
__device__ __forceinline__ int wave_incl_scan(int v, int) { return wave_scan_add(v); }

// ------------------------------------------------------------------------------------
// Per-action record table, 32 B per canonical id (433 KB: L2-resident, the hot first
// 526 ids L1-resident), filled once per device by k_build_table:
//   g_tab[2*id]     row bytes: int8 counts[15] + category (what is stored into the CSR list)
//   g_tab[2*id + 1] {nib lo, nib hi, value | len << 8 | category << 16, 0}
__device__ uint4 g_tab[2 * DDZ_NUM_ACTIONS];
// Kicker combinations: for each (remains R, kickers L) pair the action space uses, the
// L-subsets of 0..R-1 in lexicographic order (itertools.combinations, card.py:115,128,141,152),
// L positions x 4 bits per entry.  Entry j of a list is kicker set j of an id block.
//   list: 0 (14,2) 4+1+1 | 1 (13,2) 3+1 x2 | 2 (12,2) 4+2+2 | 3 (11,2) 3+2 x2 | 4 (12,3) 3+1 x3
//         5 (10,3) 3+2 x3 | 6 (11,4) 3+1 x4 | 7 (9,4) 3+2 x4 | 8 (10,5) 3+1 x5
constexpr int CL_R[9] = {14, 13, 12, 11, 12, 10, 11, 9, 10};
constexpr int CL_L[9] = {2, 2, 2, 2, 3, 3, 4, 4, 5};
constexpr int CL_N[9] = {91, 78, 66, 55, 220, 120, 330, 126, 252};
constexpr int CL_OFF[10] = {0, 91, 169, 235, 290, 510, 630, 960, 1086, 1338};
constexpr int COMBO_WORDS = 1338;
__device__ uint32_t g_combo[COMBO_WORDS];
__device__ uint64_t g_c64[COMBO_WORDS];  // the same combinations as nibble sets over the remains list (1 per kicker): team_round
// the 16-byte row (counts + category) of a canonical action id, including the joker-kicker extras of that build
__device__ __forceinline__ uint4 row_of_id(int id) {
  if (id < DDZ_NUM_ACTIONS) return g_tab[2 * id];
  const int k = id - DDZ_NUM_ACTIONS;  // 0..12 quad + jokers, 13..23 two triples + jokers
  const uint64_t jk = (1ull << 52) | (1ull << 56);
  return k < 13 ? unpack_row((4ull << (4 * k)) | jk, FOUR_TAKE_ONE) : unpack_row((0x33ull << (4 * (k - 13))) | jk, THREE_ONE_LINE);
}
__device__ uint4 g_tmp_rows[DDZ_NUM_ACTIONS];
__device__ int32_t g_tmp_ids[DDZ_NUM_ACTIONS];

// one wavefront: enumerate the full deck on lead (every action, ids 1..13526 in order)
__global__ __launch_bounds__(64) void k_build_table(int32_t* status) {
  const int lane = threadIdx.x & 63;
  if (lane < 9) {  // lane k writes combination list k
    const int R = CL_R[lane], L = CL_L[lane];
    int idx[5] = {0, 1, 2, 3, 4};
    for (int e = 0; e < CL_N[lane]; ++e) {
      uint32_t w = 0;
      uint64_t c = 0;
      for (int k = 0; k < L; ++k) { w |= (uint32_t)idx[k] << (4 * k); c += 1ull << (4 * idx[k]); }
      g_combo[CL_OFF[lane] + e] = w;
      g_c64[CL_OFF[lane] + e] = c;
      int i = L - 1;
      while (i >= 0 && idx[i] == R - L + i) --i;
      if (i >= 0) {
        ++idx[i];
        for (int j = i + 1; j < L; ++j) idx[j] = idx[j - 1] + 1;
      }
    }
  }
  const uint32_t lut = c_line_lut.v[lane];
  const uint64_t deck = 0x0114444444444444ull;  // 4 of 3..2, one of each joker
  Out o{g_tmp_rows, g_tmp_ids, 0, DDZ_NUM_ACTIONS, nullptr, nullptr, nullptr};
  Pick pk{-1, 0, 0, 0, 0};
  const int n = enumerate_table<true, false>(deck, mk_info(EMPTY, 0, 1), lut, lane, o, pk);
  __threadfence();
  __syncthreads();
  if (n != DDZ_NUM_ACTIONS - 1) {
    if (lane == 0) atomicOr(status, 8);
    return;
  }
  for (int p = lane; p < n + 1; p += 64) {
    uint4 row = make_uint4(0, 0, 0, 0);
    int id = 0;
    if (p > 0) {
      row = g_tmp_rows[p - 1];
      id = g_tmp_ids[p - 1];
      if (id != p) atomicOr(status, 8);
    }
    const uint64_t nib = pack_row(row);
    const uint32_t info = info_of_row(nib, (int)(row.w >> 24));
    g_tab[2 * id] = row;
    g_tab[2 * id + 1] = make_uint4((uint32_t)nib, (uint32_t)(nib >> 32),
                                   ((info >> 8) & 0xFFFF) | ((info & 0xFF) << 16), 0);
  }
}

// LDS-staged hot part of the record table: ids 0..525 (everything but planes-with-kickers
// and four-with-two) plus the rocket in slot 526.  A launch starts with cold L2s, so a
// record fetched from memory costs ~1000 cycles; staged once per block it costs an LDS read.
constexpr int HOT_IDS = ID_THREE_ONE_LINE;  // 526
constexpr int HOT_SLOTS = HOT_IDS + 1;
// WITH_ROWS = false: kernels that stage packed rows only (k_rollout) save 8.4 KB of LDS.  C64 (DDZ_HOT_C64): the kicker
// combinations as 8-byte nibble sets over the remains list (g_c64) instead of 4-byte position words (g_combo): a kicker
// candidate is then a splice (two masks, a shift, an or) instead of a loop over its 2 - 5 positions, for 5.3 KB more LDS
// (k_auto / k_auto2 keep the words: their blocks have no LDS left).
template <bool WITH_ROWS, bool C64 = false>
struct HotTabT {
  static constexpr bool HAS_ROWS = WITH_ROWS, HAS_C64 = C64;
  uint4 meta[HOT_SLOTS + 1];
  uint4 rows[WITH_ROWS ? HOT_SLOTS + 1 : 1];
  uint32_t combo[C64 ? 2 : COMBO_WORDS + 2];
  uint64_t c64[C64 ? COMBO_WORDS + 1 : 1];
};
#ifndef DDZ_HOT_C64
#define DDZ_HOT_C64 1
#endif
// Measured (tools/lib_ab_probe.py, both builds): the stress set of plane-rich leads through k_moves_slab 305 -> 244 us; k_rollout
// unchanged at 65,536 tables (16.06 / 16.10 us per iteration), + 0.8 % at 4096; the many-tables form of k_slab 27.4 -> 27.9 us
// (a larger fill, 85 KB of LDS): so the nibble sets are used by k_moves_slab and by the one-table-per-wave form of k_slab.
using HotTabL = HotTabT<false, DDZ_HOT_C64 != 0>;
using HotTab = HotTabT<true>;

// componentwise select (a ?: on whole uint4s is lowered to a lane-indexed scratch array)
__device__ __forceinline__ uint4 sel4(bool c, const uint4& a, const uint4& b) {
  return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}

// EM_TEAM_PLAN's record of a scan round: x = candidates - 1 (6 bits) | first id << 6 (14) | first hot slot / combination
// word << 20 (11); y = kind (0 hot ids, 1 kicker block, 2 the joker-kicker extras) | the block's shape
__device__ __forceinline__ uint32_t team_x(int left, int id, int so) {
  return (uint32_t)((left < 64 ? left : 64) - 1) | (uint32_t)id << 6 | (uint32_t)so << 20;
}
// the action a lane is looking at, however its record was obtained
template <int EM, bool IDS>
__device__ __forceinline__ int scan_emit(bool legal, int id, uint64_t nib, int cat, int vl, uint4 row, const Out& o,
                                         int n, Pick& pk) {
  const uint64_t b = __ballot(legal);
  const int k = __popcll(b);
  if (EM == EM_MASK) {
    if (legal) atomicOr(&o.mask[id >> 5], 1u << (id & 31));
  } else if (EM != EM_COUNT) {
    const int pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
    if (EM == EM_STAGE) {
      if (legal && n + pre < STAGE_CAP) {
        o.stage[n + pre] = nib | ((uint64_t)cat << 60);
        o.stage_vl[n + pre] = (uint16_t)vl;
        if (IDS) o.stage_ids[n + pre] = (uint16_t)id;
      }
    } else if (EM == EM_SLAB) {
      if (legal) {
        const int64_t pos = o.base + n + pre;
        if (pos < o.cap) {  // cap = end of this table's slab
          o.rows[pos] = unpack_row(nib, (uint32_t)cat);
          if (IDS) o.ids[pos] = id;
        }
      }
    } else {
      if (legal) {
        const int64_t pos = o.base + n + pre;
        if (pos < o.cap) {
          o.rows[pos] = row;
          if (IDS) o.ids[pos] = id;
        }
      }
      if (EM == EM_PICK) {
        const int w = pk.want - n;
        if (w >= 0 && w < k) {  // wave-uniform
          const int src = __builtin_ctzll(__ballot(legal && pre == w));
          pk.r0 = rl(row.x, src); pk.r1 = rl(row.y, src); pk.r2 = rl(row.z, src); pk.r3 = rl(row.w, src);
        }
      }
    }
  }
  return n + k;
}

// ids [id0, id0 + count) of the hot part of the action space (or the rocket): records in LDS.
//   legal <=> counter_subset(action, hand) (utils.py:16-22, SWAR per nibble) and
//             (lead or pass or bigger_than(action, last)) (utils.py:53-60, card.py:307-325)
template <int EM, bool IDS, bool LEAD, class HT>
__device__ __forceinline__ int scan_ids(int id0, int count, const HT& hot, uint64_t hand8, const Follow& f,
                                        int lane, const Out& o, int n, Pick& pk) {
  constexpr bool ROWS = EM == EM_WRITE || EM == EM_PICK;
  constexpr uint64_t H8 = 0x8888888888888888ull;
  const int slot0 = id0 == ID_BIGBANG ? HOT_IDS : id0;
  if (EM == EM_TEAM_PLAN) {  // (leads only: no follow gate) record: ids id0 + j0 ..., their hot slots
    for (int j0 = 0; j0 < count; j0 += 64, ++pk.rc)
      if (pk.rc < TEAM_ROUNDS)
        pk.desc[pk.rc] = make_uint2(team_x(count - j0, id0 + j0, slot0 + j0), 0u);
    return n;
  }
  for (int j0 = 0; j0 < count; j0 += 64) {
    const int j = j0 + lane;
    const bool in = j < count;
    const int jj = in ? j : 0, id = id0 + jj;
    const uint4 m = hot.meta[slot0 + jj];
    static_assert(!ROWS || HT::HAS_ROWS, "this emission mode reads the unpacked rows");
    const uint4 row = ROWS ? hot.rows[HT::HAS_ROWS ? slot0 + jj : 0] : make_uint4(0, 0, 0, 0);
    const uint64_t nib = (uint64_t)m.x | ((uint64_t)m.y << 32);
    const bool sub = ((hand8 - nib) & H8) == H8;
    const int val = m.z & 0xFF, len = (m.z >> 8) & 0xFF, cat = (m.z >> 16) & 0xFF;
    const bool gate = LEAD || id == 0 || cat == BIGBANG || (cat == QUADRIC && (f.lc != QUADRIC || val > f.lv)) ||
                      (cat == f.lc && f.lc != QUADRIC && len == f.ll && val > f.lv);
    n = scan_emit<EM, IDS>(in && sub && gate, id, nib, cat, (int)(m.z & 0xFFFF), row, o, n, pk);
  }
  return n;
}

// one id block of a category with kickers (card.py:110-129, :139-153): main group `mainnib`
// (ranks [s, s + gap) removed from the remains list), kicker set j = combination list entry j
// with `mult` cards per kicker; ids idb + j.  The planner has already applied the follow
// filter (category, len, value are those of the block), so legal <=> subset of the hand.
template <int EM, bool IDS, class HT>
__device__ __forceinline__ int scan_combos(int list, int count, int idb, uint64_t mainnib, int s, int gap, int mult,
                                           int cat, const HT& hot, uint64_t hand8, int lane, const Out& o,
                                           int n, Pick& pk) {
  constexpr bool ROWS = EM == EM_WRITE || EM == EM_PICK;
  constexpr uint64_t H8 = 0x8888888888888888ull;
  const int L = CL_L[list], off = CL_OFF[list];
  if (EM == EM_TEAM_PLAN) {  // record: ids idb + j0 ..., combination words off + j0 ..., the block's shape
    for (int j0 = 0; j0 < count; j0 += 64, ++pk.rc)
      if (pk.rc < TEAM_ROUNDS)
        pk.desc[pk.rc] = make_uint2(team_x(count - j0, idb + j0, off + j0),
                                    1u | (uint32_t)L << 2 | (uint32_t)s << 5 | (uint32_t)gap << 9 | (uint32_t)mult << 12 | (uint32_t)cat << 14);
    return n;
  }
  for (int j0 = 0; j0 < count; j0 += 64) {
    const int j = j0 + lane;
    const bool in = j < count;
    uint64_t nib = mainnib;
    if constexpr (HT::HAS_C64) {
      // the kicker positions as a nibble set over the REMAINS list; the main group's `gap` ranks are spliced in at nibble s,
      // `mult` cards per kicker (1 or 2: a shift)
      const uint64_t c = hot.c64[off + (in ? j : 0)], lowm = (1ull << (4 * s)) - 1ull;
      nib += ((c & lowm) | ((c & ~lowm) << (4 * gap))) << (mult - 1);
    } else {
      const uint32_t e = hot.combo[off + (in ? j : 0)];
      for (int k = 0; k < L; ++k) {
        const int pos = (e >> (4 * k)) & 15;
        nib += (uint64_t)mult << (4 * (pos < s ? pos : pos + gap));
      }
    }
    const bool sub = ((hand8 - nib) & H8) == H8;
    const uint4 row = ROWS ? unpack_row(nib, (uint32_t)cat) : make_uint4(0, 0, 0, 0);
    n = scan_emit<EM, IDS>(in && sub, idb + j, nib, cat, s | (gap << 8), row, o, n, pk);
  }
  return n;
}

// The planner: which id ranges of the action space can hold a legal move for this hand
// (wave-uniform, scalar); ranges are visited in ascending id order.  Everything a range
// admits too generously is rejected per id by scan_ids, so the planner only has to be a
// superset -- and cheap.
template <int EM, bool IDS, bool LEAD, class HT>
__device__ __forceinline__ int plan_scan_t(uint64_t hand, const Follow& f, const HT& hot, int lane, const Out& o, Pick& pk) {
  // PRUNE: skip kicker blocks / ranges that cannot hold a legal move (fewer kicker ranks in the hand than kickers needed).
  // Measured A/B (profiles/r03_notes.md): -16 % on lists of plane-rich hands (the stress leg, the tail of k_slab), but +3 %
  // on k_rollout's step, where such hands are rare and the extra scalar work sits on every lead: off for the staging
  // emission of the rollout kernel, on everywhere else.
  constexpr bool PRUNE = EM != EM_STAGE;
  const uint64_t hand8 = hand | 0x8888888888888888ull;
  const int cnt = lane < 15 ? (int)((hand >> (4 * (lane & 15))) & 15) : 0;
  const uint32_t m1 = (uint32_t)__ballot(cnt >= 1) & M15;
  const uint32_t m2 = (uint32_t)__ballot(cnt >= 2) & M13;
  const uint32_t m3 = (uint32_t)__ballot(cnt >= 3) & M13;
  const uint32_t m4 = (uint32_t)__ballot(cnt >= 4) & M13;
  int n = 0;
  auto scan = [&](int id0, int count) { n = scan_ids<EM, IDS, LEAD>(id0, count, hot, hand8, f, lane, o, n, pk); };
  if (!LEAD && f.lc == BIGBANG) {  // nothing beats the rocket: pass only (card.py:312-313)
    scan(0, 1);
    return n;
  }
  if (LEAD) scan(1, 54); else scan(0, 55);  // [pass,] singles, pairs, triples, bombs
  const uint32_t above = LEAD ? M15 : gt_mask(f.lv);
  if (LEAD || f.lc == THREE_ONE || f.lc == THREE_TWO) {  // mains m..M of card.py:69-82
    const uint32_t mains = m3 & above;
    if (mains) {
      const int lo = __builtin_ctz(mains), hi = 31 - __builtin_clz(mains);
      if (LEAD || f.lc == THREE_ONE) scan(ID_THREE_ONE + 14 * lo, 14 * (hi - lo + 1));
      if ((LEAD || f.lc == THREE_TWO) && (!PRUNE || __builtin_popcount(m2) >= 2))  // (a pair of another rank exists at all)
        scan(ID_THREE_TWO + 12 * lo, 12 * (hi - lo + 1));
    }
  }
  if (LEAD) {  // chains (card.py:86-105): the three categories are contiguous ids
    const bool a = run_starts(m1 & M12, 5) != 0, b = run_starts(m2 & M12, 3) != 0, c = run_starts(m3 & M12, 2) != 0;
    if (a || b || c) {
      const int lo = a ? ID_SINGLE_LINE : b ? ID_DOUBLE_LINE : ID_TRIPLE_LINE;
      const int hi = c ? ID_THREE_ONE_LINE : b ? ID_TRIPLE_LINE : ID_DOUBLE_LINE;
      scan(lo, hi - lo);
    }
  } else if (f.lc == SINGLE_LINE) {
    scan(ID_SINGLE_LINE, 36);
  } else if (f.lc == DOUBLE_LINE) {
    scan(ID_DOUBLE_LINE, 52);
  } else if (f.lc == TRIPLE_LINE) {
    scan(ID_TRIPLE_LINE, 45);
  }
  // planes with kickers (card.py:110-129): one id block per (start s, len L), canonical
  // order = s-major, L ascending.  Only starts of triple runs are visited; the id of a block
  // is closed-form: blocks before start s plus the shorter blocks of s.
  //   3+1: sizes C(15-L, L) = 77*, 220, 330, 252 (L = 2..5; * joker pair dropped, card.py:116)
  //   3+2: sizes C(13-L, L) = 55, 120, 126        (L = 2..4)
  auto planes = [&](int cat, int hi, int idb0, int mult, int list2, int list3, int list4, int list5) {
    const uint32_t mm = m3 & M12;
    if (!(LEAD || f.lc == cat)) return;
    for (uint32_t st = mm & (mm >> 1); st; st &= st - 1) {
      const int s = __builtin_ctz(st);
      const int runlen = __builtin_ctz(~(mm >> s));  // consecutive triples from s
      int before, sz2, sz3, sz4;
      if (cat == THREE_ONE_LINE) {
        before = 879 * (s < 8 ? s : 8) + (s > 8 ? 627 : 0) + (s > 9 ? 297 : 0);
        sz2 = 77; sz3 = 220; sz4 = 330;
      } else {
        before = 301 * (s < 9 ? s : 9) + (s > 9 ? 175 : 0);
        sz2 = 55; sz3 = 120; sz4 = 126;
      }
      const int maxL = runlen < hi ? runlen : hi;
      for (int L = 2; L <= maxL && s + L <= 12; ++L) {
        if (!(LEAD || (L == f.ll && s > f.lv))) continue;
        // L kickers of distinct ranks outside the run (card.py:110-129): without L such ranks in the hand the whole id
        // block (55-330 ids, 1-6 rounds) holds no legal move
        if (PRUNE && __builtin_popcount((mult == 1 ? m1 : m2) & ~(((1u << L) - 1u) << s)) < L) continue;
        const int idb = idb0 + before + (L > 2 ? sz2 : 0) + (L > 3 ? sz3 : 0) + (L > 4 ? sz4 : 0);
        const int size = L == 2 ? sz2 : L == 3 ? sz3 : L == 4 ? sz4 : 252;
        const int list = L == 2 ? list2 : L == 3 ? list3 : L == 4 ? list4 : list5;
        const uint64_t mainnib = ((3ull * ONES) & ((1ull << (4 * L)) - 1ull)) << (4 * s);
        n = scan_combos<EM, IDS>(list, size, idb, mainnib, s, L, mult, cat, hot, hand8, lane, o, n, pk);
      }
    }
  };
  planes(THREE_ONE_LINE, 5, ID_THREE_ONE_LINE, 1, 1, 4, 6, 8);
  planes(THREE_TWO_LINE, 4, ID_THREE_TWO_LINE, 2, 3, 5, 7, 7);
  if ((m1 & JOKERS) == JOKERS) scan(ID_BIGBANG, 1);  // rocket (card.py:134, :314-315)
  if (LEAD || f.lc == FOUR_TAKE_ONE)                // card.py:139-143 (no joker pair: :142)
    for (uint32_t qm = m4 & above; qm; qm &= qm - 1) {
      const int q = __builtin_ctz(qm);
      n = scan_combos<EM, IDS>(0, 90, ID_FOUR_TAKE_ONE + 90 * q, 4ull << (4 * q), q, 1, 1, FOUR_TAKE_ONE, hot, hand8,
                               lane, o, n, pk);
    }
  if (LEAD || f.lc == FOUR_TAKE_TWO)                // card.py:148-153
    for (uint32_t qm = m4 & above; qm; qm &= qm - 1) {
      const int q = __builtin_ctz(qm);
      if (PRUNE && __builtin_popcount(m2 & ~(1u << q)) < 2) continue;  // two pairs of other ranks, or the 66-id block is empty
      n = scan_combos<EM, IDS>(2, 66, ID_FOUR_TAKE_TWO + 66 * q, 4ull << (4 * q), q, 1, 2, FOUR_TAKE_TWO, hot, hand8,
                               lane, o, n, pk);
    }
#if DDZ_NATIVE_JOKER_KICKERS
  // optional extension: quad + both jokers (a FOUR_TAKE_ONE) and two consecutive triples + both jokers (a
  // THREE_ONE_LINE of len 2) -- the vectors server/mcts/get_moves.py:22-34 lists; one lane each, last ids
  if ((m1 & JOKERS) == JOKERS) {
    const uint32_t mm = m3 & M12;
    const uint32_t quads = (LEAD || f.lc == FOUR_TAKE_ONE) ? (m4 & above) : 0u;
    const uint32_t pairs3 = (LEAD || (f.lc == THREE_ONE_LINE && f.ll == 2)) ? (mm & (mm >> 1) & above) : 0u;
    if (EM == EM_TEAM_PLAN) {
      if (quads | pairs3) {
        if (pk.rc < TEAM_ROUNDS) pk.desc[pk.rc] = make_uint2(team_x(24, ID_JK_FOUR, 0), 2u | quads << 2 | pairs3 << 15);
        ++pk.rc;
      }
    } else if (quads | pairs3) {
      const bool isq = lane < 13;
      const int r = isq ? lane : (lane - 13) & 15;
      const bool ok = isq ? ((quads >> r) & 1u) : (lane < 24 && ((pairs3 >> r) & 1u));
      const uint64_t jk = (1ull << 52) | (1ull << 56);
      const uint64_t nib = (isq ? (4ull << (4 * r)) : (0x33ull << (4 * r))) | jk;
      const int cat = isq ? FOUR_TAKE_ONE : THREE_ONE_LINE;
      const uint4 row = (EM == EM_WRITE || EM == EM_PICK) ? unpack_row(nib, (uint32_t)cat) : make_uint4(0, 0, 0, 0);
      n = scan_emit<EM, IDS>(ok, (isq ? ID_JK_FOUR : ID_JK_PLANE) + r, nib, cat, r | ((isq ? 1 : 2) << 8), row, o, n, pk);
    }
  }
#endif
  return n;
}


// follows of a single, a pair or a triple (two thirds of all plies): the list is pass + the higher groups of
// the led size + bombs + rocket (card.py:307-325).  One round, one lane per candidate, records built
// arithmetically: no table, no planner.  lanes: 0 pass | 1..15 group of rank lane-1 | 16..28 bomb | 29 rocket
template <int EM, bool IDS>
__device__ __forceinline__ int simple_follow(uint64_t hand, const Follow& f, int lane, const Out& o, Pick& pk) {
  const int cntr = lane < 15 ? (int)((hand >> (4 * (lane & 15))) & 15) : 0;
  const uint32_t mlc = (uint32_t)__ballot(cntr >= f.lc) & (f.lc == SINGLE ? M15 : M13);
  const uint32_t mq = (uint32_t)__ballot(cntr >= 4) & M13;
  const bool rocket = ((uint32_t)__ballot(cntr >= 1) & JOKERS) == JOKERS;
  const uint32_t okm = 1u | ((mlc & gt_mask(f.lv)) << 1) | (mq << 16) | (rocket ? 1u << 29 : 0u);
  const int r = (lane < 16 ? lane - 1 : lane - 16) & 15;
  const bool grp = lane >= 1 && lane < 16, bomb = lane >= 16 && lane < 29;
  const int cat = lane == 0 ? EMPTY : grp ? f.lc : bomb ? QUADRIC : BIGBANG;
  const uint64_t nib = lane == 0 ? 0ull : lane == 29 ? ((1ull << 52) | (1ull << 56)) : (uint64_t)(grp ? f.lc : 4) << (4 * r);
  const int id = lane == 0 ? 0 : grp ? (f.lc == SINGLE ? 1 : f.lc == DOUBLE ? 16 : 29) + r : bomb ? 42 + r : ID_BIGBANG;
  const int vl = (lane == 0 ? 0 : lane == 29 ? 100 : r) | 0x100;
  const bool ok = lane < 30 && ((okm >> (lane & 31)) & 1u);
  const uint4 row = (EM == EM_WRITE || EM == EM_PICK) ? unpack_row(nib, (uint32_t)cat) : make_uint4(0, 0, 0, 0);
  return scan_emit<EM, IDS>(ok, id, nib, cat, vl, row, o, 0, pk);
}

// lead and follow are separate instantiations: on lead every gate is true at compile time
template <int EM, bool IDS, class HT>
__device__ __forceinline__ int plan_scan(uint64_t hand, uint32_t info, const HT& hot, int lane, const Out& o, Pick& pk) {
  if (hand == 0 || (info & (QF_FROZEN | QF_BADLAST))) return 0;  // utils.py:48-49
  const Follow f = follow_of(info);
  if (!f.lead && f.lc <= TRIPLE) return simple_follow<EM, IDS>(hand, f, lane, o, pk);
  return f.lead ? plan_scan_t<EM, IDS, true>(hand, f, hot, lane, o, pk) : plan_scan_t<EM, IDS, false>(hand, f, hot, lane, o, pk);
}

// every thread of the block copies its share of the hot records into LDS (callers issue
// their own independent global loads first so that all of them are in flight together)
// (OFF: the block's first OFF threads do not take part and do not call)
template <int NT, class HT, int OFF = 0>
__device__ __forceinline__ void hot_fill(HT& hot) {
#pragma unroll
  for (int i = (int)threadIdx.x - OFF; i < HOT_SLOTS; i += NT - OFF) {
    const int id = i < HOT_IDS ? i : ID_BIGBANG;
    hot.meta[i] = g_tab[2 * id + 1];
    if (HT::HAS_ROWS) hot.rows[i] = g_tab[2 * id];
  }
#pragma unroll
  for (int i = (int)threadIdx.x - OFF; i < COMBO_WORDS; i += NT - OFF) {
    if constexpr (HT::HAS_C64) hot.c64[i] = g_c64[i];
    else hot.combo[i] = g_combo[i];
  }
}

// ------------------------------------------------------------------------------------
// deal spec v2: card k = 0..53 (rank k/4, 52 = BJ, 53 = CJ) draws the key
// philox(gid, episode, 1<<16 | k/4)[k%4]; cards are ranked by (key, k); ranks 0..16 go to
// role 0 (up), 17..36 to role 1 (lord), 37..53 to role 2 (down) (envi.py:23).
// Lane-parallel: lane k owns card k; hands fall out of three ballots.
__device__ __forceinline__ uint64_t nib_from_cards(uint64_t b) {  // bit k = card k held
  uint64_t x = b & 0x000FFFFFFFFFFFFFull;
  x = x - ((x >> 1) & 0x5555555555555555ull);
  x = (x & 0x3333333333333333ull) + ((x >> 2) & 0x3333333333333333ull);
  return x | (((b >> 52) & 1ull) << 52) | (((b >> 53) & 1ull) << 56);
}
__device__ inline void deal_wave(uint64_t gid, uint32_t episode, uint32_t k0, uint32_t k1, int lane,
                                 uint64_t& o0, uint64_t& o1, uint64_t& o2) {
  const uint4 d = philox4x32_10(make_uint4((uint32_t)gid, (uint32_t)(gid >> 32), episode, (1u << 16) | (uint32_t)(lane >> 2)), k0, k1);
  const int w = lane & 3;
  const uint32_t key = w == 0 ? d.x : w == 1 ? d.y : w == 2 ? d.z : d.w;
  // rank by (key, card index): one 64-bit compare of (key << 6 | index) per card
  const uint64_t kk = ((uint64_t)key << 6) | (uint32_t)lane;
  int pos = 0;
#pragma unroll 6
  for (int j = 0; j < 54; ++j) pos += rl64(kk, j) < kk ? 1 : 0;
  const bool card = lane < 54;
  o0 = nib_from_cards(__ballot(card && pos < 17));
  o1 = nib_from_cards(__ballot(card && pos >= 17 && pos < 37));
  o2 = nib_from_cards(__ballot(card && pos >= 37));
}

// The same deal with the ranking through LDS: the 54 keys stored once, every lane reads them back two at a time (uniform
// address: a broadcast) instead of 108 v_readlane -- a lone wave on its SIMD issues one of those per ~25 cycles (k_slab with
// one table per wave, where a deal sits on the launch's critical path).  `keys`: 64 x 8 bytes of the calling wave.
__device__ inline void deal_wave_lds(uint64_t gid, uint32_t episode, uint32_t k0, uint32_t k1, int lane, uint64_t* keys,
                                     uint64_t& o0, uint64_t& o1, uint64_t& o2) {
  const uint4 d = philox4x32_10(make_uint4((uint32_t)gid, (uint32_t)(gid >> 32), episode, (1u << 16) | (uint32_t)(lane >> 2)), k0, k1);
  const int w = lane & 3;
  const uint32_t key = w == 0 ? d.x : w == 1 ? d.y : w == 2 ? d.z : d.w;
  const uint64_t kk = ((uint64_t)key << 6) | (uint32_t)lane;
  keys[lane] = kk;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  int pos = 0;
#pragma unroll 9
  for (int j = 0; j < 54; j += 2) {
    const ulonglong2 v = *(const ulonglong2*)&keys[j];
    pos += (v.x < kk ? 1 : 0) + (v.y < kk ? 1 : 0);
  }
  const bool card = lane < 54;
  o0 = nib_from_cards(__ballot(card && pos < 17));
  o1 = nib_from_cards(__ballot(card && pos >= 17 && pos < 37));
  o2 = nib_from_cards(__ballot(card && pos >= 37));
}

// the combo to beat: previous player's handout, else the one before, else lead
// (envi.py:103-109).  (n1,c1)/(n2,c2) = nib + category byte of recent[(role-1)%3] / [(role-2)%3].
__device__ __forceinline__ uint32_t last_info(uint64_t n1, int c1, uint64_t n2, int c2) {
  if (n1) return info_of_row(n1, c1);
  if (n2) return info_of_row(n2, c2);
  return mk_info(EMPTY, 0, 1);
}

// Phase stamps: s_memtime deltas per phase of a wave, written to a debug buffer nothing else reads -- in -DDDZ_STAMP
// builds only (tools/stamp_slab.py, tools/stamp_auto.py); in the product build Stamps is an empty object and every
// mark() / store() compiles to nothing.
#ifdef DDZ_STAMP
__device__ unsigned long long* g_stamps = nullptr;  // [T][16]
template <int N>
struct Stamps {
  unsigned long long acc[N] = {}, last = __builtin_amdgcn_s_memtime();
  __device__ __forceinline__ void mark(int k) {
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    acc[k] += now - last;
    last = now;
  }
  __device__ __forceinline__ void set(int k, unsigned long long v) { acc[k] = v; }
  __device__ __forceinline__ void store(int64_t slot, bool writer) const {
    if (g_stamps && writer)
      for (int q = 0; q < N; ++q) g_stamps[16 * slot + q] = acc[q];
  }
};
#else
template <int N>
struct Stamps {
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void set(int, unsigned long long) {}
  __device__ __forceinline__ void store(int64_t, bool) const {}
};
#endif
#include "ddz_auto.h"
#include "ddz_auto2.h"


struct TableArgs {
  uint8_t* state;
  int64_t T;
  int tpw;                   // tables per wave (consecutive)
  uint32_t k0, k1;           // philox key = seed
  uint64_t gid_base;
  int auto_reset;
  const void* sel;           // CHOICE: int32[T]; ROWS: int8[T][16]; RESET: u8 mask[T] or null
  int32_t* offsets;          // CSR of the current state (written by F_ENUM, read by a bare F_STEP)
  uint4* rows;
  int32_t* ids;
  int64_t cap;
  const int32_t* cur_counts; // sizes / scan of the CURRENT state's lists (previous launch)
  const int32_t* cur_local;
  const int32_t* cur_blk;
  int32_t* nxt_counts;       // same for the state this launch produces
  int32_t* nxt_local;
  int32_t* nxt_blk;
  uint8_t* done;
  int8_t* reward;
  uint8_t* illegal;
  uint4* traj;               // [T][2]
  int64_t* blk_stats;
  int32_t* status;
  int64_t* legal_rows;
  int32_t* slab_counts;      // F_SLAB: [T] list sizes; table t owns rows[t * stride ...] (no cross-table scan)
  int64_t stride;
};

template <int FLAGS, int MODE, bool IDS>
__global__ __launch_bounds__(TB, 4) void k_table(TableArgs a) {
  constexpr bool ENUM = FLAGS & F_ENUM, STEP = FLAGS & F_STEP, RESET = FLAGS & F_RESET;
  // F_SLAB: the lists live in fixed-stride slabs, so a launch applies the selections to the lists of the
  // previous launch AND writes the lists of the new state: one launch per lock-step iteration, no CSR scan
  constexpr bool SLAB = (FLAGS & F_SLAB) != 0;
  constexpr bool COUNT = !SLAB && (FLAGS & (F_STEP | F_RESET | F_COUNT)) != 0;
  constexpr bool PICK = ENUM && STEP && MODE == DDZ_STEP_RANDOM;
  Stamps<8> stamps;
  const int lane = threadIdx.x & 63;
  const int wv = (int)rfl(threadIdx.x >> 6);
  const int64_t t0 = ((int64_t)blockIdx.x * WPB + wv) * a.tpw;
  const int ntab = t0 < a.T ? (int)(a.T - t0 < a.tpw ? a.T - t0 : a.tpw) : 0;  // wave-uniform
  __shared__ HotTab hot;
  int64_t base = 0;
  int cnt_l = 0;      // lane i: size of the current list of table t0 + i
  int new_cnt_l = 0;  // lane i: size of the next list of table t0 + i
  int s_ply = 0, s_eps = 0, s_lord = 0, s_up = 0;
  int64_t slab_rows = 0;
  // all independent global loads of the prologue are issued before anything waits
  int part = 0, loc0 = 0;
  uint4 Rnext = make_uint4(0, 0, 0, 0);
  if (ntab > 0 && lane < DDZ_NFIELDS) Rnext = ((const uint4*)(a.state + t0 * STATE_ROW_BYTES))[lane];
  if (ENUM && ntab > 0) {
    // CSR base of table t0: totals of the blocks before this one + offset inside the block
    for (int j = lane; j < (int)blockIdx.x; j += 64) part += a.cur_blk[j];
    loc0 = a.cur_local[t0];
    if (lane < ntab) cnt_l = a.cur_counts[t0 + lane];
  }
  uint4 pre_row = make_uint4(0, 0, 0, 0);  // F_SLAB + CHOICE: lane i prefetches the selected row of table t0 + i
  int pre_idx = -1;
  if (SLAB && STEP && lane < ntab) {
    cnt_l = a.slab_counts[t0 + lane];
    if (cnt_l < 0 || cnt_l > a.stride) cnt_l = 0;
    if (MODE == DDZ_STEP_CHOICE) {
      pre_idx = ((const int32_t*)a.sel)[t0 + lane];
      if (pre_idx < 0 || pre_idx >= cnt_l) pre_idx = -1;
      if (pre_idx >= 0) pre_row = a.rows[(t0 + lane) * a.stride + pre_idx];
    }
  }
  stamps.mark(0);  // prologue loads issued
  hot_fill<TB>(hot);
  __syncthreads();
  stamps.mark(1);  // hot fill + barrier
  if (ENUM && ntab > 0) base = (int64_t)wave_sum(part) + loc0;
  for (int i = 0; i < ntab; ++i) {
    const int64_t t = t0 + i;
    uint4* trow = (uint4*)(a.state + t * STATE_ROW_BYTES);
    uint4 R = Rnext;
    if (i + 1 < ntab && lane < DDZ_NFIELDS) Rnext = ((const uint4*)(a.state + (t + 1) * STATE_ROW_BYTES))[lane];
    uint64_t P = pack_row(R);
    const uint32_t mx = rl(R.x, DDZ_F_META), my = rl(R.y, DDZ_F_META), mz = rl(R.z, DDZ_F_META);
    int role = mx & 0xFF;
    if (role > 2) role = 0;  // never index outside the table on a corrupted import
    bool is_done = (mx >> 8) & 0xFF, dealt = (my >> 16) & 0xFF;
    uint32_t ply = my & 0xFFFF, episode = mz;
    const uint64_t gid = a.gid_base + (uint64_t)t;
    const bool active = dealt && !is_done;
    const int rm1 = role == 0 ? 2 : role - 1, rp1 = role == 2 ? 0 : role + 1;
    uint64_t hand = rl64(P, DDZ_F_HAND0 + role);
    const uint64_t n1 = rl64(P, DDZ_F_RECENT0 + rm1);  // what the previous player played (0 = pass)
    const int cat1 = (int)(rl(R.w, DDZ_F_RECENT0 + rm1) >> 24);
    uint32_t info = last_info(n1, cat1, rl64(P, DDZ_F_RECENT0 + rp1), (int)(rl(R.w, DDZ_F_RECENT0 + rp1) >> 24));
    // the query of the NEXT actor follows from this decode (no second decode of the updated rows): its hand is
    // untouched by this ply, and it has to beat this ply's combo, or -- after a pass -- the previous player's
    uint64_t qhand = (COUNT || SLAB) ? rl64(P, DDZ_F_HAND0 + rp1) : 0;
    uint32_t qinfo = mk_info(EMPTY, 0, 1);
    int cnt = 0;
    Pick pk{-1, 0, 0, 0, 0};
    stamps.mark(2);  // decode
    if (ENUM) {
      cnt = (int)rl((uint32_t)cnt_l, i);
      if (lane == 0) a.offsets[t] = (int32_t)base;
      if (PICK && active && cnt > 0) {  // random.choice(actions), envi.py:83
        const uint4 d = philox4x32_10(make_uint4((uint32_t)gid, (uint32_t)(gid >> 32), episode, (2u << 16) | ply), a.k0, a.k1);
        pk.want = (int)__umulhi(rfl(d.x), (uint32_t)cnt);
      }
      const Out o{a.rows, a.ids, base, a.cap, nullptr, nullptr, nullptr};
      const int n = plan_scan<PICK ? EM_PICK : EM_WRITE, IDS>(hand, active ? info : QF_FROZEN, hot, lane, o, pk);
      if (lane == 0) {
        const int bits = (n != cnt ? 1 : 0) | (base + cnt > a.cap ? 2 : 0);
        if (bits) atomicOr(a.status, bits);
      }
    }
    bool changed = false;
    if (STEP) {
      int64_t off = base;
      int A = cnt;
      if (SLAB) {
        off = t * a.stride;
        A = (int)rl((uint32_t)cnt_l, i);
      } else if (!ENUM) {
        off = a.offsets[t];
        A = a.offsets[t + 1] - (int32_t)off;
      }
      if (!SLAB && (off < 0 || off + A > a.cap)) A = 0;  // list was truncated: nothing to pick from
      const bool frozen = !active || A <= 0;
      int idx = -1;
      uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;  // the chosen row, wave-uniform
      int32_t sel_id = 0;  // DDZ_STEP_IDS: canonical action id, -1 = engine RNG
      if (MODE == DDZ_STEP_IDS) sel_id = (int32_t)rfl((uint32_t)((const int32_t*)a.sel)[t]);
      if (!frozen) {
        if (MODE == DDZ_STEP_RANDOM || (MODE == DDZ_STEP_IDS && sel_id == -1)) {
          if (PICK) {
            idx = pk.want; c0 = pk.r0; c1 = pk.r1; c2 = pk.r2; c3 = pk.r3;
          } else {
            const uint4 d = philox4x32_10(make_uint4((uint32_t)gid, (uint32_t)(gid >> 32), episode, (2u << 16) | ply), a.k0, a.k1);
            idx = (int)__umulhi(rfl(d.x), (uint32_t)A);
          }
        } else if (MODE == DDZ_STEP_CHOICE) {
          if (SLAB) {
            idx = (int)rl((uint32_t)pre_idx, i);
          } else {
            idx = ((const int32_t*)a.sel)[t];
            if (idx < 0 || idx >= A) idx = -1;
          }
        } else {  // wave-parallel search of the segment for the wanted counts
          constexpr int NIDS = DDZ_NUM_ACTIONS + 24 * DDZ_NATIVE_JOKER_KICKERS;
          const bool id_ok = MODE != DDZ_STEP_IDS || (sel_id >= 0 && sel_id < NIDS);
          const uint4 want = MODE == DDZ_STEP_IDS ? row_of_id(id_ok ? sel_id : 0) : ((const uint4*)a.sel)[t];
          for (int j0 = 0; j0 < A && idx < 0 && id_ok; j0 += 64) {
            bool hit = false;
            if (j0 + lane < A) {
              const uint4 r = a.rows[off + j0 + lane];
              hit = r.x == want.x && r.y == want.y && r.z == want.z && ((r.w ^ want.w) & 0x00FFFFFFu) == 0;
            }
            const uint64_t hb = __ballot(hit);
            if (hb) idx = j0 + __builtin_ctzll(hb);
          }
        }
        if (idx >= 0 && SLAB && MODE == DDZ_STEP_CHOICE) {
          c0 = rl(pre_row.x, i); c1 = rl(pre_row.y, i); c2 = rl(pre_row.z, i); c3 = rl(pre_row.w, i);
        } else if (idx >= 0 && !PICK) {
          const uint4 r = a.rows[off + idx];
          c0 = rfl(r.x); c1 = rfl(r.y); c2 = rfl(r.z); c3 = rfl(r.w);
        }
      }
      uint4 tr0 = make_uint4(0, 0, 0, 0);
      uint4 tr1 = make_uint4((uint32_t)role, ((uint32_t)A & 0xFFFF) | (ply << 16), episode, 0xFFFFFFFFu);
      uint32_t o_done = is_done, o_illegal = 0, o_reward = 0;
      if (frozen) {
        tr1.x |= (uint32_t)is_done << 8 | 2u << 24;
      } else if (idx < 0) {
        o_done = 0; o_illegal = 1;
        tr1.x |= 1u << 24;
      } else {
        changed = true;
        const uint32_t cw3 = c3 & 0x00FFFFFFu;
        const uint64_t cn = pack_row(make_uint4(c0, c1, c2, c3));
        const uint32_t ncards = (uint32_t)nib_sum(cn);
        qinfo = cn ? info_of_row(cn, (int)(c3 >> 24)) : (n1 ? info_of_row(n1, cat1) : mk_info(EMPTY, 0, 1));
        // lane-parallel update of the table's rows (envi.py:39-43); byte-wise: every byte
        // of the hand >= the row's byte, so no borrow/carry crosses a byte
        if (lane == DDZ_F_HAND0 + role) {
          R.x -= c0; R.y -= c1; R.z -= c2; R.w -= cw3 + (ncards << 24);
        } else if (lane == DDZ_F_HIST0 + role || lane == DDZ_F_TAKEN) {
          R.x += c0; R.y += c1; R.z += c2; R.w += cw3;
        } else if (lane == DDZ_F_RECENT0 + role) {
          R = make_uint4(c0, c1, c2, c3);
        }
        const bool won = (rl(R.w, DDZ_F_HAND0 + role) >> 24) == 0;
        o_reward = won ? (role == 1 ? 0xFFu : 1u) : 0u;  // rule_play.py:14: -1 lord won, +1 farmers
        o_done = won;
        s_ply += 1; s_eps += won; s_lord += (won && role == 1); s_up += (won && role == 0);
        tr0 = make_uint4(c0, c1, c2, c3);
        tr1.x |= (uint32_t)won << 8 | o_reward << 16;
        tr1.w = (uint32_t)idx;
        const int nrole = rp1;  // lord -> down -> up, game.py:173-181
        ply += 1;
        if (won && a.auto_reset) {
          episode += 1;
          uint64_t h0, h1, h2;
          deal_wave(gid, episode, a.k0, a.k1, lane, h0, h1, h2);
          R = lane == 0 ? unpack_row(h0, 17) : lane == 1 ? unpack_row(h1, 20) : lane == 2 ? unpack_row(h2, 17)
              : lane == DDZ_F_META ? make_uint4(1u | (0xFFu << 16), 1u << 16, episode, 0) : make_uint4(0, 0, 0, 0);
          role = 1; is_done = false;
          qhand = h1; qinfo = mk_info(EMPTY, 0, 1);  // the lord leads
        } else {
          if (lane == DDZ_F_META)
            R = make_uint4((uint32_t)nrole | (won ? 1u << 8 : 0u) | ((won ? (uint32_t)role : 0xFFu) << 16) | (o_reward << 24),
                           (my & 0xFFFF0000u) | (ply & 0xFFFF), mz, R.w);
          role = nrole; is_done = won;
        }
      }
      if (lane == 0) {
        if (a.done) a.done[t] = (uint8_t)o_done;
        if (a.reward) a.reward[t] = (int8_t)o_reward;
        if (a.illegal) a.illegal[t] = (uint8_t)o_illegal;
      }
      if (a.traj && lane < 2) a.traj[2 * t + lane] = sel4(lane == 0, tr0, tr1);
    }
    if (RESET) {
      const uint8_t* mask = (const uint8_t*)a.sel;
      if (!mask || mask[t]) {
        changed = true;
        episode = dealt ? episode + 1 : 0;
        uint64_t h0, h1, h2;
        deal_wave(gid, episode, a.k0, a.k1, lane, h0, h1, h2);
        R = lane == 0 ? unpack_row(h0, 17) : lane == 1 ? unpack_row(h1, 20) : lane == 2 ? unpack_row(h2, 17)
            : lane == DDZ_F_META ? make_uint4(1u | (0xFFu << 16), 1u << 16, episode, 0) : make_uint4(0, 0, 0, 0);
        role = 1; is_done = false; dealt = true;
        qhand = h1; qinfo = mk_info(EMPTY, 0, 1);
      }
    }
    stamps.mark(3);  // select + apply + outputs
    if (changed) {
      if (lane < DDZ_NFIELDS) trow[lane] = R;  // one coalesced 176-byte store
      if (COUNT || SLAB) {                     // query of the new actor
        hand = qhand;
        info = qinfo;
      }
    }
    if (COUNT) {
      const Out none{nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr};
      Pick nopk{-1, 0, 0, 0, 0};
      const int c = (dealt && !is_done) ? plan_scan<EM_COUNT, false>(hand, info, hot, lane, none, nopk) : 0;
      if (lane == i) new_cnt_l = c;
    }
    if (SLAB) {  // the list of the (new) state, straight into the table's slab
      const int64_t sb = t * a.stride;
      const Out o{a.rows, a.ids, sb, sb + a.stride, nullptr, nullptr, nullptr};
      Pick nopk{-1, 0, 0, 0, 0};
      int n = (dealt && !is_done) ? plan_scan<EM_WRITE, IDS>(hand, info, hot, lane, o, nopk) : 0;
      if (n > a.stride) {  // cannot happen for a <= 20-card hand with the default stride
        if (lane == 0) atomicOr(a.status, 2);
        n = 0;
      }
      if (lane == 0) a.slab_counts[t] = n;
      slab_rows += n;
    }
    stamps.mark(4);  // state store + list of the new state
    base += cnt;
  }
  stamps.set(5, (unsigned long long)ntab);
  stamps.store(t0, lane == 0 && ntab > 0 && SLAB && STEP);
  if (ENUM && ntab > 0 && t0 + ntab == a.T && lane == 0) {
    a.offsets[a.T] = (int32_t)base;
    *a.legal_rows += base;
  }
  if (SLAB && ntab > 0 && lane == 0) {  // each wave owns its statistics slot (as in k_rollout)
    int64_t* ws = a.blk_stats + 4 * ((int64_t)blockIdx.x * WPB + wv);
    if (STEP) { ws[0] += s_ply; ws[1] += s_eps; ws[2] += (int64_t)s_lord | ((int64_t)s_up << 32); }
    ws[3] += slab_rows;
  }
  if (COUNT) {
    __shared__ int sh[WPB][4];
    const int incl = wave_incl_scan(new_cnt_l, lane);
    if (lane == 63) {
      sh[wv][0] = incl; sh[wv][1] = s_ply; sh[wv][2] = s_eps; sh[wv][3] = s_lord | (s_up << 16);
    }
    __syncthreads();
    int wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WPB; ++w) {
      if (w < wv) wbase += sh[w][0];
      tot += sh[w][0];
    }
    if (lane < ntab) {
      a.nxt_counts[t0 + lane] = new_cnt_l;
      a.nxt_local[t0 + lane] = wbase + incl - new_cnt_l;
    }
    if (threadIdx.x == 0) {
      a.nxt_blk[blockIdx.x] = tot;
      if (STEP) {
        int64_t* bs = a.blk_stats + 4 * (int64_t)blockIdx.x;
        int p = 0, e = 0, l = 0, u = 0;
        for (int w = 0; w < WPB; ++w) { p += sh[w][1]; e += sh[w][2]; l += sh[w][3] & 0xFFFF; u += sh[w][3] >> 16; }
        bs[0] += p; bs[1] += e; bs[2] += (int64_t)l | ((int64_t)u << 32);
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// k_rollout: the random-policy lock-step iteration (game.py:169-181 with envi.py:79-85) with
// NO dependency between tables: every table owns a fixed-stride slab of the list buffer
// (rows[t * stride ...], counts[t]), so there is no cross-table scan, no count pass and no
// block barrier after the prologue.  One wavefront per table:
//   scan once, staging nib|category of every legal move in the wave's LDS list (its length
//   is the list size, known before the pick) -> flush the list as coalesced 16-byte rows
//   -> pick by engine RNG from the staged list -> apply / terminal / deal -> store state.
struct RolloutArgs {
  uint8_t* state;
  int64_t T;
  int tpw;
  uint32_t k0, k1;
  uint64_t gid_base;
  int32_t* counts;   // [T] list sizes
  uint4* rows;       // [T][stride] list slabs
  int32_t* ids;      // [T][stride] or null
  int64_t stride;
  int64_t it_rows;   // staged form (ddz_rollout_random_csr_staged): iteration j of the launch writes its lists into slab j --
  int64_t it_counts; // rows / ids advance by it_rows, counts by it_counts per iteration (0, 0: every iteration overwrites slab 0)
  int64_t n_iters;   // lock-step iterations run inside this launch
  uint4* traj;       // [n_iters][T][2] or null
  int64_t* wave_stats;
  int32_t* status;
  int64_t* legal_rows;
};

// STAGED (ddz_rollout_random_csr_staged): iteration j of the launch writes its lists into slab j (a template parameter: the
// per-iteration address arithmetic costs the plain rollout 12 % when it is a run-time option).
// RW = wavefronts per block: 16 (one block per CU: 4 waves per SIMD), or 12 for the variants that fit 85 VGPRs -- two blocks of
// 74 KB per CU = 6 waves per SIMD (round 4: the kernel waits a third of its wave cycles; more waves in flight cover them).
template <bool IDS, bool TRAJ, bool STAGED = false, int RW = WPB>
__global__ __launch_bounds__(RW * 64, RW == WPB ? 4 : 6) void k_rollout(RolloutArgs a) {
  Stamps<12> stamps;
  __shared__ HotTabT<false> hot;
  __shared__ uint64_t s_stage[RW][STAGE_CAP];
  __shared__ uint16_t s_svl[RW][STAGE_CAP];
  __shared__ uint16_t s_sid[IDS ? RW : 1][IDS ? STAGE_CAP : 1];
  const int lane = threadIdx.x & 63;
  const int wv = (int)rfl(threadIdx.x >> 6);
  const int64_t wave = (int64_t)blockIdx.x * RW + wv;
  const int64_t t0 = wave * a.tpw;
  const int ntab = t0 < a.T ? (int)(a.T - t0 < a.tpw ? a.T - t0 : a.tpw) : 0;
  uint4 Rnext = make_uint4(0, 0, 0, 0);
  if (ntab > 0 && lane < DDZ_NFIELDS) Rnext = ((const uint4*)(a.state + t0 * STATE_ROW_BYTES))[lane];
  hot_fill<RW * 64>(hot);
  __syncthreads();
  uint64_t* stage = s_stage[wv];
  uint16_t* svl = s_svl[wv];
  uint16_t* sid = s_sid[IDS ? wv : 0];
  int s_ply = 0;
  int64_t s_rows = 0;
  // per-lane constants of the fast path (VGPRs; opaque so that they are not rebuilt from spilled lane masks)
  const int f_rr = (lane < 16 ? lane - 1 : lane - 16) & 15;
  uint32_t f_bit = lane < 32 ? 1u << lane : 0u;                     // this lane's bit in a 32-bit candidate mask
  uint32_t f_sh = 8u * (uint32_t)(f_rr & 3);                         // byte position of the rank in its row word
  uint32_t f_w0 = (f_rr >> 2) == 0 ? ~0u : 0u, f_w1 = (f_rr >> 2) == 1 ? ~0u : 0u;
  uint32_t f_w2 = (f_rr >> 2) == 2 ? ~0u : 0u, f_w3 = (f_rr >> 2) == 3 ? ~0u : 0u;
  uint32_t f_grp = (lane >= 1 && lane < 16) ? ~0u : 0u;             // lanes 1..15: a group of the led size
  uint32_t f_c4 = (lane >= 16 && lane < 29) ? 4u : 0u;              // lanes 16..28: a bomb
  uint32_t f_rk = lane == 29 ? (0x00010100u | ((uint32_t)BIGBANG << 24)) : 0u;  // lane 29: the rocket row
  asm volatile("" : "+v"(f_bit), "+v"(f_sh), "+v"(f_w0), "+v"(f_w1), "+v"(f_w2), "+v"(f_w3), "+v"(f_grp), "+v"(f_c4), "+v"(f_rk));
  for (int i = 0; i < ntab; ++i) {
    const int64_t t = t0 + i;
    stamps.mark(0);  // prologue (or previous table's tail)
    uint4* trow = (uint4*)(a.state + t * STATE_ROW_BYTES);
    uint4 R = Rnext;  // lane f < 11 holds row f of the table for the whole launch
    if (i + 1 < ntab && lane < DDZ_NFIELDS) Rnext = ((const uint4*)(a.state + (t + 1) * STATE_ROW_BYTES))[lane];
    const uint64_t gid = a.gid_base + (uint64_t)t;
    // decode once; afterwards the table's scalars are carried across the iterations
    const uint64_t P = pack_row(R);
    const uint32_t mx = rl(R.x, DDZ_F_META), my = rl(R.y, DDZ_F_META), mz = rl(R.z, DDZ_F_META);
    const uint32_t my_hi = my & 0xFFFF0000u;
    int role = mx & 0xFF;
    if (role > 2) role = 0;
    const bool active = ((my >> 16) & 0xFF) && !((mx >> 8) & 0xFF);  // dealt and not done
    uint32_t ply = my & 0xFFFF, episode = mz;
    // hands in turn order: hc = the actor's, hn = the next player's, hp = the previous player's; a ply
    // rotates the three names (register moves) instead of selecting by role twice
    uint64_t hc = rl64(P, DDZ_F_HAND0 + role), hn = rl64(P, DDZ_F_HAND0 + (role == 2 ? 0 : role + 1)),
             hp = rl64(P, DDZ_F_HAND0 + (role == 0 ? 2 : role - 1));
    // the combo to beat (envi.py:103-109) as (trick, passes since it was played)
    uint32_t trick = mk_info(EMPTY, 0, 1);
    int passes = 0;
    {
      const int rm1 = role == 0 ? 2 : role - 1, rp1 = role == 2 ? 0 : role + 1;
      const uint64_t n1 = rl64(P, DDZ_F_RECENT0 + rm1), n2 = rl64(P, DDZ_F_RECENT0 + rp1);
      if (n1) trick = info_of_row(n1, (int)(rl(R.w, DDZ_F_RECENT0 + rm1) >> 24));
      else if (n2) { trick = info_of_row(n2, (int)(rl(R.w, DDZ_F_RECENT0 + rp1) >> 24)); passes = 1; }
    }
    // engine RNG draws for 64 consecutive plies at once: lane j holds the draw of ply dbase + j
    uint32_t draws = 0, dnext = 64;  // lane dnext holds the next draw; 64 = refill
    uint4* tj = TRAJ ? a.traj + 2 * t : nullptr;  // record of (iteration, table)
    if (!active) {  // frozen table (never dealt / finished without auto-reset): empty lists, flagged records
      if (lane == 0) {
        a.counts[t] = 0;
        if (STAGED)
          for (int64_t j = 1; j < a.n_iters; ++j) a.counts[t + j * a.it_counts] = 0;
      }
      if (TRAJ) {
        const uint4 f0 = make_uint4(0, 0, 0, 0);
        const uint4 f1 = make_uint4((uint32_t)role | ((mx >> 8) & 0xFF) << 8 | 2u << 24, ply << 16, episode, 0xFFFFFFFFu);
        for (int it = (int)a.n_iters; it > 0; --it) {
          if (lane < 2) tj[lane] = sel4(lane == 0, f0, f1);
          tj += 2 * a.T;
        }
      }
      continue;
    }
    for (int it = (int)a.n_iters; it > 0; --it) {
      uint4 tr0 = make_uint4(0, 0, 0, 0);
      uint4 tr1 = make_uint4((uint32_t)role, ply << 16, episode, 0xFFFFFFFFu);
      if (dnext >= 64u) {
        dnext = 0;
        uint32_t qk0 = a.k0, qk1 = a.k1;  // opaque: round keys are computed here, not hoisted and spilled
        asm volatile("" : "+s"(qk0), "+s"(qk1));
        draws = philox4x32_10(make_uint4((uint32_t)gid, (uint32_t)(gid >> 32), episode, (2u << 16) | (ply + (uint32_t)lane)), qk0, qk1).x;
      }
      const uint64_t hand = hc;
      // rfl: the combo to beat stays wave-uniform for the compiler, so the category dispatch below is
      // scalar branches (the carried state itself lives in VGPRs: measured faster than on the scalar unit)
      const uint32_t info = rfl((passes >= 2) ? mk_info(EMPTY, 0, 1) : trick);
      const int64_t jn = STAGED ? a.n_iters - it : 0;         // iteration of this launch (staged form: its slab)
      const int64_t base = t * a.stride + (STAGED ? jn * a.it_rows : 0);
      int32_t* const cnt_p = a.counts + t + (STAGED ? jn * a.it_counts : 0);
      const uint32_t draw = rl(draws, (int)dnext);
      stamps.mark(1);  // per-iteration setup: frozen check, draw refresh, hand/info select
      int n = 0, idx = -1;
      uint4 c = make_uint4(0, 0, 0, 0);  // the chosen row, same value in every lane
      uint64_t snib = 0;                  // ... as a nib, its category and value | len << 8
      uint32_t scat = 0, svlv = 0, ncards = 0;  // ... and its number of cards
      const int lc0 = (int)(info & 0xFF);
      if (lc0 != EMPTY && lc0 <= TRIPLE) {
        // Two thirds of all plies follow a single, a pair or a triple.  Their
        // legal list is pass + the higher groups of the same size + bombs + rocket
        // (card.py:307-325): one lane per candidate, rows built arithmetically and stored
        // straight into the slab; no record table, no staging.
        //   lanes: 0 pass | 1..15 group of rank lane-1 | 16..28 bomb of rank lane-16 | 29 rocket
        const int lv0 = (int)((info >> 8) & 0xFF);
        const int cntr = lane < 15 ? (int)((hand >> (4 * (lane & 15))) & 15) : 0;
        const uint32_t mlc = (uint32_t)__ballot(cntr >= lc0) & (lc0 == SINGLE ? M15 : M13);
        const uint32_t mq = (uint32_t)__ballot(cntr >= 4) & M13;
        const bool rocket = ((uint32_t)__ballot(cntr >= 1) & JOKERS) == JOKERS;
        const uint32_t ab = gt_mask(lv0);
        const uint32_t cand = mlc & ab;
        const uint32_t bombs = mq;
        const int rr = f_rr;
        // the candidate mask is scalar: bit 0 pass | 1..15 groups | 16..28 bombs | 29 rocket
        const uint32_t okm = 1u | (cand << 1) | (bombs << 16) | (rocket ? 1u << 29 : 0u);
        const bool ok = (okm & f_bit) != 0;
        n = __builtin_popcount(okm);
        const int pre = (int)__builtin_amdgcn_mbcnt_lo(okm, 0u);
        const uint32_t copies = ((uint32_t)lc0 & f_grp) | f_c4;  // category == group size here
        const uint32_t dv = copies << f_sh;
        const uint4 row = make_uint4(dv & f_w0, dv & f_w1, dv & f_w2, (dv & f_w3) | (copies << 24) | f_rk);
        if (ok) {
          a.rows[base + pre] = row;
          if (IDS) a.ids[base + pre] = lane == 0 ? 0 : lane < 16 ? (lc0 == SINGLE ? 1 : lc0 == DOUBLE ? 16 : lc0 == TRIPLE ? 29 : 42) + rr
                                      : lane < 29 ? 42 + rr : ID_BIGBANG;
        }
        *cnt_p = n;  // every lane stores the same word: no exec-mask change
        s_rows += n;
        stamps.mark(5);
        idx = (int)__umulhi(draw, (uint32_t)n);  // random.choice(actions), envi.py:83 (n >= 1: pass)
        const int src = __builtin_ctz((uint32_t)__ballot(pre == idx) & okm);
        const int sr = (src < 16 ? src - 1 : src - 16) & 15;
#ifndef DDZ_ROLLOUT_PICK_SCALAR
        c = make_uint4(rl(row.x, src), rl(row.y, src), rl(row.z, src), rl(row.w, src));
#endif
        if (src == 0) { snib = 0; scat = EMPTY; svlv = 1u << 8; }
        else if (src == 29) { snib = (1ull << 52) | (1ull << 56); scat = BIGBANG; svlv = 100u | (1u << 8); ncards = 2; }
        else { scat = src < 16 ? (uint32_t)lc0 : (uint32_t)QUADRIC; snib = (uint64_t)scat << (4 * sr); svlv = (uint32_t)sr | (1u << 8); ncards = scat; }
#ifdef DDZ_ROLLOUT_PICK_SCALAR
        {  // the chosen row from the chosen lane's number, scalar arithmetic instead of four v_readlane: measured SLOWER (3.78
           // against 4.06 G env steps/s at 65,536 tables: the CU's scalar unit is the busier one); kept for the record
          const uint32_t cp = (src == 0 || src == 29) ? 0u : scat, dvs = cp << (8 * (sr & 3)), wq = (uint32_t)sr >> 2;
          c = make_uint4(wq == 0 ? dvs : 0u, wq == 1 ? dvs : 0u, wq == 2 ? dvs : 0u,
                         (wq == 3 ? dvs : 0u) | (cp << 24) | (src == 29 ? (0x00010100u | ((uint32_t)BIGBANG << 24)) : 0u));
        }
#endif
        stamps.mark(5);  // fast path: list + pick
      } else {
        const Out o{nullptr, nullptr, 0, 0, stage, svl, sid};
        Pick pk{-1, 0, 0, 0, 0};
        n = plan_scan<EM_STAGE, IDS>(hand, info, hot, lane, o, pk);
        __builtin_amdgcn_wave_barrier();
        if (n > STAGE_CAP || n > a.stride) {  // cannot happen for a <= 20-card hand; never index past the slab
          if (lane == 0) atomicOr(a.status, 2);
          n = 0;
        }
        stamps.mark(2);  // scan (planner + rounds + staging)
        *cnt_p = n;
        s_rows += n;
        for (int j = lane; j < n; j += 64) {  // flush: coalesced 16-byte rows
          const uint64_t e = stage[j];
          a.rows[base + j] = unpack_row(e & 0x0FFFFFFFFFFFFFFFull, (uint32_t)(e >> 60));
          if (IDS) a.ids[base + j] = sid[j];
        }
        stamps.mark(3);  // flush rows
        if (n > 0) {
          idx = (int)__umulhi(draw, (uint32_t)n);  // random.choice(actions), envi.py:83
          const uint64_t e = stage[idx];           // LDS broadcast read
          const uint64_t anib = e & 0x0FFFFFFFFFFFFFFFull;
          const uint32_t acat = (uint32_t)(e >> 60);
          c = unpack_row(anib, acat);
          snib = (uint64_t)rfl((uint32_t)anib) | ((uint64_t)rfl((uint32_t)(anib >> 32)) << 32);
          scat = rfl(acat);
          svlv = rfl((uint32_t)svl[idx]);
          ncards = (uint32_t)nib_sum(snib);
        }
        stamps.mark(10);  // generic pick
      }
      tr1.y |= (uint32_t)n & 0xFFFF;
      if (n <= 0) {
        tr1.x |= 2u << 24;
      } else {
        if (snib) {  // a pass (half of all plies) moves no card: only recent_handout and the turn change
          const uint32_t cw3 = c.w & 0x00FFFFFFu;
          if (lane == DDZ_F_HAND0 + role) {  // envi.py:39-43, byte-wise (no borrow crosses a byte)
            R.x -= c.x; R.y -= c.y; R.z -= c.z; R.w -= cw3 + (ncards << 24);
          } else if (lane == DDZ_F_HIST0 + role || lane == DDZ_F_TAKEN) {
            R.x += c.x; R.y += c.y; R.z += c.z; R.w += cw3;
          }
        }
        if (lane == DDZ_F_RECENT0 + role) R = c;
        stamps.mark(6);  // row updates
        // carried scalars
        const uint64_t hnew = hand - snib;
        if (snib) { trick = scat | (svlv << 8); passes = 0; } else { passes += 1; }
        const bool won = hnew == 0;
        const uint32_t o_reward = won ? (role == 1 ? 0xFFu : 1u) : 0u;  // rule_play.py:14
        s_ply += 1;
        tr0 = c;
        tr1.x |= (uint32_t)won << 8 | o_reward << 16;
        tr1.w = (uint32_t)idx;
        ply += 1;
        dnext += 1;
        stamps.mark(7);  // carried scalars
        if (won) {  // auto-reset: next episode of this table
          if (lane == 0) {  // the wave owns its statistics slot: plain read-modify-write
            int64_t* ws = a.wave_stats + 4 * wave;
            ws[1] += 1;
            ws[2] += role == 1 ? 1ll : role == 0 ? (1ll << 32) : 0ll;
          }
          episode += 1;
          uint32_t dk0 = a.k0, dk1 = a.k1;
          asm volatile("" : "+s"(dk0), "+s"(dk1));
          uint64_t h0, h1, h2;
#ifdef DDZ_ROLLOUT_DEAL_READLANE
          deal_wave(gid, episode, dk0, dk1, lane, h0, h1, h2);
#else
          deal_wave_lds(gid, episode, dk0, dk1, lane, stage, h0, h1, h2);  // (the staging list is idle between two lists)
#endif
          hc = h1; hn = h2; hp = h0;  // the lord (role 1) moves first, then down (2), then up (0)
          R = lane == 0 ? unpack_row(h0, 17) : lane == 1 ? unpack_row(h1, 20) : lane == 2 ? unpack_row(h2, 17)
              : lane == DDZ_F_META ? make_uint4(1u | (0xFFu << 16), 1u << 16, episode, 0) : make_uint4(0, 0, 0, 0);
          role = 1; ply = 0; trick = mk_info(EMPTY, 0, 1); passes = 0; dnext = 64;
        } else {
          role = role == 2 ? 0 : role + 1;  // lord -> down -> up, game.py:173-181
          hc = hn; hn = hp; hp = hnew;
          if (lane == DDZ_F_META) { R.x = (uint32_t)role | (0xFFu << 16); R.y = my_hi | (ply & 0xFFFF); }  // z = episode, w: unchanged
        }
        stamps.mark(8);  // deal / turn change
        if (lane < DDZ_NFIELDS) trow[lane] = R;  // one coalesced 176-byte store
        stamps.mark(9);  // state store
      }
      if (TRAJ && lane < 2) tj[lane] = sel4(lane == 0, tr0, tr1);
      if (TRAJ) tj += 2 * a.T;
      stamps.mark(4);  // pick + apply + deal + state/trajectory stores
      __builtin_amdgcn_wave_barrier();  // the staging list is reused by the next iteration / table
    }
  }
  stamps.store(t0, lane == 0 && ntab > 0);
  if (ntab > 0 && lane == 0) {  // each wave owns its statistics slot: no atomics, no barrier
    int64_t* ws = a.wave_stats + 4 * wave;
    ws[0] += s_ply; ws[3] += s_rows;
  }
}

// write-once output streams (`face`, thermometer planes): nontemporal 16-byte stores.  Measured on k_observe
// at 524,288 tables (0.6-1.2 GB per call): 3.1 TB/s with plain stores, 5.3-5.7 TB/s with these.
__device__ __forceinline__ void store_stream(float4* p, const float4 v) {
  __builtin_nontemporal_store(v.x, &p->x);
  __builtin_nontemporal_store(v.y, &p->y);
  __builtin_nontemporal_store(v.z, &p->z);
  __builtin_nontemporal_store(v.w, &p->w);
}

// ------------------------------------------------------------------------------------
// k_slab: ONE lock-step iteration of a policy-driven loop in the slab layout (ddz_step_slab): apply the selections to
// the lists the previous launch left in the slabs, then write the lists of the new states (game.py:95-106 +
// envi.py:98-116).  Same semantics as k_table<F_STEP | F_SLAB> (which it replaces on this path), built like k_rollout:
// k_table spent 315 scalar + 47 branch instructions per table-step and was bound by the ONE scalar unit of a CU
// (16 waves x 360 scalar instructions = the measured 5.9 k cycles per round of 16 tables, profiles/r02_slab_*).  Here
// the per-table decode is lane-parallel vector work (every lane packs and classifies its own row; the few values the
// control flow needs are read with single readlanes), the chosen row of CHOICE is packed / classified by the lane
// that prefetched it, and the new list is emitted by k_rollout's code (arithmetic fast path for follows of a single /
// pair / triple, planner + LDS staging + coalesced flush otherwise).
struct SlabArgs {
  uint8_t* state;
  int64_t T;
  int tpw;
  uint32_t k0, k1;
  uint64_t gid_base;
  int auto_reset;
  const void* sel;      // CHOICE: int32[T] list index; ROWS: int8[T][16]; IDS: int32[T] action id (-1 = engine RNG)
  int32_t* counts;      // [T] list sizes (in: lists of the current states, out: of the new states)
  uint4* rows;          // [T][stride]
  int32_t* ids;         // [T][stride] or null
  int64_t stride;
  uint8_t* done;
  int8_t* reward;
  uint8_t* illegal;
  uint4* traj;          // [T][2] or null
  int64_t* wave_stats;
  int32_t* status;
  // fused policy iteration (ddz_policy_step_slab): MODE = STEP_Q reads per-move values instead of selections
  uint64_t thr;         // floor(epsilon * 2^32): explore <=> draw.x < thr (as k_select)
  int32_t* choice_out;  // [T] the selected list indices, or null
  float4* face;         // [T][P][15] `face` of the NEW states, or null
  int face_variant;
  int coop;             // tpw == 1: wave 0 of a block runs the lane-parallel phases of the block's tables
  int team;             // coop: planned leads of at least this many scan rounds are written by the whole block (0 = none)
  int lpt;              // tpw >= 2: deals + lists of a block's tables go through a block work list, most expensive first
};
constexpr int STEP_Q = 4;  // internal mode of k_slab: sel = f32 q[T][stride]

// per-lane constants of the arithmetic follow list: lanes 0 pass | 1..15 group of rank lane-1 | 16..28 bomb | 29 rocket
struct FastLanes {
  int rr;
  uint32_t bit, sh, w0, w1, w2, w3, grp, c4, rk;
};
__device__ __forceinline__ FastLanes fast_lanes(int lane) {
  FastLanes f;
  f.rr = (lane < 16 ? lane - 1 : lane - 16) & 15;
  f.bit = lane < 32 ? 1u << lane : 0u;
  f.sh = 8u * (uint32_t)(f.rr & 3);
  f.w0 = (f.rr >> 2) == 0 ? ~0u : 0u; f.w1 = (f.rr >> 2) == 1 ? ~0u : 0u;
  f.w2 = (f.rr >> 2) == 2 ? ~0u : 0u; f.w3 = (f.rr >> 2) == 3 ? ~0u : 0u;
  f.grp = (lane >= 1 && lane < 16) ? ~0u : 0u;
  f.c4 = (lane >= 16 && lane < 29) ? 4u : 0u;
  f.rk = lane == 29 ? (0x00010100u | ((uint32_t)BIGBANG << 24)) : 0u;
  asm volatile("" : "+v"(f.bit), "+v"(f.sh), "+v"(f.w0), "+v"(f.w1), "+v"(f.w2), "+v"(f.w3), "+v"(f.grp), "+v"(f.c4), "+v"(f.rk));
  return f;
}

// the legal list of (hand, info) into the table's slab rows[base ...], ascending canonical id; returns its size
template <bool IDS, class HT>
__device__ __forceinline__ int slab_list(uint64_t hand, uint32_t info, int64_t base, int64_t stride, uint4* rows, int32_t* ids,
                                         const HT& hot, int lane, const FastLanes& fl, int32_t* status) {
  const int lc0 = (int)(info & 0xFF);
  int n;
  if (hand != 0 && lc0 != EMPTY && lc0 <= TRIPLE && !(info & (QF_FROZEN | QF_BADLAST))) {
    // pass + the higher groups of the led size + bombs + rocket (card.py:307-325), one lane per candidate
    const int lv0 = (int)((info >> 8) & 0xFF);
    const int cntr = lane < 15 ? (int)((hand >> (4 * (lane & 15))) & 15) : 0;
    const uint32_t mlc = (uint32_t)__ballot(cntr >= lc0) & (lc0 == SINGLE ? M15 : M13);
    const uint32_t mq = (uint32_t)__ballot(cntr >= 4) & M13;
    const bool rocket = ((uint32_t)__ballot(cntr >= 1) & JOKERS) == JOKERS;
    const uint32_t okm = 1u | ((mlc & gt_mask(lv0)) << 1) | (mq << 16) | (rocket ? 1u << 29 : 0u);
    n = __builtin_popcount(okm);
    const int pre = (int)__builtin_amdgcn_mbcnt_lo(okm, 0u);
    const uint32_t copies = ((uint32_t)lc0 & fl.grp) | fl.c4;  // category == group size here
    const uint32_t dv = copies << fl.sh;
    const uint4 row = make_uint4(dv & fl.w0, dv & fl.w1, dv & fl.w2, (dv & fl.w3) | (copies << 24) | fl.rk);
    if ((okm & fl.bit) != 0) {
      rows[base + pre] = row;
      if (IDS) ids[base + pre] = lane == 0 ? 0 : lane < 16 ? (lc0 == SINGLE ? 1 : lc0 == DOUBLE ? 16 : 29) + fl.rr
                                 : lane < 29 ? 42 + fl.rr : ID_BIGBANG;
    }
  } else {
    // every scan round stores its legal rows straight into the slab (compacted positions are consecutive: coalesced
    // 16-byte stores); nothing is staged in LDS, so there is no flush pass and no LDS round trip behind the scan
    const Out o{rows, ids, base, base + stride, nullptr, nullptr, nullptr};
    Pick pk{-1, 0, 0, 0, 0};
    n = plan_scan<EM_SLAB, IDS>(hand, info, hot, lane, o, pk);
    if (n > stride) {  // cannot happen for a <= 20-card hand (tools/max_legal_bound.c); rows past the slab were dropped
      if (lane == 0 && status) atomicOr(status, 2);
      n = 0;
    }
  }
  return n;
}

// One scan round of a lead from its EM_TEAM_PLAN record: the lanes' candidates exactly as scan_ids / scan_combos build them
// (legal <=> subset of the hand: a lead has no follow gate).  WRITE: the legal rows (+ ids) to rows[base + lanes below];
// returns the number of legal candidates.
// Rounds of a lead: ids 1..54 (1), 3+1 / 3+2 (<= 3 + 3), the chains (<= 3), one kicker block per (start, length) of a triple
// run -- a <= 20-card hand holds <= 6 triples: <= 15 blocks of <= 6 rounds per plane category --, the rocket, <= 5 quads x
// (2 + 2): ~210 at most, TEAM_ROUNDS = 256; a plan with more sets status bit 1 and writes an empty list.
template <bool WRITE, bool IDS, class HT>
__device__ __forceinline__ int team_round(uint2 d, uint64_t hand8, const HT& hot, int lane, int64_t base, int64_t cap, uint4* rows,
                                          int32_t* ids) {
  constexpr uint64_t H8 = 0x8888888888888888ull;
  const uint32_t dx = rfl(d.x), dy = rfl(d.y);
  const int nv = (int)(dx & 63u) + 1, id = (int)((dx >> 6) & 0x3FFFu) + lane, so = (int)(dx >> 20);
  const bool in = lane < nv;
  const uint32_t kind = dy & 3u;
  uint64_t nib;
  int cat;
  bool ok = in;
  if (kind == 0) {
    const uint4 m = hot.meta[so + (in ? lane : 0)];
    nib = (uint64_t)m.x | ((uint64_t)m.y << 32);
    cat = (int)((m.z >> 16) & 0xFF);
  } else if (kind == 1) {
    const int L = (int)((dy >> 2) & 7u), s_ = (int)((dy >> 5) & 15u), gap = (int)((dy >> 9) & 7u);
    const int mult = (int)((dy >> 12) & 3u);
    cat = (int)((dy >> 14) & 31u);
    const uint64_t unit = (cat == FOUR_TAKE_ONE || cat == FOUR_TAKE_TWO) ? 4ull : 3ull;  // main group: a quad, or `gap` triples
    nib = ((unit * ONES) & ((1ull << (4 * gap)) - 1ull)) << (4 * s_);
    if constexpr (HT::HAS_C64) {
      const uint64_t c = hot.c64[so + (in ? lane : 0)], lowm = (1ull << (4 * s_)) - 1ull;
      nib += ((c & lowm) | ((c & ~lowm) << (4 * gap))) << (mult - 1);
    } else {
      const uint32_t e = hot.combo[so + (in ? lane : 0)];
      for (int k = 0; k < L; ++k) {
        const int pos = (int)((e >> (4 * k)) & 15u);
        nib += (uint64_t)mult << (4 * (pos < s_ ? pos : pos + gap));
      }
    }
  } else {  // the joker-kicker extras (DDZ_NATIVE_JOKER_KICKERS builds): lanes 0..12 quad + jokers, 13..23 two triples + jokers
    const uint32_t quads = (dy >> 2) & 0x1FFFu, pairs3 = dy >> 15;
    const bool isq = lane < 13;
    const int r = isq ? lane : (lane - 13) & 15;
    ok = isq ? ((quads >> r) & 1u) : (lane < 24 && ((pairs3 >> r) & 1u));
    nib = (isq ? (4ull << (4 * r)) : (0x33ull << (4 * r))) | (1ull << 52) | (1ull << 56);
    cat = isq ? FOUR_TAKE_ONE : THREE_ONE_LINE;
  }
  const bool legal = ok && ((hand8 - nib) & H8) == H8;
  const uint64_t bl = __ballot(legal);
  if (WRITE && legal) {
    const int pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(bl >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bl, 0u));
    const int64_t pos = base + pre;
    if (pos < cap) {
      rows[pos] = unpack_row(nib, (uint32_t)cat);
      if (IDS) ids[pos] = id;
    }
  }
  return __popcll(bl);
}

// `face` of the tables of a chunk from their rows in LDS: element e = (table, plane, rank), one 16-byte store each,
// consecutive elements -> consecutive addresses.  side[i] = row index per plane kind (bytes 0..6) | the two fractions.
template <int P, uint64_t KINDS>
__device__ __forceinline__ void face_chunk(const uint4* srow, const uint4* side, int ntab, float4* out, int lane) {
  const uint8_t* rb = (const uint8_t*)srow;
  const int ne = ntab * (P * 15);
  for (int e = lane; e < ne; e += 64) {
    const int i = e / (P * 15), r = e - i * (P * 15), pl = r / 15, ri = r - pl * 15;
    const int kind = (int)((KINDS >> (4 * pl)) & 15);
    const uint4 sd = side[i];
    const uint8_t* tb = rb + i * STATE_ROW_BYTES;
    const uint64_t rowsel = (uint64_t)sd.x | ((uint64_t)sd.y << 32);
    const int f = (int)((rowsel >> (8 * (kind < 7 ? kind : 0))) & 0xFF);  // the prob planes read the actor's hand
    const int c_ = tb[f * 16 + ri];
    float4 v;
    if (kind < 7) {
      v = make_float4(c_ > 0 ? 1.f : 0.f, c_ > 1 ? 1.f : 0.f, c_ > 2 ? 1.f : 0.f, c_ > 3 ? 1.f : 0.f);
    } else {
      const int known = c_ + tb[DDZ_F_TAKEN * 16 + ri], total = ri < 13 ? 4 : 1;
      const float fr = __uint_as_float(kind == 7 ? sd.z : sd.w);
      v = make_float4((0 >= known && 0 < total) ? fr : 0.f, (1 >= known && 1 < total) ? fr : 0.f,
                      (2 >= known && 2 < total) ? fr : 0.f, (3 >= known && 3 < total) ? fr : 0.f);
    }
    store_stream(&out[e], v);
  }
}

// `face` of the chunk's tables as their rows in LDS stand (k_observe's expression; envi.py:87-96,165-217).  Per table, next
// to the rows: the row of every plane kind + the two fractions (lane i = table i), then lane-parallel over the elements.
__device__ __attribute__((noinline)) void face_phase(uint4* srow, int ntab, int lane, float4* face, int variant, int64_t t0) {
  uint4* side = srow + 16 * DDZ_NFIELDS;
  if (lane < ntab) {
    const uint8_t* rb = (const uint8_t*)(srow + lane * DDZ_NFIELDS);
    uint32_t frole = rb[DDZ_F_META * 16];
    if (frole > 2) frole = 0;
    const uint32_t fm1 = frole == 0 ? 2 : frole - 1, fp1 = frole == 2 ? 0 : frole + 1;
    const int n1 = rb[(DDZ_F_HAND0 + fp1) * 16 + 15], n2 = rb[(DDZ_F_HAND0 + fm1) * 16 + 15];
    const float f1 = n1 + n2 > 0 ? (float)n1 / (float)(n1 + n2) : 0.f, f2 = n1 + n2 > 0 ? (float)n2 / (float)(n1 + n2) : 0.f;
    side[lane] = make_uint4((DDZ_F_HAND0 + frole) | (uint32_t)DDZ_F_TAKEN << 8 | (DDZ_F_HIST0 + fm1) << 16 | (DDZ_F_HIST0 + frole) << 24,
                            (DDZ_F_HIST0 + fp1) | (DDZ_F_RECENT0 + fm1) << 8 | (DDZ_F_RECENT0 + fp1) << 16,
                            __float_as_uint(f1), __float_as_uint(f2));
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (variant == 0) face_chunk<4, 0x8710ull>(srow, side, ntab, face + t0 * (4 * 15), lane);
  else if (variant == 1) face_chunk<7, 0x8743210ull>(srow, side, ntab, face + t0 * (7 * 15), lane);
  else if (variant == 2) face_chunk<9, 0x876543210ull>(srow, side, ntab, face + t0 * (9 * 15), lane);
  else face_chunk<6, 0x876510ull>(srow, side, ntab, face + t0 * (6 * 15), lane);
}

// the same counts (the category byte of a caller's row is ignored)
__device__ __forceinline__ bool row_eq(uint4 r, uint4 w) {
  return r.x == w.x && r.y == w.y && r.z == w.z && ((r.w ^ w.w) & 0x00FFFFFFu) == 0;
}

// Round-2b layout: the decode / selection / apply of a chunk of up to SLAB_CH tables is ONE lane-parallel pass (lane i
// = table t0 + i) over the chunk's state rows staged in the wave's LDS buffer -- the staging list of the list code,
// idle in this phase -- instead of one wave-wide pass per table; only the deal of a finished game, the wave-parallel
// list search of ROWS / IDS and the lists of the new states remain per-table work.
constexpr int SLAB_CH = 16;  // 16 tables x 11 rows x 16 B = 2,816 B <= the 4,000-byte staging list of a wave
static_assert(SLAB_CH == 16, "four lanes per table of a chunk in the list search / arg-max");

#ifndef DDZ_SLAB_WAVES
#define DDZ_SLAB_WAVES 4  // waves per SIMD the register budget of k_slab allows (5 / 6 spill: measured, not faster)
#endif
template <int MODE, bool IDS, bool COOP>
__global__ __launch_bounds__(TB, DDZ_SLAB_WAVES) void k_slab(SlabArgs a) {
  Stamps<12> stamps;
  __shared__ HotTabT<false, COOP && DDZ_HOT_C64 != 0> hot;  // (the nibble-set table where the block executes planned rounds)
  __shared__ uint4 s_chunk[WPB][SLAB_CH * DDZ_NFIELDS + SLAB_CH];  // per wave: the chunk's state rows + face side records
  const int lane = threadIdx.x & 63;
  const int wv = (int)rfl(threadIdx.x >> 6);
  const int64_t wave = (int64_t)blockIdx.x * WPB + wv;
  // one table per wave (T <= 4096): the lane-parallel phases of the block's 8 tables are run by wave 0 alone (otherwise
  // 16 waves per CU each issue them for ONE useful lane), the 8 waves then write one list each
  constexpr bool coop = COOP;  // (a template parameter: the two forms share no phase after the apply, and the team lists'
                               //  registers would otherwise spill into the many-tables-per-wave form)
  const int64_t tb0 = (int64_t)blockIdx.x * WPB;  // coop: the block's tables
  const int cn = coop ? (tb0 < a.T ? (int)(a.T - tb0 < WPB ? a.T - tb0 : WPB) : 0) : 0;
  const int64_t tw0 = coop ? tb0 : wave * a.tpw;
  const int nw = coop ? (wv == 0 ? cn : 0)
                      : (tw0 < a.T ? (int)(a.T - tw0 < a.tpw ? a.T - tw0 : a.tpw) : 0);  // tables of this wave's lane-parallel phases
  __shared__ uint4 s_share[WPB];  // coop: (hand | episode to deal, combo to beat, live | deal << 1) of the block's tables for the list phase
  __shared__ uint4 s_item[WPB];   // coop: the lists the whole block writes: (hand, rounds, wave of the table)
  __shared__ int s_nheavy;
  __shared__ uint2 s_desc[COOP ? WPB : 1][TEAM_ROUNDS];     // per wave: the round records of its planned list (32 KB)
  __shared__ uint32_t s_tcnt[COOP ? WPB : 1][TEAM_ROUNDS];  // ... and the rounds' sizes, then sizes | bases << 16 (16 KB)
  __shared__ uint64_t s_keys[COOP ? WPB : 1][64];           // deal: the cards' keys (ranking by LDS broadcast reads)
  if (coop && threadIdx.x == 0) s_nheavy = 0;  // (in front of the first barrier)
  constexpr bool BYIDX = MODE == DDZ_STEP_CHOICE || MODE == STEP_Q;  // the move is an index into the current list
  constexpr bool SEARCH = MODE == DDZ_STEP_ROWS || MODE == DDZ_STEP_IDS;
  constexpr bool DRAWS = MODE == DDZ_STEP_RANDOM || MODE == DDZ_STEP_IDS;
  uint4* srow = s_chunk[wv];  // [SLAB_CH][11]: the chunk's state rows (+ SLAB_CH side records of the face phase)
  const FastLanes fl = fast_lanes(lane);
  const uint32_t LEAD = mk_info(EMPTY, 0, 1);
  int s_ply = 0, s_eps = 0, s_lord = 0, s_up = 0;
  int64_t s_rows = 0;
  const bool face_first = !((wv >> 2) & 1);  // waves wv and wv + 4 share a SIMD
  // (A dynamic queue of chunks -- waves drawing their next 4..16 tables by atomic ticket, so that the launch does not wait
  // for the unluckiest wave, whose 16 lists cost 1.76 x the mean -- was built and measured in round 3: 112-245 us instead
  // of 34 us.  Thousands of device-scope atomics on ONE address serialise at ~18 ns each on this multi-XCD part
  // (profiles/r03_notes.md); fixed shares it is.)
  // Block work list (a.lpt; every mode but the fused policy step WITH a face output).  A table whose game just ended costs its wave a deal
  // (~3-5 k cycles) and the lord's 20-card lead list (~8-10 k) where an ordinary table costs 2 k; a wave whose 16 tables held
  // three finished games took 82 k cycles against a mean of 46 k, and the launch lasts as long as its slowest wave
  // (tools/launch_floor_probe.hip: a launch of this shape whose waves all take N cycles lasts N cycles + 2.5 us).  So the
  // waves of a block publish their tables' pending work in LDS -- (hand, combo to beat, table), or (episode, table) for a
  // game to deal -- bucketed by an estimate of its cost, and then take the items one by one, MOST EXPENSIVE FIRST, with an
  // LDS ticket: the block ends with its cheapest lists (a follow of a single / pair / triple: ~0.5 k cycles), whoever owned
  // the tables.  (Round 2 tried the lists alone in table order, this round the lists alone most expensive first: no gain --
  // the deals stayed with the owner wave in front of a block barrier; profiles/r03_notes.md.)
  constexpr int WL_CLASSES = 4;
  __shared__ uint4 s_work[WL_CLASSES][WPB * SLAB_CH];
  __shared__ int s_wcnt[WL_CLASSES + 1];  // [4] = the ticket
  const bool lpt = a.lpt != 0 && !coop && !(MODE == STEP_Q && a.face);  // (with `face` the owner wave's face phase needs the
                                                                         //  dealt rows, and the launch is bound by its 94 MB of
                                                                         //  face stores: measured 48.1 -> 48.8 us with the list)
  const int64_t tblk0 = (int64_t)blockIdx.x * WPB * a.tpw;  // first table of this block
  bool first = true;
  // (with the work list every wave of a block runs the same number of rounds: the rounds contain block barriers)
  for (int c0 = 0; first || (lpt ? c0 < a.tpw : c0 < nw); c0 += SLAB_CH) {  // every wave passes the barrier of the first chunk
    const int64_t t0 = tw0 + c0;
    const int ntab = nw - c0 <= 0 ? 0 : nw - c0 < SLAB_CH ? nw - c0 : SLAB_CH;  // 0 for a wave without tables
    if (lpt) {
      if (!first) __syncthreads();  // the previous round's items are all taken
      if (threadIdx.x <= WL_CLASSES) s_wcnt[threadIdx.x] = 0;
    }
    bool deal_l = false;  // this lane's table ended and is dealt by whoever takes its work item ...
    uint32_t deal_ep = 0;  // ... as episode deal_ep
    const int nrows = ntab * DDZ_NFIELDS;
    const bool valid = lane < ntab;
    const int64_t t = t0 + lane;  // the table of this lane in the lane-parallel phases
    // ---- every independent global load of the chunk is in flight before anything waits
    uint4* sp = (uint4*)(a.state + t0 * STATE_ROW_BYTES);
    uint4 R0 = make_uint4(0, 0, 0, 0), R1 = R0, R2 = R0;
    if (lane < nrows) R0 = sp[lane];
    if (64 + lane < nrows) R1 = sp[64 + lane];
    if (128 + lane < nrows) R2 = sp[128 + lane];
    int cnt_l = 0;       // size of the current list
    int32_t sel_l = -1;  // CHOICE index / IDS action id
    uint4 c = make_uint4(0, 0, 0, 0);  // the selected row
    if (valid) {
      if (MODE == DDZ_STEP_CHOICE || MODE == DDZ_STEP_IDS) sel_l = ((const int32_t*)a.sel)[t];
      if (MODE == DDZ_STEP_ROWS) c = ((const uint4*)a.sel)[t];
      cnt_l = a.counts[t];
      if (cnt_l < 0 || cnt_l > a.stride) cnt_l = 0;
    }
    // (one table per wave: wave 0 goes straight to the block's tables, the other fifteen fill the tables)
    if (coop) { if (wv > 0) hot_fill<TB, decltype(hot), 64>(hot); }
    else if (first) hot_fill<TB>(hot);
    uint32_t o_done = 0, o_illegal = 0, o_reward = 0;
    bool live = false;         // is there a list to write afterwards
    uint64_t qhand = 0;        // ... and for which (hand, combo to beat)
    uint32_t qinfo = 0;
    if (ntab > 0) {  // (wave-uniform)
    if (MODE == STEP_Q) {
      // the (epsilon-)greedy arg-max of DQNFirst (dqn.py:50-71) over each table's list, as k_select: 4 lanes per table
      // (lane 4i + s reads entries s, s + 4, ...), a 2-step butterfly keeps (larger value, smaller index)
      static_assert(SLAB_CH * 4 == 64, "4 lanes per table of the chunk");
      const int g = lane >> 2;
      int Ag = 0;
      if (g < ntab) {
        Ag = a.counts[t0 + g];
        if (Ag < 0 || Ag > a.stride) Ag = 0;
      }
      const float* qrow = (const float*)a.sel + (t0 + g) * a.stride;
      int best = 0x7FFFFFFF;
      float bq = 0.f;
      for (int j0 = lane & 3; j0 < Ag; j0 += 16) {  // four loads in flight per trip
        const float v0 = qrow[j0];
        const float v1 = j0 + 4 < Ag ? qrow[j0 + 4] : 0.f;
        const float v2 = j0 + 8 < Ag ? qrow[j0 + 8] : 0.f;
        const float v3 = j0 + 12 < Ag ? qrow[j0 + 12] : 0.f;
        if (j0 == 0 || (v0 == v0 && (best == 0x7FFFFFFF || v0 > bq))) { bq = v0; best = j0; }
        if (j0 + 4 < Ag && v1 == v1 && (best == 0x7FFFFFFF || v1 > bq)) { bq = v1; best = j0 + 4; }
        if (j0 + 8 < Ag && v2 == v2 && (best == 0x7FFFFFFF || v2 > bq)) { bq = v2; best = j0 + 8; }
        if (j0 + 12 < Ag && v3 == v3 && (best == 0x7FFFFFFF || v3 > bq)) { bq = v3; best = j0 + 12; }
      }
#pragma unroll
      for (int dd = 1; dd < 4; dd <<= 1) {
        const float oq = __shfl_xor(bq, dd);
        const int ob = __shfl_xor(best, dd);
        const bool have = best != 0x7FFFFFFF, ohave = ob != 0x7FFFFFFF;
        if (ohave && (!have || (ob < best ? !(bq > oq) : (oq > bq)))) { bq = oq; best = ob; }
      }
      sel_l = __shfl(best, (lane & (SLAB_CH - 1)) * 4);  // lane i < 16: the index of table t0 + i
      if (cnt_l <= 0) sel_l = -1;
      if (a.thr && valid && cnt_l > 0) {
        const uint4 meta = *(const uint4*)(a.state + t * STATE_ROW_BYTES + DDZ_F_META * 16);
        const uint64_t g_ = a.gid_base + (uint64_t)t;
        const uint4 dr = philox4x32_10(make_uint4((uint32_t)g_, (uint32_t)(g_ >> 32), meta.z, (3u << 16) | (meta.y & 0xFFFF)), a.k0, a.k1);
        if ((uint64_t)dr.x < a.thr) sel_l = (int)__umulhi(dr.y, (uint32_t)cnt_l);
      }
      if (a.choice_out && valid) a.choice_out[t] = sel_l;
    }
    int idx = -1;  // list index of the move, -1 = none / not in the list
    if (BYIDX) {
      idx = (sel_l >= 0 && sel_l < cnt_l) ? sel_l : -1;
      if (idx >= 0) c = a.rows[t * a.stride + idx];
    }
    // ---- the chunk's rows into LDS
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < nrows) srow[lane] = R0;
    if (64 + lane < nrows) srow[64 + lane] = R1;
    if (128 + lane < nrows) srow[128 + lane] = R2;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    stamps.mark(0);
    // ---- decode, one table per lane
    uint4* tr = srow + (valid ? lane : 0) * DDZ_NFIELDS;
    const uint4 M = tr[DDZ_F_META];
    int role = M.x & 0xFF;
    if (role > 2) role = 0;  // never index outside the table on a corrupted import
    const bool is_done = (M.x >> 8) & 0xFF;
    const bool dealt = (M.y >> 16) & 0xFF;
    const uint32_t ply = M.y & 0xFFFF, episode = M.z;
    const uint64_t gid = a.gid_base + (uint64_t)t;
    const bool active = valid && dealt && !is_done;
    const int rm1 = role == 0 ? 2 : role - 1, rp1 = role == 2 ? 0 : role + 1;
    const uint4 Hr = tr[DDZ_F_HAND0 + role];
    // the combo the current actor has to beat (envi.py:103-109): previous player's handout, else the one before
    const uint4 Ra = tr[DDZ_F_RECENT0 + rm1], Rb = tr[DDZ_F_RECENT0 + rp1];
    const uint64_t Pa = pack_row(Ra), Pb = pack_row(Rb);
    const bool useA = Pa != 0;
    const uint64_t Pl = useA ? Pa : Pb;
    const uint32_t cur_info = Pl ? info_of_row(Pl, (int)(((useA ? Ra.w : Rb.w) >> 24) & 15)) : LEAD;
    const int A = cnt_l;
    const bool frozen = !active || A <= 0;
    // ---- selection: every mode ends as (idx, c) per lane
    if (!BYIDX) {
      bool draw = MODE == DDZ_STEP_RANDOM;
      if (MODE == DDZ_STEP_IDS) draw = sel_l == -1;
      if (DRAWS && draw && !frozen) {  // random.choice(actions), envi.py:83
        const uint4 d = philox4x32_10(make_uint4((uint32_t)gid, (uint32_t)(gid >> 32), episode, (2u << 16) | ply), a.k0, a.k1);
        idx = (int)__umulhi(d.x, (uint32_t)A);
      }
      if (SEARCH) {  // wave-parallel search of a table's list for the wanted counts, table after table
        constexpr int NIDS = DDZ_NUM_ACTIONS + 24 * DDZ_NATIVE_JOKER_KICKERS;
        bool need = !frozen && !draw;
        if (MODE == DDZ_STEP_IDS) {
          if (sel_l < 0 || sel_l >= NIDS) need = false;  // no such action: not in any list
          c = row_of_id(need ? sel_l : 0);
        }
        // every table's list is searched by 4 lanes (lane 4g + s looks at rows s, s + 4, ..., two loads in flight per
        // trip): 16 tables at once instead of one table per wave-wide pass (the arg-max of STEP_Q is laid out the same way)
        {
          static_assert(SLAB_CH * 4 == 64, "4 lanes per table of the chunk");
          const int g = lane >> 2;
          const uint4 want = make_uint4(__shfl(c.x, g), __shfl(c.y, g), __shfl(c.z, g), __shfl(c.w, g));
          // (every shuffle with all lanes active: inside a `needg ? __shfl(A, g) : 0` the source lanes of a table that does
          // not search are switched off and the bpermute returns garbage for the groups that read them)
          const int needg = __shfl(need ? 1 : 0, g);
          const int Aof = __shfl(A, g);
          const int Ag = needg ? Aof : 0;
          const uint4* lrow = a.rows + (t0 + g) * a.stride;
          int found = 0x7FFFFFFF;
          for (int j0 = lane & 3; j0 < Ag && found == 0x7FFFFFFF; j0 += 8) {
            const uint4 r0 = lrow[j0];
            const uint4 r1 = j0 + 4 < Ag ? lrow[j0 + 4] : make_uint4(~0u, 0, 0, 0);
            if (j0 + 4 < Ag && row_eq(r1, want)) found = j0 + 4;
            if (row_eq(r0, want)) found = j0;
          }
#pragma unroll
          for (int dd = 1; dd < 4; dd <<= 1) {
            const int o_ = __shfl_xor(found, dd);
            found = o_ < found ? o_ : found;
          }
          const int mine = __shfl(found, (lane & (SLAB_CH - 1)) * 4);
          if (need) idx = mine == 0x7FFFFFFF ? -1 : mine;
        }
      }
      if (idx >= 0) c = a.rows[t * a.stride + idx];  // the listed row carries the category byte
    }
    uint32_t cinfo = 0;  // category | value << 8 | len << 16 | number of cards << 24 of the move
    if (idx >= 0) {
      const uint64_t pn = pack_row(c);
      cinfo = info_of_row(pn, (int)(c.w >> 24)) | ((uint32_t)nib_sum(pn) << 24);
    }
    stamps.mark(2);
    // ---- apply (envi.py:38-43 _update + native step), outputs, trajectory record
    uint4 tr0 = make_uint4(0, 0, 0, 0);
    uint4 tr1 = make_uint4((uint32_t)role, ((uint32_t)A & 0xFFFF) | (ply << 16), episode, 0xFFFFFFFFu);
    o_done = is_done;
    live = active;
    qinfo = cur_info;
    bool changed = false, won = false;
    if (frozen) {
      tr1.x |= (uint32_t)is_done << 8 | 2u << 24;
      if (active) qhand = pack_row(Hr);  // a live table whose list was reported empty: write it afresh
    } else if (idx < 0) {  // not in the list: table untouched, flagged; its list is written again as it was
      o_done = 0; o_illegal = 1;
      tr1.x |= 1u << 24;
      qhand = pack_row(Hr);
    } else {
      changed = true;
      const uint32_t ncards = cinfo >> 24;
      const uint32_t cw3 = c.w & 0x00FFFFFFu;
      won = (Hr.w >> 24) == ncards;
      o_reward = won ? (role == 1 ? 0xFFu : 1u) : 0u;  // rule_play.py:14: -1 lord won, +1 farmers
      o_done = won;
      tr0 = c;
      tr1.x |= (uint32_t)won << 8 | o_reward << 16;
      tr1.w = (uint32_t)idx;
      if (!(won && a.auto_reset)) {
        // byte-wise: every byte of the hand >= the row's byte, so no borrow/carry crosses a byte
        tr[DDZ_F_HAND0 + role] = make_uint4(Hr.x - c.x, Hr.y - c.y, Hr.z - c.z, Hr.w - (cw3 + (ncards << 24)));
        const uint4 Hi = tr[DDZ_F_HIST0 + role], Tk = tr[DDZ_F_TAKEN];
        tr[DDZ_F_HIST0 + role] = make_uint4(Hi.x + c.x, Hi.y + c.y, Hi.z + c.z, Hi.w + cw3);
        tr[DDZ_F_TAKEN] = make_uint4(Tk.x + c.x, Tk.y + c.y, Tk.z + c.z, Tk.w + cw3);
        tr[DDZ_F_RECENT0 + role] = c;
        tr[DDZ_F_META] = make_uint4((uint32_t)rp1 | (won ? 1u << 8 : 0u) | ((won ? (uint32_t)role : 0xFFu) << 16) | (o_reward << 24),
                                    (M.y & 0xFFFF0000u) | ((ply + 1) & 0xFFFF), M.z, M.w);
        live = !won;
        // the next actor (lord -> down -> up, game.py:173-181) has to beat this ply's combo, or -- after a pass -- the
        // previous player's: its hand is untouched by this ply
        qhand = pack_row(tr[DDZ_F_HAND0 + rp1]);
        qinfo = ncards ? (cinfo & 0x00FFFFFFu) : (useA ? cur_info : LEAD);
      }
    }
    {
      const uint64_t chm = __ballot(changed), wm = __ballot(changed && won);
      s_ply += __builtin_popcountll(chm);
      s_eps += __builtin_popcountll(wm);
      s_lord += __builtin_popcountll(wm & __ballot(role == 1));
      s_up += __builtin_popcountll(wm & __ballot(role == 0));
      if (a.traj && valid) {
        a.traj[2 * t] = tr0;
        a.traj[2 * t + 1] = tr1;
      }
      // finished games with auto-reset: the next episode's deal (wave-wide), the lord leads
      uint64_t wr = a.auto_reset ? wm : 0ull;
      // (work list: dealt in the list phase, by any wave of the block; coop: by the table's own wave, the block's deals side
      //  by side instead of one after the other in wave 0)
      const bool defer = lpt || (coop && !(MODE == STEP_Q && a.face));  // (with `face` wave 0's face phase needs the dealt rows)
      const uint64_t deferred = defer ? wr : 0ull;
      deal_l = (deferred >> lane) & 1ull;
      deal_ep = episode + 1u;
      if (defer) wr = 0ull;
      while (wr) {
        const int i = __builtin_ctzll(wr);
        wr &= wr - 1;
        const uint32_t ep = rl(episode, i) + 1u;
        uint64_t h0, h1, h2;
        deal_wave(a.gid_base + (uint64_t)(t0 + i), ep, a.k0, a.k1, lane, h0, h1, h2);
        if (lane < DDZ_NFIELDS)
          srow[i * DDZ_NFIELDS + lane] = lane == 0 ? unpack_row(h0, 17) : lane == 1 ? unpack_row(h1, 20) : lane == 2 ? unpack_row(h2, 17)
                                         : lane == DDZ_F_META ? make_uint4(1u | (0xFFu << 16), 1u << 16, ep, 0) : make_uint4(0, 0, 0, 0);
        if (lane == i) { qhand = h1; qinfo = LEAD; }
      }
      // the rows of the tables that moved, back to the state: coalesced 16-byte stores
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint64_t stm = chm & ~deferred;  // (a deferred deal writes its table's rows itself)
      if (lane < nrows && ((stm >> (lane / DDZ_NFIELDS)) & 1)) sp[lane] = srow[lane];
      if (64 + lane < nrows && ((stm >> ((64 + lane) / DDZ_NFIELDS)) & 1)) sp[64 + lane] = srow[64 + lane];
      if (128 + lane < nrows && ((stm >> ((128 + lane) / DDZ_NFIELDS)) & 1)) sp[128 + lane] = srow[128 + lane];
    }
    if (coop && valid)
      s_share[lane] = deal_l ? make_uint4(deal_ep, 0u, LEAD, 3u)
                             : make_uint4((uint32_t)qhand, (uint32_t)(qhand >> 32), qinfo, live ? 1u : 0u);
    }  // ntab > 0
    stamps.mark(3);
    if (first || lpt) __syncthreads();  // the hot records are in LDS (and the work list's counters are zero)
    if (lpt) {
      // publish: lane i = table t0 + i.  Cost classes: 0 a game to deal (+ the lord's 20-card lead) or a lead from a hand
      // with two triples / a bomb (planes, four-with-two: many kicker rounds), 1 any other lead, 2 a follow of a chain /
      // plane / ..., 3 a follow of a single / pair / triple
      if (live) {
        const int lc = (int)(qinfo & 0xFF);
        const bool lead = lc == EMPTY;
        const bool rich = __builtin_popcount(ge_mask(qhand, 3)) >= 2 || ge_mask(qhand, 4) != 0;
        const int cls = deal_l ? 0 : lead ? (rich ? 0 : 1) : lc <= TRIPLE ? 3 : 2;
        const int pos = atomicAdd(&s_wcnt[cls], 1);
        s_work[cls][pos] = deal_l ? make_uint4(deal_ep, 0u, LEAD, (uint32_t)(t - tblk0) | 0x80000000u)
                                  : make_uint4((uint32_t)qhand, (uint32_t)(qhand >> 32), qinfo, (uint32_t)(t - tblk0));
      }
      __syncthreads();
    }
    if (MODE == STEP_Q && a.face && face_first && ntab > 0) {
      face_phase(srow, ntab, lane, a.face, a.face_variant, t0);
      stamps.mark(6);
    }
    // ---- the lists of the (new) states, straight into the tables' slabs (nothing is staged: the rows in LDS stay valid)
    int n_l = 0;
    if (coop) {
      stamps.mark(1);
      bool heavy = false;  // this wave's list is written by the block below
      if (wv < cn) {  // this wave's table of the block
        const uint4 sh = s_share[wv];
        const uint32_t fw = rfl(sh.w);
        const int64_t tt = tb0 + wv;
        int n = 0;
        if (fw & 1u) {
          uint64_t hand = (uint64_t)rfl(sh.x) | ((uint64_t)rfl(sh.y) << 32);
          const uint32_t qi = rfl(sh.z);
          if (fw & 2u) {  // a finished game: the next episode's deal (native prepare(), spec v2), the lord leads
            const uint32_t ep = rfl(sh.x);
            uint64_t h0, h1, h2;
            deal_wave_lds(a.gid_base + (uint64_t)tt, ep, a.k0, a.k1, lane, s_keys[wv], h0, h1, h2);
            if (lane < DDZ_NFIELDS)
              ((uint4*)(a.state + tt * STATE_ROW_BYTES))[lane] =
                  lane == 0 ? unpack_row(h0, 17) : lane == 1 ? unpack_row(h1, 20) : lane == 2 ? unpack_row(h2, 17)
                  : lane == DDZ_F_META ? make_uint4(1u | (0xFFu << 16), 1u << 16, ep, 0) : make_uint4(0, 0, 0, 0);
            hand = h1;
            stamps.mark(7);
          }
          // A lead from a hand with two triples or a bomb (kicker blocks: many scan rounds) is planned into round records;
          // from TEAM_MIN rounds on the block executes them together -- a launch of <= 4096 tables lasts as long as its
          // slowest list -- below that the wave executes its records itself.
          // (measured, tools/slab_coop_probe.py: planning every lead with a triple, or every lead, changes nothing -- 10.1 us
          //  per launch at 4096 tables either way; thresholds 3 .. 8 are equal, 16 costs 1.6 us, no block lists 1.2 us)
          const int thr = a.team;
          if (thr && hand != 0 && (qi & 0xFF) == EMPTY && !(qi & (QF_FROZEN | QF_BADLAST)) &&
              (__builtin_popcount(ge_mask(hand, 3)) >= 2 || ge_mask(hand, 4) != 0)) {
            const Out o{a.rows, a.ids, tt * a.stride, (tt + 1) * a.stride, nullptr, nullptr, nullptr};
            stamps.mark(4);
            Pick pk{-1, 0, 0, 0, 0};
            pk.desc = s_desc[wv];
            plan_scan_t<EM_TEAM_PLAN, IDS, true>(hand, follow_of(qi), hot, lane, o, pk);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int R = pk.rc;
            stamps.mark(8);
            if (R > TEAM_ROUNDS) {  // cannot happen (see team_round): flagged, an empty list
              if (lane == 0 && a.status) atomicOr(a.status, 2);
            } else if (R >= thr) {
              heavy = true;
              if (lane == 0) s_item[atomicAdd(&s_nheavy, 1)] = make_uint4((uint32_t)hand, (uint32_t)(hand >> 32), (uint32_t)R, (uint32_t)wv);
            } else {
              const uint64_t hand8 = hand | 0x8888888888888888ull;
              for (int r = 0; r < R; ++r)
                n += team_round<true, IDS>(s_desc[wv][r], hand8, hot, lane, o.base + n, o.cap, a.rows, a.ids);
              if (n > a.stride) {
                if (lane == 0 && a.status) atomicOr(a.status, 2);
                n = 0;
              }
              stamps.mark(9);
            }
          } else {
            n = slab_list<IDS>(hand, qi, tt * a.stride, a.stride, a.rows, a.ids, hot, lane, fl, a.status);
          }
        }
        if (lane == 0 && !heavy) a.counts[tt] = n;
        s_rows += n;
      }
      stamps.mark(4);
      __syncthreads();
      const int nh = s_nheavy;
      stamps.mark(6);
      if (nh > 0) {  // (block-uniform)
        // pass 1: the rounds of every heavy list dealt out over the sixteen waves, their sizes into s_tcnt
        for (int h = 0; h < nh; ++h) {
          const uint4 e = s_item[h];
          const uint64_t hand8 = ((uint64_t)rfl(e.x) | ((uint64_t)rfl(e.y) << 32)) | 0x8888888888888888ull;
          const int R = (int)rfl(e.z), ow = (int)rfl(e.w);
          for (int r = (wv + 5 * h) & (WPB - 1); r < R; r += WPB) {
            const int k = team_round<false, IDS>(s_desc[ow][r], hand8, hot, lane, 0, 0, a.rows, a.ids);
            if (lane == 0) s_tcnt[ow][r] = (uint32_t)k;
          }
        }
        stamps.mark(10);
        __syncthreads();
        // exclusive scan of a list's round sizes, in place (wave h: list h); its size to counts[]
        if (wv < nh) {
          const uint4 e = s_item[wv];
          const int R = (int)rfl(e.z), ow = (int)rfl(e.w);
          int run = 0;
          for (int r0 = 0; r0 < R; r0 += 64) {
            const int i = r0 + lane;
            const int v = i < R ? (int)s_tcnt[ow][i] : 0;
            const int inc = wave_scan_add(v);
            if (i < R) s_tcnt[ow][i] = (uint32_t)v | (uint32_t)(run + inc - v) << 16;  // size | base << 16
            run += (int)rl((uint32_t)inc, 63);
          }
          if (run > a.stride) {  // cannot happen for a <= 20-card hand (tools/max_legal_bound.c)
            if (lane == 0 && a.status) atomicOr(a.status, 2);
            run = 0;
          }
          if (lane == 0) a.counts[tb0 + ow] = run;
          s_rows += run;
        }
        __syncthreads();
        // pass 2: the same rounds again, rows written at their bases
        for (int h = 0; h < nh; ++h) {
          const uint4 e = s_item[h];
          const uint64_t hand8 = ((uint64_t)rfl(e.x) | ((uint64_t)rfl(e.y) << 32)) | 0x8888888888888888ull;
          const int R = (int)rfl(e.z), ow = (int)rfl(e.w);
          const int64_t base = (tb0 + ow) * a.stride;
          for (int r = (wv + 5 * h) & (WPB - 1); r < R; r += WPB) {
            const uint32_t cb = rfl(s_tcnt[ow][r]);
            if (cb & 0xFFFFu)  // (a round without a legal id -- most rounds of a kicker block -- is not repeated)
              team_round<true, IDS>(s_desc[ow][r], hand8, hot, lane, base + (int)(cb >> 16), base + a.stride, a.rows, a.ids);
          }
        }
      }
      stamps.mark(11);
    }
    if (lpt) {
      const int n0 = s_wcnt[0], n1 = n0 + s_wcnt[1], n2 = n1 + s_wcnt[2], n3 = n2 + s_wcnt[3];
      for (;;) {
        int k = 0;
        if (lane == 0) k = atomicAdd(&s_wcnt[WL_CLASSES], 1);
        k = (int)rfl((uint32_t)k);
        if (k >= n3) break;
        const int cls = k < n0 ? 0 : k < n1 ? 1 : k < n2 ? 2 : 3;
        const uint4 e = s_work[cls][k - (cls == 0 ? 0 : cls == 1 ? n0 : cls == 2 ? n1 : n2)];
        const uint32_t ew = rfl(e.w);
        const int64_t tt = tblk0 + (int64_t)(ew & 0x7FFFFFFFu);
        uint64_t hand = (uint64_t)rfl(e.x) | ((uint64_t)rfl(e.y) << 32);
        if (ew >> 31) {  // a finished game: the next episode's deal (native prepare(), spec v2), the lord leads
          const uint32_t ep = rfl(e.x);
          uint64_t h0, h1, h2;
#ifdef DDZ_SLAB_DEAL_READLANE
          deal_wave(a.gid_base + (uint64_t)tt, ep, a.k0, a.k1, lane, h0, h1, h2);
#else
          // (ranking through LDS: the wave's chunk rows are dead here -- stored back before the items were published)
          deal_wave_lds(a.gid_base + (uint64_t)tt, ep, a.k0, a.k1, lane, (uint64_t*)srow, h0, h1, h2);
#endif
          if (lane < DDZ_NFIELDS)
            ((uint4*)(a.state + tt * STATE_ROW_BYTES))[lane] =
                lane == 0 ? unpack_row(h0, 17) : lane == 1 ? unpack_row(h1, 20) : lane == 2 ? unpack_row(h2, 17)
                : lane == DDZ_F_META ? make_uint4(1u | (0xFFu << 16), 1u << 16, ep, 0) : make_uint4(0, 0, 0, 0);
          hand = h1;
        }
        const int n = slab_list<IDS>(hand, rfl(e.z), tt * a.stride, a.stride, a.rows, a.ids, hot, lane, fl, a.status);
        if (lane == 0) a.counts[tt] = n;
        s_rows += n;
      }
    }
    uint64_t lv = (coop || lpt) ? 0ull : __ballot(live);
    while (lv) {
      const int i = __builtin_ctzll(lv);
      lv &= lv - 1;
      const int n = slab_list<IDS>(rl64(qhand, i), rl(qinfo, i), (t0 + i) * a.stride, a.stride, a.rows, a.ids, hot, lane, fl,
                                   a.status);
      if (lane == i) n_l = n;
      s_rows += n;
    }
    if (MODE == STEP_Q && a.face && !face_first && ntab > 0) {
      // this half of the waves writes `face` after its lists (the rows of the new states are still in LDS), so that at any
      // time some waves of a SIMD are in the HBM-bound phase and the others in the issue-bound one
      face_phase(srow, ntab, lane, a.face, a.face_variant, t0);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      stamps.mark(6);
    }
    if (valid) {  // the per-table outputs: consecutive addresses, one store each
      if (lpt) { if (!live) a.counts[t] = 0; }  // (a live table's size is written by the wave that took its item)
      else if (!coop) a.counts[t] = n_l;
      if (a.done) a.done[t] = (uint8_t)o_done;
      if (a.reward) a.reward[t] = (int8_t)o_reward;
      if (a.illegal) a.illegal[t] = (uint8_t)o_illegal;
    }
    stamps.mark(4);
    first = false;
  }
  stamps.set(5, (unsigned long long)(coop ? 1 : nw));
  stamps.store(coop ? tb0 + wv : tw0, lane == 0 && (coop ? wv < cn : nw > 0));
  if ((coop ? wv < cn : (nw > 0 || lpt)) && lane == 0) {  // each wave owns its statistics slot (as in k_rollout)
    int64_t* ws = a.wave_stats + 4 * wave;
    ws[0] += s_ply; ws[1] += s_eps; ws[2] += (int64_t)s_lord | ((int64_t)s_up << 32); ws[3] += s_rows;
  }
}

// stateless r.get_moves(hand15, last15) in the slab layout: query i owns rows[i * stride ...], ONE launch, no size pass
// and no scan over the queries (k_moves needs both for its packed CSR output); the list code is k_slab's.
template <bool IDS>
__global__ __launch_bounds__(TB, 4) void k_moves_slab(const uint4* __restrict__ hands, const uint4* __restrict__ lasts,
                                                      int64_t n, int tpw, int32_t* __restrict__ counts,
                                                      uint4* __restrict__ rows, int32_t* __restrict__ ids, int64_t stride,
                                                      int32_t* __restrict__ status) {
  __shared__ HotTabL hot;
  const int lane = threadIdx.x & 63;
  const int wv = (int)rfl(threadIdx.x >> 6);
  const int64_t t0 = ((int64_t)blockIdx.x * WPB + wv) * tpw;
  const int ntab = t0 < n ? (int)(n - t0 < tpw ? n - t0 : tpw) : 0;
  uint4 hr = make_uint4(0, 0, 0, 0), lr = hr;  // lane i: the rows of query t0 + i
  if (lane < ntab) { hr = hands[t0 + lane]; lr = lasts[t0 + lane]; }
  hot_fill<TB>(hot);
  __syncthreads();
  // pack / classify lane-parallel, then one readlane per value and query
  const uint64_t hn = pack_row(hr);
  uint32_t li = classify(pack_row(lr));
  if (li == INFO_INVALID || ge_mask(hn, 5) || (hn >> 60)) li = QF_BADLAST | QF_FROZEN;
  const FastLanes fl = fast_lanes(lane);
  int n_l = 0;
  bool bad = false;
  for (int i = 0; i < ntab; ++i) {
    const uint64_t hand = rl64(hn, i);
    const uint32_t info = rl(li, i);
    bad = bad || (info & QF_BADLAST);
    const int m = slab_list<IDS>(hand, info, (t0 + i) * stride, stride, rows, ids, hot, lane, fl, status);
    if (lane == i) n_l = m;
  }
  if (lane < ntab) counts[t0 + lane] = n_l;
  if (bad && lane == 0 && status) atomicOr(status, 4);
}

// slab -> CSR: the fixed-stride lists packed into offsets[n+1] / rows[sum A] / ids[sum A] (what a ragged NN forward
// consumes) -- the "prefix-sum compaction of the variable-length legal-action list" as its own cheap pass, so that the
// stepping kernels never wait on a scan over all tables.  Two launches: sizes -> block scans, then offsets + row copy.
constexpr int CSR_BT = 256;  // tables per block
// blockIdx.y = one of several slabs compacted by the same launch (ddz_rollout_random_csr_staged: the iterations of a batch;
// CsrBatch = how far the buffers advance per slab; all zero for a single slab)
struct CsrBatch { int64_t counts, rows, scan, blk; int first; };
__global__ __launch_bounds__(CSR_BT) void k_csr_scan(const int32_t* __restrict__ counts, int64_t n, int64_t stride,
                                                     int32_t* __restrict__ local_off, int32_t* __restrict__ blk_tot, CsrBatch bt) {
  {
    const int64_t y = bt.first + (int64_t)blockIdx.y;
    counts += y * bt.counts; local_off += y * bt.scan; blk_tot += y * bt.blk;
  }
  __shared__ int sh[CSR_BT / 64];
  const int64_t t = (int64_t)blockIdx.x * CSR_BT + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int c = t < n ? counts[t] : 0;
  if (c < 0 || c > stride) c = 0;
  const int incl = wave_incl_scan(c, lane);
  if (lane == 63) sh[wv] = incl;
  __syncthreads();
  int wbase = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < CSR_BT / 64; ++w) {
    if (w < wv) wbase += sh[w];
    tot += sh[w];
  }
  if (t < n) local_off[t] = wbase + incl - c;
  if (threadIdx.x == 0) blk_tot[blockIdx.x] = tot;
}

template <bool IDS>
__global__ __launch_bounds__(CSR_BT) void k_csr_copy(const int32_t* __restrict__ counts, const uint4* __restrict__ rows,
                                                     const int32_t* __restrict__ ids, int64_t n, int64_t stride,
                                                     const int32_t* __restrict__ local_off, const int32_t* __restrict__ blk_tot,
                                                     int32_t* __restrict__ offsets, uint4* __restrict__ rows_out,
                                                     int32_t* __restrict__ ids_out, int64_t cap, int32_t* __restrict__ status,
                                                     CsrBatch bt) {
  {
    const int64_t y = bt.first + (int64_t)blockIdx.y;
    counts += y * bt.counts; rows += y * bt.rows; local_off += y * bt.scan; blk_tot += y * bt.blk;
    if (IDS) ids += y * bt.rows;
  }
  __shared__ long long sh[CSR_BT / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  long long part = 0;
  for (int j = threadIdx.x; j < (int)blockIdx.x; j += CSR_BT) part += blk_tot[j];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
  if (lane == 0) sh[wv] = part;
  __syncthreads();
  long long base = 0;
#pragma unroll
  for (int w = 0; w < CSR_BT / 64; ++w) base += sh[w];
  const int64_t t = (int64_t)blockIdx.x * CSR_BT + threadIdx.x;
  int c = t < n ? counts[t] : 0;
  if (c < 0 || c > stride) c = 0;
  const long long off = base + (t < n ? local_off[t] : 0);
  // offsets are clamped to the capacity: a consumer that indexes rows_out[offsets[t] .. offsets[t+1]) never leaves the
  // buffer (a truncated list is shorter / empty; status bit 1 says that it happened)
  if (t < n) offsets[t] = (int32_t)(off < cap ? off : cap);
  if (t == n - 1) {
    offsets[n] = (int32_t)(off + c < cap ? off + c : cap);
    if (off + c > cap && status) atomicOr(status, 2);
  }
  // every wave copies the lists of its 64 tables flat: output row r of the wave belongs to the last table whose
  // exclusive offset is <= r (binary search over 64 LDS words), so the loads of one trip are independent
  __shared__ int s_excl[CSR_BT / 64][64];
  const long long woff = (long long)rl64((uint64_t)off, 0);
  const int64_t t0 = (int64_t)blockIdx.x * CSR_BT + wv * 64;
  const int nv = n - t0 >= 64 ? 64 : (n > t0 ? (int)(n - t0) : 0);  // tables of this wave that exist
  const int excl = lane < nv ? (int)(off - woff) : 0x7FFFFFFF;
  s_excl[wv][lane] = excl;
  const int W = nv ? (int)rl((uint32_t)(excl + c), nv - 1) : 0;
  __builtin_amdgcn_wave_barrier();
  for (int r = lane; r < W; r += 64) {
    int i = 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
      if (s_excl[wv][i + d] <= r) i += d;
    const int j = r - s_excl[wv][i];
    if (woff + r < cap) {
      rows_out[woff + r] = rows[(t0 + i) * stride + j];
      if (IDS) ids_out[woff + r] = ids[(t0 + i) * stride + j];
    }
  }
}

// ------------------------------------------------------------------------------------
// k_mask: the legal moves of every table as a dense 0/1 mask over the action space -- the form the reference's
// rules produce (get_mask, rule_based/utils/utils.py:45-63; mask[0] = pass) and what a policy head with one logit
// per action consumes.  Bit-packed: MASK_WORDS u32 per table, bit (id & 31) of word id >> 5.  One wavefront per
// table: bits are set in an LDS mask during the scan, then streamed out (write-once: nontemporal stores).
constexpr int MASK_WORDS = (DDZ_NUM_ACTIONS + 24 + 31) / 32;  // 424: also covers the joker-kicker build

__global__ __launch_bounds__(TB, 4) void k_mask(const uint8_t* __restrict__ state, int64_t T, int tpw,
                                                uint32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wv = (int)rfl(threadIdx.x >> 6);
  const int64_t t0 = ((int64_t)blockIdx.x * WPB + wv) * tpw;
  const int ntab = t0 < T ? (int)(T - t0 < tpw ? T - t0 : tpw) : 0;
  __shared__ HotTabT<false> hot;
  static_assert(MASK_WORDS % 4 == 0, "the mask is moved in 16-byte pieces");
  __shared__ uint4 s_mask[WPB][MASK_WORDS / 4];
  uint4 Rnext = make_uint4(0, 0, 0, 0);
  if (ntab > 0 && lane < DDZ_NFIELDS) Rnext = ((const uint4*)(state + t0 * STATE_ROW_BYTES))[lane];
  hot_fill<TB>(hot);
  __syncthreads();
  uint4* mask4 = s_mask[wv];
  uint32_t* mask = (uint32_t*)mask4;
  for (int i = 0; i < ntab; ++i) {
    const int64_t t = t0 + i;
    const uint4 R = Rnext;
    if (i + 1 < ntab && lane < DDZ_NFIELDS) Rnext = ((const uint4*)(state + (t + 1) * STATE_ROW_BYTES))[lane];
    for (int w = lane; w < MASK_WORDS / 4; w += 64) mask4[w] = make_uint4(0, 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
    const uint64_t P = pack_row(R);
    const uint32_t mx = rl(R.x, DDZ_F_META), my = rl(R.y, DDZ_F_META);
    int role = mx & 0xFF;
    if (role > 2) role = 0;
    const bool active = ((my >> 16) & 0xFF) && !((mx >> 8) & 0xFF);  // dealt and not done
    const int rm1 = role == 0 ? 2 : role - 1, rp1 = role == 2 ? 0 : role + 1;
    const uint64_t hand = rl64(P, DDZ_F_HAND0 + role);
    const uint32_t info = last_info(rl64(P, DDZ_F_RECENT0 + rm1), (int)(rl(R.w, DDZ_F_RECENT0 + rm1) >> 24),
                                    rl64(P, DDZ_F_RECENT0 + rp1), (int)(rl(R.w, DDZ_F_RECENT0 + rp1) >> 24));
    const Out o{nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, mask};
    Pick pk{-1, 0, 0, 0, 0};
    plan_scan<EM_MASK, false>(hand, active ? info : QF_FROZEN, hot, lane, o, pk);
    __builtin_amdgcn_wave_barrier();
    uint4* dst = (uint4*)(out + t * MASK_WORDS);  // 1696 B per table: 16-byte aligned
    for (int w = lane; w < MASK_WORDS / 4; w += 64) {
      const uint4 v = mask4[w];
      __builtin_nontemporal_store(v.x, &dst[w].x); __builtin_nontemporal_store(v.y, &dst[w].y);
      __builtin_nontemporal_store(v.z, &dst[w].z); __builtin_nontemporal_store(v.w, &dst[w].w);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------
// stateless path: r.get_moves(hand15, last15) for n independent (hand, last) pairs, one
// wavefront per query (tpw consecutive queries per wave); pass 1 (WRITE = false) sizes the
// lists and scans them per block, pass 2 writes the CSR list.
constexpr int MW = BLOCK / 64;  // waves per block of k_moves

template <bool WRITE, bool IDS>
__global__ __launch_bounds__(BLOCK) void k_moves(const uint4* __restrict__ hands, const uint4* __restrict__ lasts,
                                                 int64_t n, int tpw, int32_t* __restrict__ counts,
                                                 int32_t* __restrict__ local_off, int32_t* __restrict__ blk_tot,
                                                 int32_t* __restrict__ offsets, uint4* __restrict__ rows,
                                                 int32_t* __restrict__ ids, int64_t cap, int32_t* __restrict__ status) {
  const int lane = threadIdx.x & 63;
  const int wv = (int)rfl(threadIdx.x >> 6);
  const int64_t t0 = ((int64_t)blockIdx.x * MW + wv) * tpw;
  const int ntab = t0 < n ? (int)(n - t0 < tpw ? n - t0 : tpw) : 0;
  __shared__ HotTabT<true, DDZ_HOT_C64 != 0> hot;  // (nibble-set combinations: the stress set of plane-rich leads)
  int64_t base = 0;
  int cnt_l = 0, new_cnt_l = 0, part = 0, loc0 = 0;
  if (WRITE && ntab > 0) {
    for (int j = lane; j < (int)blockIdx.x; j += 64) part += blk_tot[j];
    loc0 = local_off[t0];
    if (lane < ntab) cnt_l = counts[t0 + lane];
  }
  hot_fill<BLOCK>(hot);
  __syncthreads();
  if (WRITE && ntab > 0) base = (int64_t)wave_sum(part) + loc0;
  for (int i = 0; i < ntab; ++i) {
    const int64_t t = t0 + i;
    const uint4 hr = hands[t], lr = lasts[t];
    const uint64_t hand = pack_row(make_uint4(rfl(hr.x), rfl(hr.y), rfl(hr.z), rfl(hr.w)));
    uint32_t info = classify(pack_row(make_uint4(rfl(lr.x), rfl(lr.y), rfl(lr.z), rfl(lr.w))));
    if (info == INFO_INVALID || ge_mask(hand, 5) || (hand >> 60)) info = QF_BADLAST | QF_FROZEN;
    Pick pk{-1, 0, 0, 0, 0};
    if (WRITE) {
      const int cnt = (int)rl((uint32_t)cnt_l, i);
      if (lane == 0) offsets[t] = (int32_t)base;
      const Out o{rows, ids, base, cap, nullptr, nullptr, nullptr};
      const int m = plan_scan<EM_WRITE, IDS>(hand, info, hot, lane, o, pk);
      if (lane == 0) {
        const int bits = (m != cnt ? 1 : 0) | (base + cnt > cap ? 2 : 0) | ((info & QF_BADLAST) ? 4 : 0);
        if (bits) atomicOr(status, bits);
      }
      base += cnt;
    } else {
      const Out none{nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr};
      const int c = plan_scan<EM_COUNT, false>(hand, info, hot, lane, none, pk);
      if (lane == i) new_cnt_l = c;
    }
  }
  if (WRITE) {
    if (ntab > 0 && t0 + ntab == n && lane == 0) offsets[n] = (int32_t)base;
  } else {
    __shared__ int sh[MW];
    const int incl = wave_incl_scan(new_cnt_l, lane);
    if (lane == 63) sh[wv] = incl;
    __syncthreads();
    int wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < MW; ++w) {
      if (w < wv) wbase += sh[w];
      tot += sh[w];
    }
    if (lane < ntab) {
      counts[t0 + lane] = new_cnt_l;
      local_off[t0 + lane] = wbase + incl - new_cnt_l;
    }
    if (threadIdx.x == 0) blk_tot[blockIdx.x] = tot;
  }
}

// action selection of DQNFirst.greedy_action / e_greedy_action (dqn.py:50-71) for every table:
// choice = first index of the maximum q of the table's list (torch.argmax, dqn.py:60,70); with probability epsilon
// (engine RNG domain 3, one Philox call per table and ply) a uniform index instead (dqn.py:57-58).
// thr = floor(epsilon * 2^32): explore <=> draw.x < thr.
// Sixteen lanes per table: lane s reads entries s, s + 16, ... of the segment (consecutive lanes read consecutive
// floats: coalesced whatever the list size; the slab form used to stride 2 KB per thread), then a 4-step butterfly
// keeps (larger value, on ties the smaller index).
constexpr int SEL_G = 16;
__global__ __launch_bounds__(BLOCK) void k_select(const uint8_t* __restrict__ state, int64_t T, uint32_t k0, uint32_t k1,
                                                  uint64_t gid_base, const float* __restrict__ q,
                                                  const int32_t* __restrict__ offsets, uint64_t thr,
                                                  int32_t* __restrict__ choice, const int32_t* __restrict__ counts,
                                                  int64_t stride) {
  const int64_t gtid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t t = gtid / SEL_G;
  const int sub = (int)(gtid % SEL_G);
  if (t >= T) return;  // whole 16-lane groups leave together (BLOCK is a multiple of 16)
  // CSR: segment [offsets[t], offsets[t+1]); slab (counts != null): [t * stride, t * stride + counts[t])
  const int64_t off = counts ? t * stride : (int64_t)offsets[t];
  int32_t A = counts ? counts[t] : offsets[t + 1] - offsets[t];
  if (counts && A > stride) A = 0;
  if (A <= 0) {
    if (sub == 0) choice[t] = -1;
    return;
  }
  // lane-local first maximum over j = sub, sub + 16, ...; a lane without an entry holds index INT_MAX
  int best = 0x7FFFFFFF;
  float bq = 0.f;
  for (int j = sub; j < A; j += SEL_G) {
    const float v = q[off + j];
    // the sequential rule "replace when strictly greater": a NaN never replaces anything, and a NaN in entry 0 is
    // never replaced -- so NaNs behind entry 0 are skipped here, entry 0 is taken as it is
    if (j == 0 || (v == v && (best == 0x7FFFFFFF || v > bq))) { bq = v; best = j; }
  }
#pragma unroll
  for (int d = 1; d < SEL_G; d <<= 1) {
    const float oq = __shfl_xor(bq, d, SEL_G);
    const int ob = __shfl_xor(best, d, SEL_G);
    // the sequential scan keeps the first entry unless a later one is strictly greater (dqn.py:60 torch.argmax on
    // distinct values; NaNs never win): between two candidates the earlier index stays unless the later is greater
    const bool have = best != 0x7FFFFFFF, ohave = ob != 0x7FFFFFFF;
    const bool take = ohave && (!have || (ob < best ? !(bq > oq) : (oq > bq)));
    if (take) { bq = oq; best = ob; }
  }
  if (sub != 0) return;
  if (thr) {
    const uint4 meta = *(const uint4*)(state + t * STATE_ROW_BYTES + DDZ_F_META * 16);
    const uint64_t gid = gid_base + (uint64_t)t;
    const uint4 d = philox4x32_10(make_uint4((uint32_t)gid, (uint32_t)(gid >> 32), meta.z, (3u << 16) | (meta.y & 0xFFFF)), k0, k1);
    if ((uint64_t)d.x < thr) best = (int)__umulhi(d.y, (uint32_t)A);
  }
  choice[t] = best;
}

// ------------------------------------------------------------------------------------
// k_q_slab: the per-row stage of the ragged Q forward (game.py:95-104 -> dqn.py:56,67: policy_net(face, actions) over
// ALL legal actions of a state; net.py:99-101 relu(fc1) -> fc2) over the slab lists, with the first layer factorised
// per (rank, count) by the host glue (doudizhu-rl_amd/dqn_glue.py FactorisedQ.tables):
//     q[t][j] = b2 + w2 . relu( sum_{r = 0..14} (U[r][cnt_r][t][:] + Z[r][cnt_r][:]) ),   cnt_r = count of rank r in row j of table t
// U f32 [15][5][T][256] (per table), Z f32 [15][5][256] (weights only: the action plane through conv_shunzi and fc1; Z[r][0] = 0).
// One wavefront per table (tpw consecutive tables per wave), lane l owns hidden units 4l..4l+3: the all-zero-count sum
// once per table (15 coalesced 1-KB reads), then per row three 1-KB reads per rank the action touches, a 4-wide dot and
// a DPP wave reduction.  The lists are read where ddz_step_slab left them (counts / rows): no CSR, no padding rows, no
// host sync.
constexpr int QH = 256;  // hidden units of fc1 (net.py:147)
// PACKED (ddz_q_slab_packed): u holds only the (rank, count, table) rows a legal move can use -- rank r's rows start at
// row0[r], its first T rows are count 0 of tables 0..T-1, the row of (r, c >= 1, t) is pidx[t][QP_OF(r, c)] -- and the
// per-table term (fc1 bias + the face part of conv_shunzi) comes as its own [T][256] array instead of riding on rank 0.
constexpr int QP_COLS = 64;  // pidx row: (rank r < 13, count c = 1..4) at 4 r + c - 1, the jokers' count 1 at 52, 53
__device__ __forceinline__ int qp_col(int r, int c) { return r < 13 ? 4 * r + c - 1 : 52 + (r - 13); }
struct QRow0 { int64_t v[16]; };  // first packed row of each rank; [15] = the number of rows
template <bool PACKED>
__global__ __launch_bounds__(TB, 4) void k_q_slab(const float4* __restrict__ U, const float4* __restrict__ Z, int64_t T, int tpw,
                                                 const float4* __restrict__ w2, const float* __restrict__ b2,
                                                 const int32_t* __restrict__ counts, const uint4* __restrict__ rows, int64_t stride,
                                                 float* __restrict__ q, const int32_t* __restrict__ pidx, QRow0 row0,
                                                 const float4* __restrict__ tab, int32_t* __restrict__ status) {
  const int lane = threadIdx.x & 63;
  const int wv = (int)rfl(threadIdx.x >> 6);
  const int64_t t0 = ((int64_t)blockIdx.x * WPB + wv) * tpw;
  const int ntab = t0 < T ? (int)(T - t0 < tpw ? T - t0 : tpw) : 0;
  const float4 w = w2[lane];
  const float bias = b2[0];
  const int64_t cstride = T * (QH / 4);  // float4s between the counts of a rank; 5 of them between two ranks
  for (int i = 0; i < ntab; ++i) {
    const int64_t t = t0 + i;
    int n = (int)rfl((uint32_t)counts[t]);
    if (n < 0 || n > stride) n = 0;
    const float4* ut = U + (PACKED ? 0 : t * (QH / 4)) + lane;
    const float4* zl = Z + lane;
    const uint4* lrow = rows + t * stride;
    float* qt = q + t * stride;
    if (n == 0) continue;
    int32_t myidx = -1;  // PACKED: lane l holds pidx[t][l] (one coalesced 256-byte read per table)
    if (PACKED) myidx = pidx[t * QP_COLS + lane];
    // h0 = the sum with every count 0 (the pass): once per table; a row then swaps in the terms of the ranks it touches
    // (an action touches 1.3 ranks on average)
    float4 h0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PACKED && tab) h0 = tab[t * (QH / 4) + lane];
#pragma unroll
    for (int r = 0; r < 15; ++r) {
      const float4 v = PACKED ? ut[(row0.v[r] + t) * (QH / 4)] : ut[r * 5 * cstride];
      h0.x += v.x; h0.y += v.y; h0.z += v.z; h0.w += v.w;
    }
    for (int j0 = 0; j0 < n; j0 += 64) {
      const int m = n - j0 < 64 ? n - j0 : 64;
      uint4 myrow = make_uint4(0, 0, 0, 0);
      if (lane < m) myrow = lrow[j0 + lane];  // one coalesced read of up to 64 rows; row jj is handed round by readlane
      const uint64_t mynib = pack_row(myrow);
      float res = 0.f;
      for (int jj = 0; jj < m; ++jj) {
        const uint64_t nib = rl64(mynib, jj);  // wave-uniform
        float4 h = h0;
        for (uint32_t tm = ge_mask(nib, 1); tm; tm &= tm - 1) {  // the ranks the action touches
          const int r = __builtin_ctz(tm);
          uint32_t c = (uint32_t)(nib >> (4 * r)) & 15u;
          c = c > 4u ? 4u : c;
          if (r >= 13 && c > 1u) c = 1u;  // (a joker exists once; u holds counts 0 and 1 for ranks 13, 14)
          float4 v, z0;
          if (PACKED) {
            // (a count the actor does not hold has no row: such a move is not legal -- read rank r's count-0 row instead
            // of faulting on a list that does not belong to this state)
            // a row index at or beyond n_rows (a row_index that does not belong to this u: a stale pack, a caller's
            // bug) is never dereferenced: count-0 row + status bit 5
            const int32_t pr = (int32_t)__builtin_amdgcn_readlane(myidx, qp_col(r, (int)c));
            z0 = ut[(row0.v[r] + t) * (QH / 4)];
            const bool inside = (uint32_t)pr < (uint32_t)row0.v[15];
            v = inside ? ut[(int64_t)pr * (QH / 4)] : z0;
            if (pr >= 0 && !inside && lane == 0) atomicOr(status, 32);
          } else {
            v = ut[(r * 5 + (int64_t)c) * cstride]; z0 = ut[r * 5 * cstride];
          }
          const float4 zz = zl[(r * 5 + (int)c) * (QH / 4)];
          h.x += v.x - z0.x + zz.x; h.y += v.y - z0.y + zz.y; h.z += v.z - z0.z + zz.z; h.w += v.w - z0.w + zz.w;
        }
        float p = fmaxf(h.x, 0.f) * w.x + fmaxf(h.y, 0.f) * w.y + fmaxf(h.z, 0.f) * w.z + fmaxf(h.w, 0.f) * w.w;
        p = wave_sum_f32(p);
        if (lane == jj) res = p + bias;
      }
      if (lane < m) qt[j0 + lane] = res;  // coalesced
    }
  }
}

// k_q_feat: the first layer of the same forward per (table, rank, count) -- conv1..conv4 + the (1,4) max-pool of
// net.py:92-94 (each conv_k is a (1,k) window, stride 4, on a width-4 input: ONE output column per rank; the pool is the
// max over the four convs) -- evaluated from `face` alone, for every count cnt = 0..4 an action could take of the rank:
//     Y[r][cnt][t][c] = max_k ( bias_k[c] + sum_{plane, slot < k} W_k[c][plane][slot] * face[t][plane][r][slot] + A[cnt][k][c] )
// (cnt-major inside a rank: the fc1 GEMM then takes counts 0..4 of ranks 3..2 and only counts 0..1 of the two jokers)
// A[cnt][k][c] = the action plane's thermometer (envi.py:139-146: slots < cnt set) through conv_k.  One block of 256
// threads = the 256 channels; a thread keeps its channel's weights in registers (P * 10 + 4 + 16 floats); the block's 16
// tables of `face` are staged in LDS and read back as broadcasts; the loop is rank-major so that the stores of a (rank,
// count) are one 16-KB run of y.  Written once, read once by the fc1 GEMM: bound by its 5 x 1 KB of stores per pair.  (The torch statement of the same stage -- FactorisedQ.tables(fused=False) --
// reads and writes the [T, 15, 4, 256] conv output ten times; this kernel never materialises it.)
constexpr int QF_TILE = 16;  // tables per block
// PACKED (ddz_q_features_packed): only the rows a legal move can use are written, at the packed positions k_q_slab reads.
template <int P, bool PACKED>
__global__ __launch_bounds__(QH) void k_q_feat(const float4* __restrict__ face, int64_t T, const float* __restrict__ wf,
                                               const float* __restrict__ bias, const float* __restrict__ acnt,
                                               float* __restrict__ y, int64_t ystride, const int32_t* __restrict__ pidx,
                                               QRow0 row0) {
  const int c = threadIdx.x;
  // the block's tile of `face` (QF_TILE tables x P x 15 float4: one contiguous piece) goes through LDS: coalesced loads,
  // then wave-uniform (broadcast) LDS reads per (table, rank) pair
  __shared__ float4 s_face[QF_TILE * P * 15];
  const int64_t tb = (int64_t)blockIdx.x * QF_TILE;
  const int nt = (int)(T - tb < QF_TILE ? T - tb : QF_TILE);
  for (int i = threadIdx.x; i < nt * P * 15; i += QH) s_face[i] = face[tb * P * 15 + i];
  __shared__ __attribute__((aligned(16))) int32_t s_pidx[PACKED ? QF_TILE * QP_COLS : 4];  // PACKED: the tile's rows of pidx (one coalesced 4-KB read)
  if (PACKED)
    for (int i = threadIdx.x; i < nt * QP_COLS; i += QH) s_pidx[i] = pidx[tb * QP_COLS + i];
  // this channel's weights: wf [P * 4][4 * 256] (row = plane * 4 + slot, column = k * 256 + c; slots >= k + 1 are zero)
  float w[P][10];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    int q = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j <= k; ++j) w[p][q++] = wf[(int64_t)(p * 4 + j) * (4 * QH) + k * QH + c];
  }
  float b[4], a[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    b[k] = bias[k * QH + c];
#pragma unroll
    for (int n = 0; n < 4; ++n) a[n][k] = acnt[((n + 1) * 4 + k) * QH + c];  // counts 1..4 (count 0 adds nothing)
  }
  __syncthreads();
  const int64_t cs = T * ystride;  // between two counts of a rank
  // rank-major over the tile: for a (rank, count) the tile's tables are consecutive rows of y -- QF_TILE KB per run
  for (int r = 0; r < 15; ++r) {
    const int ncnt = r < 13 ? 4 : 1;  // a joker exists once: counts 2..4 of ranks 13, 14 are never read
    float* dst = y + (PACKED ? row0.v[r] + tb : (int64_t)r * 5 * T + tb) * ystride + c;
    // (two tables per trip with the sums as halves of packed-fp32 registers -- v_pk_fma_f32, half the FMA instructions --
    // measured slower: 0.79 -> 1.30 ms for the packed form at 65,536 tables, 196 VGPRs and ~130 moves to pair the operands)
    for (int ti = 0; ti < nt; ++ti) {
      float s0 = b[0], s1 = b[1], s2 = b[2], s3 = b[3];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const float4 x = s_face[(ti * P + p) * 15 + r];  // wave-uniform address: an LDS broadcast
        s0 += w[p][0] * x.x;
        s1 += w[p][1] * x.x + w[p][2] * x.y;
        s2 += w[p][3] * x.x + w[p][4] * x.y + w[p][5] * x.z;
        s3 += w[p][6] * x.x + w[p][7] * x.y + w[p][8] * x.z + w[p][9] * x.w;
      }
      // (plain stores: y is read back by the fc1 GEMM right behind this kernel)
      float* d = dst + ti * ystride;
      d[0] = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
      if (PACKED) {
        // the rows of counts 1..4 of this (table, rank): ONE 16-byte LDS broadcast (columns 4 r .. 4 r + 3; a joker has
        // one), the tests and the addresses stay in vector registers (the scalar unit was this kernel's second bound:
        // 60 scalar instructions per (table, rank) against 93 vector ones -- 790 -> 605 us at 65,536 tables)
        int4 pr4 = make_int4(-1, -1, -1, -1);
        if (r < 13) pr4 = *(const int4*)&s_pidx[ti * QP_COLS + 4 * r];
        else pr4.x = s_pidx[ti * QP_COLS + 52 + (r - 13)];
        const uint32_t ys = (uint32_t)ystride;  // (rows x stride < 2^31: checked by the caller)
        // (unsigned compares: -1 = not held and anything at or beyond n_rows -- a row_index that does not belong to this
        // y -- are skipped alike, nothing outside y[:n_rows] is ever written)
        const uint32_t nr = (uint32_t)row0.v[15];
        if ((uint32_t)pr4.x < nr) y[(uint32_t)pr4.x * ys + c] = fmaxf(fmaxf(s0 + a[0][0], s1 + a[0][1]), fmaxf(s2 + a[0][2], s3 + a[0][3]));
        if ((uint32_t)pr4.y < nr) y[(uint32_t)pr4.y * ys + c] = fmaxf(fmaxf(s0 + a[1][0], s1 + a[1][1]), fmaxf(s2 + a[1][2], s3 + a[1][3]));
        if ((uint32_t)pr4.z < nr) y[(uint32_t)pr4.z * ys + c] = fmaxf(fmaxf(s0 + a[2][0], s1 + a[2][1]), fmaxf(s2 + a[2][2], s3 + a[2][3]));
        if ((uint32_t)pr4.w < nr) y[(uint32_t)pr4.w * ys + c] = fmaxf(fmaxf(s0 + a[3][0], s1 + a[3][1]), fmaxf(s2 + a[3][2], s3 + a[3][3]));
      } else {
        for (int n = 0; n < ncnt; ++n)
          d[(n + 1) * cs] = fmaxf(fmaxf(s0 + a[n][0], s1 + a[n][1]), fmaxf(s2 + a[n][2], s3 + a[n][3]));
      }
    }
  }
}

#include "ddz_qnet.h"

__global__ __launch_bounds__(BLOCK) void k_classify(const uint4* __restrict__ rows, int64_t n,
                                                    uint32_t* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (t < n) out[t] = classify(pack_row(rows[t]));
}

// stats[0..5] += {plies, episodes, legal rows, lord wins, up wins, down wins}; slots are cleared
__global__ __launch_bounds__(BLOCK) void k_reduce_stats(Scratch sc, int64_t nslots, int64_t* stats) {
  __shared__ long long sh[5][BLOCK];
  long long v[5] = {0, 0, 0, 0, 0};  // plies, episodes, lord wins, rows, up wins
  for (int64_t b = threadIdx.x; b < nslots; b += BLOCK) {
    const long long w = sc.blk_stats[4 * b + 2];
    v[0] += sc.blk_stats[4 * b]; v[1] += sc.blk_stats[4 * b + 1]; v[3] += sc.blk_stats[4 * b + 3];
    v[2] += w & 0xFFFFFFFFll; v[4] += w >> 32;
    for (int k = 0; k < 4; ++k) sc.blk_stats[4 * b + k] = 0;
  }
  for (int k = 0; k < 5; ++k) sh[k][threadIdx.x] = v[k];
  __syncthreads();
  for (int d = BLOCK / 2; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d)
      for (int k = 0; k < 5; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    stats[0] += sh[0][0];
    stats[1] += sh[1][0];
    stats[2] += *sc.legal_rows + sh[3][0];
    stats[3] += sh[2][0];
    stats[4] += sh[4][0];
    stats[5] += sh[1][0] - sh[2][0] - sh[4][0];
    *sc.legal_rows = 0;
  }
}

// ------------------------------------------------------------------------------------
// face: f32 [T][P][15][4], one thread per (table, plane, rank) -> one 16-byte store.
// plane kinds: 0 hand 1 taken 2..4 history of (role-1, role, role+1) 5,6 recent handout of
// (role-1, role-2) 7,8 prob planes (spec v1; native get_state_prob is absent).
// variant 0: {0,1,7,8}  1: {0,1,2,3,4,7,8}  2: {0,..,8}  3: {0,1,5,6,7,8}   (envi.py:87-96,165-217)

// VARIANT is a template parameter: the divisions by P * 15 and by 15 are by constants (a 64-bit runtime
// division per 16-byte store made the first version instruction-bound at half the HBM write rate) and the
// plane kinds are immediates.
template <int VARIANT>
__global__ __launch_bounds__(BLOCK) void k_observe(const uint8_t* __restrict__ state, int64_t T,
                                                   float4* __restrict__ out) {
  constexpr int P = VARIANT == 0 ? 4 : VARIANT == 1 ? 7 : VARIANT == 2 ? 9 : 6;
  // kinds of the planes, 4 bits each, plane 0 in the low nibble
  constexpr uint64_t KINDS = VARIANT == 0 ? 0x8710ull : VARIANT == 1 ? 0x8743210ull
                           : VARIANT == 2 ? 0x876543210ull : 0x876510ull;
  const int64_t idx = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (idx >= T * (P * 15)) return;
  int64_t t;
  int rem;
  if (T * (P * 15) <= 0x7FFFFFFFll) {  // wave-uniform: 32-bit index arithmetic
    const uint32_t i32 = (uint32_t)idx, t32 = i32 / (uint32_t)(P * 15);
    t = t32;
    rem = (int)(i32 - t32 * (uint32_t)(P * 15));
  } else {
    t = idx / (P * 15);
    rem = (int)(idx - t * (P * 15));
  }
  const int p = rem / 15, i = rem - p * 15;
  const uint8_t* row = state + t * STATE_ROW_BYTES;
  auto byte = [&](int f, int k) { return (int)row[f * 16 + k]; };
  int role = byte(DDZ_F_META, 0);
  if (role > 2) role = 0;
  const int kind = (int)((KINDS >> (4 * p)) & 15);
  const int rm1 = role == 0 ? 2 : role - 1, rp1 = role == 2 ? 0 : role + 1;
  float4 v;
  if (kind < 7) {
    const int f = kind == 0 ? DDZ_F_HAND0 + role : kind == 1 ? DDZ_F_TAKEN
                : kind == 2 ? DDZ_F_HIST0 + rm1 : kind == 3 ? DDZ_F_HIST0 + role
                : kind == 4 ? DDZ_F_HIST0 + rp1 : kind == 5 ? DDZ_F_RECENT0 + rm1 : DDZ_F_RECENT0 + rp1;
    const int c = byte(f, i);  // thermometer: slot j set iff count > j (envi.py:139-146)
    v = make_float4(c > 0 ? 1.f : 0.f, c > 1 ? 1.f : 0.f, c > 2 ? 1.f : 0.f, c > 3 ? 1.f : 0.f);
  } else {
    const int n1 = byte(DDZ_F_HAND0 + rp1, 15), n2 = byte(DDZ_F_HAND0 + rm1, 15);
    const int known = byte(DDZ_F_HAND0 + role, i) + byte(DDZ_F_TAKEN, i), total = i < 13 ? 4 : 1;
    const float fr = n1 + n2 > 0 ? (float)(kind == 7 ? n1 : n2) / (float)(n1 + n2) : 0.f;
    auto slot = [&](int j) { return (j >= known && j < total) ? fr : 0.f; };
    v = make_float4(slot(0), slot(1), slot(2), slot(3));
  }
  store_stream(&out[idx], v);
}

// get_state_prob_manual(known60, size1, size2) (server/core.py:26-33; native in the reference, prob planes spec v1):
// one thread per (query, rank); known = thermometer u8 [n][60] of own cards + cards played, decoded by its row sum
// (onehot2arr, envi.py:148-157); out f32 [n][2][15][4] -- the same expression k_observe evaluates for its two planes.
__global__ __launch_bounds__(BLOCK) void k_state_prob(const uint8_t* __restrict__ known60, const int32_t* __restrict__ sizes,
                                                      int64_t n, float4* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (idx >= n * 15) return;
  const int64_t q = idx / 15;
  const int i = (int)(idx - q * 15);
  const uchar4 k4 = *(const uchar4*)(known60 + q * 60 + 4 * i);
  const int known = (k4.x != 0) + (k4.y != 0) + (k4.z != 0) + (k4.w != 0), total = i < 13 ? 4 : 1;
  const int n1 = sizes[2 * q], n2 = sizes[2 * q + 1];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float fr = n1 + n2 > 0 ? (float)(p == 0 ? n1 : n2) / (float)(n1 + n2) : 0.f;
    auto slot = [&](int j) { return (j >= known && j < total) ? fr : 0.f; };
    out[(q * 2 + p) * 15 + i] = make_float4(slot(0), slot(1), slot(2), slot(3));
  }
}

__global__ __launch_bounds__(BLOCK) void k_onehot(const uint8_t* __restrict__ rows, int64_t n,
                                                  float4* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (idx >= n * 15) return;
  const int64_t r = idx / 15;
  const int c = rows[r * 16 + (idx - r * 15)];
  store_stream(&out[idx], make_float4(c > 0 ? 1.f : 0.f, c > 1 ? 1.f : 0.f, c > 2 ? 1.f : 0.f, c > 3 ? 1.f : 0.f));
}

// ------------------------------------------------------------------------------------
// action table export and the compact trajectory record.
constexpr int NUM_ACTIONS_X = DDZ_NUM_ACTIONS + 24 * DDZ_NATIVE_JOKER_KICKERS;
__device__ uint64_t g_sorted_nib[NUM_ACTIONS_X];  // packed rows of all actions, ascending (host-sorted once)
__device__ int32_t g_sorted_id[NUM_ACTIONS_X];    // ... and their canonical ids

// rows[id] = int8 counts[15] + category of action id (card.py:34-159 order; + the joker-kicker rows)
__global__ __launch_bounds__(BLOCK) void k_export_table(uint4* __restrict__ rows) {
  const int id = (int)(blockIdx.x * BLOCK + threadIdx.x);
  if (id >= NUM_ACTIONS_X) return;
  rows[id] = row_of_id(id);
}

// 32-byte trajectory record -> 8 bytes (what has to cross xGMI): the action as its canonical id
//   w0: id (14 bits) | n_legal << 14 (9) | role << 23 (2) | done << 25 | reward code << 26 (0: 0, 1: +1, 2: -1) | flags << 28 (2)
//   w1: choice + 1 (10 bits) | ply << 10 (8) | episode << 18 (low 14 bits)
__global__ __launch_bounds__(BLOCK) void k_pack_traj(const uint4* __restrict__ traj, int64_t n, uint2* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint4 row = traj[2 * i], m = traj[2 * i + 1];
  const uint64_t nib = pack_row(row);
  // 0x3FFF: not an action -- a row outside the action space, or a record of a frozen table / an illegal selection
  // (flags != 0: its all-zero row is NOT the pass)
  int lo = 0, hi = ((m.x >> 24) & 3) ? -1 : NUM_ACTIONS_X - 1, id = 0x3FFF;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const uint64_t v = g_sorted_nib[mid];
    if (v == nib) { id = g_sorted_id[mid]; break; }
    if (v < nib) lo = mid + 1; else hi = mid - 1;
  }
  const uint32_t role = m.x & 0xFF, done = (m.x >> 8) & 1, rew = (m.x >> 16) & 0xFF, flags = (m.x >> 24) & 3;
  const uint32_t rc = rew == 0 ? 0u : rew == 1 ? 1u : 2u;
  const uint32_t w0 = (uint32_t)id | ((m.y & 0x1FF) << 14) | ((role & 3) << 23) | (done << 25) | (rc << 26) | (flags << 28);
  const uint32_t w1 = (((uint32_t)((int32_t)m.w + 1)) & 0x3FF) | (((m.y >> 16) & 0xFF) << 10) | ((m.z & 0x3FFF) << 18);
  out[i] = make_uint2(w0, w1);
}

// ------------------------------------------------------------------------------------
// host side
thread_local int g_last_hip = 0;
constexpr uint32_t MAGIC = 0xDD2E0001u;

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; }
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

inline int hip_fail(hipError_t e) {
  g_last_hip = (int)e;
  return DDZ_EHIP;
}
inline int check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DDZ_OK : hip_fail(e);
}

// tables per wave: one table per wave until the chip is full (256 CUs x 16 waves), then
// consecutive tables share a wave so that the per-wave CSR prefix stays short.  (No environment switches in the
// library: tests and diagnostics change the geometry of a handle through ddz_debug_set_geometry.)
inline int pick_tpw(int64_t T) {
  int64_t v = (T + 4095) / 4096;
  return (int)(v < 1 ? 1 : v > 32 ? 32 : v);
}

int launch_moves(const Scratch& sc, const int8_t* hands, const int8_t* lasts, int64_t n, int32_t* offsets,
                 int8_t* rows, int32_t* ids, int64_t cap, hipStream_t st) {
  int64_t v = (n + 16383) / 16384;
  const int tpw = (int)(v < 1 ? 1 : v > 16 ? 16 : v);
  const int64_t per_block = (int64_t)MW * tpw;
  const dim3 grid((unsigned)((n + per_block - 1) / per_block)), block(BLOCK);
  hipLaunchKernelGGL((k_moves<false, false>), grid, block, 0, st, (const uint4*)hands, (const uint4*)lasts, n, tpw,
                     sc.counts[0], sc.local_off[0], sc.blk_tot[0], offsets, (uint4*)rows, ids, cap, sc.status);
  int rc = check_launch();
  if (rc) return rc;
  if (ids)
    hipLaunchKernelGGL((k_moves<true, true>), grid, block, 0, st, (const uint4*)hands, (const uint4*)lasts, n, tpw,
                       sc.counts[0], sc.local_off[0], sc.blk_tot[0], offsets, (uint4*)rows, ids, cap, sc.status);
  else
    hipLaunchKernelGGL((k_moves<true, false>), grid, block, 0, st, (const uint4*)hands, (const uint4*)lasts, n, tpw,
                       sc.counts[0], sc.local_off[0], sc.blk_tot[0], offsets, (uint4*)rows, ids, cap, sc.status);
  return check_launch();
}

// the record table is built once per device, on the null stream, the first time a handle
// (or a stateless call) needs it; this is the only place the library synchronises by itself
constexpr int MAX_DEVICES = 64;
// The stateless entry points (ddz_get_moves, ddz_action_table, ddz_pack_trajectory, ddz_auto_choose) carry no handle,
// so two host threads may make the first call together: one mutex per device serialises the build, the flag is
// read with acquire / written with release so that a reader never sees a half-built table.
std::mutex g_table_mutex[MAX_DEVICES];
std::atomic<bool> g_table_ready[MAX_DEVICES];
extern uint32_t* g_ticket_base[MAX_DEVICES];
extern int32_t* g_device_status_ptr[MAX_DEVICES];
int build_table_locked(int device);
int ensure_table(int device) {
  if (device < 0 || device >= MAX_DEVICES) return DDZ_ENODEV;
  if (g_table_ready[device].load(std::memory_order_acquire)) return DDZ_OK;
  std::lock_guard<std::mutex> lock(g_table_mutex[device]);
  if (g_table_ready[device].load(std::memory_order_relaxed)) return DDZ_OK;
  const int rc = build_table_locked(device);
  if (rc == DDZ_OK) g_table_ready[device].store(true, std::memory_order_release);
  return rc;
}
int build_table_locked(int device) {
  {
    void* tk = nullptr;
    const hipError_t r0 = hipGetSymbolAddress(&tk, HIP_SYMBOL(g_tickets));
    if (r0 != hipSuccess) return hip_fail(r0);
    g_ticket_base[device] = (uint32_t*)tk;
    void* ds = nullptr;
    const hipError_t r1 = hipGetSymbolAddress(&ds, HIP_SYMBOL(g_device_status));
    if (r1 != hipSuccess) return hip_fail(r1);
    g_device_status_ptr[device] = (int32_t*)ds;
  }
  int32_t* flag = nullptr;
  hipError_t r = hipHostMalloc((void**)&flag, sizeof(int32_t), 0);  // host-pinned status word
  if (r != hipSuccess) return hip_fail(r);
  *flag = 0;
  hipLaunchKernelGGL(k_build_table, dim3(1), dim3(64), 0, (hipStream_t)0, flag);
  int rc = check_launch();
  r = hipStreamSynchronize((hipStream_t)0);
  if (rc == DDZ_OK && r != hipSuccess) rc = hip_fail(r);
  if (rc == DDZ_OK && *flag != 0) rc = DDZ_EHIP;
  (void)hipHostFree(flag);
  if (rc == DDZ_OK) {  // nib -> id lookup for the compact trajectory record: sorted once on the host
    struct Rec { uint64_t nib; int32_t id; };
    Rec* recs = (Rec*)malloc(sizeof(Rec) * NUM_ACTIONS_X);
    uint4* tab = (uint4*)malloc(sizeof(uint4) * 2 * DDZ_NUM_ACTIONS);
    uint64_t* nibs = (uint64_t*)malloc(sizeof(uint64_t) * NUM_ACTIONS_X);
    int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * NUM_ACTIONS_X);
    if (!recs || !tab || !nibs || !ids) rc = DDZ_EINVAL;
    if (rc == DDZ_OK) {
      r = hipMemcpyFromSymbol(tab, HIP_SYMBOL(g_tab), sizeof(uint4) * 2 * DDZ_NUM_ACTIONS);
      if (r != hipSuccess) rc = hip_fail(r);
    }
    if (rc == DDZ_OK) {
      for (int i = 0; i < DDZ_NUM_ACTIONS; ++i) recs[i] = Rec{(uint64_t)tab[2 * i + 1].x | ((uint64_t)tab[2 * i + 1].y << 32), i};
      const uint64_t jk = (1ull << 52) | (1ull << 56);
      for (int k = 0; k < NUM_ACTIONS_X - DDZ_NUM_ACTIONS; ++k)
        recs[DDZ_NUM_ACTIONS + k] = Rec{(k < 13 ? (4ull << (4 * k)) : (0x33ull << (4 * (k - 13)))) | jk, DDZ_NUM_ACTIONS + k};
      qsort(recs, NUM_ACTIONS_X, sizeof(Rec), [](const void* a, const void* b) {
        const uint64_t x = ((const Rec*)a)->nib, y = ((const Rec*)b)->nib;
        return x < y ? -1 : x > y ? 1 : 0;
      });
      for (int i = 0; i < NUM_ACTIONS_X; ++i) { nibs[i] = recs[i].nib; ids[i] = recs[i].id; }
      r = hipMemcpyToSymbol(HIP_SYMBOL(g_sorted_nib), nibs, sizeof(uint64_t) * NUM_ACTIONS_X);
      if (r == hipSuccess) r = hipMemcpyToSymbol(HIP_SYMBOL(g_sorted_id), ids, sizeof(int32_t) * NUM_ACTIONS_X);
      if (r != hipSuccess) rc = hip_fail(r);
    }
    free(recs); free(tab); free(nibs); free(ids);
  }
  return rc;
}

}  // namespace

struct ddz_env {
  uint32_t magic;
  int64_t T;
  uint64_t seed, gid_base;
  int device;
  uint8_t* state;
  void* scratch;
  Layout lay;
  Scratch sc;
  int tpw;            // launch geometry of k_table (fixed per handle: the scan buffers depend on it)
  int64_t nblocks;
  int parity;         // which scan buffer describes the current state
  bool counts_valid;
  int64_t legal_cap;  // capacity of the row buffer the last ddz_legal wrote
  int slab_coop;      // k_slab with one table per wave: wave 0 of a block runs the block's lane-parallel phases
  int slab_team;      // ... and the block writes its heavy lists together
  uint32_t auto_next; // next slot of the k_auto2 queue ring
  int slab_lpt;       // k_slab: block work list of deals + lists, heaviest first (tpw >= 2)
  int auto_teams;     // k_auto2: waves without tables help the searches of their block
  int rollout_waves;  // k_rollout without ids / records: wavefronts per block (12: two blocks per CU; 16: one) -- ddz_debug_set_geometry
};

namespace {
inline bool good(const ddz_env* e) { return e && e->magic == MAGIC; }
// buffers moved as 16-byte (rows, records, planes, masks) / 8-byte / 4-byte words: a misaligned pointer from the
// caller would be a device memory fault, so it is an argument error instead (null = optional buffer, fine)
inline bool al(const void* p, uintptr_t a) { return ((uintptr_t)p & (a - 1)) == 0; }

struct Io {  // optional buffers of one k_table launch
  const void* sel = nullptr;
  int32_t* offsets = nullptr;
  int8_t* rows = nullptr;
  int32_t* ids = nullptr;
  int64_t cap = 0;
  int auto_reset = 0;
  uint8_t* done = nullptr;
  int8_t* reward = nullptr;
  uint8_t* illegal = nullptr;
  uint8_t* traj = nullptr;
  int32_t* slab_counts = nullptr;
  int64_t stride = 0;
};

template <int FLAGS, int MODE>
int launch_table(ddz_env* e, const Io& io, hipStream_t st) {
  TableArgs a;
  a.state = e->state; a.T = e->T; a.tpw = e->tpw;
  a.k0 = (uint32_t)e->seed; a.k1 = (uint32_t)(e->seed >> 32); a.gid_base = e->gid_base;
  a.auto_reset = io.auto_reset; a.sel = io.sel;
  a.offsets = io.offsets; a.rows = (uint4*)io.rows; a.ids = io.ids; a.cap = io.cap;
  const int cur = e->parity, nxt = cur ^ 1;
  a.cur_counts = e->sc.counts[cur]; a.cur_local = e->sc.local_off[cur]; a.cur_blk = e->sc.blk_tot[cur];
  a.nxt_counts = e->sc.counts[nxt]; a.nxt_local = e->sc.local_off[nxt]; a.nxt_blk = e->sc.blk_tot[nxt];
  a.done = io.done; a.reward = io.reward; a.illegal = io.illegal; a.traj = (uint4*)io.traj;
  a.blk_stats = e->sc.blk_stats; a.status = e->sc.status; a.legal_rows = e->sc.legal_rows;
  a.slab_counts = io.slab_counts; a.stride = io.stride;
  const dim3 grid((unsigned)e->nblocks), block(TB);
  if ((FLAGS & (F_ENUM | F_SLAB)) && io.ids)
    hipLaunchKernelGGL((k_table<FLAGS, MODE, true>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((k_table<FLAGS, MODE, false>), grid, block, 0, st, a);
  int rc = check_launch();
  if (rc == DDZ_OK && (FLAGS & F_SLAB)) {
    if (FLAGS & F_STEP) e->counts_valid = false;  // the state moved on without refreshing the CSR scan buffers
  } else if (rc == DDZ_OK && (FLAGS & (F_STEP | F_RESET | F_COUNT))) {
    e->parity = nxt;
    e->counts_valid = true;
  }
  return rc;
}

int ensure_counts(ddz_env* e, hipStream_t st) {
  if (e->counts_valid) return DDZ_OK;
  return launch_table<F_COUNT, 0>(e, Io{}, st);
}

}  // namespace

namespace {
// k_auto2 keeps ~150 KB of LDS per block: one block per CU, tables round-robin over the resident waves.
// DDZ_AUTO_KERNEL=1 selects the sequential reference kernel k_auto (diagnostics; same results).
static int auto_blocks(int device, int64_t n) {
  static int cus[MAX_DEVICES] = {};
  if (device >= 0 && device < MAX_DEVICES && cus[device] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v <= 0) v = 256;
    cus[device] = v;
  }
  const int64_t want = (n + A2_WPB - 1) / A2_WPB;
  const int64_t cap = device >= 0 && device < MAX_DEVICES ? cus[device] : 256;
  return (int)(want < cap ? want : cap);
}
// k_auto2 hands its tables out through a ticket word (next table to take).  Every launch gets its OWN word from a
// per-device ring of device globals, zeroed on the launch stream right before the kernel: no re-arm protocol, nothing a
// failed / rejected / concurrent launch (another stream of the same handle, the stateless entry point) can leave behind.
std::atomic<uint32_t> g_ticket_next[MAX_DEVICES];
int32_t* g_device_status_ptr[MAX_DEVICES];  // device address of g_device_status (the stateless entry points' status word)
uint32_t* g_ticket_base[MAX_DEVICES];  // device address of g_tickets, resolved once per device (under the table mutex)
constexpr int AUTO_K_SEQUENTIAL = 1, AUTO_K_LANES = 2, AUTO_K_LANES_TABLE_ORDER = 3;
// the handle's form: heaviest hands first (k_auto_order into a slot of the handle's ring), then k_auto2 over that queue
static int launch_auto_ordered(ddz_env* e, AutoArgs& a, hipStream_t st) {
  uint8_t* slot_mem = (uint8_t*)e->scratch + e->lay.off_auto + (int64_t)(e->auto_next++ % AUTO_SLOTS) * e->lay.auto_slot_bytes;
  AutoOrder* slot = (AutoOrder*)slot_mem;
  int32_t* order = (int32_t*)(slot_mem + AO_HDR_BYTES);
  const hipError_t r = hipMemsetAsync(slot, 0, sizeof(AutoOrder), st);
  if (r != hipSuccess) return hip_fail(r);
  const dim3 grid((unsigned)((a.T + AO_BT - 1) / AO_BT)), block(AO_BT);
  hipLaunchKernelGGL(k_auto_order<0>, grid, block, 0, st, a.state, a.T, a.auto_roles, slot, order, a.ids, a.stats);
  hipLaunchKernelGGL(k_auto_order<1>, grid, block, 0, st, a.state, a.T, a.auto_roles, slot, order, a.ids, a.stats);
  int rc = check_launch();
  if (rc) return rc;
  a.ticket = &slot->ticket; a.order = order; a.order_hdr = (const uint32_t*)slot;
  static_assert(offsetof(AutoOrder, total) == 0 && offsetof(AutoOrder, nsingle) == 8, "k_auto2 reads words 0 and 2");
  hipLaunchKernelGGL(k_auto2<true>, dim3((unsigned)auto_blocks(e->device, a.T)), dim3(A2_TB), 0, st, a);
  return check_launch();
}

template <bool STATE>
static int launch_auto(int device, AutoArgs& a, hipStream_t st, int kernel = AUTO_K_LANES) {
  if (kernel == AUTO_K_SEQUENTIAL) {  // the sequential cross-check kernel (ddz_debug_auto_choose_state only)
    hipLaunchKernelGGL(k_auto<STATE>, dim3((unsigned)((a.T + WPB - 1) / WPB)), dim3(TB), 0, st, a);
    return check_launch();
  }
  if (device < 0 || device >= MAX_DEVICES || !g_ticket_base[device]) return DDZ_ENODEV;
  const uint32_t slot = g_ticket_next[device].fetch_add(1u, std::memory_order_relaxed) % A2_TICKET_SLOTS;
  a.ticket = g_ticket_base[device] + slot;
  const hipError_t r = hipMemsetAsync(a.ticket, 0, sizeof(uint32_t), st);
  if (r != hipSuccess) return hip_fail(r);
  hipLaunchKernelGGL(k_auto2<STATE>, dim3((unsigned)auto_blocks(device, a.T)), dim3(A2_TB), 0, st, a);
  return check_launch();
}

// rank_row0[15] (host): first packed row of each rank, in any order; validated against the table count: every rank owns at
// least its T count-0 rows, inside [0, n_rows), and no two ranks' count-0 rows overlap
static bool q_row0(const int64_t* rank_row0, int64_t n_rows, int64_t T, QRow0& out) {
  if (!rank_row0 || n_rows <= 0 || n_rows > ((int64_t)1 << 31) - 1) return false;
  for (int r = 0; r < 15; ++r) {
    out.v[r] = rank_row0[r];
    if (rank_row0[r] < 0 || rank_row0[r] > n_rows - T) return false;
    for (int q = 0; q < r; ++q) {
      const int64_t d = rank_row0[r] > rank_row0[q] ? rank_row0[r] - rank_row0[q] : rank_row0[q] - rank_row0[r];
      if (d < T) return false;
    }
  }
  out.v[15] = n_rows;
  return true;
}

template <bool PACKED>
static int launch_q_features(int device, const float* face, int64_t n_tables, int planes, const float* wf, const float* bias,
                             const float* acnt, float* y, int64_t y_row_stride, const int32_t* pidx, const QRow0& row0,
                             void* stream) {
  if (!al(face, 16) || !al(wf, 4) || !al(bias, 4) || !al(acnt, 4) || !al(y, 4)) return DDZ_EINVAL;
  if (!face || !wf || !bias || !acnt || !y || n_tables <= 0 || y_row_stride < QH) return DDZ_EINVAL;
  if (n_tables > ((int64_t)1 << 30)) return DDZ_ECAP;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  const dim3 grid((unsigned)((n_tables + QF_TILE - 1) / QF_TILE)), block(QH);
  hipStream_t st = (hipStream_t)stream;
  const float4* f = (const float4*)face;
  switch (planes) {
    case 4: hipLaunchKernelGGL((k_q_feat<4, PACKED>), grid, block, 0, st, f, n_tables, wf, bias, acnt, y, y_row_stride, pidx, row0); break;
    case 6: hipLaunchKernelGGL((k_q_feat<6, PACKED>), grid, block, 0, st, f, n_tables, wf, bias, acnt, y, y_row_stride, pidx, row0); break;
    case 7: hipLaunchKernelGGL((k_q_feat<7, PACKED>), grid, block, 0, st, f, n_tables, wf, bias, acnt, y, y_row_stride, pidx, row0); break;
    case 9: hipLaunchKernelGGL((k_q_feat<9, PACKED>), grid, block, 0, st, f, n_tables, wf, bias, acnt, y, y_row_stride, pidx, row0); break;
    default: return DDZ_EINVAL;
  }
  return check_launch();
}

static void fill_round_penalty(AutoArgs& a) {
  // rule_based_model.py:57: round_penalty = 15 - 12 * min_oppo_cards / 20 (Python: int product, true division)
  for (int m = 0; m < 24; ++m) a.rp[m] = 15 - 12 * m / 20.0;
}

}  // namespace

extern "C" {

int ddz_abi_version(void) { return DDZ_ABI_VERSION; }
int ddz_num_actions(void) { return DDZ_NUM_ACTIONS + 24 * DDZ_NATIVE_JOKER_KICKERS; }

const char* ddz_strerror(int code) {
  switch (code) {
    case DDZ_OK: return "ok";
    case DDZ_EINVAL: return "invalid argument";
    case DDZ_EHANDLE: return "bad handle";
    case DDZ_EHIP: return "HIP runtime error";
    case DDZ_ECAP: return "row capacity not indexable with int32";
    case DDZ_ENODEV: return "no usable device";
    default: return "unknown error";
  }
}

int ddz_last_hip_error(void) { return g_last_hip; }

int64_t ddz_state_bytes(int64_t T) { return T < 0 ? DDZ_EINVAL : T * DDZ_NFIELDS * DDZ_ROW; }
int64_t ddz_scratch_bytes(int64_t T) { return T < 0 ? DDZ_EINVAL : make_layout(T).bytes; }
int ddz_face_planes(int v) {
  static const int p[4] = {4, 7, 9, 6};
  return v >= 0 && v < 4 ? p[v] : DDZ_EINVAL;
}

int ddz_create(ddz_env_t** out, int64_t T, uint64_t seed, uint64_t gid_base, int device, void* state,
               int64_t state_bytes, void* scratch, int64_t scratch_bytes) {
  if (!out || T <= 0 || !state || !scratch) return DDZ_EINVAL;
  if (T > (int64_t)1 << 30) return DDZ_EINVAL;
  if (state_bytes < ddz_state_bytes(T) || scratch_bytes < ddz_scratch_bytes(T)) return DDZ_EINVAL;
  if (((uintptr_t)state | (uintptr_t)scratch) & 15) return DDZ_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return DDZ_ENODEV;
  {
    DeviceGuard g(device);
    if (!g.ok) return DDZ_ENODEV;
    int rc = ensure_table(device);
    if (rc) return rc;
  }
  ddz_env* e = (ddz_env*)calloc(1, sizeof(ddz_env));
  if (!e) return DDZ_EINVAL;
  e->magic = MAGIC; e->T = T; e->seed = seed; e->gid_base = gid_base; e->device = device;
  e->state = (uint8_t*)state; e->scratch = scratch;
  e->lay = make_layout(T); e->sc = bind(scratch, e->lay);
  e->tpw = pick_tpw(T);
  e->nblocks = (T + (int64_t)WPB * e->tpw - 1) / ((int64_t)WPB * e->tpw);
  e->parity = 0; e->counts_valid = false; e->legal_cap = 0;
  e->slab_coop = e->tpw == 1;
  e->slab_lpt = e->tpw >= 2;
  e->slab_team = 5;
  e->rollout_waves = 12;
  e->auto_teams = 2;   // teams at the queue's end (1 = also team-first for the predicted-heaviest: measured, no gain -- DESIGN.md 9)
  *out = e;
  return DDZ_OK;
}

int ddz_destroy(ddz_env_t* e) {
  if (!good(e)) return DDZ_EHANDLE;
  e->magic = 0;
  free(e);
  return DDZ_OK;
}

int ddz_invalidate(ddz_env_t* e) {
  if (!good(e)) return DDZ_EHANDLE;
  e->counts_valid = false;
  return DDZ_OK;
}

int ddz_reset(ddz_env_t* e, const uint8_t* mask, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  Io io;
  io.sel = mask;
  return launch_table<F_RESET, 0>(e, io, (hipStream_t)stream);
}

int ddz_legal(ddz_env_t* e, int32_t* offsets, int8_t* rows, int32_t* ids, int64_t cap, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(rows, 16) || !al(offsets, 4) || !al(ids, 4)) return DDZ_EINVAL;
  if (!offsets || !rows || cap < 0) return DDZ_EINVAL;
  if (cap > 0x7FFFFFFF) return DDZ_ECAP;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  int rc = ensure_counts(e, (hipStream_t)stream);
  if (rc) return rc;
  e->legal_cap = cap;
  Io io;
  io.offsets = offsets; io.rows = rows; io.ids = ids; io.cap = cap;
  return launch_table<F_ENUM, 0>(e, io, (hipStream_t)stream);
}

int ddz_step(ddz_env_t* e, int mode, const void* sel, const int32_t* offsets, const int8_t* rows, int auto_reset,
             uint8_t* done, int8_t* reward, uint8_t* illegal, uint8_t* traj, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(rows, 16) || !al(offsets, 4) || !al(traj, 16) || (mode == DDZ_STEP_ROWS ? !al(sel, 16) : !al(sel, 4))) return DDZ_EINVAL;
  if (mode < DDZ_STEP_RANDOM || mode > DDZ_STEP_IDS || !offsets || !rows) return DDZ_EINVAL;
  if (mode != DDZ_STEP_RANDOM && !sel) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  Io io;
  io.sel = sel; io.offsets = (int32_t*)offsets; io.rows = (int8_t*)rows; io.cap = e->legal_cap;
  io.auto_reset = auto_reset ? 1 : 0; io.done = done; io.reward = reward; io.illegal = illegal; io.traj = traj;
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case DDZ_STEP_RANDOM: return launch_table<F_STEP, DDZ_STEP_RANDOM>(e, io, st);
    case DDZ_STEP_CHOICE: return launch_table<F_STEP, DDZ_STEP_CHOICE>(e, io, st);
    case DDZ_STEP_IDS: return launch_table<F_STEP, DDZ_STEP_IDS>(e, io, st);
    default: return launch_table<F_STEP, DDZ_STEP_ROWS>(e, io, st);
  }
}

int ddz_legal_slab(ddz_env_t* e, int32_t* counts, int8_t* rows, int32_t* ids, int64_t stride, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(rows, 16) || !al(counts, 4) || !al(ids, 4)) return DDZ_EINVAL;
  if (!counts || !rows || stride < DDZ_SLAB_MIN_STRIDE) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  Io io;
  io.rows = rows; io.ids = ids; io.slab_counts = counts; io.stride = stride;
  return launch_table<F_SLAB, 0>(e, io, (hipStream_t)stream);
}

int ddz_step_slab(ddz_env_t* e, int mode, const void* sel, int32_t* counts, int8_t* rows, int32_t* ids, int64_t stride,
                  int auto_reset, uint8_t* done, int8_t* reward, uint8_t* illegal, uint8_t* traj, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(rows, 16) || !al(counts, 4) || !al(ids, 4) || !al(traj, 16) || (mode == DDZ_STEP_ROWS ? !al(sel, 16) : !al(sel, 4))) return DDZ_EINVAL;
  if (mode < DDZ_STEP_RANDOM || mode > DDZ_STEP_IDS || !counts || !rows || stride < DDZ_SLAB_MIN_STRIDE) return DDZ_EINVAL;
  if (mode != DDZ_STEP_RANDOM && !sel) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  SlabArgs a;
  a.state = e->state; a.T = e->T; a.tpw = e->tpw;
  a.k0 = (uint32_t)e->seed; a.k1 = (uint32_t)(e->seed >> 32); a.gid_base = e->gid_base;
  a.auto_reset = auto_reset ? 1 : 0; a.sel = sel; a.counts = counts; a.rows = (uint4*)rows; a.ids = ids; a.stride = stride;
  a.done = done; a.reward = reward; a.illegal = illegal; a.traj = (uint4*)traj;
  a.wave_stats = e->sc.blk_stats; a.status = e->sc.status;
  a.thr = 0; a.choice_out = nullptr; a.face = nullptr; a.face_variant = 0;
  a.coop = e->slab_coop; a.lpt = e->slab_lpt; a.team = e->slab_team;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)e->nblocks), block(TB);
#define DDZ_LAUNCH_SLAB(M)                                                                    \
  do {                                                                                        \
    if (a.coop) {                                                                             \
      if (ids) hipLaunchKernelGGL((k_slab<M, true, true>), grid, block, 0, st, a);            \
      else hipLaunchKernelGGL((k_slab<M, false, true>), grid, block, 0, st, a);               \
    } else {                                                                                  \
      if (ids) hipLaunchKernelGGL((k_slab<M, true, false>), grid, block, 0, st, a);           \
      else hipLaunchKernelGGL((k_slab<M, false, false>), grid, block, 0, st, a);              \
    }                                                                                         \
  } while (0)
  switch (mode) {
    case DDZ_STEP_RANDOM: DDZ_LAUNCH_SLAB(DDZ_STEP_RANDOM); break;
    case DDZ_STEP_CHOICE: DDZ_LAUNCH_SLAB(DDZ_STEP_CHOICE); break;
    case DDZ_STEP_IDS: DDZ_LAUNCH_SLAB(DDZ_STEP_IDS); break;
    default: DDZ_LAUNCH_SLAB(DDZ_STEP_ROWS); break;
  }
#undef DDZ_LAUNCH_SLAB
  e->counts_valid = false;  // the state moved on without refreshing the CSR scan buffers
  return check_launch();
}

int ddz_policy_step_slab(ddz_env_t* e, const float* q, double epsilon, int32_t* counts, int8_t* rows, int32_t* ids,
                         int64_t stride, int auto_reset, uint8_t* done, int8_t* reward, uint8_t* illegal, uint8_t* traj,
                         int32_t* choice, int face_variant, float* face, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(rows, 16) || !al(counts, 4) || !al(ids, 4) || !al(traj, 16) || !al(q, 4) || !al(choice, 4) || !al(face, 16)) return DDZ_EINVAL;
  if (!q || !counts || !rows || stride < DDZ_SLAB_MIN_STRIDE || !(epsilon >= 0.0) || epsilon > 1.0) return DDZ_EINVAL;
  if (face && ddz_face_planes(face_variant) < 0) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  SlabArgs a;
  a.state = e->state; a.T = e->T; a.tpw = e->tpw;
  a.k0 = (uint32_t)e->seed; a.k1 = (uint32_t)(e->seed >> 32); a.gid_base = e->gid_base;
  a.auto_reset = auto_reset ? 1 : 0; a.sel = q; a.counts = counts; a.rows = (uint4*)rows; a.ids = ids; a.stride = stride;
  a.done = done; a.reward = reward; a.illegal = illegal; a.traj = (uint4*)traj;
  a.wave_stats = e->sc.blk_stats; a.status = e->sc.status;
  a.thr = (uint64_t)(epsilon * 4294967296.0); a.choice_out = choice; a.face = (float4*)face; a.face_variant = face_variant;
  a.coop = e->slab_coop; a.lpt = e->slab_lpt; a.team = e->slab_team;
  const dim3 grid((unsigned)e->nblocks), block(TB);
  if (a.coop) {
    if (ids) hipLaunchKernelGGL((k_slab<STEP_Q, true, true>), grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_slab<STEP_Q, false, true>), grid, block, 0, (hipStream_t)stream, a);
  } else {
    if (ids) hipLaunchKernelGGL((k_slab<STEP_Q, true, false>), grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_slab<STEP_Q, false, false>), grid, block, 0, (hipStream_t)stream, a);
  }
  e->counts_valid = false;
  return check_launch();
}

int ddz_mask_words(void) { return MASK_WORDS; }

int ddz_legal_mask(ddz_env_t* e, uint32_t* mask, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(mask, 16)) return DDZ_EINVAL;
  if (!mask) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  hipLaunchKernelGGL(k_mask, dim3((unsigned)e->nblocks), dim3(TB), 0, (hipStream_t)stream, (const uint8_t*)e->state,
                     e->T, e->tpw, mask);
  return check_launch();
}

int ddz_observe(ddz_env_t* e, int variant, float* face, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(face, 16)) return DDZ_EINVAL;
  const int P = ddz_face_planes(variant);
  if (P < 0 || !face) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  const int64_t n = e->T * P * 15;
  const dim3 grid((unsigned)((n + BLOCK - 1) / BLOCK)), block(BLOCK);
  const uint8_t* st = (const uint8_t*)e->state;
  switch (variant) {
    case 0: hipLaunchKernelGGL(k_observe<0>, grid, block, 0, (hipStream_t)stream, st, e->T, (float4*)face); break;
    case 1: hipLaunchKernelGGL(k_observe<1>, grid, block, 0, (hipStream_t)stream, st, e->T, (float4*)face); break;
    case 2: hipLaunchKernelGGL(k_observe<2>, grid, block, 0, (hipStream_t)stream, st, e->T, (float4*)face); break;
    default: hipLaunchKernelGGL(k_observe<3>, grid, block, 0, (hipStream_t)stream, st, e->T, (float4*)face); break;
  }
  return check_launch();
}

int ddz_rows_to_onehot(int device, const int8_t* rows, int64_t n, float* out, void* stream) {
  if (!al(rows, 16) || !al(out, 16)) return DDZ_EINVAL;
  if (n < 0 || (n > 0 && (!rows || !out))) return DDZ_EINVAL;
  if (n == 0) return DDZ_OK;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  const int64_t m = n * 15;
  hipLaunchKernelGGL(k_onehot, dim3((unsigned)((m + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     (const uint8_t*)rows, n, (float4*)out);
  return check_launch();
}

int ddz_observe_actions(ddz_env_t* e, int variant, float* face, const int8_t* rows, int64_t n, float* onehot, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (n < 0 || (n > 0 && (!rows || !onehot)) || !al(rows, 16) || !al(onehot, 16)) return DDZ_EINVAL;
  const int rc = ddz_observe(e, variant, face, stream);
  if (rc != DDZ_OK || n == 0) return rc;
  return ddz_rows_to_onehot(e->device, rows, n, onehot, stream);
}

int ddz_state_prob(int device, const uint8_t* known60, const int32_t* sizes, int64_t n, float* out, void* stream) {
  if (!al(known60, 4) || !al(sizes, 4) || !al(out, 16)) return DDZ_EINVAL;
  if (n < 0 || (n > 0 && (!known60 || !sizes || !out))) return DDZ_EINVAL;
  if (n == 0) return DDZ_OK;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  const int64_t m = n * 15;
  hipLaunchKernelGGL(k_state_prob, dim3((unsigned)((m + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     known60, sizes, n, (float4*)out);
  return check_launch();
}

int ddz_get_moves(int device, const int8_t* hands, const int8_t* lasts, int64_t n, int32_t* offsets, int8_t* rows,
                  int32_t* ids, int64_t cap, void* scratch, int64_t scratch_bytes, void* stream) {
  if (!al(hands, 16) || !al(lasts, 16) || !al(rows, 16) || !al(offsets, 4) || !al(ids, 4) || !al(scratch, 16)) return DDZ_EINVAL;
  if (n <= 0 || !hands || !lasts || !offsets || !rows || !scratch || cap < 0) return DDZ_EINVAL;
  if (cap > 0x7FFFFFFF) return DDZ_ECAP;
  const Layout l = make_layout(n);
  if (scratch_bytes < l.bytes || ((uintptr_t)scratch & 15)) return DDZ_EINVAL;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  int rc = ensure_table(device);
  if (rc) return rc;
  return launch_moves(bind(scratch, l), hands, lasts, n, offsets, rows, ids, cap, (hipStream_t)stream);
}

int ddz_get_moves_slab(int device, const int8_t* hands, const int8_t* lasts, int64_t n, int32_t* counts, int8_t* rows,
                       int32_t* ids, int64_t stride, int32_t* status, void* stream) {
  if (!al(hands, 16) || !al(lasts, 16) || !al(rows, 16) || !al(counts, 4) || !al(ids, 4) || !al(status, 4)) return DDZ_EINVAL;
  if (n <= 0 || !hands || !lasts || !counts || !rows || stride < DDZ_SLAB_MIN_STRIDE) return DDZ_EINVAL;
  if (n > ((int64_t)1 << 30) || n * stride > 0x7FFFFFFFll * 16) return DDZ_ECAP;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  int rc = ensure_table(device);
  if (rc) return rc;
  int64_t v = (n + 4095) / 4096;
  const int tpw = (int)(v < 1 ? 1 : v > 32 ? 32 : v);
  const int64_t per_block = (int64_t)WPB * tpw;
  const dim3 grid((unsigned)((n + per_block - 1) / per_block)), block(TB);
  if (ids)
    hipLaunchKernelGGL((k_moves_slab<true>), grid, block, 0, (hipStream_t)stream, (const uint4*)hands, (const uint4*)lasts, n,
                       tpw, counts, (uint4*)rows, ids, stride, status);
  else
    hipLaunchKernelGGL((k_moves_slab<false>), grid, block, 0, (hipStream_t)stream, (const uint4*)hands, (const uint4*)lasts, n,
                       tpw, counts, (uint4*)rows, ids, stride, status);
  return check_launch();
}

int ddz_slab_to_csr(ddz_env_t* e, const int32_t* counts, const int8_t* rows, const int32_t* ids, int64_t stride,
                    int32_t* offsets, int8_t* rows_out, int32_t* ids_out, int64_t row_capacity, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(rows, 16) || !al(rows_out, 16) || !al(counts, 4) || !al(ids, 4) || !al(ids_out, 4) || !al(offsets, 4)) return DDZ_EINVAL;
  if (!counts || !rows || !offsets || !rows_out || stride < 1 || row_capacity < 0 || (ids_out && !ids)) return DDZ_EINVAL;
  if (row_capacity > 0x7FFFFFFF) return DDZ_ECAP;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nb = (e->T + CSR_BT - 1) / CSR_BT;  // <= lay.nblk = ceil(T / 4)
  e->counts_valid = false;                             // the scan buffers of the CSR path are reused here
  hipLaunchKernelGGL(k_csr_scan, dim3((unsigned)nb), dim3(CSR_BT), 0, st, counts, e->T, stride, e->sc.local_off[0], e->sc.blk_tot[0],
                     CsrBatch{});
  int rc = check_launch();
  if (rc) return rc;
  if (ids_out)
    hipLaunchKernelGGL((k_csr_copy<true>), dim3((unsigned)nb), dim3(CSR_BT), 0, st, counts, (const uint4*)rows, ids, e->T, stride,
                       e->sc.local_off[0], e->sc.blk_tot[0], offsets, (uint4*)rows_out, ids_out, row_capacity, e->sc.status, CsrBatch{});
  else
    hipLaunchKernelGGL((k_csr_copy<false>), dim3((unsigned)nb), dim3(CSR_BT), 0, st, counts, (const uint4*)rows, ids, e->T, stride,
                       e->sc.local_off[0], e->sc.blk_tot[0], offsets, (uint4*)rows_out, ids_out, row_capacity, e->sc.status, CsrBatch{});
  return check_launch();
}

int ddz_read_stats(ddz_env_t* e, int64_t* stats, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(stats, 8)) return DDZ_EINVAL;
  if (!stats) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(BLOCK), 0, (hipStream_t)stream, e->sc, e->T, stats);
  return check_launch();
}

static int launch_rollout(ddz_env* e, int64_t n_iters, int32_t* counts, int8_t* rows, int32_t* ids, int64_t stride,
                          uint8_t* traj, hipStream_t st, int64_t it_rows = 0, int64_t it_counts = 0) {
  RolloutArgs a;
  a.state = e->state; a.T = e->T; a.tpw = e->tpw;
  a.k0 = (uint32_t)e->seed; a.k1 = (uint32_t)(e->seed >> 32); a.gid_base = e->gid_base;
  a.counts = counts; a.rows = (uint4*)rows; a.ids = ids; a.stride = stride;
  a.it_rows = it_rows; a.it_counts = it_counts;
  a.wave_stats = e->sc.blk_stats; a.status = e->sc.status; a.legal_rows = e->sc.legal_rows;
  const dim3 grid((unsigned)e->nblocks), block(TB);
  // the variants without ids fit 85 VGPRs: 12-wave blocks, two per CU, tables per wave so that the grid
  // is one round of 6 waves per SIMD (256 CUs x 24 waves)
  constexpr int RW12 = 12;
  const int tpw12 = (int)((e->T + 6143) / 6144 < 1 ? 1 : (e->T + 6143) / 6144);
  const int64_t waves12 = (e->T + tpw12 - 1) / tpw12;
  // ... when that grid fills the chip (>= 88 % of the 512 block slots: at 4096 tables a third of the CUs would hold two
  // blocks and the others one -- measured 2.66 against 3.47 G steps/s; at 65,536 tables 3.98 against 3.59 G)
  const bool dense = !ids && e->rollout_waves == RW12 && waves12 >= 5400;   // (with records: 80 VGPRs + two spilled dwords)
  const dim3 grid12((unsigned)((waves12 + RW12 - 1) / RW12)), block12(RW12 * 64);
  if (dense) a.tpw = tpw12;
  // the kernel counts iterations and plies in 32 bits: at most 2^20 iterations per launch (about a second)
  constexpr int64_t CHUNK = 1 << 20;
  for (int64_t done = 0; done < n_iters; done += CHUNK) {
    a.n_iters = n_iters - done < CHUNK ? n_iters - done : CHUNK;
    a.traj = traj ? (uint4*)(traj + done * e->T * DDZ_TRAJ_BYTES) : nullptr;
    if (dense) {
      if (it_rows && traj) hipLaunchKernelGGL((k_rollout<false, true, true, RW12>), grid12, block12, 0, st, a);
      else if (it_rows) hipLaunchKernelGGL((k_rollout<false, false, true, RW12>), grid12, block12, 0, st, a);
      else if (traj) hipLaunchKernelGGL((k_rollout<false, true, false, RW12>), grid12, block12, 0, st, a);
      else hipLaunchKernelGGL((k_rollout<false, false, false, RW12>), grid12, block12, 0, st, a);
    } else if (it_rows) {
      if (ids && traj) hipLaunchKernelGGL((k_rollout<true, true, true>), grid, block, 0, st, a);
      else if (ids) hipLaunchKernelGGL((k_rollout<true, false, true>), grid, block, 0, st, a);
      else if (traj) hipLaunchKernelGGL((k_rollout<false, true, true>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((k_rollout<false, false, true>), grid, block, 0, st, a);
    } else if (ids && traj) hipLaunchKernelGGL((k_rollout<true, true>), grid, block, 0, st, a);
    else if (ids) hipLaunchKernelGGL((k_rollout<true, false>), grid, block, 0, st, a);
    else if (traj) hipLaunchKernelGGL((k_rollout<false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_rollout<false, false>), grid, block, 0, st, a);
  }
  e->counts_valid = false;  // the state moved on without refreshing the CSR scan buffers
  return check_launch();
}

int ddz_rollout_random(ddz_env_t* e, int64_t n_iters, int32_t* counts, int8_t* rows, int32_t* ids, int64_t stride,
                       int64_t* stats, uint8_t* traj, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(rows, 16) || !al(counts, 4) || !al(ids, 4) || !al(traj, 16) || !al(stats, 8)) return DDZ_EINVAL;
  if (n_iters < 0 || !counts || !rows || stride < DDZ_SLAB_MIN_STRIDE) return DDZ_EINVAL;
  if (e->T * stride > 0x7FFFFFFF) return DDZ_ECAP;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  hipStream_t st = (hipStream_t)stream;
  if (n_iters > 0) {  // tables do not depend on each other: all iterations run inside one launch
    int rc = launch_rollout(e, n_iters, counts, rows, ids, stride, traj, st);
    if (rc) return rc;
  }
  if (stats) {
    hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(BLOCK), 0, st, e->sc, e->T, stats);
    return check_launch();
  }
  return DDZ_OK;
}

int ddz_rollout_random_csr(ddz_env_t* e, int64_t n_iters, int32_t* offsets, int8_t* rows, int32_t* ids, int64_t cap,
                           uint8_t* traj, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(rows, 16) || !al(offsets, 4) || !al(ids, 4) || !al(traj, 16)) return DDZ_EINVAL;
  if (n_iters < 0 || !offsets || !rows || cap < 0) return DDZ_EINVAL;
  if (cap > 0x7FFFFFFF) return DDZ_ECAP;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  hipStream_t st = (hipStream_t)stream;
  e->legal_cap = cap;
  Io io;
  io.offsets = offsets; io.rows = rows; io.ids = ids; io.cap = cap; io.auto_reset = 1;
  for (int64_t it = 0; it < n_iters; ++it) {
    int rc = ensure_counts(e, st);
    if (rc) return rc;
    io.traj = traj ? traj + it * e->T * DDZ_TRAJ_BYTES : nullptr;
    rc = launch_table<F_ENUM | F_STEP, DDZ_STEP_RANDOM>(e, io, st);  // CSR needs the scan of the previous launch
    if (rc) return rc;
  }
  return DDZ_OK;
}

// The CSR rollout without a launch per iteration: the lists of a BATCH of iterations are staged as slabs by ONE k_rollout
// launch (state in registers, no table waits for another one), then compacted to CSR by one k_csr_scan + one k_csr_copy
// launch over (blocks x iterations of the batch) -- the cross-table prefix of an iteration is computed after the fact, so
// it is nobody's launch boundary.  Every iteration's lists are written to offsets / rows / ids at their CSR positions as
// ddz_legal would write them; the last iteration of the call is compacted by a launch of its own behind all the others, so
// the buffers end up holding exactly its lists (earlier iterations of a batch are written concurrently).
int64_t ddz_rollout_csr_staging_bytes(int64_t n_tables, int batch, int want_ids) {
  if (n_tables <= 0 || batch < 1) return 0;
  const int64_t nb = (n_tables + CSR_BT - 1) / CSR_BT;
  int64_t o = 0;
  o = align_up(o + (int64_t)batch * n_tables * 4, 256);                        // counts
  o = align_up(o + (int64_t)batch * n_tables * 4, 256);                        // local offsets
  o = align_up(o + (int64_t)batch * nb * 4, 256);                              // block totals
  o = align_up(o + (int64_t)batch * n_tables * DDZ_SLAB_MIN_STRIDE * 16, 256); // rows
  if (want_ids) o = align_up(o + (int64_t)batch * n_tables * DDZ_SLAB_MIN_STRIDE * 4, 256);
  return o;
}

int ddz_rollout_random_csr_staged(ddz_env_t* e, int64_t n_iters, int batch, void* staging, int64_t staging_bytes, int32_t* offsets,
                                  int8_t* rows, int32_t* ids, int64_t cap, uint8_t* traj, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(rows, 16) || !al(offsets, 4) || !al(ids, 4) || !al(traj, 16) || !al(staging, 256)) return DDZ_EINVAL;
  if (n_iters < 0 || batch < 1 || !staging || !offsets || !rows || cap < 0) return DDZ_EINVAL;
  if (cap > 0x7FFFFFFF || (int64_t)batch * e->T * DDZ_SLAB_MIN_STRIDE > 0x7FFFFFFF) return DDZ_ECAP;
  if (staging_bytes < ddz_rollout_csr_staging_bytes(e->T, batch, ids != nullptr)) return DDZ_ECAP;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  hipStream_t st = (hipStream_t)stream;
  const int64_t T = e->T, nb = (T + CSR_BT - 1) / CSR_BT, S = DDZ_SLAB_MIN_STRIDE;
  uint8_t* p = (uint8_t*)staging;
  int64_t o = 0;
  int32_t* s_counts = (int32_t*)(p + o);   o = align_up(o + (int64_t)batch * T * 4, 256);
  int32_t* s_local = (int32_t*)(p + o);    o = align_up(o + (int64_t)batch * T * 4, 256);
  int32_t* s_blk = (int32_t*)(p + o);      o = align_up(o + (int64_t)batch * nb * 4, 256);
  int8_t* s_rows = (int8_t*)(p + o);       o = align_up(o + (int64_t)batch * T * S * 16, 256);
  int32_t* s_ids = ids ? (int32_t*)(p + o) : nullptr;
  const CsrBatch bt0{T, T * S, T, nb, 0};
  auto copy = [&](int first, int count) {
    CsrBatch bt = bt0;
    bt.first = first;
    const dim3 grid((unsigned)nb, (unsigned)count);
    if (ids)
      hipLaunchKernelGGL((k_csr_copy<true>), grid, dim3(CSR_BT), 0, st, (const int32_t*)s_counts, (const uint4*)s_rows, (const int32_t*)s_ids,
                         T, S, (const int32_t*)s_local, (const int32_t*)s_blk, offsets, (uint4*)rows, ids, cap, e->sc.status, bt);
    else
      hipLaunchKernelGGL((k_csr_copy<false>), grid, dim3(CSR_BT), 0, st, (const int32_t*)s_counts, (const uint4*)s_rows, (const int32_t*)s_ids,
                         T, S, (const int32_t*)s_local, (const int32_t*)s_blk, offsets, (uint4*)rows, ids, cap, e->sc.status, bt);
  };
  for (int64_t done = 0; done < n_iters; done += batch) {
    const int b = (int)(n_iters - done < batch ? n_iters - done : batch);
    int rc = launch_rollout(e, b, s_counts, s_rows, s_ids, S, traj ? traj + done * T * DDZ_TRAJ_BYTES : nullptr, st, T * S, T);
    if (rc) return rc;
    hipLaunchKernelGGL(k_csr_scan, dim3((unsigned)nb, (unsigned)b), dim3(CSR_BT), 0, st, (const int32_t*)s_counts, T, S, s_local, s_blk, bt0);
    const bool last = done + b >= n_iters;
    if (!last) copy(0, b);
    else {
      if (b > 1) copy(0, b - 1);
      copy(b - 1, 1);             // the call's last iteration: alone, behind everything else -- its lists are what stays
    }
    rc = check_launch();
    if (rc) return rc;
  }
  e->legal_cap = cap;
  return DDZ_OK;
}

int ddz_rollout_random_timed(ddz_env_t* e, int64_t n_iters, int32_t* counts, int8_t* rows, int32_t* ids,
                             int64_t stride, double* ms, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(rows, 16) || !al(counts, 4) || !al(ids, 4)) return DDZ_EINVAL;
  if (n_iters <= 0 || !counts || !rows || stride < DDZ_SLAB_MIN_STRIDE || !ms) return DDZ_EINVAL;
  if (e->T * stride > 0x7FFFFFFF) return DDZ_ECAP;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t ev[2];
  if (hipEventCreate(&ev[0]) != hipSuccess || hipEventCreate(&ev[1]) != hipSuccess) return hip_fail(hipGetLastError());
  (void)hipEventRecord(ev[0], st);
  int rc = launch_rollout(e, n_iters, counts, rows, ids, stride, nullptr, st);
  (void)hipEventRecord(ev[1], st);
  hipError_t r = hipStreamSynchronize(st);
  if (rc == DDZ_OK && r != hipSuccess) rc = hip_fail(r);
  if (rc == DDZ_OK) {
    float x = 0;
    (void)hipEventElapsedTime(&x, ev[0], ev[1]);
    ms[0] = x;  // duration of the one k_rollout launch that ran the n_iters iterations
    ms[1] = (double)n_iters;
  }
  (void)hipEventDestroy(ev[0]);
  (void)hipEventDestroy(ev[1]);
  return rc;
}

int ddz_auto_choose_state(ddz_env_t* e, int auto_roles, int32_t* ids, int64_t* stats, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!ids || !al(ids, 4) || !al(stats, 8) || auto_roles < 0 || auto_roles > 7) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  AutoArgs a{};
  a.state = e->state; a.T = e->T; a.tpw = 1; a.auto_roles = auto_roles; a.ids = ids; a.stats = stats;
  a.status = e->sc.status; a.teams = e->auto_teams != 0; a.team_first = e->auto_teams == 1;
  fill_round_penalty(a);
  return launch_auto_ordered(e, a, (hipStream_t)stream);
}

// test hook: ddz_auto_choose_state with an explicit kernel -- 1 = k_auto (sequential walk, wave-uniform control, full
// enumeration: the cross-check), 2 = k_auto2 as the product runs it (heaviest hands first), 3 = k_auto2 in table order.
// Same ids by construction; tests compare them.
int ddz_debug_auto_choose_state(ddz_env_t* e, int kernel, int auto_roles, int32_t* ids, int64_t* stats, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!ids || !al(ids, 4) || !al(stats, 8) || auto_roles < 0 || auto_roles > 7) return DDZ_EINVAL;
  if (kernel != AUTO_K_SEQUENTIAL && kernel != AUTO_K_LANES && kernel != AUTO_K_LANES_TABLE_ORDER) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  AutoArgs a{};
  a.state = e->state; a.T = e->T; a.tpw = 1; a.auto_roles = auto_roles; a.ids = ids; a.stats = stats;
  a.status = e->sc.status; a.teams = e->auto_teams != 0; a.team_first = e->auto_teams == 1;
  fill_round_penalty(a);
  if (kernel == AUTO_K_LANES) return launch_auto_ordered(e, a, (hipStream_t)stream);
  return launch_auto<true>(e->device, a, (hipStream_t)stream, kernel == AUTO_K_SEQUENTIAL ? AUTO_K_SEQUENTIAL : AUTO_K_LANES);
}

// test hook: the launch geometry of a handle's table kernels -- tables per wave (1..64, 0 = keep) and whether k_slab runs
// the one-table-per-wave block-cooperative form (0 / 1, 2 = without the block-written heavy lists, n > 2 = those from n scan rounds on (default 5), -1 = keep).  Results never depend on either (tests sweep them).
int ddz_debug_set_geometry(ddz_env_t* e, int tables_per_wave, int slab_coop, int slab_work_list) {
  if (!good(e)) return DDZ_EHANDLE;
  if (tables_per_wave < 0 || tables_per_wave > 64 || slab_coop < -1 || slab_coop > TEAM_ROUNDS || slab_work_list < -1 || slab_work_list > 1)
    return DDZ_EINVAL;
  if (tables_per_wave > 0) {
    e->tpw = tables_per_wave;
    e->nblocks = (e->T + (int64_t)WPB * e->tpw - 1) / ((int64_t)WPB * e->tpw);
    e->counts_valid = false;  // the scan buffers depend on the geometry
    e->slab_lpt = e->tpw >= 2;
    e->rollout_waves = WPB;   // (an explicit geometry also holds for the rollout: 16-wave blocks, this many tables per wave)
  }
  e->slab_coop = e->tpw == 1 && (slab_coop < 0 ? e->slab_coop || tables_per_wave > 0 : slab_coop) ? 1 : 0;
  if (slab_coop >= 0) e->slab_team = slab_coop == 2 ? 0 : slab_coop > 2 ? slab_coop : 5;
  if (slab_work_list >= 0) e->slab_lpt = slab_work_list && e->tpw >= 2;
  return DDZ_OK;
}

int ddz_debug_set_auto_teams(ddz_env_t* e, int on) {
  if (!good(e)) return DDZ_EHANDLE;
  if (on < 0 || on > 2) return DDZ_EINVAL;   // 0 off, 1 teams + team-first for the heaviest decisions (default), 2 teams at the queue's end only
  e->auto_teams = on;
  return DDZ_OK;
}

int ddz_auto_choose(int device, const int8_t* hands, const int8_t* lasts, const uint8_t* info, int64_t n, int32_t* ids,
                    int64_t* stats, void* stream) {
  if (!al(hands, 16) || !al(lasts, 16) || !al(info, 4) || !al(ids, 4) || !al(stats, 8)) return DDZ_EINVAL;
  if (n <= 0 || n > ((int64_t)1 << 30) || !hands || !lasts || !info || !ids) return DDZ_EINVAL;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  int rc = ensure_table(device);
  if (rc) return rc;
  AutoArgs a{};
  a.hands = (const uint4*)hands; a.lasts = (const uint4*)lasts; a.info = (const uint32_t*)info;
  a.T = n; a.tpw = 1; a.ids = ids; a.stats = stats; a.status = g_device_status_ptr[device]; a.teams = 1;
  fill_round_penalty(a);
  return launch_auto<false>(device, a, (hipStream_t)stream);
}

int ddz_debug_cards_value(int device, int8_t* out, void* stream) {
  if (!out) return DDZ_EINVAL;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  int rc = ensure_table(device);
  if (rc) return rc;
  hipLaunchKernelGGL(k_cards_value, dim3((DDZ_NUM_ACTIONS + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)stream, out);
  return check_launch();
}

int ddz_debug_auto_leaf(int device, const int32_t* in, const double* rp, int64_t n, double* value, int32_t* move, void* stream) {
  if (!in || !rp || !value || !move || n <= 0 || !al(in, 16) || !al(rp, 8) || !al(value, 8) || !al(move, 4)) return DDZ_EINVAL;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  hipLaunchKernelGGL(k_debug_leaf, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     (const int4*)in, rp, n, value, move);
  return check_launch();
}

int ddz_status(ddz_env_t* e, int32_t* out, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!out) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  hipError_t r = hipMemcpyAsync(out, e->sc.status, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (r != hipSuccess) return hip_fail(r);
  r = hipStreamSynchronize((hipStream_t)stream);
  return r == hipSuccess ? DDZ_OK : hip_fail(r);
}

int ddz_sync(int device, void* stream) {
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  const hipError_t r = hipStreamSynchronize((hipStream_t)stream);
  return r == hipSuccess ? DDZ_OK : hip_fail(r);
}

int ddz_device_status(int device, int32_t* out, void* stream) {
  if (!out) return DDZ_EINVAL;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  int rc = ensure_table(device);
  if (rc) return rc;
  hipError_t r = hipMemcpyAsync(out, g_device_status_ptr[device], 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (r != hipSuccess) return hip_fail(r);
  r = hipMemsetAsync(g_device_status_ptr[device], 0, 4, (hipStream_t)stream);
  if (r != hipSuccess) return hip_fail(r);
  r = hipStreamSynchronize((hipStream_t)stream);
  return r == hipSuccess ? DDZ_OK : hip_fail(r);
}

#ifdef DDZ_STAMP
int ddz_debug_set_stamps(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : DDZ_EHIP;
}
#endif

int ddz_select(ddz_env_t* e, const float* q, const int32_t* offsets, double epsilon, int32_t* choice, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(q, 4) || !al(offsets, 4) || !al(choice, 4)) return DDZ_EINVAL;
  if (!q || !offsets || !choice || !(epsilon >= 0.0) || epsilon > 1.0) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  const uint64_t thr = (uint64_t)(epsilon * 4294967296.0);
  hipLaunchKernelGGL(k_select, dim3((unsigned)((e->T * SEL_G + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     (const uint8_t*)e->state, e->T, (uint32_t)e->seed, (uint32_t)(e->seed >> 32), e->gid_base, q, offsets,
                     thr, choice, (const int32_t*)nullptr, (int64_t)0);
  return check_launch();
}

int ddz_select_slab(ddz_env_t* e, const float* q, const int32_t* counts, int64_t stride, double epsilon, int32_t* choice,
                    void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(q, 4) || !al(counts, 4) || !al(choice, 4)) return DDZ_EINVAL;
  if (!q || !counts || stride <= 0 || !choice || !(epsilon >= 0.0) || epsilon > 1.0) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  const uint64_t thr = (uint64_t)(epsilon * 4294967296.0);
  hipLaunchKernelGGL(k_select, dim3((unsigned)((e->T * SEL_G + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     (const uint8_t*)e->state, e->T, (uint32_t)e->seed, (uint32_t)(e->seed >> 32), e->gid_base, q,
                     (const int32_t*)nullptr, thr, choice, counts, stride);
  return check_launch();
}

int ddz_q_features(int device, const float* face, int64_t n_tables, int planes, const float* wf, const float* bias,
                   const float* acnt, float* y, int64_t y_row_stride, void* stream) {
  return launch_q_features<false>(device, face, n_tables, planes, wf, bias, acnt, y, y_row_stride, nullptr, QRow0{}, stream);
}

int ddz_q_features_packed(int device, const float* face, int64_t n_tables, int planes, const float* wf, const float* bias,
                          const float* acnt, const int32_t* row_index, const int64_t* rank_row0, int64_t n_rows, float* y,
                          int64_t y_row_stride, void* stream) {
  QRow0 row0;
  if (!row_index || !al(row_index, 16) || !q_row0(rank_row0, n_rows, n_tables, row0)) return DDZ_EINVAL;
  if (y_row_stride > 0 && n_rows > (((int64_t)1 << 31) - 1) / y_row_stride) return DDZ_ECAP;  // k_q_feat indexes y with 32 bits
  return launch_q_features<true>(device, face, n_tables, planes, wf, bias, acnt, y, y_row_stride, row_index, row0, stream);
}

int ddz_q_slab(ddz_env_t* e, const float* u, const float* z, int64_t hidden, const float* w2, const float* b2,
               const int32_t* counts, const int8_t* rows, int64_t stride, float* q, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(u, 16) || !al(z, 16) || !al(w2, 16) || !al(b2, 4) || !al(counts, 4) || !al(rows, 16) || !al(q, 4)) return DDZ_EINVAL;
  if (!u || !z || !w2 || !b2 || !counts || !rows || !q || hidden != QH || stride < 1) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  // one table per wave up to 16 waves per CU, then consecutive tables per wave (as the stepping kernels)
  int64_t v = (e->T + 4095) / 4096;
  const int tpw = (int)(v < 1 ? 1 : v > 8 ? 8 : v);
  const int64_t per_block = (int64_t)WPB * tpw;
  hipLaunchKernelGGL(k_q_slab<false>, dim3((unsigned)((e->T + per_block - 1) / per_block)), dim3(TB), 0, (hipStream_t)stream,
                     (const float4*)u, (const float4*)z, e->T, tpw, (const float4*)w2, b2, counts, (const uint4*)rows, stride, q,
                     (const int32_t*)nullptr, QRow0{}, (const float4*)nullptr, e->sc.status);
  return check_launch();
}

int ddz_q_slab_packed(ddz_env_t* e, const float* u, const int32_t* row_index, const int64_t* rank_row0, int64_t n_rows,
                      const float* table_term, const float* z, int64_t hidden, const float* w2, const float* b2,
                      const int32_t* counts, const int8_t* rows, int64_t stride, float* q, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(u, 16) || !al(z, 16) || !al(w2, 16) || !al(b2, 4) || !al(counts, 4) || !al(rows, 16) || !al(q, 4) || !al(row_index, 4) ||
      !al(table_term, 16))
    return DDZ_EINVAL;
  if (!u || !z || !w2 || !b2 || !counts || !rows || !q || !row_index || hidden != QH || stride < 1) return DDZ_EINVAL;
  QRow0 row0;
  if (!q_row0(rank_row0, n_rows, e->T, row0)) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  int64_t v = (e->T + 4095) / 4096;
  const int tpw = (int)(v < 1 ? 1 : v > 8 ? 8 : v);
  const int64_t per_block = (int64_t)WPB * tpw;
  hipLaunchKernelGGL(k_q_slab<true>, dim3((unsigned)((e->T + per_block - 1) / per_block)), dim3(TB), 0, (hipStream_t)stream,
                     (const float4*)u, (const float4*)z, e->T, tpw, (const float4*)w2, b2, counts, (const uint4*)rows, stride, q,
                     row_index, row0, (const float4*)table_term, e->sc.status);
  return check_launch();
}

// ---- the "needed rows" form of the ragged Q forward (ddz_qnet.h) ----
int ddz_q_fc1_tile_rows(void) { return FC_M; }

int64_t ddz_q_need_scratch_bytes(int64_t n_tables) {
  if (n_tables <= 0) return 0;
  const int64_t nblk = (n_tables + QN_TPB - 1) / QN_TPB;
  return align_up(8 * n_tables, 256) + align_up(nblk * 16 * 4, 256);
}

int ddz_q_need(ddz_env_t* e, const int32_t* counts, const int8_t* rows, int64_t stride, int64_t row_capacity, void* scratch,
               int64_t scratch_bytes, int32_t* row_index, int32_t* seg, uint8_t* row_cnt, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!counts || !rows || !scratch || !row_index || !seg || !row_cnt || stride < 1) return DDZ_EINVAL;
  if (!al(counts, 4) || !al(rows, 16) || !al(scratch, 256) || !al(row_index, 16) || !al(seg, 4)) return DDZ_EINVAL;
  if (row_capacity < 15 * FC_M || row_capacity % FC_M || row_capacity > (((int64_t)1 << 31) - 1) / QH) return DDZ_EINVAL;
  if (scratch_bytes < ddz_q_need_scratch_bytes(e->T)) return DDZ_ECAP;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nblk = (e->T + QN_TPB - 1) / QN_TPB;
  uint64_t* need = (uint64_t*)scratch;
  int32_t* blk = (int32_t*)((uint8_t*)scratch + align_up(8 * e->T, 256));
  hipLaunchKernelGGL(k_q_need_mask, dim3((unsigned)nblk), dim3(256), 0, st, counts, (const uint4*)rows, stride, e->T, need, blk);
  hipLaunchKernelGGL(k_q_need_scan, dim3(1), dim3(256), 0, st, blk, nblk, seg, row_capacity);
  hipLaunchKernelGGL(k_q_need_assign, dim3((unsigned)nblk), dim3(256), 0, st, (const uint64_t*)need, e->T, (const int32_t*)blk,
                     (const int32_t*)seg, row_index, row_cnt, e->sc.status);
  return check_launch();
}

int ddz_q_features_needed(int device, const float* face, int64_t n_tables, int planes, const float* wf, const float* bias,
                          const float* acnt, const int32_t* row_index, float* y0, float* dy, int64_t row_capacity, void* stream) {
  if (!face || !wf || !bias || !acnt || !row_index || !dy || n_tables <= 0) return DDZ_EINVAL;   // (y0 may be null: section 5)
  if (!al(face, 16) || !al(wf, 4) || !al(bias, 4) || !al(acnt, 4) || !al(row_index, 16) || !al(y0, 4) || !al(dy, 4)) return DDZ_EINVAL;
  if (row_capacity < 1 || row_capacity > (((int64_t)1 << 31) - 1) / QH || n_tables > ((int64_t)1 << 30)) return DDZ_ECAP;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  const dim3 grid((unsigned)((n_tables + QF_TILE - 1) / QF_TILE)), block(QH);
  hipStream_t st = (hipStream_t)stream;
  const float4* f = (const float4*)face;
  switch (planes) {
    case 4: hipLaunchKernelGGL(k_q_feat_needed<4>, grid, block, 0, st, f, n_tables, wf, bias, acnt, y0, dy, row_capacity, row_index); break;
    case 6: hipLaunchKernelGGL(k_q_feat_needed<6>, grid, block, 0, st, f, n_tables, wf, bias, acnt, y0, dy, row_capacity, row_index); break;
    case 7: hipLaunchKernelGGL(k_q_feat_needed<7>, grid, block, 0, st, f, n_tables, wf, bias, acnt, y0, dy, row_capacity, row_index); break;
    case 9: hipLaunchKernelGGL(k_q_feat_needed<9>, grid, block, 0, st, f, n_tables, wf, bias, acnt, y0, dy, row_capacity, row_index); break;
    default: return DDZ_EINVAL;
  }
  return check_launch();
}

int ddz_q_fc1_dense(int device, const float* a, int64_t n_rows, int64_t k, const float* w, float* c, void* stream) {
  if (!a || !w || !c || n_rows <= 0 || k < FC_K || k % FC_K || k > (1 << 20)) return DDZ_EINVAL;
  if (!al(a, 16) || !al(w, 16) || !al(c, 4)) return DDZ_EINVAL;
  if (n_rows > ((int64_t)1 << 31) - FC_M) return DDZ_ECAP;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  hipLaunchKernelGGL(k_fc1<false>, dim3((unsigned)((n_rows + FC_M - 1) / FC_M)), dim3(FC_THREADS), 0, (hipStream_t)stream, a, k, w, c,
                     n_rows, (int)k, (const int32_t*)nullptr, (const float*)nullptr, (const uint8_t*)nullptr);
  return check_launch();
}

int ddz_q_fc1_rows(int device, const float* dy, const int32_t* seg, const uint8_t* row_cnt, const float* w2, const float* z, float* d,
                   int64_t row_capacity, void* stream) {
  if (!dy || !seg || !w2 || !d || ((z == nullptr) != (row_cnt == nullptr))) return DDZ_EINVAL;   // (z and row_cnt both null: no fold)
  if (!al(dy, 16) || !al(w2, 16) || !al(d, 4) || !al(seg, 4) || !al(z, 4)) return DDZ_EINVAL;
  if (row_capacity < FC_M || row_capacity % FC_M || row_capacity > (((int64_t)1 << 31) - 1) / QH) return DDZ_EINVAL;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  hipLaunchKernelGGL(k_fc1<true>, dim3((unsigned)(row_capacity / FC_M)), dim3(FC_THREADS), 0, (hipStream_t)stream, dy, (int64_t)QH, w2, d,
                     (int64_t)0, QH, seg, z, row_cnt);
  return check_launch();
}

// the shared-rows form of H0 (ddz_qnet.h section 5)
int64_t ddz_q_shared_ws_bytes(void) { return (int64_t)QSH_WS_INTS * 4; }
int ddz_q_shared_rows(ddz_env_t* e, void* ws, int64_t ws_bytes, int64_t row_capacity, int32_t* rows, int32_t* rep, int32_t* seg,
                      void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!ws || !rows || !rep || !seg || !al(ws, 16) || !al(rows, 16) || !al(rep, 4) || !al(seg, 4)) return DDZ_EINVAL;
  if (ws_bytes < ddz_q_shared_ws_bytes() || row_capacity % FC_M || e->T > ((int64_t)1 << 26)) return DDZ_ECAP;
  const int64_t most = e->T * 15 < (int64_t)QSH_KEYS ? e->T * 15 : (int64_t)QSH_KEYS;   // distinct (rank, column) pairs at most
  if (row_capacity < most + 15 * FC_M || row_capacity > (((int64_t)1 << 31) - 1) / QH) return DDZ_ECAP;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  hipStream_t st = (hipStream_t)stream;
  int32_t* slots = (int32_t*)ws;
  int32_t* cnt = slots + QSH_KEYS;
  int32_t* base = cnt + 15 * QSH_CPR;
  if (hipMemsetAsync(slots, 0, (size_t)QSH_KEYS * 4, st) != hipSuccess) return DDZ_EHIP;
  if (hipMemsetAsync(rep, 0xFF, (size_t)row_capacity * 4, st) != hipSuccess) return DDZ_EHIP;
  const unsigned nb = (unsigned)((e->T * 16 + 255) / 256);
  hipLaunchKernelGGL(k_qs_mark, dim3(nb), dim3(256), 0, st, (const uint8_t*)e->state, e->T, slots, rows);
  hipLaunchKernelGGL(k_qs_count, dim3(15 * QSH_CPR), dim3(256), 0, st, (const int32_t*)slots, cnt);
  hipLaunchKernelGGL(k_qs_seg, dim3(1), dim3(64), 0, st, (const int32_t*)cnt, base, seg, (int32_t)row_capacity);
  hipLaunchKernelGGL(k_qs_assign, dim3(15 * QSH_CPR), dim3(256), 0, st, slots, (const int32_t*)base, rep, (int32_t)row_capacity);
  hipLaunchKernelGGL(k_qs_rows, dim3(nb), dim3(256), 0, st, (const int32_t*)slots, e->T, rows);
  return check_launch();
}
int ddz_q_features_rows(int device, const float* face, int64_t n_tables, int planes, const float* wf, const float* bias,
                        const int32_t* rep, const int32_t* seg, float* ys, int64_t ys_ld, int64_t row_capacity, const float* mz, float* g,
                        void* stream) {
  if (!face || !wf || !bias || !rep || !seg || !ys || n_tables <= 0 || ((mz == nullptr) != (g == nullptr))) return DDZ_EINVAL;
  if (ys_ld != QH && ys_ld != QH + 32) return DDZ_EINVAL;
  if (!al(mz, 4) || !al(g, 4)) return DDZ_EINVAL;
  if (!al(face, 16) || !al(wf, 4) || !al(bias, 4) || !al(rep, 4) || !al(seg, 4) || !al(ys, 4)) return DDZ_EINVAL;
  if (row_capacity < QR_TILE || row_capacity % QR_TILE || row_capacity > (((int64_t)1 << 31) - 1) / (QH + 32) || n_tables > ((int64_t)1 << 26))
    return DDZ_ECAP;
  if (planes != 6) return DDZ_EINVAL;   // EnvCooperationSimplify's six planes: the only face whose columns ddz_q_shared_rows keys
  DeviceGuard gd(device);
  if (!gd.ok) return DDZ_ENODEV;
  hipLaunchKernelGGL(k_q_feat_rows<6>, dim3((unsigned)(row_capacity / QR_TILE)), dim3(QH), 0, (hipStream_t)stream, (const float4*)face,
                     n_tables, wf, bias, rep, seg, ys, (int)ys_ld, mz, g);
  return check_launch();
}
// the needed rows D shared as well (ddz_qnet.h section 6)
int64_t ddz_q_shared_need_ws_bytes(int64_t shared_row_capacity) {
  if (shared_row_capacity <= 0 || shared_row_capacity % FC_M) return DDZ_EINVAL;
  return (shared_row_capacity * 4 + 2 * (shared_row_capacity / FC_M)) * 4;   // dslot | cnt[tiles] | base[tiles]
}
int ddz_q_shared_need(ddz_env_t* e, const int32_t* row_index, const int32_t* rows, const int32_t* sseg, int64_t shared_row_capacity,
                      void* ws, int64_t ws_bytes, int64_t row_capacity, int32_t* row_index2, int32_t* drep, int32_t* dseg,
                      uint8_t* row_cnt, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!row_index || !rows || !sseg || !ws || !row_index2 || !drep || !dseg || !row_cnt) return DDZ_EINVAL;
  if (!al(row_index, 4) || !al(rows, 4) || !al(sseg, 4) || !al(ws, 16) || !al(row_index2, 4) || !al(drep, 4) || !al(dseg, 4)) return DDZ_EINVAL;
  if (shared_row_capacity <= 0 || shared_row_capacity % FC_M || shared_row_capacity > ((int64_t)1 << 28)) return DDZ_ECAP;
  if (ws_bytes < ddz_q_shared_need_ws_bytes(shared_row_capacity)) return DDZ_ECAP;
  if (row_capacity < 15 * FC_M || row_capacity % FC_M || row_capacity > (((int64_t)1 << 31) - 1) / QH) return DDZ_ECAP;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  hipStream_t st = (hipStream_t)stream;
  int32_t* dslot = (int32_t*)ws;
  const int64_t tiles = shared_row_capacity / FC_M;
  int32_t* cnt = dslot + shared_row_capacity * 4;
  int32_t* base = cnt + tiles;
  if (hipMemsetAsync(dslot, 0, (size_t)shared_row_capacity * 16, st) != hipSuccess) return DDZ_EHIP;
  if (hipMemsetAsync(drep, 0xFF, (size_t)row_capacity * 4, st) != hipSuccess) return DDZ_EHIP;
  const unsigned nb = (unsigned)((e->T * QP_COLS + 255) / 256);
  hipLaunchKernelGGL(k_qd_mark, dim3(nb), dim3(256), 0, st, row_index, rows, e->T, dslot, shared_row_capacity);
  hipLaunchKernelGGL(k_qd_count, dim3((unsigned)tiles), dim3(256), 0, st, (const int32_t*)dslot, sseg, cnt);
  hipLaunchKernelGGL(k_qd_seg, dim3(1), dim3(64), 0, st, (const int32_t*)cnt, sseg, base, dseg, (int32_t)row_capacity, e->sc.status);
  hipLaunchKernelGGL(k_qd_assign, dim3((unsigned)tiles), dim3(256), 0, st, dslot, sseg, (const int32_t*)base, drep, row_cnt,
                     (int32_t)row_capacity);
  hipLaunchKernelGGL(k_qd_remap, dim3(nb), dim3(256), 0, st, row_index, rows, e->T, (const int32_t*)dslot, shared_row_capacity, row_index2);
  return check_launch();
}
int ddz_q_features_drows(int device, const float* face, int64_t n_tables, int planes, const float* wf, const float* bias,
                         const float* acnt, const int32_t* rep, int64_t shared_row_capacity, const int32_t* drep, const int32_t* dseg,
                         float* dy, int64_t row_capacity, void* stream) {
  if (!face || !wf || !bias || !acnt || !rep || !drep || !dseg || !dy || n_tables <= 0 || shared_row_capacity <= 0) return DDZ_EINVAL;
  if (!al(face, 16) || !al(wf, 4) || !al(bias, 4) || !al(acnt, 4) || !al(rep, 4) || !al(drep, 4) || !al(dseg, 4) || !al(dy, 4)) return DDZ_EINVAL;
  if (row_capacity < QR_TILE || row_capacity % QR_TILE || row_capacity > (((int64_t)1 << 31) - 1) / QH || n_tables > ((int64_t)1 << 26))
    return DDZ_ECAP;
  if (planes != 6) return DDZ_EINVAL;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  hipLaunchKernelGGL(k_q_feat_drows<6>, dim3((unsigned)(row_capacity / QR_TILE)), dim3(QH), 0, (hipStream_t)stream, (const float4*)face,
                     n_tables, wf, bias, acnt, rep, shared_row_capacity, drep, dseg, dy);
  return check_launch();
}
int ddz_q_gather_h0(int device, const float* g, int64_t g_rows, const int32_t* rows, int64_t n_tables, const float* base, float* h0,
                    void* stream) {
  if (!g || !rows || !h0 || n_tables <= 0 || g_rows <= 0) return DDZ_EINVAL;
  if (!al(g, 16) || !al(rows, 16) || !al(h0, 16) || !al(base, 16)) return DDZ_EINVAL;
  DeviceGuard gd(device);
  if (!gd.ok) return DDZ_ENODEV;
  hipLaunchKernelGGL(k_qs_gather, dim3((unsigned)((n_tables + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)g, g_rows,
                     rows, n_tables, (float4*)h0, (const float4*)base);
  return check_launch();
}

int ddz_q_fc1_rows_k(int device, const float* y, int64_t k, const int32_t* seg, const float* w2k, float* g, int64_t row_capacity,
                      int accumulate, void* stream) {
  if (!y || !seg || !w2k || !g) return DDZ_EINVAL;
  if (!al(y, 16) || !al(w2k, 16) || !al(g, 4) || !al(seg, 4)) return DDZ_EINVAL;
  if (k < FC_K || k % FC_K || k > 4096) return DDZ_EINVAL;
  if (row_capacity < FC_M || row_capacity % FC_M || row_capacity > (((int64_t)1 << 31) - 1) / k) return DDZ_EINVAL;
  DeviceGuard gd(device);
  if (!gd.ok) return DDZ_ENODEV;
  hipLaunchKernelGGL(k_fc1<true>, dim3((unsigned)(row_capacity / FC_M)), dim3(FC_THREADS), 0, (hipStream_t)stream, y, k, w2k, g,
                     (int64_t)0, (int)k, seg, (const float*)nullptr, (const uint8_t*)nullptr, accumulate ? 1 : 0);
  return check_launch();
}

int ddz_q_slab_needed(ddz_env_t* e, const float* h0, const float* d, int64_t row_capacity, const int32_t* row_index,
                      int64_t hidden, const float* w2, const float* b2, const int32_t* counts, const int8_t* rows, int64_t stride,
                      float* q, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!al(h0, 16) || !al(d, 16) || !al(w2, 16) || !al(b2, 4) || !al(counts, 4) || !al(rows, 16) || !al(q, 4) || !al(row_index, 4))
    return DDZ_EINVAL;
  if (!h0 || !d || !w2 || !b2 || !counts || !rows || !q || !row_index || hidden != QH || stride < 1 || row_capacity < 1) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  // (64 tables per block at most: the blocks are the unit the hardware balances over the CUs, the tables inside a block are
  // handed out by an LDS ticket)
  int64_t v = (e->T + 4095) / 4096;
  int tpw = (int)(v < 1 ? 1 : v > 4 ? 4 : v);
#ifdef DDZ_QS_TPW_ENV   // (a timing experiment: tables per wave of the row stage from the environment; tools/row_stage_probe.py)
  if (const char* s_ = getenv("DDZ_QS_TPW")) { const int x = atoi(s_); if (x >= 1 && x <= 8) tpw = x; }
#endif
  const int64_t per_block = (int64_t)WPB * tpw;
  hipLaunchKernelGGL(k_q_slab_needed, dim3((unsigned)((e->T + per_block - 1) / per_block)), dim3(TB), 0, (hipStream_t)stream,
                     (const float4*)h0, (const float4*)d, row_capacity, e->T, tpw, (const float4*)w2, b2, counts,
                     (const uint4*)rows, stride, q, row_index, e->sc.status);
  return check_launch();
}

int ddz_action_table(int device, int8_t* rows, void* stream) {
  if (!al(rows, 16)) return DDZ_EINVAL;
  if (!rows) return DDZ_EINVAL;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  int rc = ensure_table(device);
  if (rc) return rc;
  hipLaunchKernelGGL(k_export_table, dim3((NUM_ACTIONS_X + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)stream,
                     (uint4*)rows);
  return check_launch();
}

int ddz_pack_trajectory(int device, const uint8_t* traj, int64_t n_records, uint8_t* packed, void* stream) {
  if (!al(traj, 16) || !al(packed, 8)) return DDZ_EINVAL;
  if (n_records < 0 || (n_records > 0 && (!traj || !packed))) return DDZ_EINVAL;
  if (n_records == 0) return DDZ_OK;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  int rc = ensure_table(device);
  if (rc) return rc;
  hipLaunchKernelGGL(k_pack_traj, dim3((unsigned)((n_records + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     (const uint4*)traj, n_records, (uint2*)packed);
  return check_launch();
}

// debug/test entry: classify(rows) -> info words (category | value << 8 | len << 16, 0xFF invalid)
int ddz_debug_classify(int device, const int8_t* rows, int64_t n, uint32_t* out, void* stream) {
  if (!al(rows, 16) || !al(out, 4)) return DDZ_EINVAL;
  if (n <= 0 || !rows || !out) return DDZ_EINVAL;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  hipLaunchKernelGGL(k_classify, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     (const uint4*)rows, n, out);
  return check_launch();
}

}  // extern "C"
