// ddz_engine.hip -- gfx950 kernels + C ABI (include/ddz_env.h) of the batched Doudizhu engine.
//
// Kernel map (reference paths relative to /root/reference):
//   k_step    one thread per table: pick (engine RNG / index / row), apply the action
//             (envi.py:38-43 _update + native step_manual), terminal + reward
//             (rule_play.py:14), auto-reset deal (native prepare(), spec v1), then the
//             closed-form size of the next legal list and a block scan of those sizes.
//             Also serves reset (deal only) and count-only passes.
//   k_enum    one wavefront per table: the combo enumerator + follow filter
//             (r.get_moves, envi.py:111; rules card.py:34-159, :307-325,
//             utils.py:45-63), structurally per category, __ballot + mbcnt compaction
//             into the CSR row list, 16-byte coalesced row stores, canonical id order.
//   k_query   stateless (hand,last) queries -> same records k_enum consumes (r.get_moves).
//   k_observe the `face` tensors (envi.py:87-96,165-217), k_onehot batch_arr2onehot (:139-146).
// No table of the action space is read: ids/rows are computed from rank masks.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ddz_env.h"
#include "ddz_device.h"

using namespace ddz;

namespace {

constexpr int BLOCK = 256;  // threads per block for every kernel
constexpr int MODE_RESET = 3, MODE_COUNT = 4;

// ------------------------------------------------------------------------------------
// scratch layout (caller-owned, zero-filled at create)
struct Layout {
  int64_t T, nblk;
  int64_t off_q, off_counts, off_local, off_blk_tot, off_blk_stats, off_status, bytes;
};
__host__ __device__ inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
inline Layout make_layout(int64_t T) {
  Layout l;
  l.T = T;
  l.nblk = (T + BLOCK - 1) / BLOCK;
  int64_t o = 0;
  l.off_q = o;          o = align_up(o + T * 16, 256);
  l.off_counts = o;     o = align_up(o + T * 4, 256);
  l.off_local = o;      o = align_up(o + T * 4, 256);
  l.off_blk_tot = o;    o = align_up(o + l.nblk * 4, 256);
  l.off_blk_stats = o;  o = align_up(o + l.nblk * 32, 256);
  l.off_status = o;     o = align_up(o + 64, 256);
  l.bytes = o;
  return l;
}
struct Scratch {  // device pointers into the scratch buffer
  uint4* q;              // per table: {hand nib lo, hi, info|flags, 0}
  int32_t* counts;       // size of each table's legal list
  int32_t* local_off;    // exclusive scan of counts inside the table's 256-block
  int32_t* blk_tot;      // sum of counts per 256-block
  int64_t* blk_stats;    // [nblk][4] plies, episodes, lord wins, -
  int32_t* status;       // [0] status bits
  int64_t* legal_rows;   // running total of rows produced by k_enum
};
inline Scratch bind(void* scratch, const Layout& l) {
  uint8_t* p = (uint8_t*)scratch;
  Scratch s;
  s.q = (uint4*)(p + l.off_q);
  s.counts = (int32_t*)(p + l.off_counts);
  s.local_off = (int32_t*)(p + l.off_local);
  s.blk_tot = (int32_t*)(p + l.off_blk_tot);
  s.blk_stats = (int64_t*)(p + l.off_blk_stats);
  s.status = (int32_t*)(p + l.off_status);
  s.legal_rows = (int64_t*)(p + l.off_status + 16);
  return s;
}

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

// ------------------------------------------------------------------------------------
// wave-cooperative emission: lanes holding a legal candidate append their row at
// base + n + (number of legal lanes below), in lane order.
struct Out {
  uint4* rows;
  int32_t* ids;
  int64_t base, cap;
};

template <bool IDS>
__device__ __forceinline__ int emit(bool legal, uint64_t nib, int cat, int id, const Out& o, int n) {
  const uint64_t b = __ballot(legal);
  if (legal) {
    const int pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32),
                                              __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
    const int64_t pos = o.base + n + pre;
    if (pos < o.cap) {
      o.rows[pos] = unpack_row(nib, (uint32_t)cat);
      if (IDS) o.ids[pos] = id;
    }
  }
  return n + __popcll(b);
}

__device__ __forceinline__ int binom_sel(int a, int b) {  // a uniform, b per lane (-1..4)
  const int c2 = a * (a - 1) / 2, c3 = c2 * (a - 2) / 3, c4 = c3 * (a - 3) / 4;
  int r = b == 0 ? 1 : b == 1 ? a : b == 2 ? c2 : b == 3 ? c3 : b == 4 ? c4 : 0;
  return a < b ? 0 : r;
}

// every L-subset K of the ranks in `av`, in lexicographic order, as
// row = mainnib + mult * K; id = idbase + lex rank of K among the L-subsets of `rm`
// (itertools.combinations(remains, L), card.py:115,128,141,152).
template <bool IDS>
__device__ int emit_combos(uint32_t av, uint32_t rm, int L, uint64_t mainnib, int mult, int cat,
                           int idbase, bool skipj, int lane, const Out& o, int n) {
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
    n = emit<IDS>(legal, mainnib + (uint64_t)mult * spread15(K), cat, idbase + lex, o, n);
  }
  return n;
}

// The enumerator for one table, executed by one wavefront.  `hand`, `info` are
// wave-uniform.  Returns the number of rows (uniform).  Emission order == ascending
// canonical id == index order of card.py:get_action_space().
template <bool IDS>
__device__ int enumerate_table(uint64_t hand, uint32_t info, uint32_t lut, int lane, const Out& o) {
  if (hand == 0 || (info & (QF_FROZEN | QF_BADLAST))) return 0;
  const Follow f = follow_of(info);
  // rank masks by ballot: lane r < 15 looks at rank r
  const int cnt = lane < 15 ? (int)((hand >> (4 * (lane & 15))) & 15) : 0;
  const uint32_t m1 = (uint32_t)__ballot(cnt >= 1) & M15;
  const uint32_t m2 = (uint32_t)__ballot(cnt >= 2) & M13;
  const uint32_t m3 = (uint32_t)__ballot(cnt >= 3) & M13;
  const uint32_t m4 = (uint32_t)__ballot(cnt >= 4) & M13;
  int n = 0;
  if (!f.lead && f.lc == BIGBANG) return emit<IDS>(lane == 0, 0, EMPTY, 0, o, n);  // card.py:312-313

  {  // ids 0..54: pass, singles, pairs, triples, bombs -- one lane each
    const int g = lane == 0 ? 0 : lane < 16 ? 1 : lane < 29 ? 2 : lane < 42 ? 3 : lane < 55 ? 4 : 5;
    const int r = lane - (g == 0 ? 0 : g == 1 ? 1 : g == 2 ? 16 : g == 3 ? 29 : 42);
    const uint32_t mk = g == 1   ? (m1 & value_gate(f, SINGLE))
                        : g == 2 ? (m2 & value_gate(f, DOUBLE))
                        : g == 3 ? (m3 & value_gate(f, TRIPLE))
                        : g == 4 ? (m4 & value_gate(f, QUADRIC))
                                 : 0u;
    const bool legal = g == 0 ? !f.lead : ((mk >> (r & 15)) & 1u);
    n = emit<IDS>(legal, (uint64_t)(g & 7) << (4 * (r & 15)), g, lane, o, n);
  }
  if (f.lead || f.lc == THREE_ONE) {  // card.py:69-73, four mains per round
    const uint32_t mains = m3 & value_gate(f, THREE_ONE);
    for (int it = 0; it < 4; ++it) {
      if (((mains >> (4 * it)) & 15u) == 0) continue;
      const int q = lane / 15, k = lane - 15 * q, main = 4 * it + q;
      const bool legal = lane < 60 && main < 13 && ((mains >> main) & 1u) && k != main && ((m1 >> k) & 1u);
      n = emit<IDS>(legal, (3ull << (4 * (main & 15))) + (1ull << (4 * k)), THREE_ONE,
                    ID_THREE_ONE + main * 14 + (k < main ? k : k - 1), o, n);
    }
  }
  if (f.lead || f.lc == THREE_TWO) {  // card.py:78-82
    const uint32_t mains = m3 & value_gate(f, THREE_TWO);
    for (int it = 0; it < 4; ++it) {
      if (((mains >> (4 * it)) & 15u) == 0) continue;
      const int q = lane / 13, k = lane - 13 * q, main = 4 * it + q;
      const bool legal = lane < 52 && main < 13 && ((mains >> main) & 1u) && k != main && ((m2 >> k) & 1u);
      n = emit<IDS>(legal, (3ull << (4 * (main & 15))) + (2ull << (4 * k)), THREE_TWO,
                    ID_THREE_TWO + main * 12 + (k < main ? k : k - 1), o, n);
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
    n = emit<IDS>(legal, nib, cat, idbase + lane, o, n);
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
          n = emit_combos<IDS>(kick & ranks & ~run, ranks & ~run, L,
                               ((3ull * ONES) & ((1ull << (4 * L)) - 1ull)) << (4 * s), mult, cat, idb,
                               skipj, lane, o, n);
        idb += binom(R, L) - ((skipj && L == 2) ? 1 : 0);
      }
  };
  planes(THREE_ONE_LINE, m1, M15, 5, 1, true, ID_THREE_ONE_LINE);
  planes(THREE_TWO_LINE, m2, M13, 4, 2, false, ID_THREE_TWO_LINE);
  // rocket (card.py:134); beats everything (card.py:314-315)
  if ((m1 & JOKERS) == JOKERS) n = emit<IDS>(lane == 0, (1ull << 52) | (1ull << 56), BIGBANG, ID_BIGBANG, o, n);
  // four with two kickers (card.py:139-153)
  auto fours = [&](int cat, uint32_t kick, uint32_t ranks, int mult, bool skipj, int idbase, int per) {
    if (!(f.lead || f.lc == cat)) return;
    for (uint32_t qm = m4 & value_gate(f, cat); qm; qm &= qm - 1) {
      const int q = __builtin_ctz(qm);
      n = emit_combos<IDS>(kick & ranks & ~(1u << q), ranks & ~(1u << q), 2, 4ull << (4 * q), mult, cat,
                           idbase + q * per, skipj, lane, o, n);
    }
  };
  fours(FOUR_TAKE_ONE, m1, M15, 1, true, ID_FOUR_TAKE_ONE, 90);
  fours(FOUR_TAKE_TWO, m2, M13, 2, false, ID_FOUR_TAKE_TWO, 66);
  return n;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// one wavefront per table (tpw consecutive tables per wave)
template <bool IDS>
__global__ __launch_bounds__(BLOCK) void k_enum(const uint4* __restrict__ q,
                                                const int32_t* __restrict__ counts,
                                                const int32_t* __restrict__ local_off,
                                                const int32_t* __restrict__ blk_tot, int64_t T, int tpw,
                                                int32_t* __restrict__ offsets, uint4* __restrict__ rows,
                                                int32_t* __restrict__ ids, int64_t cap,
                                                int32_t* __restrict__ status,
                                                int64_t* __restrict__ legal_rows) {
  const int lane = threadIdx.x & 63;
  // readfirstlane: the wave index is uniform, tell the compiler so (scalar loads, s_branches)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t wave = (int64_t)blockIdx.x * (BLOCK / 64) + wv;
  const int64_t t0 = wave * tpw;
  if (t0 >= T) return;
  const uint32_t lut = c_line_lut.v[lane];
  // CSR base of table t0: totals of the 256-blocks before it + its offset inside its block
  const int blk = (int)(t0 / BLOCK);
  int part = 0;
  for (int j = lane; j < blk; j += 64) part += blk_tot[j];
  int64_t base = (int64_t)wave_sum(part) + local_off[t0];
  const int ntab = (int)(T - t0 < tpw ? T - t0 : tpw);
  uint4 rec = make_uint4(0, 0, 0, 0);
  int cnt_l = 0;
  if (lane < ntab) {
    rec = q[t0 + lane];
    cnt_l = counts[t0 + lane];
  }
  Out o{rows, ids, 0, cap};
  for (int i = 0; i < ntab; ++i) {
    const uint64_t hand = (uint64_t)__builtin_amdgcn_readlane(rec.x, i) |
                          ((uint64_t)__builtin_amdgcn_readlane(rec.y, i) << 32);
    const uint32_t info = __builtin_amdgcn_readlane(rec.z, i);
    const int cnt = __builtin_amdgcn_readlane(cnt_l, i);
    o.base = base;
    if (lane == 0) offsets[t0 + i] = (int32_t)base;
    const int n = enumerate_table<IDS>(hand, info, lut, lane, o);
    if (lane == 0) {
      int bits = (n != cnt ? 1 : 0) | (base + cnt > cap ? 2 : 0) | ((info & QF_BADLAST) ? 4 : 0);
      if (bits) atomicOr(status, bits);
    }
    base += cnt;
  }
  if (t0 + ntab == T && lane == 0) {
    offsets[T] = (int32_t)base;
    *legal_rows += base;
  }
}

// ------------------------------------------------------------------------------------
__device__ __forceinline__ int block_excl_scan(int v, int* total) {
  __shared__ int wsum[BLOCK / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[w] = x;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < BLOCK / 64; ++i) {
    if (i < w) off += wsum[i];
    tot += wsum[i];
  }
  *total = tot;
  return off + x - v;
}

struct StateView {
  uint8_t* s;
  int64_t T;
  __device__ __forceinline__ uint4* row(int f, int64_t t) const { return (uint4*)(s + ((int64_t)f * T + t) * 16); }
};

// deal spec v1: cards k = 0..53 in rank order; card k goes to the role picked by
// x = (u32 * remaining) >> 32 against the remaining capacities {17 up, 20 lord, 17 down}
// (envi.py:23).  Philox counter = (gid, episode, 1<<16 | block), key = seed.
__device__ inline void deal(uint64_t gid, uint32_t episode, uint32_t k0, uint32_t k1, uint64_t h[3]) {
  int cap0 = 17, cap1 = 20;
  uint64_t h0 = 0, h1 = 0, h2 = 0;
  for (uint32_t b = 0; b < 14; ++b) {
    const uint4 d = philox4x32_10(make_uint4((uint32_t)gid, (uint32_t)(gid >> 32), episode, (1u << 16) | b), k0, k1);
    const uint32_t dr[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 4 * (int)b + j;
      if (k < 54) {
        const uint32_t x = __umulhi(dr[j], (uint32_t)(54 - k));
        const int rank = k < 52 ? k >> 2 : k - 39;
        const uint64_t one = 1ull << (4 * rank);
        if (x < (uint32_t)cap0) { h0 += one; --cap0; }
        else if (x < (uint32_t)(cap0 + cap1)) { h1 += one; --cap1; }
        else { h2 += one; }
      }
    }
  }
  h[0] = h0; h[1] = h1; h[2] = h2;
}

// the combo to beat: previous player's handout, else the one before, else lead
// (envi.py:103-109).  b1/b2 = recent rows of (role-1)%3 and (role-2)%3.
__device__ __forceinline__ uint32_t last_info(uint4 b1, uint4 b2) {
  const uint64_t n1 = pack_row(b1), n2 = pack_row(b2);
  if (n1) return info_of_row(n1, (int)(b1.w >> 24));
  if (n2) return info_of_row(n2, (int)(b2.w >> 24));
  return mk_info(EMPTY, 0, 1);
}

struct StepArgs {
  uint8_t* state;
  int64_t T;
  uint32_t k0, k1;    // philox key = seed
  uint64_t gid_base;
  int mode, auto_reset;
  const void* sel;           // CHOICE: int32[T]; ROWS: int8[T][16]; RESET: u8 mask[T] or null
  const int32_t* offsets;    // CSR of the current state
  const uint4* rows;
  int64_t cap;               // rows actually backed by memory
  uint8_t* done;
  int8_t* reward;
  uint8_t* illegal;
  uint4* traj;               // [T][2]
  Scratch sc;
};

__global__ __launch_bounds__(BLOCK) void k_step(StepArgs a) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const bool in = t < a.T;
  const StateView S{a.state, a.T};
  uint64_t qhand = 0;
  uint32_t qinfo = QF_FROZEN;
  int st_ply = 0, st_eps = 0, st_lord = 0;
  if (in) {
    uint4 meta = *S.row(DDZ_F_META, t);
    int role = meta.x & 0xFF;
    if (role > 2) role = 0;  // never index outside the state on a corrupted import
    const bool was_done = (meta.x >> 8) & 0xFF, dealt = (meta.y >> 16) & 0xFF;
    uint32_t ply = meta.y & 0xFFFF, episode = meta.z;
    const uint64_t gid = a.gid_base + (uint64_t)t;
    bool redeal = false;
    if (a.mode == MODE_RESET) {
      const uint8_t* mask = (const uint8_t*)a.sel;
      if (!mask || mask[t]) {
        redeal = true;
        episode = dealt ? episode + 1 : 0;
      }
    } else if (a.mode != MODE_COUNT) {
      const int32_t off = a.offsets[t];
      int32_t A = a.offsets[t + 1] - off;
      if (off < 0 || (int64_t)off + A > a.cap) A = 0;  // list was truncated: nothing to pick from
      const bool frozen = was_done || !dealt || A <= 0;
      int32_t idx = -1;
      if (!frozen) {
        if (a.mode == DDZ_STEP_RANDOM) {  // random.choice(actions), envi.py:83
          const uint4 d = philox4x32_10(make_uint4((uint32_t)gid, (uint32_t)(gid >> 32), episode, (2u << 16) | ply), a.k0, a.k1);
          idx = (int32_t)__umulhi(d.x, (uint32_t)A);
        } else if (a.mode == DDZ_STEP_CHOICE) {
          idx = ((const int32_t*)a.sel)[t];
          if (idx < 0 || idx >= A) idx = -1;
        } else {
          const uint4 want = ((const uint4*)a.sel)[t];
          for (int32_t j = 0; j < A && idx < 0; ++j) {
            const uint4 r = a.rows[off + j];
            if (r.x == want.x && r.y == want.y && r.z == want.z && ((r.w ^ want.w) & 0x00FFFFFFu) == 0) idx = j;
          }
        }
      }
      uint4 tr0 = make_uint4(0, 0, 0, 0);
      uint4 tr1 = make_uint4((uint32_t)role, ((uint32_t)A & 0xFFFF) | (ply << 16), episode, 0xFFFFFFFFu);
      uint8_t o_done = was_done, o_illegal = 0;
      int8_t o_reward = 0;
      if (frozen) {
        tr1.x |= (uint32_t)was_done << 8 | 2u << 24;
      } else if (idx < 0) {
        o_done = 0; o_illegal = 1;
        tr1.x |= 1u << 24;
      } else {
        const uint4 row = a.rows[off + idx];
        const uint4 rowc = make_uint4(row.x, row.y, row.z, row.w & 0x00FFFFFFu);
        const int ncards = nib_sum(pack_row(row));
        uint4* ph = S.row(DDZ_F_HAND0 + role, t);
        uint4* pi = S.row(DDZ_F_HIST0 + role, t);
        uint4* pt = S.row(DDZ_F_TAKEN, t);
        uint4 h = *ph, hi = *pi, tk = *pt;
        // byte-wise: every byte of the hand >= the row's byte, so no borrows cross bytes
        h.x -= rowc.x; h.y -= rowc.y; h.z -= rowc.z; h.w -= rowc.w + ((uint32_t)ncards << 24);
        hi.x += rowc.x; hi.y += rowc.y; hi.z += rowc.z; hi.w += rowc.w;
        tk.x += rowc.x; tk.y += rowc.y; tk.z += rowc.z; tk.w += rowc.w;
        *ph = h; *pi = hi; *pt = tk;                 // envi.py:39-41
        *S.row(DDZ_F_RECENT0 + role, t) = row;       // envi.py:43
        const bool won = (h.w >> 24) == 0;
        o_reward = won ? (role == 1 ? -1 : 1) : 0;   // rule_play.py:14
        o_done = won;
        st_ply = 1; st_eps = won; st_lord = won && role == 1;
        tr0 = row;
        tr1.x |= (uint32_t)won << 8 | ((uint32_t)(uint8_t)o_reward) << 16;
        tr1.w = (uint32_t)idx;
        const int nrole = role == 2 ? 0 : role + 1;  // lord -> down -> up, game.py:173-181
        ply += 1;
        if (won && a.auto_reset) {
          redeal = true;
          episode += 1;
        } else {
          meta.x = (uint32_t)nrole | (won ? 1u << 8 : 0u) | ((won ? (uint32_t)role : 0xFFu) << 16) |
                   ((uint32_t)(uint8_t)o_reward << 24);
          meta.y = (meta.y & 0xFFFF0000u) | (ply & 0xFFFF);
          *S.row(DDZ_F_META, t) = meta;
          if (!won) {  // query of the next actor: its hand + the two most recent handouts
            const uint4 nh = *S.row(DDZ_F_HAND0 + nrole, t);
            const uint4 b2 = *S.row(DDZ_F_RECENT0 + (role == 0 ? 2 : role - 1), t);
            qhand = pack_row(nh);
            qinfo = last_info(row, b2);
          }
        }
      }
      if (a.done) a.done[t] = o_done;
      if (a.reward) a.reward[t] = o_reward;
      if (a.illegal) a.illegal[t] = o_illegal;
      if (a.traj) { a.traj[2 * t] = tr0; a.traj[2 * t + 1] = tr1; }
      if (frozen || idx < 0) {  // state unchanged: recount it below as MODE_COUNT does
        if (!was_done && dealt) {
          const uint4 nh = *S.row(DDZ_F_HAND0 + role, t);
          qhand = pack_row(nh);
          qinfo = last_info(*S.row(DDZ_F_RECENT0 + (role + 2) % 3, t), *S.row(DDZ_F_RECENT0 + (role + 1) % 3, t));
        }
      }
    }
    if (redeal) {
      uint64_t h[3];
      deal(gid, episode, a.k0, a.k1, h);
      const uint4 z = make_uint4(0, 0, 0, 0);
      *S.row(DDZ_F_HAND0 + 0, t) = unpack_row(h[0], 17);
      *S.row(DDZ_F_HAND0 + 1, t) = unpack_row(h[1], 20);
      *S.row(DDZ_F_HAND0 + 2, t) = unpack_row(h[2], 17);
#pragma unroll
      for (int f = DDZ_F_HIST0; f <= DDZ_F_TAKEN; ++f) *S.row(f, t) = z;  // envi.py:32-35
      *S.row(DDZ_F_META, t) = make_uint4(1u | (0xFFu << 16), 1u << 16, episode, 0);
      qhand = h[1];  // lord leads (game.py:173)
      qinfo = mk_info(EMPTY, 0, 1);
    } else if (a.mode == MODE_COUNT || a.mode == MODE_RESET) {
      if (!was_done && dealt) {
        qhand = pack_row(*S.row(DDZ_F_HAND0 + role, t));
        qinfo = last_info(*S.row(DDZ_F_RECENT0 + (role + 2) % 3, t), *S.row(DDZ_F_RECENT0 + (role + 1) % 3, t));
      }
    }
  }
  const int cnt = (in && !(qinfo & QF_FROZEN)) ? count_legal(qhand, qinfo) : 0;
  int total;
  const int loff = block_excl_scan(cnt, &total);
  if (in) {
    a.sc.q[t] = make_uint4((uint32_t)qhand, (uint32_t)(qhand >> 32), qinfo, 0);
    a.sc.counts[t] = cnt;
    a.sc.local_off[t] = loff;
  }
  // per-block statistics (no atomics: each block owns its slot)
  const uint64_t bp = __ballot(st_ply), be = __ballot(st_eps), bl = __ballot(st_lord);
  __shared__ int sst[BLOCK / 64][3];
  if ((threadIdx.x & 63) == 0) {
    sst[threadIdx.x >> 6][0] = __popcll(bp);
    sst[threadIdx.x >> 6][1] = __popcll(be);
    sst[threadIdx.x >> 6][2] = __popcll(bl);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.sc.blk_tot[blockIdx.x] = total;
    int64_t* bs = a.sc.blk_stats + 4 * (int64_t)blockIdx.x;
    for (int k = 0; k < 3; ++k) bs[k] += sst[0][k] + sst[1][k] + sst[2][k] + sst[3][k];
  }
}

// stateless queries: r.get_moves(hand15, last15) for n independent (hand, last) pairs
__global__ __launch_bounds__(BLOCK) void k_query(const uint4* __restrict__ hands,
                                                 const uint4* __restrict__ lasts, int64_t n, Scratch sc) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  uint64_t qhand = 0;
  uint32_t qinfo = QF_FROZEN;
  if (t < n) {
    qhand = pack_row(hands[t]);
    qinfo = classify(pack_row(lasts[t]));
    if (qinfo == INFO_INVALID) qinfo = QF_BADLAST | QF_FROZEN;
    if (ge_mask(qhand, 5) || (qhand >> 60)) qinfo = QF_BADLAST | QF_FROZEN;
  }
  const int cnt = (t < n && !(qinfo & QF_FROZEN)) ? count_legal(qhand, qinfo) : 0;
  int total;
  const int loff = block_excl_scan(cnt, &total);
  if (t < n) {
    sc.q[t] = make_uint4((uint32_t)qhand, (uint32_t)(qhand >> 32), qinfo, 0);
    sc.counts[t] = cnt;
    sc.local_off[t] = loff;
  }
  if (threadIdx.x == 0) sc.blk_tot[blockIdx.x] = total;
}

__global__ __launch_bounds__(BLOCK) void k_classify(const uint4* __restrict__ rows, int64_t n,
                                                    uint32_t* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (t < n) out[t] = classify(pack_row(rows[t]));
}

// stats[0..3] += {plies, episodes, legal rows, lord wins}; block slots are cleared
__global__ __launch_bounds__(BLOCK) void k_reduce_stats(Scratch sc, int64_t nblk, int64_t* stats) {
  __shared__ long long sh[3][BLOCK];
  long long v[3] = {0, 0, 0};
  for (int64_t b = threadIdx.x; b < nblk; b += BLOCK)
    for (int k = 0; k < 3; ++k) {
      v[k] += sc.blk_stats[4 * b + k];
      sc.blk_stats[4 * b + k] = 0;
    }
  for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] = v[k];
  __syncthreads();
  for (int d = BLOCK / 2; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d)
      for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    stats[0] += sh[0][0];
    stats[1] += sh[1][0];
    stats[2] += *sc.legal_rows;
    stats[3] += sh[2][0];
    *sc.legal_rows = 0;
  }
}

// ------------------------------------------------------------------------------------
// face: f32 [T][P][15][4], one thread per (table, plane, rank) -> one 16-byte store.
// plane kinds: 0 hand 1 taken 2..4 history of (role-1, role, role+1) 5,6 recent handout of
// (role-1, role-2) 7,8 prob planes (spec v1; native get_state_prob is absent).
__constant__ uint8_t c_face_kind[4][9] = {{0, 1, 7, 8, 0, 0, 0, 0, 0},
                                          {0, 1, 2, 3, 4, 7, 8, 0, 0},
                                          {0, 1, 2, 3, 4, 5, 6, 7, 8},
                                          {0, 1, 5, 6, 7, 8, 0, 0, 0}};

__global__ __launch_bounds__(BLOCK) void k_observe(const uint8_t* __restrict__ state, int64_t T, int variant,
                                                   int P, float4* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (idx >= T * P * 15) return;
  const int64_t t = idx / (P * 15);
  const int rem = (int)(idx - t * (P * 15)), p = rem / 15, i = rem - p * 15;
  auto byte = [&](int f, int k) { return (int)state[((int64_t)f * T + t) * 16 + k]; };
  int role = byte(DDZ_F_META, 0);
  if (role > 2) role = 0;
  const int kind = c_face_kind[variant][p];
  const int rm1 = (role + 2) % 3, rp1 = (role + 1) % 3;
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
  out[idx] = v;
}

__global__ __launch_bounds__(BLOCK) void k_onehot(const uint8_t* __restrict__ rows, int64_t n,
                                                  float4* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (idx >= n * 15) return;
  const int64_t r = idx / 15;
  const int c = rows[r * 16 + (idx - r * 15)];
  out[idx] = make_float4(c > 0 ? 1.f : 0.f, c > 1 ? 1.f : 0.f, c > 2 ? 1.f : 0.f, c > 3 ? 1.f : 0.f);
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

inline int pick_tpw(int64_t T) {
  if (const char* s = getenv("DDZ_TPW")) {
    int v = atoi(s);
    if (v >= 1 && v <= 64) return v;
  }
  int64_t v = (T + 16383) / 16384;
  return (int)(v < 1 ? 1 : v > 16 ? 16 : v);
}

int launch_enum(const Scratch& sc, int64_t T, int32_t* offsets, int8_t* rows, int32_t* ids, int64_t cap,
                hipStream_t st) {
  if (T == 0) return DDZ_OK;
  const int tpw = pick_tpw(T);
  const int64_t waves = (T + tpw - 1) / tpw;
  const dim3 grid((unsigned)((waves + BLOCK / 64 - 1) / (BLOCK / 64)));
  if (ids)
    hipLaunchKernelGGL(k_enum<true>, grid, dim3(BLOCK), 0, st, sc.q, sc.counts, sc.local_off, sc.blk_tot, T, tpw,
                       offsets, (uint4*)rows, ids, cap, sc.status, sc.legal_rows);
  else
    hipLaunchKernelGGL(k_enum<false>, grid, dim3(BLOCK), 0, st, sc.q, sc.counts, sc.local_off, sc.blk_tot, T, tpw,
                       offsets, (uint4*)rows, (int32_t*)nullptr, cap, sc.status, sc.legal_rows);
  return check_launch();
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
  bool counts_valid;
  int64_t legal_cap;  // capacity of the row buffer the last ddz_legal wrote
};

namespace {
inline bool good(const ddz_env* e) { return e && e->magic == MAGIC; }

int launch_step(ddz_env* e, int mode, const void* sel, const int32_t* offsets, const int8_t* rows, int auto_reset,
                uint8_t* done, int8_t* reward, uint8_t* illegal, uint8_t* traj, hipStream_t st) {
  StepArgs a;
  a.state = e->state; a.T = e->T;
  a.k0 = (uint32_t)e->seed; a.k1 = (uint32_t)(e->seed >> 32);
  a.gid_base = e->gid_base; a.mode = mode; a.auto_reset = auto_reset; a.sel = sel;
  a.offsets = offsets; a.rows = (const uint4*)rows; a.cap = e->legal_cap; a.done = done; a.reward = reward; a.illegal = illegal;
  a.traj = (uint4*)traj; a.sc = e->sc;
  hipLaunchKernelGGL(k_step, dim3((unsigned)e->lay.nblk), dim3(BLOCK), 0, st, a);
  int rc = check_launch();
  if (rc == DDZ_OK) e->counts_valid = true;
  return rc;
}
}  // namespace

extern "C" {

int ddz_abi_version(void) { return DDZ_ABI_VERSION; }

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
  ddz_env* e = (ddz_env*)calloc(1, sizeof(ddz_env));
  if (!e) return DDZ_EINVAL;
  e->magic = MAGIC; e->T = T; e->seed = seed; e->gid_base = gid_base; e->device = device;
  e->state = (uint8_t*)state; e->scratch = scratch;
  e->lay = make_layout(T); e->sc = bind(scratch, e->lay); e->counts_valid = false; e->legal_cap = 0;
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
  return launch_step(e, MODE_RESET, mask, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

int ddz_legal(ddz_env_t* e, int32_t* offsets, int8_t* rows, int32_t* ids, int64_t cap, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!offsets || !rows || cap < 0) return DDZ_EINVAL;
  if (cap > 0x7FFFFFFF) return DDZ_ECAP;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  if (!e->counts_valid) {
    int rc = launch_step(e, MODE_COUNT, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream);
    if (rc) return rc;
  }
  e->legal_cap = cap;
  return launch_enum(e->sc, e->T, offsets, rows, ids, cap, (hipStream_t)stream);
}

int ddz_step(ddz_env_t* e, int mode, const void* sel, const int32_t* offsets, const int8_t* rows, int auto_reset,
             uint8_t* done, int8_t* reward, uint8_t* illegal, uint8_t* traj, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (mode < DDZ_STEP_RANDOM || mode > DDZ_STEP_ROWS || !offsets || !rows) return DDZ_EINVAL;
  if (mode != DDZ_STEP_RANDOM && !sel) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  return launch_step(e, mode, sel, offsets, rows, auto_reset ? 1 : 0, done, reward, illegal, traj, (hipStream_t)stream);
}

int ddz_observe(ddz_env_t* e, int variant, float* face, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  const int P = ddz_face_planes(variant);
  if (P < 0 || !face) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  const int64_t n = e->T * P * 15;
  hipLaunchKernelGGL(k_observe, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     (const uint8_t*)e->state, e->T, variant, P, (float4*)face);
  return check_launch();
}

int ddz_rows_to_onehot(int device, const int8_t* rows, int64_t n, float* out, void* stream) {
  if (n < 0 || (n > 0 && (!rows || !out))) return DDZ_EINVAL;
  if (n == 0) return DDZ_OK;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  const int64_t m = n * 15;
  hipLaunchKernelGGL(k_onehot, dim3((unsigned)((m + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     (const uint8_t*)rows, n, (float4*)out);
  return check_launch();
}

int ddz_get_moves(int device, const int8_t* hands, const int8_t* lasts, int64_t n, int32_t* offsets, int8_t* rows,
                  int32_t* ids, int64_t cap, void* scratch, int64_t scratch_bytes, void* stream) {
  if (n <= 0 || !hands || !lasts || !offsets || !rows || !scratch || cap < 0) return DDZ_EINVAL;
  if (cap > 0x7FFFFFFF) return DDZ_ECAP;
  const Layout l = make_layout(n);
  if (scratch_bytes < l.bytes || ((uintptr_t)scratch & 15)) return DDZ_EINVAL;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  const Scratch sc = bind(scratch, l);
  hipLaunchKernelGGL(k_query, dim3((unsigned)l.nblk), dim3(BLOCK), 0, (hipStream_t)stream, (const uint4*)hands,
                     (const uint4*)lasts, n, sc);
  int rc = check_launch();
  if (rc) return rc;
  return launch_enum(sc, n, offsets, rows, ids, cap, (hipStream_t)stream);
}

int ddz_read_stats(ddz_env_t* e, int64_t* stats, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (!stats) return DDZ_EINVAL;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(BLOCK), 0, (hipStream_t)stream, e->sc, e->lay.nblk, stats);
  return check_launch();
}

int ddz_rollout_random(ddz_env_t* e, int64_t n_iters, int32_t* offsets, int8_t* rows, int32_t* ids, int64_t cap,
                       int64_t* stats, uint8_t* traj, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (n_iters < 0 || !offsets || !rows || cap < 0) return DDZ_EINVAL;
  if (cap > 0x7FFFFFFF) return DDZ_ECAP;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  hipStream_t st = (hipStream_t)stream;
  e->legal_cap = cap;
  for (int64_t it = 0; it < n_iters; ++it) {
    if (!e->counts_valid) {
      int rc = launch_step(e, MODE_COUNT, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, st);
      if (rc) return rc;
    }
    int rc = launch_enum(e->sc, e->T, offsets, rows, ids, cap, st);
    if (rc) return rc;
    rc = launch_step(e, DDZ_STEP_RANDOM, nullptr, offsets, rows, 1, nullptr, nullptr, nullptr,
                     traj ? traj + it * e->T * DDZ_TRAJ_BYTES : nullptr, st);
    if (rc) return rc;
  }
  if (stats) {
    hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(BLOCK), 0, st, e->sc, e->lay.nblk, stats);
    return check_launch();
  }
  return DDZ_OK;
}

int ddz_rollout_random_timed(ddz_env_t* e, int64_t n_iters, int32_t* offsets, int8_t* rows, int32_t* ids,
                             int64_t cap, double* ms, void* stream) {
  if (!good(e)) return DDZ_EHANDLE;
  if (n_iters <= 0 || n_iters > 100000 || !offsets || !rows || cap < 0 || !ms) return DDZ_EINVAL;
  if (cap > 0x7FFFFFFF) return DDZ_ECAP;
  DeviceGuard g(e->device);
  if (!g.ok) return DDZ_ENODEV;
  hipStream_t st = (hipStream_t)stream;
  e->legal_cap = cap;
  if (!e->counts_valid) {
    int rc = launch_step(e, MODE_COUNT, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, st);
    if (rc) return rc;
  }
  const int64_t nev = 2 * n_iters + 1;
  hipEvent_t* ev = (hipEvent_t*)calloc((size_t)nev, sizeof(hipEvent_t));
  if (!ev) return DDZ_EINVAL;
  int rc = DDZ_OK;
  int64_t made = 0;
  for (; made < nev; ++made)
    if (hipEventCreate(&ev[made]) != hipSuccess) { rc = hip_fail(hipGetLastError()); break; }
  if (rc == DDZ_OK) {
    (void)hipEventRecord(ev[0], st);
    for (int64_t it = 0; it < n_iters && rc == DDZ_OK; ++it) {
      rc = launch_enum(e->sc, e->T, offsets, rows, ids, cap, st);
      (void)hipEventRecord(ev[2 * it + 1], st);
      if (rc == DDZ_OK)
        rc = launch_step(e, DDZ_STEP_RANDOM, nullptr, offsets, rows, 1, nullptr, nullptr, nullptr, nullptr, st);
      (void)hipEventRecord(ev[2 * it + 2], st);
    }
    hipError_t r = hipStreamSynchronize(st);
    if (r != hipSuccess) rc = hip_fail(r);
    if (rc == DDZ_OK) {
      double a = 0, b = 0;
      for (int64_t it = 0; it < n_iters; ++it) {
        float x = 0, y = 0;
        (void)hipEventElapsedTime(&x, ev[2 * it], ev[2 * it + 1]);
        (void)hipEventElapsedTime(&y, ev[2 * it + 1], ev[2 * it + 2]);
        a += x; b += y;
      }
      ms[0] = a; ms[1] = b;
    }
  }
  for (int64_t i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
  free(ev);
  return rc;
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

// debug/test entry: classify(rows) -> info words (category | value << 8 | len << 16, 0xFF invalid)
int ddz_debug_classify(int device, const int8_t* rows, int64_t n, uint32_t* out, void* stream) {
  if (n <= 0 || !rows || !out) return DDZ_EINVAL;
  DeviceGuard g(device);
  if (!g.ok) return DDZ_ENODEV;
  hipLaunchKernelGGL(k_classify, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     (const uint4*)rows, n, out);
  return check_launch();
}

}  // extern "C"
