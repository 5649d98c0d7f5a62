// ddz_device.h -- device-side building blocks of the gfx950 Doudizhu engine.
//
// Representation: a card-count vector (ranks 3..K,A,2,BJ,CJ; envi.py:122-124) lives in
// HBM as a 16-byte row (int8 counts[15] + 1 aux byte) and in registers as a
// nibble-packed u64 ("nib"): rank i at bits [4i, 4i+4).  All rule predicates are SWAR
// arithmetic on nibs or bit tests on 15-bit rank masks.  Canonical action ids follow the
// order of rule_based/utils/card.py:34-159 (reference paths relative to /root/reference).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef DDZ_NATIVE_JOKER_KICKERS
#define DDZ_NATIVE_JOKER_KICKERS 0  // 1: libddz_hip_jk.so, the rule set plus the 24 joker-kicker rows (include/ddz_env.h)
#endif

namespace ddz {

constexpr uint64_t ONES = 0x1111111111111111ull;
constexpr uint32_t M15 = 0x7FFF;   // all 15 ranks
constexpr uint32_t M13 = 0x1FFF;   // 3..2 (ranks that exist four times)
constexpr uint32_t M12 = 0x0FFF;   // 3..A (ranks allowed in chains, card.py:86-87)
constexpr uint32_t JOKERS = 0x6000;

// categories, card.py:13-28
enum : int {
  EMPTY = 0, SINGLE, DOUBLE, TRIPLE, QUADRIC, THREE_ONE, THREE_TWO, SINGLE_LINE, DOUBLE_LINE,
  TRIPLE_LINE, THREE_ONE_LINE, THREE_TWO_LINE, BIGBANG, FOUR_TAKE_ONE, FOUR_TAKE_TWO
};
// first canonical id of each category (card.py Category2Range)
constexpr int ID_THREE_ONE = 55, ID_THREE_TWO = 237, ID_SINGLE_LINE = 393, ID_DOUBLE_LINE = 429,
              ID_TRIPLE_LINE = 481, ID_THREE_ONE_LINE = 526, ID_THREE_TWO_LINE = 8559,
              ID_BIGBANG = 11498, ID_FOUR_TAKE_ONE = 11499, ID_FOUR_TAKE_TWO = 12669;

constexpr uint32_t INFO_INVALID = 0xFFu;
constexpr uint32_t QF_FROZEN = 1u << 24;   // empty legal list (done / not dealt)
constexpr uint32_t QF_BADLAST = 2u << 24;  // `last` is no combo of the action space

__host__ __device__ __forceinline__ uint32_t mk_info(int cat, int value, int len) {
  return (uint32_t)cat | ((uint32_t)value << 8) | ((uint32_t)len << 16);
}

// ---- byte row <-> nibble word ------------------------------------------------------
__device__ __forceinline__ uint32_t squeeze8(uint64_t x) {  // 8 bytes (each <= 15) -> 8 nibbles
  x &= 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
  x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
  x = (x | (x >> 16));
  return (uint32_t)x;
}
__device__ __forceinline__ uint64_t spread8(uint32_t v) {  // 8 nibbles -> 8 bytes
  uint64_t x = v;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
  return x;
}
__device__ __forceinline__ uint64_t pack_row(uint4 r) {  // byte 15 (aux) is dropped
  uint64_t lo = (uint64_t)r.x | ((uint64_t)r.y << 32);
  uint64_t hi = (uint64_t)r.z | ((uint64_t)(r.w & 0x00FFFFFFu) << 32);
  return (uint64_t)squeeze8(lo) | ((uint64_t)squeeze8(hi) << 32);
}
__device__ __forceinline__ uint32_t spread4(uint32_t n16) {  // 4 nibbles -> 4 bytes
  uint32_t w = (n16 | (n16 << 8)) & 0x00FF00FFu;
  return (w | (w << 4)) & 0x0F0F0F0Fu;
}
__device__ __forceinline__ uint4 unpack_row(uint64_t nib, uint32_t aux) {
  const uint32_t lo = (uint32_t)nib, hi = (uint32_t)(nib >> 32);
  return make_uint4(spread4(lo & 0xFFFFu), spread4(lo >> 16), spread4(hi & 0xFFFFu),
                    spread4((hi >> 16) & 0x0FFFu) | (aux << 24));
}
__device__ __forceinline__ int nib_sum(uint64_t nib) {  // number of cards
  uint64_t b = (nib & 0x0F0F0F0F0F0F0F0Full) + ((nib >> 4) & 0x0F0F0F0F0F0F0F0Full);
  return (int)((b * 0x0101010101010101ull) >> 56);
}
// 15-bit mask of the ranks whose count is >= k (k in 1..5, counts <= 7)
__device__ __forceinline__ uint32_t ge_mask(uint64_t nib, int k) {
  uint64_t y = ((nib + (uint64_t)(8 - k) * ONES) >> 3) & ONES;
  y = (y | (y >> 3)) & 0x0303030303030303ull;
  y = (y | (y >> 6)) & 0x000F000F000F000Full;
  y = (y | (y >> 12)) & 0x000000FF000000FFull;
  y = (y | (y >> 24));
  return (uint32_t)y & M15;
}
__device__ __forceinline__ uint64_t spread15(uint32_t m) {  // bit r -> bit 4r
  uint64_t x = m & M15;
  x = (x | (x << 24)) & 0x000000FF000000FFull;
  x = (x | (x << 12)) & 0x000F000F000F000Full;
  x = (x | (x << 6)) & 0x0303030303030303ull;
  x = (x | (x << 3)) & ONES;
  return x;
}

// ---- wave64 scans / reductions with DPP row operations ------------------------------------------------------------
// (v_*_dpp reads a neighbour lane's VGPR inside the VALU: ~1 issue slot per step, against an LDS-crossbar round trip of
// ~100 cycles for every ds_bpermute a __shfl compiles to.)  Kogge-Stone inside the rows of 16 lanes (row_shr 1, 2, 4, 8:
// a lane whose source is outside its row keeps `identity`), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into
// rows 2 and 3: an inclusive scan of the 64 lanes in six steps; lane 63 holds the reduction.
#define DDZ_DPP(identity, v, ctrl, row_mask) __builtin_amdgcn_update_dpp((identity), (v), (ctrl), (row_mask), 0xf, false)
__device__ __forceinline__ int wave_scan_add(int v) {  // inclusive prefix sum over the lanes
  v += DDZ_DPP(0, v, 0x111, 0xf);
  v += DDZ_DPP(0, v, 0x112, 0xf);
  v += DDZ_DPP(0, v, 0x114, 0xf);
  v += DDZ_DPP(0, v, 0x118, 0xf);
  v += DDZ_DPP(0, v, 0x142, 0xa);
  v += DDZ_DPP(0, v, 0x143, 0xc);
  return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) { return __builtin_amdgcn_readlane(wave_scan_add(v), 63); }
__device__ __forceinline__ float wave_sum_f32(float v) {  // the same value in every lane; the order of the adds is fixed
#define DDZ_STEP(ctrl, rm) v += __int_as_float(DDZ_DPP(0, __float_as_int(v), ctrl, rm));
  DDZ_STEP(0x111, 0xf) DDZ_STEP(0x112, 0xf) DDZ_STEP(0x114, 0xf) DDZ_STEP(0x118, 0xf) DDZ_STEP(0x142, 0xa) DDZ_STEP(0x143, 0xc)
#undef DDZ_STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_max_i32(int v) {  // the same value in every lane (wave-uniform)
  constexpr int ID = (int)0x80000000;
#define DDZ_STEP(ctrl, rm) { const int o_ = DDZ_DPP(ID, v, ctrl, rm); v = o_ > v ? o_ : v; }
  DDZ_STEP(0x111, 0xf) DDZ_STEP(0x112, 0xf) DDZ_STEP(0x114, 0xf) DDZ_STEP(0x118, 0xf) DDZ_STEP(0x142, 0xa) DDZ_STEP(0x143, 0xc)
#undef DDZ_STEP
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {  // wave-uniform
#define DDZ_STEP(ctrl, rm) { const uint32_t o_ = (uint32_t)DDZ_DPP((int)0xFFFFFFFF, (int)v, ctrl, rm); v = o_ < v ? o_ : v; }
  DDZ_STEP(0x111, 0xf) DDZ_STEP(0x112, 0xf) DDZ_STEP(0x114, 0xf) DDZ_STEP(0x118, 0xf) DDZ_STEP(0x142, 0xa) DDZ_STEP(0x143, 0xc)
#undef DDZ_STEP
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ double wave_max_f64(double v) {  // NaN-free inputs (-inf allowed); wave-uniform result
  constexpr int NINF_HI = (int)0xFFF00000;
#define DDZ_STEP(ctrl, rm)                                                                         \
  {                                                                                                \
    const int lo_ = DDZ_DPP(0, __double2loint(v), ctrl, rm), hi_ = DDZ_DPP(NINF_HI, __double2hiint(v), ctrl, rm); \
    const double o_ = __hiloint2double(hi_, lo_);                                                  \
    v = o_ > v ? o_ : v;                                                                           \
  }
  DDZ_STEP(0x111, 0xf) DDZ_STEP(0x112, 0xf) DDZ_STEP(0x114, 0xf) DDZ_STEP(0x118, 0xf) DDZ_STEP(0x142, 0xa) DDZ_STEP(0x143, 0xc)
#undef DDZ_STEP
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

__device__ __forceinline__ uint32_t gt_mask(int v) {  // ranks strictly above v (v may be 100)
  return v >= 14 ? 0u : (M15 & ~((2u << v) - 1u));
}
// starts of runs of at least L consecutive set bits
__device__ __forceinline__ uint32_t run_starts(uint32_t m, int L) {
  uint32_t x = m;
  for (int i = 1; i < L; ++i) x &= m >> i;
  return x;
}
__device__ __forceinline__ int binom(int a, int b) {  // C(a,b), a <= 15, b <= 5
  if (b < 0 || a < b) return 0;
  switch (b) {
    case 0: return 1;
    case 1: return a;
    case 2: return a * (a - 1) / 2;
    case 3: return a * (a - 1) * (a - 2) / 6;
    case 4: return a * (a - 1) * (a - 2) * (a - 3) / 24;
    default: return a * (a - 1) * (a - 2) * (a - 3) * (a - 4) / 120;
  }
}

// (category, value, len) of an action row given its category byte, as
// CardGroup.to_cardgroup reports them (card.py:327-335; value = rank of the main group's
// first card, len = chain length, rocket value 100).
__device__ __forceinline__ uint32_t info_of_row(uint64_t nib, int cat) {
  if (cat == EMPTY) return mk_info(EMPTY, 0, 1);
  if (cat == BIGBANG) return mk_info(BIGBANG, 100, 1);
  // main-group multiplicity per category, 4 bits each; cat 14..0: 4,4,1,3,3,3,2,1,3,3,4,3,2,1,0
  const uint64_t MC = 0x0441333213343210ull;
  int mc = (int)((MC >> (4 * cat)) & 15);
  uint32_t m = ge_mask(nib, mc);
  return mk_info(cat, __builtin_ctz(m | 0x8000u), __builtin_popcount(m));
}

// exact-structure classifier of an arbitrary count vector: the (category, value, len)
// CardGroup.to_cardgroup (card.py:327-335 over analyze :372-527) gives for the rows of
// the action space, INFO_INVALID for every other vector.
__device__ inline uint32_t classify(uint64_t nib) {
  if (nib == 0) return mk_info(EMPTY, 0, 1);
  uint32_t g1 = ge_mask(nib, 1), g2 = ge_mask(nib, 2), g3 = ge_mask(nib, 3), g4 = ge_mask(nib, 4);
  if (ge_mask(nib, 5) || (g2 & JOKERS) || (nib >> 60)) return INFO_INVALID;
  uint32_t e1 = g1 & ~g2, e2 = g2 & ~g3, e3 = g3 & ~g4, e4 = g4;
  int n1 = __builtin_popcount(e1), n2 = __builtin_popcount(e2), n3 = __builtin_popcount(e3),
      n4 = __builtin_popcount(e4);
  auto chain = [](uint32_t m) {  // contiguous and within 3..A
    uint32_t s = m >> __builtin_ctz(m);
    return (s & (s + 1)) == 0 && (m & ~M12) == 0;
  };
  if (!e2 && !e3 && !e4) {
    if (n1 == 1) return mk_info(SINGLE, __builtin_ctz(e1), 1);
    if (e1 == JOKERS) return mk_info(BIGBANG, 100, 1);
    if (n1 >= 5 && chain(e1)) return mk_info(SINGLE_LINE, __builtin_ctz(e1), n1);
    return INFO_INVALID;
  }
  if (!e1 && !e3 && !e4) {
    if (n2 == 1) return mk_info(DOUBLE, __builtin_ctz(e2), 1);
    if (n2 >= 3 && n2 <= 10 && chain(e2)) return mk_info(DOUBLE_LINE, __builtin_ctz(e2), n2);
    return INFO_INVALID;
  }
  if (!e1 && !e2 && !e4) {
    if (n3 == 1) return mk_info(TRIPLE, __builtin_ctz(e3), 1);
    if (n3 >= 2 && n3 <= 6 && chain(e3)) return mk_info(TRIPLE_LINE, __builtin_ctz(e3), n3);
    return INFO_INVALID;
  }
  if (!e1 && !e2 && !e3) return n4 == 1 ? mk_info(QUADRIC, __builtin_ctz(e4), 1) : INFO_INVALID;
  if (e3 && !e4) {
    if (!e2 && n1 == n3) {
      if (n3 == 1) return mk_info(THREE_ONE, __builtin_ctz(e3), 1);
      if (n3 <= 5 && chain(e3) && (DDZ_NATIVE_JOKER_KICKERS || !(n3 == 2 && e1 == JOKERS)))
        return mk_info(THREE_ONE_LINE, __builtin_ctz(e3), n3);
    }
    if (!e1 && n2 == n3) {
      if (n3 == 1) return mk_info(THREE_TWO, __builtin_ctz(e3), 1);
      if (n3 <= 4 && chain(e3)) return mk_info(THREE_TWO_LINE, __builtin_ctz(e3), n3);
    }
    return INFO_INVALID;
  }
  if (n4 == 1 && !e3) {
    if (!e2 && n1 == 2 && (DDZ_NATIVE_JOKER_KICKERS || e1 != JOKERS)) return mk_info(FOUR_TAKE_ONE, __builtin_ctz(e4), 1);
    if (!e1 && n2 == 2) return mk_info(FOUR_TAKE_TWO, __builtin_ctz(e4), 1);
  }
  return INFO_INVALID;
}

// ---- follow filter (CardGroup.bigger_than, card.py:307-325) ------------------------
// `L` = info of the combo to beat (cat 0 = lead).  allowed(c, v, l): may a combo of
// category c, value v, len l be played?  (pass is handled by the callers.)
struct Follow {
  int lc, lv, ll;
  bool lead;
};
__device__ __forceinline__ Follow follow_of(uint32_t info) {
  Follow f;
  f.lc = info & 0xFF; f.lv = (info >> 8) & 0xFF; f.ll = (info >> 16) & 0xFF;
  f.lead = f.lc == EMPTY;
  return f;
}
// ranks whose value may be played as the main rank of category c (before the len test)
__device__ __forceinline__ uint32_t value_gate(const Follow& f, int c) {
  if (f.lead) return M15;
  if (c == QUADRIC) return f.lc == QUADRIC ? gt_mask(f.lv) : (f.lc == BIGBANG ? 0u : M15);
  return c == f.lc ? gt_mask(f.lv) : 0u;
}

// ---- Philox4x32-10 -----------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t h0 = __umulhi(0xD2511F53u, c.x), l0 = 0xD2511F53u * c.x;
    uint32_t h1 = __umulhi(0xCD9E8D57u, c.z), l1 = 0xCD9E8D57u * c.z;
    c = make_uint4(h1 ^ c.y ^ k0, l1, h0 ^ c.w ^ k1, l0);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

}  // namespace ddz
