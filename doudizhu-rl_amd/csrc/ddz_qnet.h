// ddz_qnet.h -- the "needed rows" form of the ragged Q forward (BASELINE configs[2]: NetCooperationSimplify inference in the
// loop; net.py:81-102, dqn.py:50-71, game.py:95-104), device side.  Included by ddz_engine.hip inside its anonymous
// namespace, after k_q_slab / k_q_feat (it uses QH, QP_COLS, qp_col, pack_row, ge_mask, rl64, rfl, wave_sum_f32).
//
// With the first layer factorised per (table, rank, count) (ddz_engine.hip k_q_feat, dqn_glue.FactorisedQ):
//     fc1 pre-activation of action j of table t = tab[t] + sum_r W2[r]^T Y[t][r][cnt_jr]          (+ Z[r][cnt_jr], weights only)
//                                               = H0[t]  + sum_{r touched by j} ( D[t][r][cnt_jr] + Z[r][cnt_jr] )
//     H0[t]      = tab[t] + sum_r W2[r]^T Y[t][r][0]            ONE dense GEMM [T, 15 * 256] x [15 * 256, 256], K = 3840
//     D[t][r][c] = W2[r]^T (Y[t][r][c] - Y[t][r][0]),  c >= 1   only for the (r, c) some LEGAL MOVE of table t uses
// The rows of D are the "needed rows": a follow needs one or two (the ranks / counts of the moves that beat `last`), a lead
// about as many as the actor holds cards; 15 count-0 rows per table collapse into H0.  Against the packed form of round 3
// (15 + cards-in-hand rows per table through fifteen GEMMs, segment sizes crossing to the host every iteration) this is
// ~40 % fewer fc1 FLOP, no per-row count-0 reads in the row stage, and NOTHING on the host: the needed rows are found on the
// device from the slab lists (k_q_need_*), laid out in rank segments whose starts live in device memory, and multiplied by a
// hand-written fp32 MFMA kernel (k_fc1: v_mfma_f32_32x32x2_f32, exact f32 = a k-ordered fmaf chain) that reads the segment
// table itself.  The same kernel does the dense GEMM.
#pragma once

// ---- 1. which rows are needed: three small launches, deterministic layout ---------------------------------------------
// need[t] = 64-bit set over the columns of row_index (QP_COLS: (r < 13, c = 1..4) at 4 r + c - 1, the jokers' count 1 at
// 52 / 53): column set <=> some legal move of table t takes exactly c cards of rank r.
constexpr int QN_TPB = 64;           // tables per block of the need kernels (four lanes per table)
constexpr int QN_SEG_WORDS = 40;     // seg[0..15] first row of rank r's segment (multiples of FC_M), [15] = rows in use,
                                     // seg[16..31] first TILE of rank r, [31] = tiles in use, seg[32] = rows needed, [33] = overflow
// k_fc1's geometry (compile-time; tools/fc1_probe.py builds and times the variants): FC_WAVES wavefronts per block, each
// owning FC_TM x FC_TN tiles of 32 x 32; the block covers all 256 outputs, so FC_WN = 8 / FC_TN waves sit side by side and
// the tile is FC_M = (FC_WAVES / FC_WN) * 32 * FC_TM rows; K in chunks of FC_K; FC_OCC blocks per CU (launch bounds)
#ifndef DDZ_FC_WAVES
#define DDZ_FC_WAVES 4
#define DDZ_FC_TM 1
#define DDZ_FC_TN 8
#define DDZ_FC_KC 16
#define DDZ_FC_OCC 2
#endif
constexpr int FC_WAVES = DDZ_FC_WAVES, FC_TM = DDZ_FC_TM, FC_TN = DDZ_FC_TN, FC_K = DDZ_FC_KC, FC_OCC = DDZ_FC_OCC, FC_N = 256;
constexpr int FC_WN = 8 / FC_TN, FC_M = FC_WAVES / FC_WN * 32 * FC_TM;

__device__ __forceinline__ uint64_t q_need_of_row(uint64_t nib) {
  // nibble c in 1..4 of rank r < 13 -> bit c - 1 of the same nibble (SWAR over the 13 nibbles); jokers -> bits 52, 53
  const uint64_t M = 0x0001111111111111ull;  // bit 0 of nibbles 0..12
  const uint64_t b0 = nib & M, b1 = (nib >> 1) & M, b2 = (nib >> 2) & M;
  const uint64_t e1 = b0 & ~b1 & ~b2, e2 = b1 & ~b0 & ~b2, e3 = b0 & b1 & ~b2, e4 = b2;
  uint64_t out = e1 | (e2 << 1) | (e3 << 2) | (e4 << 3);
  out |= (uint64_t)(((nib >> 52) & 15u) != 0) << 52;
  out |= (uint64_t)(((nib >> 56) & 15u) != 0) << 53;
  return out;
}
// per-rank row counts of a need set, packed: 15 fields of 12 bits in three words (5 ranks each; a block of 64 tables sums to
// at most 256 per rank)
__device__ __forceinline__ void q_need_counts(uint64_t need, uint64_t w[3]) {
  w[0] = w[1] = w[2] = 0;
#pragma unroll
  for (int r = 0; r < 15; ++r) {
    const uint32_t c = r < 13 ? (uint32_t)__popc((uint32_t)(need >> (4 * r)) & 15u) : (uint32_t)((need >> (52 + r - 13)) & 1u);
    w[r / 5] |= (uint64_t)c << (12 * (r % 5));
  }
}
__device__ __forceinline__ uint32_t q_need_field(const uint64_t w[3], int r) { return (uint32_t)(w[r / 5] >> (12 * (r % 5))) & 0xFFFu; }

// (a) four lanes per table (a wave = 16 tables, a block = 64): lane q of a table ORs rows q, q + 4, ... of its slab list (the
// four 16-byte loads of a quad are one 64-byte run), two DPP steps join the quad -> need[t]; per block the per-rank row
// counts -> blk[b][16]
__device__ __forceinline__ uint64_t quad_or(uint64_t v) {   // OR over the 4 lanes of a quad, the result in all four
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo |= (uint32_t)__builtin_amdgcn_mov_dpp((int)lo, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  hi |= (uint32_t)__builtin_amdgcn_mov_dpp((int)hi, 0xB1, 0xf, 0xf, false);
  lo |= (uint32_t)__builtin_amdgcn_mov_dpp((int)lo, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  hi |= (uint32_t)__builtin_amdgcn_mov_dpp((int)hi, 0x4E, 0xf, 0xf, false);
  return ((uint64_t)hi << 32) | lo;
}
__global__ __launch_bounds__(256) void k_q_need_mask(const int32_t* __restrict__ counts, const uint4* __restrict__ rows,
                                                     int64_t stride, int64_t T, uint64_t* __restrict__ need,
                                                     int32_t* __restrict__ blk) {
  const int q = threadIdx.x & 3;
  const int64_t t = (int64_t)blockIdx.x * QN_TPB + (threadIdx.x >> 2);
  uint64_t m = 0;
  if (t < T) {
    int n = counts[t];
    if (n < 0 || n > stride) n = 0;
    const uint4* lr = rows + t * stride;
    for (int j = q; j < n; j += 4) m |= q_need_of_row(pack_row(lr[j]));
  }
  m = quad_or(m);
  if (t < T && q == 0) need[t] = m;
  uint64_t w[3];
  q_need_counts(q == 0 ? m : 0, w);   // (one lane per table counts)
  __shared__ uint64_t s[3][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 3; ++k) {  // wave sum of the packed fields (no field overflows: <= 16 * 4 per wave)
    uint64_t v = w[k];
    for (int d = 32; d > 0; d >>= 1) v += ((uint64_t)__shfl_xor((uint32_t)(v >> 32), d) << 32) | (uint64_t)__shfl_xor((uint32_t)v, d);
    if (lane == 0) s[k][wv] = v;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    uint64_t tot[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k)
      for (int i = 0; i < 4; ++i) tot[k] += s[k][i];
    blk[(int64_t)blockIdx.x * 16 + threadIdx.x] = threadIdx.x < 15 ? (int32_t)q_need_field(tot, (int)threadIdx.x) : 0;
  }
}

// (b) one block: exclusive scan of blk[][r] over the blocks (in place) and the segment table.  Rank r's segment starts at a
// multiple of FC_M (a k_fc1 tile never straddles two ranks); rows behind a segment's last needed row are padding.
__global__ __launch_bounds__(256) void k_q_need_scan(int32_t* __restrict__ blk, int64_t nblk, int32_t* __restrict__ seg,
                                                     int64_t row_capacity) {
  __shared__ int32_t s_part[256][16];
  __shared__ int32_t s_tot[16];
  const int tid = threadIdx.x;
  const int64_t per = (nblk + 255) / 256;   // blocks per thread, consecutive
  int32_t acc[15];
#pragma unroll
  for (int r = 0; r < 15; ++r) acc[r] = 0;
  for (int64_t b = tid * per; b < (tid + 1) * per && b < nblk; ++b)
#pragma unroll
    for (int r = 0; r < 15; ++r) acc[r] += blk[b * 16 + r];
#pragma unroll
  for (int r = 0; r < 15; ++r) s_part[tid][r] = acc[r];
  __syncthreads();
  {  // exclusive scan over the 256 partials of every rank: wave w scans ranks 4 w .. 4 w + 3, 64 lanes x 4 partials each
    const int lane = tid & 63, wv = tid >> 6;
    for (int r = 4 * wv; r < 4 * wv + 4 && r < 15; ++r) {
      int32_t v[4], sum = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = s_part[4 * lane + i][r]; sum += v[i]; }
      const int32_t inc = wave_scan_add(sum);
      int32_t run = inc - sum;
#pragma unroll
      for (int i = 0; i < 4; ++i) { s_part[4 * lane + i][r] = run; run += v[i]; }
      if (lane == 63) s_tot[r] = inc;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 15; ++r) acc[r] = s_part[tid][r];
  for (int64_t b = tid * per; b < (tid + 1) * per && b < nblk; ++b)
#pragma unroll
    for (int r = 0; r < 15; ++r) { const int32_t v = blk[b * 16 + r]; blk[b * 16 + r] = acc[r]; acc[r] += v; }
  if (tid == 0) {
    int64_t row = 0, needed = 0;
    int32_t over = 0;
    for (int r = 0; r < 15; ++r) {
      seg[r] = (int32_t)row;
      seg[16 + r] = (int32_t)(row / FC_M);
      needed += s_tot[r];
      int64_t len = ((int64_t)s_tot[r] + FC_M - 1) / FC_M * FC_M;
      if (row + len > row_capacity) { len = (row_capacity - row) / FC_M * FC_M; over = 1; }  // (k_q_need_assign drops what does not fit)
      row += len;
    }
    seg[15] = (int32_t)row;
    seg[31] = (int32_t)(row / FC_M);
    seg[32] = (int32_t)needed;
    seg[33] = over;
  }
}

// (c) four lanes per table: row of every needed (r, c) = seg[r] + rows of rank r in the blocks before + in the tables before
// inside the block + position inside the table (ascending c); lane q of a table writes columns 16 q .. 16 q + 15 of
// row_index[t][64] (-1 = not needed): a wave stores 16 tables x 256 bytes = one 4-KB run.
__global__ __launch_bounds__(256) void k_q_need_assign(const uint64_t* __restrict__ need, int64_t T, const int32_t* __restrict__ blk,
                                                       const int32_t* __restrict__ seg, int32_t* __restrict__ row_index,
                                                       uint8_t* __restrict__ row_cnt, int32_t* __restrict__ status) {
  // (row_cnt, may be null: the count c of every needed row -- k_fc1<true> adds z[rank][c] to the row it computes)
  const int q = threadIdx.x & 3;
  const int64_t t = (int64_t)blockIdx.x * QN_TPB + (threadIdx.x >> 2);
  const uint64_t m = t < T ? need[t] : 0;
  uint64_t w[3], inc[3];
  q_need_counts(m, w);                 // (the same in the four lanes of a table)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __shared__ uint64_t s[3][4];
#pragma unroll
  for (int k = 0; k < 3; ++k) {  // inclusive scan over the wave's 16 tables (quads): shifts by 4, 8, 16, 32 lanes
    uint64_t v = w[k];
    for (int d = 4; d < 64; d <<= 1) {
      const uint64_t o = ((uint64_t)__shfl_up((uint32_t)(v >> 32), d) << 32) | (uint64_t)__shfl_up((uint32_t)v, d);
      if (lane >= d) v += o;
    }
    inc[k] = v;
    if (lane == 63) s[k][wv] = v;
  }
  __syncthreads();
  uint64_t ex[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    uint64_t before = 0;
    for (int i = 0; i < wv; ++i) before += s[k][i];
    ex[k] = before + inc[k] - w[k];
  }
  if (t >= T) return;
  int32_t out[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) out[i] = -1;
  bool dropped = false;
#pragma unroll
  for (int i = 0; i < 4; ++i) {        // lane q: ranks 4 q .. 4 q + 3 (q = 3: rank 12, then the two jokers' single columns)
    const int r = 4 * q + i;
    if (r < 13) {
      const int32_t hi = seg[r + 1];
      int32_t row = seg[r] + blk[(int64_t)blockIdx.x * 16 + r] + (int32_t)q_need_field(ex, r);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if ((m >> (4 * r + c)) & 1u) {
          if (row < hi) { out[4 * i + c] = row; if (row_cnt) row_cnt[row] = (uint8_t)(c + 1); }
          else dropped = true;
          ++row;
        }
    }
  }
  if (q == 3) {
#pragma unroll
    for (int r = 13; r < 15; ++r)
      if ((m >> (52 + r - 13)) & 1u) {
        const int32_t row = seg[r] + blk[(int64_t)blockIdx.x * 16 + r] + (int32_t)q_need_field(ex, r);
        if (row < seg[r + 1]) { out[4 + r - 13] = row; if (row_cnt) row_cnt[row] = 1; }
        else dropped = true;
      }
  }
  int4* dst = (int4*)(row_index + t * QP_COLS + 16 * q);
#pragma unroll
  for (int i = 0; i < 4; ++i) dst[i] = make_int4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
  if (dropped) atomicOr(status, 2);   // row capacity overflow (cannot happen with capacity >= 20 T + 15 * FC_M)
}

// ---- 2. first layer for the needed rows ----------------------------------------------------------------------------------
// As k_q_feat (ddz_engine.hip), writing  y0[t][r][c] = Y[t][r][0][c]  (dense [T][15 * 256]: the A operand of the K = 3840 GEMM)
// and, for every needed (r, cnt >= 1) of the table,  dy[row][c] = Y[t][r][cnt][c] - Y[t][r][0][c]  at row = row_index[t][col].
// (Two channels per thread as the halves of packed-fp32 registers -- 30 v_pk_fma_f32 instead of 60 v_fma_f32 per (table, rank),
// the `face` value broadcast by op_sel, no moves, bit-identical results -- measured SLOWER: 524 against 481 us at 65,536 tables
// (190 VGPRs: two waves per SIMD instead of four, and the packed instruction does not issue faster than two plain ones).)
template <int P>
__global__ __launch_bounds__(QH) void k_q_feat_needed(   // ((QH, 5): 96 VGPRs + three spilled dwords, five waves per SIMD -- 536 against 466 us)
    const float4* __restrict__ face, int64_t T, const float* __restrict__ wf,
                                                      const float* __restrict__ bias, const float* __restrict__ acnt,
                                                      float* __restrict__ y0, float* __restrict__ dy, int64_t dy_rows,
                                                      const int32_t* __restrict__ pidx) {
  const int c = threadIdx.x;
  __shared__ float4 s_face[QF_TILE * P * 15];
  const int64_t tb = (int64_t)blockIdx.x * QF_TILE;
  const int nt = (int)(T - tb < QF_TILE ? T - tb : QF_TILE);
  for (int i = threadIdx.x; i < nt * P * 15; i += QH) s_face[i] = face[tb * P * 15 + i];
  __shared__ __attribute__((aligned(16))) int32_t s_pidx[QF_TILE * QP_COLS];
  for (int i = threadIdx.x; i < nt * QP_COLS; i += QH) s_pidx[i] = pidx[tb * QP_COLS + i];
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
    for (int n = 0; n < 4; ++n) a[n][k] = acnt[((n + 1) * 4 + k) * QH + c];
  }
  __syncthreads();
  const uint32_t nr = (uint32_t)dy_rows;
  for (int ti = 0; ti < nt; ++ti) {   // table-major: a table's fifteen count-0 rows are one 15-KB run of y0
    float* d0 = y0 + (tb + ti) * (15 * QH) + c;
#pragma unroll 3
    for (int r = 0; r < 15; ++r) {   // (three ranks per trip: their LDS broadcasts and FMA chains interleave)
      int4 pr4 = make_int4(-1, -1, -1, -1);
      if (r < 13) pr4 = *(const int4*)&s_pidx[ti * QP_COLS + 4 * r];
      else pr4.x = s_pidx[ti * QP_COLS + 52 + (r - 13)];
      // y0 == null (the shared-rows form, section 5: the count-0 rows come from k_q_feat_rows): only the ranks some legal move
      // takes cards of are evaluated (block-uniform: every thread of the block looks at the same table and rank)
      if (!y0 && (pr4.x & pr4.y & pr4.z & pr4.w) == -1) continue;
      float s0 = b[0], s1 = b[1], s2 = b[2], s3 = b[3];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const float4 x = s_face[(ti * P + p) * 15 + r];
        s0 += w[p][0] * x.x;
        s1 += w[p][1] * x.x + w[p][2] * x.y;
        s2 += w[p][3] * x.x + w[p][4] * x.y + w[p][5] * x.z;
        s3 += w[p][6] * x.x + w[p][7] * x.y + w[p][8] * x.z + w[p][9] * x.w;
      }
      const float v0 = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
      if (y0) d0[r * QH] = v0;
      // (unsigned compares: -1 and anything beyond the buffer are skipped alike; row_index is device data)
      if ((uint32_t)pr4.x < nr) dy[(uint32_t)pr4.x * (uint32_t)QH + c] = fmaxf(fmaxf(s0 + a[0][0], s1 + a[0][1]), fmaxf(s2 + a[0][2], s3 + a[0][3])) - v0;
      if ((uint32_t)pr4.y < nr) dy[(uint32_t)pr4.y * (uint32_t)QH + c] = fmaxf(fmaxf(s0 + a[1][0], s1 + a[1][1]), fmaxf(s2 + a[1][2], s3 + a[1][3])) - v0;
      if ((uint32_t)pr4.z < nr) dy[(uint32_t)pr4.z * (uint32_t)QH + c] = fmaxf(fmaxf(s0 + a[2][0], s1 + a[2][1]), fmaxf(s2 + a[2][2], s3 + a[2][3])) - v0;
      if ((uint32_t)pr4.w < nr) dy[(uint32_t)pr4.w * (uint32_t)QH + c] = fmaxf(fmaxf(s0 + a[3][0], s1 + a[3][1]), fmaxf(s2 + a[3][2], s3 + a[3][3])) - v0;
    }
  }
}

// ---- 3. fc1 on the matrix cores, exact f32 ---------------------------------------------------------------------------------
// C[m][0..255] (+)= sum_k A[m][k] * B[k][0..255] with v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit for bit a k-ordered
// fmaf chain; 64 cycles per instruction and SIMD = the fp32 vector peak, 157.3 TFLOP/s).  Block = 256 threads = 4 wavefronts,
// ONE block per CU (one wave per SIMD, the whole register file: 256 accumulator registers per lane), tile = 256 rows x all 256
// outputs (A is read from HBM once), wave (wm, wn) owns rows 128 wm .. + 127 x columns 128 wn .. + 127 = 4 x 4 accumulators of
// 16 VGPRs: per k-slice eight operand registers feed sixteen MFMAs.  K runs in chunks of FC_K = 16 through two LDS buffers (A
// 256 x 17, B 16 x 288 floats each, padded so that the operand reads are conflict-free: lane l reads A[l & 31][k + (l >> 5)]
// and B[k + (l >> 5)][32 n + (l & 31)]); the global loads of chunk i + 1 are issued into registers before chunk i is
// multiplied and stored to the OTHER buffer behind its MFMAs: one barrier per 128 MFMAs of a wave.
// (First version: 128-row tiles, wave tile 32 x 256, two blocks per CU: 125 TFLOP/s on the dense GEMM, its matrix pipes busy
// 79 % of the cycles at 2.22 GHz -- nine operand reads per eight MFMAs and a barrier per 64; profiles/r04_dqn_pmc.json.)
//   ROWS = false: the dense GEMM  H0 [M][256] += Y0 [M][K] x Wd [K][256]  (K = 3840: 240 chunks per tile; C holds the
//                 per-table term on entry and is added in the epilogue).
//   ROWS = true : D [row][256] = dY [row][256] x W2[rank of the row][256][256]: tile b belongs to the rank r with
//                 seg[16 + r] <= b < seg[16 + r + 1] (device memory: nothing crosses to the host), tiles >= seg[31] exit.
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;   // (native vectors: HIP's float4 struct arrays end up in scratch here)
constexpr int FC_AS = FC_K + 1;      // LDS row stride of A (floats)
constexpr int FC_BS = FC_N + 32;     // ... of B
constexpr int FC_THREADS = 64 * FC_WAVES;
constexpr int FC_NA = FC_M * FC_K / 4 / FC_THREADS, FC_NB = FC_K * FC_N / 4 / FC_THREADS;   // float4 per thread and chunk
static_assert(FC_NA >= 1 && FC_NB >= 1 && FC_M * FC_K / 4 % FC_THREADS == 0 && FC_K * FC_N / 4 % FC_THREADS == 0, "chunk / block shape");
__device__ __forceinline__ void fc1_load(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t m0, int64_t M,
                                         int k0, int tid, f32x4 (&ra)[FC_NA], f32x4 (&rb)[FC_NB]) {
  // global -> registers: A row f / (FC_K / 4), k 4 (f % (FC_K / 4)); B row f / 64, columns 4 (f % 64)
#pragma unroll
  for (int i = 0; i < FC_NA; ++i) {
    const int f = tid + FC_THREADS * i, row = f / (FC_K / 4), kq = f % (FC_K / 4);
    const int64_t m = m0 + row < M ? m0 + row : M - 1;   // (rows beyond M: some valid row, never stored -- no branch)
    ra[i] = *(const f32x4*)(A + m * lda + k0 + 4 * kq);
  }
#pragma unroll
  for (int i = 0; i < FC_NB; ++i) {
    const int f = tid + FC_THREADS * i, row = f >> 6, c4 = f & 63;
    rb[i] = *(const f32x4*)(B + (int64_t)(k0 + row) * FC_N + 4 * c4);
  }
}
__device__ __forceinline__ void fc1_stage(float* __restrict__ sA, float* __restrict__ sB, int tid, const f32x4 (&ra)[FC_NA],
                                          const f32x4 (&rb)[FC_NB]) {
#pragma unroll
  for (int i = 0; i < FC_NA; ++i) {
    const int f = tid + FC_THREADS * i, row = f / (FC_K / 4), kq = f % (FC_K / 4);
    float* d = sA + row * FC_AS + 4 * kq;
    d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
  }
#pragma unroll
  for (int i = 0; i < FC_NB; ++i) {
    const int f = tid + FC_THREADS * i, row = f >> 6, c4 = f & 63;
    *(f32x4*)(sB + row * FC_BS + 4 * c4) = rb[i];
  }
}
// one chunk of FC_K: a wave multiplies its FC_TM x FC_TN tiles of 32 x 32 -- per k-slice of 2, FC_TM + FC_TN operand registers
// feed FC_TM x FC_TN MFMAs; the operands of k-slice kk + 2 are read from LDS before the MFMAs of slice kk are issued (explicit
// double buffer: an LDS read has the slice's matrix work to land in)
__device__ __forceinline__ void fc1_chunk(const float* __restrict__ pa, const float* __restrict__ pb, f32x16 (&acc)[FC_TM][FC_TN]) {
  float a0[FC_TM], b0[FC_TN], a1[FC_TM], b1[FC_TN];
#pragma unroll
  for (int i = 0; i < FC_TM; ++i) { a0[i] = pa[32 * i * FC_AS]; a1[i] = 0.f; }
#pragma unroll
  for (int j = 0; j < FC_TN; ++j) { b0[j] = pb[32 * j]; b1[j] = 0.f; }
#pragma unroll
  for (int kk = 0; kk < FC_K; kk += 4) {
#pragma unroll
    for (int i = 0; i < FC_TM; ++i) a1[i] = pa[32 * i * FC_AS + kk + 2];
#pragma unroll
    for (int j = 0; j < FC_TN; ++j) b1[j] = pb[(kk + 2) * FC_BS + 32 * j];
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler would sink the reads next to their MFMAs to save registers)
#pragma unroll
    for (int i = 0; i < FC_TM; ++i)
#pragma unroll
      for (int j = 0; j < FC_TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (kk + 4 < FC_K) {
#pragma unroll
      for (int i = 0; i < FC_TM; ++i) a0[i] = pa[32 * i * FC_AS + kk + 4];
#pragma unroll
      for (int j = 0; j < FC_TN; ++j) b0[j] = pb[(kk + 4) * FC_BS + 32 * j];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < FC_TM; ++i)
#pragma unroll
      for (int j = 0; j < FC_TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <bool ROWS>
__global__ __launch_bounds__(FC_THREADS, FC_OCC) void k_fc1(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                          float* __restrict__ C, int64_t M, int K, const int32_t* __restrict__ seg,
                                                          const float* __restrict__ zfold, const uint8_t* __restrict__ row_cnt,
                                                          int accum = 0) {   // ROWS: accum != 0 adds the product to C (section 5)
  // ROWS with zfold / row_cnt: D[row] = dY[row] x fc1[rank] + z[rank][count of the row] -- the weights-only term of the action
  // plane rides on the row, so the row stage reads ONE 1-KB row per (move, rank) and keeps its LDS for the table's rows
  __shared__ float sA[2][FC_M * FC_AS];
  __shared__ __attribute__((aligned(16))) float sB[2][FC_K * FC_BS];
  __shared__ float sZ[ROWS ? 5 * QH : 1];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv / FC_WN, wn = wv % FC_WN;
  const int64_t m0 = (int64_t)blockIdx.x * FC_M;
  if (ROWS) {
    if ((int)blockIdx.x >= seg[31]) return;
    int r = 0;
#pragma unroll
    for (int q = 1; q < 15; ++q) r += (int)blockIdx.x >= seg[16 + q];
    B += (int64_t)r * K * FC_N;     // (K rows per rank: 256, or 288 with the table term's 24 column rows appended -- section 5)
    M = seg[15];
    if (zfold)
      for (int i = tid; i < 5 * QH; i += FC_THREADS) sZ[i] = zfold[(int64_t)r * 5 * QH + i];   // (visible behind the first barrier)
  }
  f32x4 ra[FC_NA], rb[FC_NB];
  fc1_load(A, lda, B, m0, M, 0, tid, ra, rb);
  f32x16 acc[FC_TM][FC_TN];
  const int crow = 4 * (lane >> 5), ccol = lane & 31;   // C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < FC_TM; ++i)
#pragma unroll
    for (int j = 0; j < FC_TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;
  fc1_stage(sA[0], sB[0], tid, ra, rb);
  __syncthreads();
  const int oa = (32 * FC_TM * wm + (lane & 31)) * FC_AS + (lane >> 5), ob = (lane >> 5) * FC_BS + 32 * FC_TN * wn + (lane & 31);
  int cur = 0;
  for (int k0 = FC_K; k0 < K; k0 += FC_K) {
    // chunk k0 - FC_K is in LDS buffer `cur`; chunk k0 travels through the registers meanwhile and lands in the other
    // buffer behind the MFMAs: ONE barrier per chunk (nobody reads the other buffer before it, nobody writes `cur`)
    fc1_load(A, lda, B, m0, M, k0, tid, ra, rb);
    fc1_chunk(sA[cur] + oa, sB[cur] + ob, acc);
    fc1_stage(sA[cur ^ 1], sB[cur ^ 1], tid, ra, rb);
    __syncthreads();
    cur ^= 1;
  }
  fc1_chunk(sA[cur] + oa, sB[cur] + ob, acc);
#pragma unroll
  for (int i = 0; i < FC_TM; ++i)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int64_t row = m0 + 32 * FC_TM * wm + 32 * i + (g & 3) + 8 * (g >> 2) + crow;
      int zoff = -1;
      if (ROWS && zfold && row < M) {
        uint32_t c = row_cnt[row];
        zoff = (int)(c > 4u ? 0u : c) * QH;     // (padding rows carry whatever count: their result is never read)
      }
#pragma unroll
      for (int j = 0; j < FC_TN; ++j) {
        const int col = 32 * FC_TN * wn + 32 * j + ccol;
        if (row < M)
          C[row * FC_N + col] = ROWS ? (zoff >= 0 ? acc[i][j][g] + sZ[zoff + col] : acc[i][j][g]) + (accum ? C[row * FC_N + col] : 0.f)
                                     : acc[i][j][g] + C[row * FC_N + col];
      }
    }
}

// ---- 4. the per-row stage over the needed rows -------------------------------------------------------------------------------
// q[t][j] = b2 + w2 . relu( H0[t] + sum over the ranks r move j touches of D'[row_index[t][col(r, cnt)]] ),  D' = D + Z[r][cnt]
// (the weights-only term of the action plane is folded into the row by k_fc1<true>).
// One wavefront per table at a time (tpw consecutive tables per wave); the wave's four 16-lane rows work on FOUR MOVES at
// once: lane l of a row owns hidden units {4 (l + 16 k) .. + 3, k = 0..3} (a 1-KB row of H0 / D = four coalesced 256-byte
// reads per 16 lanes), a move's dot product is reduced with four DPP row steps.  THE TABLE'S NEEDED ROWS ARE READ ONCE: a
// (rank, count) row is used by every move that takes that count of the rank (a single by every straight through it ...) -- 4 x
// on average, and between two uses the row had left the caches (1.08 GB fetched for 0.26 GB of rows, profiles/r04_dqn_pmc.json:
// the kernel ran at the speed of those re-reads).  The wave stages the first QS_ROWS needed rows of its table in LDS (slot =
// rank of the row's column among the table's needed columns); rows beyond that come from memory as before.  A column that
// is not set (a list that does not belong to this row_index) or points beyond the buffer contributes nothing and raises
// status bit 5.
constexpr int QS_ROWS = 8;    // needed rows of a table kept in the wave's LDS cache (8 KB per wave; mean 3.8 per table)
constexpr int QS_BIG = 24;    // ... in the block's cache of a HEAVY table (a move takes what the actor holds: <= 20 rows)
constexpr int QS_HEAVY_MOVES = 64;
// the moves [j_first, n) of one table, four at a time (the wave's 16-lane rows), stepping j_step; rows cached in `cache`
// (slot < ncache) or read from memory
__device__ __forceinline__ void q_slab_moves(const float4* __restrict__ D, const uint4* __restrict__ lrow, float* __restrict__ qt, int n,
                                             int j_first, int j_step, int32_t myidx, uint64_t needm, const float4 (&h0)[4],
                                             const float4 (&w)[4], float bias, const float4 (*cache)[QH / 4], int ncache, int lane,
                                             int32_t* __restrict__ status) {
  const int l16 = lane & 15, quarter = lane >> 4;
  uint4 row_cur = j_first + quarter < n ? lrow[j_first + quarter] : make_uint4(0, 0, 0, 0);   // (a row's 16 lanes read the same 16 bytes)
  for (int j0 = j_first; j0 < n; j0 += j_step) {
    const int j = j0 + quarter;
    const bool have = j < n;
    const uint64_t nib = have ? pack_row(row_cur) : 0;
    if (j + j_step < n) row_cur = lrow[j + j_step];        // the next trip's move, in flight during this one
    float4 h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = h0[k];
    uint32_t tm = ge_mask(nib, 1);                        // the ranks this row's move touches
    while (__ballot(tm != 0)) {                           // wave-uniform trip count: the row with the most ranks
      const bool act = tm != 0;
      const int r = act ? __builtin_ctz(tm) : 0;
      uint32_t c = (uint32_t)(nib >> (4 * r)) & 15u;
      c = c > 4u ? 4u : c;
      if (r >= 13 && c > 1u) c = 1u;
      const int col = act ? qp_col(r, (int)c) : 63;       // (column 63 is never set)
      const int32_t pr = __shfl(myidx, col);              // every lane active here
      if (act) {
        const bool ok = (needm >> col) & 1ull;
        const int slot = __popcll(needm & ((1ull << col) - 1ull));
        if (!ok) {
          if (l16 == 0) atomicOr(status, 32);
        } else if (slot < ncache) {
          const float4* cr = cache[slot] + l16;
#pragma unroll
          for (int k = 0; k < 4; ++k) { const float4 v = cr[16 * k]; h[k].x += v.x; h[k].y += v.y; h[k].z += v.z; h[k].w += v.w; }
        } else {
          const float4* dr = D + (int64_t)pr * (QH / 4) + l16;
#pragma unroll
          for (int k = 0; k < 4; ++k) { const float4 v = dr[16 * k]; h[k].x += v.x; h[k].y += v.y; h[k].z += v.z; h[k].w += v.w; }
        }
      }
      tm &= tm - 1;
    }
    float p = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      p += fmaxf(h[k].x, 0.f) * w[k].x + fmaxf(h[k].y, 0.f) * w[k].y + fmaxf(h[k].z, 0.f) * w[k].z + fmaxf(h[k].w, 0.f) * w[k].w;
    // sum over the 16 lanes of the row: four DPP row shifts, the total lands in lane 15 of the row
    p += __int_as_float(DDZ_DPP(0, __float_as_int(p), 0x111, 0xf));
    p += __int_as_float(DDZ_DPP(0, __float_as_int(p), 0x112, 0xf));
    p += __int_as_float(DDZ_DPP(0, __float_as_int(p), 0x114, 0xf));
    p += __int_as_float(DDZ_DPP(0, __float_as_int(p), 0x118, 0xf));
    if (have && l16 == 15) qt[j] = p + bias;
  }
}
__global__ __launch_bounds__(TB, 4) void k_q_slab_needed(const float4* __restrict__ H0, const float4* __restrict__ D, int64_t d_rows,
                                                        int64_t T, int tpw, const float4* __restrict__ w2,
                                                        const float* __restrict__ b2, const int32_t* __restrict__ counts,
                                                        const uint4* __restrict__ rows, int64_t stride, float* __restrict__ q,
                                                        const int32_t* __restrict__ pidx, int32_t* __restrict__ status) {
  __shared__ float4 s_cache[WPB][QS_ROWS][QH / 4];
  __shared__ float4 s_big[QS_BIG][QH / 4];
  __shared__ int32_t s_heavy[WPB * 8];   // tables of this block left to the whole block (tpw <= 8)
  __shared__ uint32_t s_nheavy, s_next;
  const int lane = threadIdx.x & 63, l16 = lane & 15;
  const int wv = (int)rfl(threadIdx.x >> 6);
  if (threadIdx.x == 0) { s_nheavy = 0; s_next = WPB; }
  __syncthreads();
  const int64_t tb = (int64_t)blockIdx.x * WPB * tpw;    // the block's first table
  const int nblk_tab = tb < T ? (int)(T - tb < (int64_t)WPB * tpw ? T - tb : (int64_t)WPB * tpw) : 0;   // ... and how many it has
  float4 (*cache)[QH / 4] = s_cache[wv];
  float4 w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = w2[l16 + 16 * k];
  const float bias = b2[0];
  // ---- phase 1: the block's tables are handed out one by one (an LDS ticket: a table costs between one memory round trip
  // and dozens -- with a fixed share per wave, half of the wave-slots of the launch stood idle behind the unluckiest wave of
  // every block).  A HEAVY table -- more needed rows than the wave's cache holds, or a long list: a fresh deal's 20-card
  // lead has ~20 rows and 70 .. 400 moves -- is left to phase 2.  The next table's list size and row_index travel while the
  // current table is evaluated.
  int cur = wv < nblk_tab ? wv : -1;                     // the first WPB tables need no ticket
  int n_next = cur >= 0 ? counts[tb + cur] : 0;
  int32_t idx_next = cur >= 0 ? pidx[(tb + cur) * QP_COLS + lane] : -1;
  while (cur >= 0) {
    const int64_t t = tb + cur;
    int n = (int)rfl((uint32_t)n_next);
    const int32_t myidx = idx_next;                     // lane L holds column L of the table's row_index
    {
      uint32_t nx = 0;
      if (lane == 0) nx = atomicAdd(&s_next, 1u);
      nx = rfl(nx);
      cur = nx < (uint32_t)nblk_tab ? (int)nx : -1;
      if (cur >= 0) { n_next = counts[tb + cur]; idx_next = pidx[(tb + cur) * QP_COLS + lane]; }
    }
    if (n < 0 || n > stride) n = 0;
    if (n == 0) continue;
    const bool mine = (uint32_t)myidx < (uint32_t)d_rows;
    const uint64_t needm = __ballot(mine);              // the table's needed columns (with a valid row)
    if (__popcll(needm) > QS_ROWS || n > QS_HEAVY_MOVES) {
      if (lane == 0) s_heavy[atomicAdd(&s_nheavy, 1u)] = (int32_t)(t - tb);
      continue;
    }
    // stage the needed rows: slot s = the s-th set column.  ALL loads are issued before the first one is stored (a
    // load-store pair per trip would cost a memory round trip per row)
    float4 h0[4];
    {
      uint64_t m = needm;
      float4 tmp[QS_ROWS];
#pragma unroll
      for (int s_ = 0; s_ < QS_ROWS; ++s_) {
        const bool on = m != 0;                           // wave-uniform
        const int col = on ? __builtin_ctzll(m) : 0;
        const int32_t pr = (int32_t)__builtin_amdgcn_readlane(myidx, col);
        tmp[s_] = on ? D[(int64_t)pr * (QH / 4) + lane] : make_float4(0.f, 0.f, 0.f, 0.f);   // one coalesced 1-KB read
        m &= m - 1;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) h0[k] = H0[t * (QH / 4) + l16 + 16 * k];
      __builtin_amdgcn_wave_barrier();                  // (the previous table's reads of the cache are done: same wave, in order)
#pragma unroll
      for (int s_ = 0; s_ < QS_ROWS; ++s_) cache[s_][lane] = tmp[s_];
      __builtin_amdgcn_wave_barrier();
    }
    q_slab_moves(D, rows + t * stride, q + t * stride, n, 0, 4, myidx, needm, h0, w, bias, cache, QS_ROWS, lane, status);
  }
  __syncthreads();
  // ---- phase 2: the block's heavy tables, one after the other, by ALL its waves: the rows staged once in the block's cache
  // (wave w loads rows w, w + 16), wave w evaluates moves 4 w .. 4 w + 3, then + 64, ...
  const int nheavy = (int)s_nheavy;                      // (block-uniform)
  for (int hi = 0; hi < nheavy; ++hi) {
    const int64_t t = tb + s_heavy[hi];
    int n = counts[t];
    if (n < 0 || n > stride) n = 0;
    const int32_t myidx = pidx[t * QP_COLS + lane];
    const bool mine = (uint32_t)myidx < (uint32_t)d_rows;
    const uint64_t needm = __ballot(mine);
    for (int s_ = wv; s_ < QS_BIG; s_ += WPB) {          // the s-th set column, if there is one
      uint64_t m = needm;
      for (int k = 0; k < s_ && m; ++k) m &= m - 1;
      if (m) {
        const int32_t pr = (int32_t)__builtin_amdgcn_readlane(myidx, __builtin_ctzll(m));
        s_big[s_][lane] = D[(int64_t)pr * (QH / 4) + lane];
      }
    }
    float4 h0[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) h0[k] = H0[t * (QH / 4) + l16 + 16 * k];
    __syncthreads();
    q_slab_moves(D, rows + t * stride, q + t * stride, n, 4 * wv, 4 * WPB, myidx, needm, h0, w, bias, s_big, QS_BIG, lane, status);
    __syncthreads();                                     // (the cache is free for the next heavy table)
  }
}

// ---- 5. the count-0 term H0 from SHARED rows (faces of EnvCooperationSimplify, envi.py:201-217) ------------------------------
// H0[t] = tab[t] + sum_r Y[t][r][0] x fc1[r]  is a [T][3840] x [3840][256] product -- but Y[t][r][0] depends only on the face
// COLUMN of rank r, and that column is a function of (hand_r, taken_r, b1_r, b2_r, n1, n2) alone (k_observe<3>: four thermometer
// planes and the two prob planes  [known <= j < total] n / (n1 + n2)  with known = hand_r + taken_r).  Across the tables of a
// batch the columns of a rank repeat: 65,536 tables hold 983,040 (table, rank) columns and ~5 % as many DISTINCT (rank, column)
// pairs (measured mid-game under the greedy network; 9 % under the random policy).  So:
//   one row per distinct (rank, column):  first layer (k_q_feat_rows) and  G[row] = Y[row] x fc1[rank]  (k_fc1<true>, rank
//   segments as for the needed rows) over ~5 * 10^4 rows instead of the K = 3840 product over every table, and
//   H0[t] = tab[t] + sum_r G[row(t, r)]   (k_qs_gather: fifteen 1-KB rows per table, summed in rank order).
// Every table still gets its exact H0 every iteration from the current weights -- nothing is cached across calls; what is
// removed is the recomputation of identical subexpressions (equal to the dense form up to fp32 summation order: fifteen K = 256
// chains added in rank order instead of one K = 3840 chain; tests: 1e-5).
// Rows are found by direct addressing: slot[key], key = rank * QSH_COLS + column code (4,134,375 int32 slots, 16.5 MB):
//   k_qs_mark   every (table, rank) writes its instance number into its slot (any winner: equal keys <=> equal columns)
//   k_qs_count / k_qs_seg / k_qs_assign   occupied slots per 2048-key chunk -> rank segments (starts multiples of the GEMM tile)
//               and chunk bases -> row number of every occupied slot in KEY ORDER (deterministic), rep[row] = an instance
//   k_qs_rows   rows[t][r] = slot[key(t, r)] - 1
constexpr int QSH_COLS = 625 * 441;                              // (hand, taken, b1, b2) in 0..4 each x (n1, n2) in 0..20 each
constexpr int QSH_KEYS = 15 * QSH_COLS;
constexpr int QSH_CHUNK = 2048;                                  // keys per block of the count / assign kernels (8 per thread)
constexpr int QSH_CPR = (QSH_COLS + QSH_CHUNK - 1) / QSH_CHUNK;     // chunks per rank (135): a chunk never straddles two ranks
constexpr int QSH_WS_INTS = QSH_KEYS + 2 * 15 * QSH_CPR;           // workspace: slots | cnt[15][QSH_CPR] | base[15][QSH_CPR]

__global__ __launch_bounds__(256) void k_qs_mark(const uint8_t* __restrict__ state, int64_t T, int32_t* __restrict__ slots,
                                                 int32_t* __restrict__ rows) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= T * 16) return;
  const int64_t t = idx >> 4;
  const int r = (int)(idx & 15);
  if (r == 15) { rows[idx] = -1; return; }
  const uint8_t* row = state + t * STATE_ROW_BYTES;
  int role = row[DDZ_F_META * 16];
  if (role > 2) role = 0;
  const int rm1 = role == 0 ? 2 : role - 1, rp1 = role == 2 ? 0 : role + 1;
  auto c4 = [](int v) { return v > 4 ? 4 : v; };   // (a thermometer saturates at 4; never index outside on a corrupted import)
  const int hand = c4(row[(DDZ_F_HAND0 + role) * 16 + r]), taken = c4(row[DDZ_F_TAKEN * 16 + r]);
  const int b1 = c4(row[(DDZ_F_RECENT0 + rm1) * 16 + r]), b2 = c4(row[(DDZ_F_RECENT0 + rp1) * 16 + r]);
  int n1 = row[(DDZ_F_HAND0 + rp1) * 16 + 15], n2 = row[(DDZ_F_HAND0 + rm1) * 16 + 15];
  n1 = n1 > 20 ? 20 : n1; n2 = n2 > 20 ? 20 : n2;
  // canonical (n1, n2): the prob planes hold n / (n1 + n2) -- the same float for (2, 4) and (1, 2) (one correctly rounded
  // division of the same rational) -- and only in the slots known <= j < total: none when hand + taken >= total.  Halves the
  // distinct rows (measured on the oracle's states: 86,689 -> 43,232 at 65,536 tables).
  {
    int g = n1, b = n2;
    while (b) { const int m = g % b; g = b; b = m; }
    if (g > 1) { n1 /= g; n2 /= g; }
  }
  const int ncode = hand + taken >= (r < 13 ? 4 : 1) ? 0 : n1 * 21 + n2;
  const int key = r * QSH_COLS + (((hand * 5 + taken) * 5 + b1) * 5 + b2) * 441 + ncode;
  slots[key] = (int32_t)idx + 1;
  rows[idx] = key;
}
__global__ __launch_bounds__(256) void k_qs_count(const int32_t* __restrict__ slots, int32_t* __restrict__ cnt) {
  __shared__ int s_n[4];
  const int r = blockIdx.x / QSH_CPR, ch = blockIdx.x % QSH_CPR;
  const int k0 = ch * QSH_CHUNK + (int)threadIdx.x * 8;
  int n = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) n += (k0 + i < QSH_COLS && slots[r * QSH_COLS + k0 + i] != 0) ? 1 : 0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d);
  if ((threadIdx.x & 63) == 0) s_n[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) cnt[blockIdx.x] = s_n[0] + s_n[1] + s_n[2] + s_n[3];
}
// one block: rank totals -> seg (the layout k_fc1<true> reads, QN_SEG_WORDS ints), chunk bases.  row_capacity >= 15 T + 15 tiles
// always suffices (checked on the host), so nothing can overflow; seg[33] stays 0.
__global__ __launch_bounds__(64) void k_qs_seg(const int32_t* __restrict__ cnt, int32_t* __restrict__ base, int32_t* __restrict__ seg,
                                               int32_t row_capacity) {
  __shared__ int s_tot[16], s_start[16];
  const int r = threadIdx.x;
  if (r < 15) {
    int n = 0;
    for (int c = 0; c < QSH_CPR; ++c) n += cnt[r * QSH_CPR + c];
    s_tot[r] = n;
  }
  __syncthreads();
  if (r == 0) {
    int start = 0, need = 0;
    for (int q = 0; q < 15; ++q) {
      s_start[q] = start;
      seg[q] = start; seg[16 + q] = start / FC_M;
      start += (s_tot[q] + FC_M - 1) / FC_M * FC_M;
      need += s_tot[q];
    }
    const int over = start > row_capacity ? 1 : 0;
    if (over) start = row_capacity / FC_M * FC_M;
    seg[15] = start; seg[31] = start / FC_M; seg[32] = need; seg[33] = over;
  }
  __syncthreads();
  if (r < 15) {
    int b = s_start[r];
    for (int c = 0; c < QSH_CPR; ++c) { base[r * QSH_CPR + c] = b; b += cnt[r * QSH_CPR + c]; }
  }
}
__global__ __launch_bounds__(256) void k_qs_assign(int32_t* __restrict__ slots, const int32_t* __restrict__ base,
                                                   int32_t* __restrict__ rep, int32_t row_capacity) {
  __shared__ int s_w[4];
  const int r = blockIdx.x / QSH_CPR, ch = blockIdx.x % QSH_CPR;
  const int k0 = ch * QSH_CHUNK + (int)threadIdx.x * 8;
  int v[8], n = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = k0 + i < QSH_COLS ? slots[r * QSH_COLS + k0 + i] : 0; n += v[i] != 0; }
  // exclusive scan of n over the block: inside the wave by DPP-free shuffles, across the four waves through LDS
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  if (lane == 63) s_w[wv] = inc;
  __syncthreads();
  int off = base[blockIdx.x] + inc - n;
  for (int w = 0; w < wv; ++w) off += s_w[w];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (v[i] != 0) {
      if (off < row_capacity) { rep[off] = v[i] - 1; slots[r * QSH_COLS + k0 + i] = off + 1; }
      else slots[r * QSH_COLS + k0 + i] = 0;    // (cannot happen with the documented capacity; never a row beyond it)
      ++off;
    }
}
__global__ __launch_bounds__(256) void k_qs_rows(const int32_t* __restrict__ slots, int64_t T, int32_t* __restrict__ rows) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= T * 16) return;
  const int key = rows[idx];
  if (key >= 0) rows[idx] = (key < QSH_KEYS ? slots[key] : 0) - 1;
}

// first layer of the distinct rows: ys[row][c] = Y[t][r][0][c] for (t, r) = rep[row] (padding rows of a segment: zeros);
// 256 threads = 256 channels, a tile of QR_TILE rows per block, their face columns staged in LDS
constexpr int QR_TILE = 64;    // rows per block (128: fewer, longer blocks -- measured slower: 102 against 88 us for 7.3 * 10^4 rows)
template <int P>
__global__ __launch_bounds__(QH) void k_q_feat_rows(const float4* __restrict__ face, int64_t T, const float* __restrict__ wf,
                                                    const float* __restrict__ bias, const int32_t* __restrict__ rep,
                                                    const int32_t* __restrict__ seg, float* __restrict__ ys, int ys_ld,
                                                    const float* __restrict__ mz, float* __restrict__ g) {
  // ys_ld > 256 (QS_K = 288): the row's 24 COLUMN values (plane-major, then 8 zeros) are appended behind its 256 first-layer
  // values -- the table term (linear in the face) then is 24 more rows of the rank's fc1 block: ONE K = 288 GEMM gives
  // G[row] = Y[row] x fc1[rank] + column x Mz[rank], nothing is accumulated (mz / g: the earlier form, kept).
  // mz / g (both or neither): the TABLE TERM folded into the rows -- the face part of conv_shunzi through fc1 is linear in the
  // face, i.e. a sum over the ranks of (column of rank r) x mz[rows p * 60 + 4 r + w] (mz f32 [P * 60][256], the operand of the
  // per-table GEMM [T, 60 P] x [60 P, 256] it replaces): g[row] = that product for the row's column and rank; the rows GEMM
  // then ACCUMULATES into g, and H0[t] = base + sum_r g[row(t, r)].
  const int c = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * QR_TILE;
  if (row0 >= seg[15]) return;
  int rk = 0;                       // the tile's rank (a tile never straddles two 128-aligned segments)
#pragma unroll
  for (int q = 1; q < 15; ++q) rk += row0 >= seg[q];
  float m[P][4];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int k = 0; k < 4; ++k) m[p][k] = mz ? mz[(int64_t)(p * 60 + 4 * rk + k) * QH + c] : 0.f;
  __shared__ float4 s_col[QR_TILE * P];
  __shared__ int s_ok[QR_TILE];
  for (int i = threadIdx.x; i < QR_TILE * P; i += QH) {
    const int j = i / P, p = i - j * P;
    const int inst = rep[row0 + j];
    const int64_t t = inst >> 4;
    const bool ok = inst >= 0 && t < T && (inst & 15) < 15;
    s_col[i] = ok ? face[(t * P + p) * 15 + (inst & 15)] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (p == 0) s_ok[j] = ok ? 1 : 0;
  }
  float w[P][10];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    int q = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j <= k; ++j) w[p][q++] = wf[(int64_t)(p * 4 + j) * (4 * QH) + k * QH + c];
  }
  float b[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) b[k] = bias[k * QH + c];
  __syncthreads();
#pragma unroll 4
  for (int j = 0; j < QR_TILE; ++j) {
    float s0 = b[0], s1 = b[1], s2 = b[2], s3 = b[3];
#pragma unroll
    for (int p = 0; p < P; ++p) {   // (the same expression, in the same order, as k_q_feat_needed: bit-identical Y)
      const float4 x = s_col[j * P + p];
      s0 += w[p][0] * x.x;
      s1 += w[p][1] * x.x + w[p][2] * x.y;
      s2 += w[p][3] * x.x + w[p][4] * x.y + w[p][5] * x.z;
      s3 += w[p][6] * x.x + w[p][7] * x.y + w[p][8] * x.z + w[p][9] * x.w;
    }
    ys[(row0 + j) * ys_ld + c] = s_ok[j] ? fmaxf(fmaxf(s0, s1), fmaxf(s2, s3)) : 0.f;
    if (c < ys_ld - QH) ys[(row0 + j) * ys_ld + QH + c] = c < 4 * P ? ((const float*)&s_col[j * P])[c] : 0.f;   // (zeros where !ok)
    if (g) {
      float lin = 0.f;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const float4 x = s_col[j * P + p];
        lin += m[p][0] * x.x + m[p][1] * x.y + m[p][2] * x.z + m[p][3] * x.w;
      }
      g[(row0 + j) * QH + c] = s_ok[j] ? lin : 0.f;
    }
  }
}

// H0[t] += sum_r G[rows[t][r]] in rank order; one wavefront per table, lane l owns hidden units 4 l .. 4 l + 3
__global__ __launch_bounds__(256) void k_qs_gather(const float4* __restrict__ G, int64_t g_rows, const int32_t* __restrict__ rows,
                                                   int64_t T, float4* __restrict__ h0, const float4* __restrict__ base) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const int mine = lane < 16 ? rows[t * 16 + lane] : -1;
  float4 acc = base ? base[lane] : h0[t * 64 + lane];   // (base f32 [256]: H0 = base + the rows; else H0 += the rows)
  float4 g[15];
#pragma unroll
  for (int r = 0; r < 15; ++r) {
    const int row = __builtin_amdgcn_readlane(mine, r);
    g[r] = (row >= 0 && row < g_rows) ? G[(int64_t)row * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int r = 0; r < 15; ++r) { acc.x += g[r].x; acc.y += g[r].y; acc.z += g[r].z; acc.w += g[r].w; }
  h0[t * 64 + lane] = acc;
}

// ---- 6. the needed rows D, shared as well ---------------------------------------------------------------------------------------
// D[(t, r, c)] = (Y[t][r][c] - Y[t][r][0]) x fc1[r] + z[r][c] depends on (rank, column, c) only: one D row per distinct
// (shared row s of section 5, count c) that SOME table needs, instead of one per (table, rank, count):
//   k_qd_mark    every needed (t, r, c) (ddz_q_need's row_index >= 0) marks slot 4 s + c - 1, s = rows[t][r]
//   k_qd_count / k_qd_seg / k_qd_assign   marked slots per tile of 128 shared rows (a tile never straddles two ranks) -> rank
//                segments of the D rows + tile bases -> D row of every marked slot in slot order, drep[drow] = its slot,
//                row_cnt[drow] = c (k_fc1<true> folds z[rank][c])
//   k_qd_remap   row_index2[t][col] = D row of (rows[t][r], c)       (what the row stage k_q_slab_needed indexes D with)
//   k_q_feat_drows   dy[drow] = Y[c] - Y[0] of the slot's column (read at the shared row's representative)
constexpr int QD_SLOTS_PER_TILE = 4 * FC_M;      // 512 slots per tile of shared rows: 256 threads x 2
static_assert(QR_TILE <= FC_M && FC_M % QR_TILE == 0, "a first-layer tile lies inside one rank segment");
static_assert(QD_SLOTS_PER_TILE == 512, "k_qd_count / k_qd_assign take two slots per thread");
__device__ __forceinline__ void qd_col(int col, int& r, int& c) {   // row_index column -> (rank, count)
  r = col < 52 ? col >> 2 : 13 + (col - 52);
  c = col < 52 ? (col & 3) + 1 : 1;
}
__global__ __launch_bounds__(256) void k_qd_mark(const int32_t* __restrict__ row_index, const int32_t* __restrict__ rows, int64_t T,
                                                 int32_t* __restrict__ dslot, int64_t s_rows) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= T * QP_COLS) return;
  const int col = (int)(idx & (QP_COLS - 1));
  if (col >= 54 || row_index[idx] < 0) return;
  int r, c;
  qd_col(col, r, c);
  const int s = rows[(idx >> 6) * 16 + r];
  if (s >= 0 && s < s_rows) dslot[(int64_t)s * 4 + c - 1] = 1;
}
__global__ __launch_bounds__(256) void k_qd_count(const int32_t* __restrict__ dslot, const int32_t* __restrict__ sseg,
                                                  int32_t* __restrict__ cnt) {
  __shared__ int s_n[4];
  if ((int)blockIdx.x >= sseg[31]) return;            // tiles of shared rows in use
  const int64_t e0 = (int64_t)blockIdx.x * QD_SLOTS_PER_TILE + 2 * threadIdx.x;
  int n = (dslot[e0] != 0) + (dslot[e0 + 1] != 0);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d);
  if ((threadIdx.x & 63) == 0) s_n[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) cnt[blockIdx.x] = s_n[0] + s_n[1] + s_n[2] + s_n[3];
}
// one block: thread r walks the tiles of rank r's shared-row segment (sseg[16 + r] .. sseg[16 + r + 1]): totals -> dseg, tile bases
__global__ __launch_bounds__(64) void k_qd_seg(const int32_t* __restrict__ cnt, const int32_t* __restrict__ sseg, int32_t* __restrict__ base,
                                               int32_t* __restrict__ dseg, int32_t row_capacity, int32_t* __restrict__ status) {
  __shared__ int s_tot[16], s_start[16];
  const int r = threadIdx.x;
  const int t0 = r < 15 ? sseg[16 + r] : 0, t1 = r < 15 ? (r < 14 ? sseg[16 + r + 1] : sseg[31]) : 0;
  if (r < 15) {
    int n = 0;
    for (int t = t0; t < t1; ++t) n += cnt[t];
    s_tot[r] = n;
  }
  __syncthreads();
  if (r == 0) {
    int start = 0, need = 0;
    for (int q = 0; q < 15; ++q) {
      s_start[q] = start;
      dseg[q] = start; dseg[16 + q] = start / FC_M;
      start += (s_tot[q] + FC_M - 1) / FC_M * FC_M;
      need += s_tot[q];
    }
    const int over = start > row_capacity ? 1 : 0;   // (cannot happen: distinct (row, count) pairs <= needed (t, r, c) triples)
    if (over) { start = row_capacity / FC_M * FC_M; atomicOr(status, 2); }
    dseg[15] = start; dseg[31] = start / FC_M; dseg[32] = need; dseg[33] = over;
  }
  __syncthreads();
  if (r < 15) {
    int b = s_start[r];
    for (int t = t0; t < t1; ++t) { base[t] = b; b += cnt[t]; }
  }
}
__global__ __launch_bounds__(256) void k_qd_assign(int32_t* __restrict__ dslot, const int32_t* __restrict__ sseg,
                                                   const int32_t* __restrict__ base, int32_t* __restrict__ drep,
                                                   uint8_t* __restrict__ row_cnt, int32_t row_capacity) {
  __shared__ int s_w[4];
  if ((int)blockIdx.x >= sseg[31]) return;
  const int64_t e0 = (int64_t)blockIdx.x * QD_SLOTS_PER_TILE + 2 * threadIdx.x;
  const int v0 = dslot[e0] != 0, v1 = dslot[e0 + 1] != 0, n = v0 + v1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  if (lane == 63) s_w[wv] = inc;
  __syncthreads();
  int off = base[blockIdx.x] + inc - n;
  for (int w = 0; w < wv; ++w) off += s_w[w];
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (i == 0 ? v0 : v1) {
      if (off < row_capacity) { drep[off] = (int32_t)(e0 + i); row_cnt[off] = (uint8_t)(((e0 + i) & 3) + 1); dslot[e0 + i] = off + 1; }
      else dslot[e0 + i] = 0;
      ++off;
    }
}
__global__ __launch_bounds__(256) void k_qd_remap(const int32_t* __restrict__ row_index, const int32_t* __restrict__ rows, int64_t T,
                                                  const int32_t* __restrict__ dslot, int64_t s_rows, int32_t* __restrict__ row_index2) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= T * QP_COLS) return;
  const int col = (int)(idx & (QP_COLS - 1));
  int out = -1;
  if (col < 54 && row_index[idx] >= 0) {
    int r, c;
    qd_col(col, r, c);
    const int s = rows[(idx >> 6) * 16 + r];
    if (s >= 0 && s < s_rows) out = dslot[(int64_t)s * 4 + c - 1] - 1;
  }
  row_index2[idx] = out;
}
// dy[drow][ch] = Y[c] - Y[0] of the column of shared row s = drep[drow] >> 2, c = (drep[drow] & 3) + 1 (padding rows: zeros)
template <int P>
__global__ __launch_bounds__(QH) void k_q_feat_drows(const float4* __restrict__ face, int64_t T, const float* __restrict__ wf,
                                                     const float* __restrict__ bias, const float* __restrict__ acnt,
                                                     const int32_t* __restrict__ rep, int64_t s_rows, const int32_t* __restrict__ drep,
                                                     const int32_t* __restrict__ dseg, float* __restrict__ dy) {
  const int ch = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * QR_TILE;
  if (row0 >= dseg[15]) return;
  __shared__ float4 s_col[QR_TILE * P];
  __shared__ int s_c[QR_TILE];                 // count of the row (0: padding)
  for (int i = threadIdx.x; i < QR_TILE * P; i += QH) {
    const int j = i / P, p = i - j * P;
    const int e = drep[row0 + j];
    const int64_t s = e >> 2;
    const int inst = (e >= 0 && s < s_rows) ? rep[s] : -1;
    const int64_t t = inst >> 4;
    const bool ok = inst >= 0 && t < T && (inst & 15) < 15;
    s_col[i] = ok ? face[(t * P + p) * 15 + (inst & 15)] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (p == 0) s_c[j] = ok ? (e & 3) + 1 : 0;
  }
  float w[P][10];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    int q = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j <= k; ++j) w[p][q++] = wf[(int64_t)(p * 4 + j) * (4 * QH) + k * QH + ch];
  }
  float b[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) b[k] = bias[k * QH + ch];
  __syncthreads();
#pragma unroll 2
  for (int j = 0; j < QR_TILE; ++j) {
    float s0 = b[0], s1 = b[1], s2 = b[2], s3 = b[3];
#pragma unroll
    for (int p = 0; p < P; ++p) {   // (the expression and order of k_q_feat_needed: bit-identical dY)
      const float4 x = s_col[j * P + p];
      s0 += w[p][0] * x.x;
      s1 += w[p][1] * x.x + w[p][2] * x.y;
      s2 += w[p][3] * x.x + w[p][4] * x.y + w[p][5] * x.z;
      s3 += w[p][6] * x.x + w[p][7] * x.y + w[p][8] * x.z + w[p][9] * x.w;
    }
    const float v0 = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
    const int c = s_c[j];                      // (block-uniform)
    float out = 0.f;
    if (c > 0) {
      const float* a = acnt + (int64_t)(c * 4) * QH + ch;   // acnt[c][k][ch], c = 1..4
      out = fmaxf(fmaxf(s0 + a[0], s1 + a[QH]), fmaxf(s2 + a[2 * QH], s3 + a[3 * QH])) - v0;
    }
    dy[(row0 + j) * QH + ch] = out;
  }
}
