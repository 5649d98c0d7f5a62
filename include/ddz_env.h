/*
 * ddz_env.h -- C ABI of libddz_hip.so, the MI355X (gfx950) batched Doudizhu engine.
 *
 * This is the drop-in boundary for the hot path named by BASELINE.json:north_star:
 * what the reference reaches through its (absent) pybind modules `env` and `r`
 * (envi.py:10-13) plus the per-game Python adapter envi.py, re-cut as batched
 * verbs over T independent tables.  Plain pointers and sizes only: every buffer
 * is DEVICE memory owned by the caller (PyTorch tensors in the host mirror
 * doudizhu-rl_amd/envi.py); the library allocates nothing on the device, never
 * synchronises, and enqueues all work on the hipStream_t passed as `stream`
 * (void* so that this header needs no HIP include).  All entry points return 0
 * or a negative DDZ_E* code and never throw.  A handle is not thread-safe.
 *
 * Reference citations are relative to /root/reference.
 */
#ifndef DDZ_ENV_H
#define DDZ_ENV_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDZ_ABI_VERSION 1
#define DDZ_NUM_ACTIONS 13527 /* rule_based/utils/card.py:34-159 */
/* Optional rule-set extension, a second build of the same source (libddz_hip_jk.so,
 * -DDDZ_NATIVE_JOKER_KICKERS=1; default library: off).  It adds the 24 vectors that card.py:116,142 exclude
 * but the reference's native r.get_moves is known to return -- server/mcts/get_moves.py:22-34 builds
 * exactly them ("sidaihuojian": quad + both jokers, 13; "sandaihuojian": two consecutive triples + both
 * jokers, 11) to filter them out.  Ids 13527 + quad rank (category FOUR_TAKE_ONE) and 13540 + start rank
 * (category THREE_ONE_LINE, len 2): last in every list.  ddz_num_actions() tells the builds apart.     */
#define DDZ_ROW 16            /* packed row: int8 counts[15] (3..K,A,2,BJ,CJ; envi.py:122-124) + 1 aux byte */
#define DDZ_NFIELDS 11
#define DDZ_TRAJ_BYTES 32
/* slab list layout: rows per table.  497 = the largest legal list of any hand of <= 20 cards, proven by exhaustive
 * enumeration of every 20-card hand (tools/max_legal_bound.c, tests/test_rules_bounds.py); the joker-kicker rule set
 * adds at most 24 rows to a list and the same exhaustive run (built with -DDDZ_JK_RULES) proves its maximum too: also 497 (the worst hands hold no jokers).
 * Smaller strides are DDZ_EINVAL. */
#define DDZ_SLAB_MIN_STRIDE 512

/* error codes */
#define DDZ_OK 0
#define DDZ_EINVAL (-1)   /* bad argument / shape / null pointer          */
#define DDZ_EHANDLE (-2)  /* bad or destroyed handle                       */
#define DDZ_EHIP (-3)     /* a HIP runtime call failed (see ddz_last_hip_error) */
#define DDZ_ECAP (-4)     /* row capacity cannot be indexed with int32     */
#define DDZ_ENODEV (-5)   /* no usable gfx950 device                       */

/* state fields: state is uint8 [T][DDZ_NFIELDS][16], table-major (the 11 rows of a table are 176 contiguous bytes).
 *   0..2  hand of role 0 up / 1 lord / 2 down (envi.py:24); byte 15 = cards left (envi.py:23)
 *   3..5  history: cumulative cards played by the role (envi.py:41)
 *   6..8  recent_handout of the role, zeros for a pass (envi.py:43); byte 15 = category
 *   9     taken: all cards played (envi.py:40)
 *   10    meta: [0] role to move, [1] done, [2] winner (0xFF running), [3] i8 last r,
 *               [4..5] u16 ply, [6] dealt, [8..11] u32 episode                          */
enum { DDZ_F_HAND0 = 0, DDZ_F_HIST0 = 3, DDZ_F_RECENT0 = 6, DDZ_F_TAKEN = 9, DDZ_F_META = 10 };

/* step modes */
#define DDZ_STEP_RANDOM 0 /* uniform index into the legal list, engine RNG   (envi.py:79-85 step_random) */
#define DDZ_STEP_CHOICE 1 /* sel = const int32_t[T], index into each table's legal segment              */
#define DDZ_STEP_ROWS 2   /* sel = const int8_t[T][16] count rows (envi.py:63-70 step_manual), validated
                             against the legal segment; no match -> illegal flag, table untouched       */
#define DDZ_STEP_IDS 3    /* sel = const int32_t[T] canonical action ids (index into card.py:get_action_space();
                             what ddz_auto_choose_state writes, or the arg-max of a policy head over
                             ddz_legal_mask), validated like ROWS; -1 = engine RNG for that table (RANDOM);
                             any other value that is no action id (DDZ_AUTO_INVALID) -> illegal flag         */
#define DDZ_AUTO_INVALID (-2) /* ddz_auto_choose[_state]: the query was invalid (no combo, role > 2, > 20 cards)   */

/* face variants (envi.py:87-96, :165-178, :182-198, :202-217) -> planes P = 4, 7, 9, 6 */
#define DDZ_FACE_ENV 0
#define DDZ_FACE_COMPLICATED 1
#define DDZ_FACE_COOPERATION 2
#define DDZ_FACE_COOPERATION_SIMPLIFY 3

typedef struct ddz_env ddz_env_t;

int ddz_abi_version(void);
int ddz_num_actions(void); /* 13527 (default) or 13551 (joker-kicker build) */
const char* ddz_strerror(int code);
/* last hipError_t seen by this library on the calling thread (0 = hipSuccess) */
int ddz_last_hip_error(void);

/* sizes of the caller-owned device buffers for T tables */
int64_t ddz_state_bytes(int64_t n_tables);   /* DDZ_NFIELDS * T * 16 */
int64_t ddz_scratch_bytes(int64_t n_tables); /* per-table query records, counts, scan partials, status */
int ddz_face_planes(int variant);            /* 4 / 7 / 9 / 6, or DDZ_EINVAL */

/* Replaces `Env(seed=)` construction (envi.py:17-28): binds caller-owned device
 * buffers to a handle.  `state` must be zero-filled (or hold an exported state);
 * table t of this handle is global table `table_id_base + t` for the RNG, so a
 * sharded run deals the same cards whatever the GPU count.                      */
int ddz_create(ddz_env_t** out, int64_t n_tables, uint64_t seed, uint64_t table_id_base,
               int device_id, void* state, int64_t state_bytes, void* scratch,
               int64_t scratch_bytes);
int ddz_destroy(ddz_env_t* env);
/* tell the engine that `state` was overwritten behind its back (checkpoint load) */
int ddz_invalidate(ddz_env_t* env);

/* Replaces Env.reset() + native prepare() (envi.py:30-36, game.py:170-171): clear the
 * bookkeeping and deal 17/20/17 (lord = role 1 moves first) for every table whose mask
 * byte is non-zero (mask NULL = all).  Deal/RNG = spec v2.                */
int ddz_reset(ddz_env_t* env, const uint8_t* table_mask, void* stream);

/* Replaces Env.valid_actions(tensor=False) / r.get_moves (envi.py:98-116) for all tables:
 * CSR list in ascending canonical action id (pass first when following).
 *   offsets int32[T+1]; rows int8[row_capacity][16] (byte 15 = category);
 *   ids int32[row_capacity] canonical action ids, may be NULL.
 * Writes beyond row_capacity are dropped and status bit 1 is raised.                  */
int ddz_legal(ddz_env_t* env, int32_t* offsets, int8_t* rows, int32_t* ids,
              int64_t row_capacity, void* stream);

/* Replaces Env.step_manual / step_random (+ _update) and the native step
 * (envi.py:38-43, 63-70, 79-85): apply one action per table, taken from the CSR list
 * produced by ddz_legal for the *current* state.  Outputs may be NULL:
 *   done u8[T]; reward i8[T] (-1 lord won, +1 farmers won, 0 running: rule_play.py:14);
 *   illegal u8[T]; traj u8[T][32].  With auto_reset a finished table is
 *   re-dealt inside the same call (episode + 1).                                       */
int ddz_step(ddz_env_t* env, int mode, const void* sel, const int32_t* offsets,
             const int8_t* rows, int auto_reset, uint8_t* done, int8_t* reward,
             uint8_t* illegal, uint8_t* traj, void* stream);

/* get_mask (rule_based/utils/utils.py:45-63) for every table: the legal moves as a dense 0/1 mask over the
 * action space (bit id; bit 0 = pass, forced 0 on lead), bit-packed into ddz_mask_words() = 424 uint32 per table
 * (bit id & 31 of word id >> 5; the tail bits are 0).  mask: [T][424] uint32, device memory.                  */
int ddz_mask_words(void);
int ddz_legal_mask(ddz_env_t* env, uint32_t* mask, void* stream);

/* Slab variant of ddz_legal / ddz_step (fixed-stride list layout, as ddz_rollout_random uses): table t owns
 * rows[t * stride ... t * stride + counts[t]).  Without the CSR prefix there is no dependency between tables,
 * so ONE launch per lock-step iteration does both halves of the reference's loop body (game.py:95-106 +
 * envi.py:98-116): ddz_step_slab applies the selections -- indices into (CHOICE) or rows of (ROWS) the lists that
 * are in the buffers -- and overwrites them with the lists of the new states.  ddz_legal_slab fills the buffers
 * for the current states (after ddz_reset / ddz_create / a state import).  stride >= 512 holds any list of a
 * <= 20-card hand; a list that does not fit raises status bit 1 and is reported empty.                       */
int ddz_legal_slab(ddz_env_t* env, int32_t* counts, int8_t* rows, int32_t* ids, int64_t stride, void* stream);
int ddz_step_slab(ddz_env_t* env, int mode, const void* sel, int32_t* counts, int8_t* rows, int32_t* ids,
                  int64_t stride, int auto_reset, uint8_t* done, int8_t* reward, uint8_t* illegal, uint8_t* traj,
                  void* stream);

/* The environment side of one lock-step iteration of a value-based policy in ONE launch (game.py:95-104 with
 * dqn.py:50-71): choose each table's move as the (epsilon-)greedy arg-max of q over its current slab list
 * (= ddz_select_slab: q f32 [T][stride], first maximum, engine RNG domain 3), apply it (= ddz_step_slab CHOICE), write the
 * lists of the new states and -- face != NULL -- their `face` tensors (= ddz_observe(face_variant): f32 [T][P][15][4]).
 * choice (may be NULL) receives the selected list indices.  Bit-identical to the three separate calls.          */
int ddz_policy_step_slab(ddz_env_t* env, const float* q, double epsilon, int32_t* counts, int8_t* rows, int32_t* ids,
                         int64_t stride, int auto_reset, uint8_t* done, int8_t* reward, uint8_t* illegal, uint8_t* traj,
                         int32_t* choice, int face_variant, float* face, void* stream);

/* Slab lists -> CSR (what ddz_legal writes, what a ragged NN forward over all legal moves consumes): offsets
 * int32[T+1], rows_out int8[row_capacity][16], ids_out int32[row_capacity] (NULL: no ids).  The prefix-sum compaction of
 * the variable-length lists as its own cheap pass (two small launches), so that ddz_step_slab / ddz_rollout_random
 * never wait on a scan over all tables: legal lists in CSR order cost ddz_step_slab + this instead of ddz_legal +
 * ddz_step.  A list index is the same in both layouts (ddz_select's choice feeds ddz_step_slab CHOICE).  Rows beyond
 * row_capacity are dropped and status bit 1 is raised; the offsets written are clamped to row_capacity, so every
 * segment [offsets[t], offsets[t+1]) stays inside rows_out (truncated lists are shorter or empty, never out of bounds). */
int ddz_slab_to_csr(ddz_env_t* env, const int32_t* counts, const int8_t* rows, const int32_t* ids, int64_t stride,
                    int32_t* offsets, int8_t* rows_out, int32_t* ids_out, int64_t row_capacity, void* stream);

/* Replaces the `face` property of the four Env classes: f32 [T][P][15][4].            */
int ddz_observe(ddz_env_t* env, int variant, float* face, void* stream);

/* What an agent reads per ply -- `face` (envi.py:87-96,165-217) and valid_actions(tensor=True) (envi.py:98-116) -- in ONE
 * call: ddz_observe(env, variant, face) followed by ddz_rows_to_onehot(rows, n, onehot) on the same stream (n = 0: the
 * face only).  The N = 1 `Env` view's per-ply call; the two launches are the ones of the separate entry points.        */
int ddz_observe_actions(ddz_env_t* env, int variant, float* face, const int8_t* rows, int64_t n, float* onehot, void* stream);

/* Replaces the native get_state_prob_manual(known60, size1, size2) (server/core.py:26-33; Env.get_state_prob(),
 * envi.py:94, is the same function of the live table): known60 u8[n][60] = thermometer of the cards the actor can see
 * (own hand + everything played), sizes int32[n][2] = cards left of the next and the next-but-one player;
 * out f32[n][2][15][4] = the two probability planes of `face` (prob planes spec v1, DESIGN.md 4: PARITY UNPINNED,
 * bit-identical to the last two planes ddz_observe writes for the same table).                                  */
int ddz_state_prob(int device_id, const uint8_t* known60, const int32_t* sizes, int64_t n, float* out, void* stream);

/* Replaces batch_arr2onehot (envi.py:139-146) on device: rows int8[n][16] -> f32 [n][15][4] */
int ddz_rows_to_onehot(int device_id, const int8_t* rows, int64_t n, float* out, void* stream);

/* Replaces r.get_moves(hand15, last15) (envi.py:111, server/core.py:65, server/CFR.py:55)
 * for n independent queries: hands/lasts int8[n][16] (byte 15 ignored; `last` all-zero =
 * lead; a `last` that is no combo of the action space yields an empty list and status
 * bit 2).  scratch: ddz_scratch_bytes(n) bytes, zero-filled by the caller; its status
 * word (bits as ddz_status) is the int32 at byte ddz_scratch_bytes(n) - 256.           */
int ddz_get_moves(int device_id, const int8_t* hands, const int8_t* lasts, int64_t n,
                  int32_t* offsets, int8_t* rows, int32_t* ids, int64_t row_capacity,
                  void* scratch, int64_t scratch_bytes, void* stream);

/* ddz_get_moves in the slab layout (as ddz_legal_slab): query i owns rows[i * stride ... i * stride + counts[i]),
 * ascending canonical id, pass first when following; stride >= DDZ_SLAB_MIN_STRIDE.  ONE launch, no scratch, no size
 * pass: the low-latency form for a serving path (server/core.py:56-67 valid_actions).  status (device int32, may be
 * NULL) gets bit 2 when a `last` is no combo of the action space (that query's list is empty) and bit 1 when a list
 * does not fit (a hand of more than 20 cards can have more than 512 moves: use ddz_get_moves for those).            */
int ddz_get_moves_slab(int device_id, const int8_t* hands, const int8_t* lasts, int64_t n, int32_t* counts, int8_t* rows,
                       int32_t* ids, int64_t stride, int32_t* status, void* stream);

/* The lock-step loop of Game.play under a random policy (game.py:169-181 with
 * envi.py:79-85): n_iters iterations of {legal list, step_random(auto_reset)} in ONE kernel
 * launch.  The lists are written in the SLAB layout: table t owns
 * rows[t * stride .. t * stride + counts[t]) (ascending canonical id, same rows as
 * ddz_legal), so no table depends on another: a wavefront keeps its table's rows in
 * registers across the iterations and stores the list, the state and the trajectory
 * record of EVERY iteration (each iteration overwrites the table's slab; after the call
 * counts/rows hold the lists of the last pre-step states).
 *   counts int32[T]; rows int8[T * stride][16]; ids int32[T * stride] or NULL;
 *   stride >= 512 covers every list of a <= 20-card hand (497 is the maximum);
 *   stats (device, int64[8], may be NULL) accumulates {plies, finished episodes, total legal
 *   rows, lord wins, up wins, down wins, -, -} (Game.compete's per-role win counts,
 *   game.py:258-290); traj (may be NULL) is u8[n_iters][T][32].                            */
int ddz_rollout_random(ddz_env_t* env, int64_t n_iters, int32_t* counts, int8_t* rows,
                       int32_t* ids, int64_t stride, int64_t* stats, uint8_t* traj,
                       void* stream);

/* The same random-policy loop with the lists in the CSR layout of ddz_legal (offsets/rows/ids
 * packed across tables).  CSR bases need a scan over all tables, so this variant is one
 * launch per iteration: the fused kernel writes the list of the current state, steps, and
 * sizes + scans the lists of the new state for the next launch.  Same trajectories, same
 * states as ddz_rollout_random; after the call offsets/rows hold the last pre-step lists.     */
int ddz_rollout_random_csr(ddz_env_t* env, int64_t n_iters, int32_t* offsets, int8_t* rows,
                           int32_t* ids, int64_t row_capacity, uint8_t* traj, void* stream);
/* The same loop, same outputs (offsets / rows / ids hold the CSR lists of the last iteration's pre-step states, exactly as
 * ddz_legal writes them; every iteration's lists are written there at their CSR positions), WITHOUT a launch per iteration:
 * the lists of `batch` iterations are staged as slabs by one rollout launch and compacted by two more (the cross-table
 * prefix of an iteration is nobody's launch boundary).  staging: caller-owned device scratch of
 * ddz_rollout_csr_staging_bytes(n_tables, batch, ids != NULL) bytes (batch x n_tables x 512 rows of 16 (+ 4) bytes: size
 * the batch for the memory you have), 256-byte aligned.  Same states, trajectories and RNG draws as ddz_rollout_random.  */
int64_t ddz_rollout_csr_staging_bytes(int64_t n_tables, int batch, int want_ids);
int ddz_rollout_random_csr_staged(ddz_env_t* env, int64_t n_iters, int batch, void* staging, int64_t staging_bytes,
                                  int32_t* offsets, int8_t* rows, int32_t* ids, int64_t row_capacity, uint8_t* traj, void* stream);

/* Measurement aid: the same loop between two hipEvents on `stream`.  ms (HOST, double[2])
 * receives {elapsed ms of the launch, n_iters}; synchronises the stream.
 * Used by bench.py for the roofline figure.                                                */
int ddz_rollout_random_timed(ddz_env_t* env, int64_t n_iters, int32_t* counts, int8_t* rows,
                             int32_t* ids, int64_t stride, double* ms, void* stream);

/* stats (device, int64[8]) += {plies, finished episodes, legal rows, lord wins, up wins, down
 * wins, -, -} accumulated by ddz_step / ddz_legal / the rollouts since the last read; the
 * internal accumulators are cleared.                                                      */
int ddz_read_stats(ddz_env_t* env, int64_t* stats, void* stream);

/* Replaces the action selection of DQNFirst.greedy_action / e_greedy_action (dqn.py:50-71)
 * for all tables: q float[offsets[T]] holds one value per legal row in CSR order;
 * choice int32[T] = first index of the segment's maximum (torch.argmax, dqn.py:60,70), -1 for
 * an empty list.  With epsilon > 0 a table explores with that probability (engine RNG
 * domain 3, keyed by table id / episode / ply): uniform index (dqn.py:57-58).             */
int ddz_select(ddz_env_t* env, const float* q, const int32_t* offsets, double epsilon,
               int32_t* choice, void* stream);

/* ddz_select for the slab layout: q is f32 [T][stride] (values beyond counts[t] are ignored). */
int ddz_select_slab(ddz_env_t* env, const float* q, const int32_t* counts, int64_t stride, double epsilon,
                    int32_t* choice, void* stream);

/* First layer of the same forward per (table, rank, count), from `face` alone: conv1..conv4 (net.py:141-144: a (1,k)
 * window, stride 4, on the width-4 input = one output column per rank) + the (1,4) max-pool (net.py:93-94 = the max over
 * the four convs), for every count cnt = 0..4 an action could take of that rank (its thermometer, envi.py:139-146, is the
 * action plane net.py:89-90 appends):
 *   y[((r * 5 + cnt) * T + t) * y_row_stride + c] = max_k (bias[k][c] + sum_{p, j <= k} wf[p * 4 + j][k * 256 + c] * face[t][p][r][j]
 *                                                          + acnt[cnt][k][c]),   c < 256
 * (the two joker ranks r = 13, 14 exist once: only cnt = 0, 1 are written for them).
 * face f32 [T][planes][15][4] (ddz_observe / ddz_policy_step_slab), wf f32 [planes * 4][1024], bias f32 [1024], acnt f32
 * [5][4][256] (weight-only tables, built by FactorisedQ.refresh from the network's conv weights); planes in {4, 6, 7, 9}.
 * The host glue multiplies y by fc1 per rank (a batched GEMM) into the u of ddz_q_slab.  Stateless; fp32.            */
int ddz_q_features(int device_id, const float* face, int64_t n_tables, int planes, const float* wf, const float* bias,
                   const float* acnt, float* y, int64_t y_row_stride, void* stream);

/* The per-row stage of the reference's ragged Q forward -- policy_net(face, actions) over ALL legal actions of a state
 * (game.py:95-104, dqn.py:56,67; net.py:99-101 relu(fc1) -> fc2) -- for every table at once, over the slab lists as
 * ddz_step_slab / ddz_legal_slab left them.  The host glue evaluates the first layer factorised per (rank, count)
 * (doudizhu-rl_amd/dqn_glue.py FactorisedQ.tables: dense per-table GEMMs, no ragged dimension) into
 *   u f32 [15][5][T][hidden]: fc1's pre-activation contribution of rank r when the action takes cnt cards of it (per table),
 *   z f32 [15][5][hidden]:    the same for the action plane's own path through conv_shunzi (weights only; z[r][0] = 0),
 * and this writes q[t * stride + j] = b2[0] + w2 . relu(sum_r (u[r][cnt_r][t][:] + z[r][cnt_r][:])), cnt_r = the count of rank
 * r in row j of table t, for j < counts[t] (entries beyond counts[t] are left alone) -- what ddz_policy_step_slab /
 * ddz_select_slab take.  No CSR, no padded rows, no host sync.  hidden must be 256 (net.py:147); w2 f32 [hidden], b2 f32
 * [1]: DEVICE memory.  fp32 (tests: tolerance 1e-5 against the literal nn.Conv2d evaluation).                     */
int ddz_q_slab(ddz_env_t* env, const float* u, const float* z, int64_t hidden, const float* w2, const float* b2,
               const int32_t* counts, const int8_t* rows, int64_t stride, float* q, void* stream);

/* The same two stages over PACKED rows: only the (rank, count, table) triples a legal move of table t can use exist -- count 0
 * of every rank, and count c >= 1 of rank r where the actor holds at least c cards of it (15 + cards-in-hand rows per table
 * instead of 69: a third of the fc1 GEMM).  Layout (built by the host glue from the actors' hands, FactorisedQ.pack):
 *   rank r's rows start at rank_row0[r] (HOST memory, 15 entries read; the ranks' segments in any order, not overlapping,
 *   inside [0, n_rows); rows between segments are padding the glue's batched GEMMs may compute on): the first T rows of a
 *   segment are count 0 of tables 0..T-1, then the held counts in any order;
 *   row_index int32 [T][64] (device): row of (r < 13, c = 1..4) at column 4 r + c - 1, of a joker's count 1 at column 52 /
 *   53; -1 = not held (ddz_q_features_packed skips it; ddz_q_slab_packed reads the count-0 row instead: no legal move of
 *   the table takes that count).  Entries are device data and are never trusted as addresses: one at or beyond n_rows is
 *   treated like -1 by both functions (nothing outside y[:n_rows] / u[:n_rows] is touched) and ddz_q_slab_packed raises
 *   status bit 5.
 * y / u f32 [n_rows][y_row_stride / hidden]; the glue multiplies each rank's rows by that rank's fc1 block (15 GEMMs).
 * table_term f32 [T][hidden] or NULL: the per-table term (fc1 bias + the face part of conv_shunzi), added once per table
 * (the unpacked form carries it on rank 0's rows).  Results equal the unpacked functions' up to the GEMM's summation order. */
int ddz_q_features_packed(int device_id, const float* face, int64_t n_tables, int planes, const float* wf, const float* bias,
                          const float* acnt, const int32_t* row_index, const int64_t* rank_row0, int64_t n_rows, float* y,
                          int64_t y_row_stride, void* stream);
int ddz_q_slab_packed(ddz_env_t* env, const float* u, const int32_t* row_index, const int64_t* rank_row0, int64_t n_rows,
                      const float* table_term, const float* z, int64_t hidden, const float* w2, const float* b2,
                      const int32_t* counts, const int8_t* rows, int64_t stride, float* q, void* stream);

/* The same forward over NEEDED rows only -- nothing on the host, no host-side sizes (doudizhu-rl_amd/csrc/ddz_qnet.h;
 * BASELINE configs[2]: net.py inference in the loop, game.py:95-104 / dqn.py:56,67 for every table at once):
 *   fc1 pre-activation of move j of table t = H0[t] + sum over the ranks r the move touches of D[row(t, r, cnt_jr)]
 *   H0[t] = table_term[t] + sum_r fc1_r^T Y[t][r][0]               one dense GEMM, K = 15 * 256
 *   D[row] = fc1_r^T (Y[t][r][c] - Y[t][r][0]) + z[r][c], c >= 1   only for the (r, c) some LEGAL MOVE of table t takes
 * ddz_q_need: finds those (r, c) from the slab lists (counts / rows as ddz_step_slab left them) and lays their rows out in
 *   fifteen rank segments: row_index int32 [T][64] (columns as above; -1 = not needed) and seg int32 [40] (DEVICE memory:
 *   [r] first row of rank r's segment -- a multiple of the tile, ddz_q_fc1_tile_rows() = 128 --, [15] rows in use, [16 + r] first tile of rank r, [31]
 *   tiles in use, [32] rows needed, [33] 1 if row_capacity was too small -- then status bit 1 is raised and the rows that did
 *   not fit are -1); row_cnt uint8 [row_capacity]: the count c of every needed row (ddz_q_fc1_rows adds z[rank][c]).
 *   row_capacity: rows of dy / d, a multiple of the tile, >= 15 tiles; 20 T + 15 tiles always suffices (a move
 *   takes at most what the actor holds: <= 20 cards).  scratch: ddz_q_need_scratch_bytes(T) bytes, 256-byte aligned.
 * ddz_q_features_needed: the first layer (as ddz_q_features) into y0 f32 [T][15 * 256] (count 0 of every rank: the dense
 *   GEMM's left operand) and dy f32 [row_capacity][256] (Y[t][r][c] - Y[t][r][0] at the row of every needed (t, r, c)).
 * ddz_q_fc1_dense: c f32 [n_rows][256] += a f32 [n_rows][k] x w f32 [k][256] (k a multiple of 16); ddz_q_fc1_rows:
 *   d[row] = dy[row] x w2[rank of the row] + z[rank][row_cnt[row]] (w2 f32 [15][256][256], input-major; z f32 [15][5][256]: the
 *   action plane's own path through conv_shunzi and fc1, weights only), rows and ranks from seg -- both one launch
 *   of a hand-written fp32 MFMA kernel (v_mfma_f32_32x32x2_f32: exact f32, a k-ordered fmaf chain; no library GEMM).
 * ddz_q_slab_needed: the per-row stage (as ddz_q_slab) from h0 f32 [T][256], d, row_index (a table's needed rows are staged in
 *   LDS once: every move that takes that count of the rank uses the row); a move whose (r, c) has no row
 *   (a list that does not belong to this row_index) contributes nothing for that rank and raises status bit 5.
 * fp32 throughout; results equal ddz_q_slab's up to summation order (tests: 1e-5 against the literal nn.Conv2d network). */
int ddz_q_fc1_tile_rows(void);   /* rows per tile of the fc1 kernel: segment starts and row_capacity are multiples of it */
int64_t ddz_q_need_scratch_bytes(int64_t n_tables);
int ddz_q_need(ddz_env_t* env, const int32_t* counts, const int8_t* rows, int64_t stride, int64_t row_capacity, void* scratch,
               int64_t scratch_bytes, int32_t* row_index, int32_t* seg, uint8_t* row_cnt, void* stream);
int ddz_q_features_needed(int device_id, const float* face, int64_t n_tables, int planes, const float* wf, const float* bias,
                          const float* acnt, const int32_t* row_index, float* y0, float* dy, int64_t row_capacity, void* stream);
int ddz_q_fc1_dense(int device_id, const float* a, int64_t n_rows, int64_t k, const float* w, float* c, void* stream);
int ddz_q_fc1_rows(int device_id, const float* dy, const int32_t* seg, const uint8_t* row_cnt, const float* w2, const float* z,
                   float* d, int64_t row_capacity, void* stream);
int ddz_q_slab_needed(ddz_env_t* env, const float* h0, const float* d, int64_t row_capacity, const int32_t* row_index,
                      int64_t hidden, const float* w2, const float* b2, const int32_t* counts, const int8_t* rows,
                      int64_t stride, float* q, void* stream);

/* H0 from SHARED rows (doudizhu-rl_amd/csrc/ddz_qnet.h section 5; the same forward, net.py:81-102, for faces of
 * EnvCooperationSimplify, envi.py:201-217): Y[t][r][0] depends only on the face column of rank r, i.e. on (hand_r, taken_r,
 * b1_r, b2_r) and -- through n / (n1 + n2), and only where hand_r + taken_r < total -- on (n1, n2) reduced by their gcd; across
 * the tables of a batch those columns repeat (3.6 % distinct at 65,536 tables).  One row per
 * DISTINCT (rank, column): first layer + G[row] = Y[row] x fc1[rank] (ddz_q_fc1_rows with z = row_cnt = NULL), then
 * H0[t] = table_term[t] + sum_r G[rows[t][r]] -- instead of the K = 15 * 256 dense product.  Exact per call (nothing is kept
 * between calls), equal to the dense form up to fp32 summation order.
 * ddz_q_shared_rows: from env's CURRENT state: rows int32 [T][16] (row of (t, r); column 15 = -1), rep int32 [row_capacity]
 *   (row -> a (table, rank) instance 16 t + r that has this column; -1 = padding), seg int32 [40] as ddz_q_need's (rank
 *   segments, starts multiples of the tile).  row_capacity: a multiple of the tile, >= min(15 T, 4134375) + 15 tiles (then
 *   nothing can overflow).  ws: ddz_q_shared_ws_bytes() bytes (16.6 MB: one int32 slot per possible (rank, column)), 16-byte
 *   aligned, contents irrelevant on entry.  Row numbers follow the key order: deterministic.
 * ddz_q_features_rows: ys f32 [row_capacity][ys_ld] (ys_ld 256 or 288) = first layer (count 0) of every row's column, read from `face` f32
 *   [T][6][15][4] at rep[row]; padding rows = 0.  planes must be 6.
 * ddz_q_gather_h0: h0 f32 [T][256] += sum over r = 0..14 (in this order) of g[rows[t][r]] (g f32 [g_rows][256]; rows < 0 or
 *   >= g_rows contribute nothing).
 * ddz_q_features_needed with y0 = NULL then evaluates only the ranks a legal move takes cards of (dy alone). */
/* The needed rows shared as well (ddz_qnet.h section 6): D[(t, r, c)] depends on (rank, column, c) only -- one D row per
 * distinct (shared row, count) some table needs.
 * ddz_q_shared_need: row_index (ddz_q_need's: >= 0 <=> (t, r, c) is needed), rows / sseg (ddz_q_shared_rows') -> row_index2
 *   int32 [T][64] (the D row of every needed (t, r, c); -1 otherwise: what ddz_q_slab_needed indexes d with), drep int32
 *   [row_capacity] (D row -> 4 * shared row + c - 1; -1 = padding), dseg int32 [40] (rank segments of the D rows, as seg),
 *   row_cnt uint8 [row_capacity] (c of every D row, for ddz_q_fc1_rows' z fold).  row_capacity as ddz_q_need's (distinct
 *   pairs never outnumber the needed triples).  ws: ddz_q_shared_need_ws_bytes(shared_row_capacity) bytes, 16-byte aligned.
 * ddz_q_features_drows: dy f32 [row_capacity][256] = Y[c] - Y[0] of every D row's column (face read at the shared row's
 *   representative rep[]); padding rows = 0.  planes must be 6. */
int64_t ddz_q_shared_need_ws_bytes(int64_t shared_row_capacity);
int ddz_q_shared_need(ddz_env_t* env, const int32_t* row_index, const int32_t* rows, const int32_t* sseg,
                      int64_t shared_row_capacity, void* ws, int64_t ws_bytes, int64_t row_capacity, int32_t* row_index2,
                      int32_t* drep, int32_t* dseg, uint8_t* row_cnt, void* stream);
int ddz_q_features_drows(int device_id, const float* face, int64_t n_tables, int planes, const float* wf, const float* bias,
                         const float* acnt, const int32_t* rep, int64_t shared_row_capacity, const int32_t* drep,
                         const int32_t* dseg, float* dy, int64_t row_capacity, void* stream);
int64_t ddz_q_shared_ws_bytes(void);
int ddz_q_shared_rows(ddz_env_t* env, void* ws, int64_t ws_bytes, int64_t row_capacity, int32_t* rows, int32_t* rep,
                      int32_t* seg, void* stream);
int ddz_q_features_rows(int device_id, const float* face, int64_t n_tables, int planes, const float* wf, const float* bias,
                        const int32_t* rep, const int32_t* seg, float* ys, int64_t ys_ld, int64_t row_capacity, const float* mz,
                        float* g, void* stream);
int ddz_q_gather_h0(int device_id, const float* g, int64_t g_rows, const int32_t* rows, int64_t n_tables, const float* base,
                    float* h0, void* stream);
/* The table term folded into the rows: it is linear in the face (fc1 bias + the face part of conv_shunzi through fc1), i.e.
 * base + sum_r column_r x mz[p * 60 + 4 r + w] (mz f32 [60 * 6][256]: the operand of the [T, 360] x [360, 256] GEMM it
 * replaces).  ddz_q_features_rows with ys_ld = 288 appends the row's 24 column values (plane-major; then 8 zeros) behind its
 * 256 first-layer values; ddz_q_fc1_rows_k(k = 288): g[row] (+)= y[row] x w2k[rank] with w2k f32 [15][k][256] = fc1's block of
 * the rank, then the rank's 24 rows of mz, then zeros -- ONE product per row gives both terms; ddz_q_gather_h0 with base f32
 * [256] (or NULL: h0 += ...): h0[t] = base + sum_r g[rows[t][r]].  (ys_ld = 256 with mz / g: the product of the column is
 * written into g instead and the GEMM accumulates -- the same values, one more pass over g.) */
int ddz_q_fc1_rows_k(int device_id, const float* y, int64_t k, const int32_t* seg, const float* w2k, float* g,
                     int64_t row_capacity, int accumulate, void* stream);

/* The canonical action table: rows[ddz_num_actions()][16] = int8 counts[15] + category of action id
 * (rule_based/utils/card.py:34-159 order), device memory. */
int ddz_action_table(int device_id, int8_t* rows, void* stream);

/* 32-byte trajectory records -> 8-byte records (for the end-of-batch gather over xGMI: 4x fewer bytes).
 *   word 0: action id (14 bits; 0x3FFF = not an action: flags != 0, or a row outside the action space) | n_legal << 14 (9 bits) | role << 23 (2) | done << 25 |
 *           reward code << 26 (0: 0, 1: +1, 2: -1) | flags << 28 (bit0 illegal, bit1 frozen)
 *   word 1: choice + 1 (10 bits) | ply << 10 (8 bits) | episode << 18 (low 14 bits)
 * traj: u8[n][32], packed: u8[n][8], device memory. */
#define DDZ_TRAJ_PACKED_BYTES 8
int ddz_pack_trajectory(int device_id, const uint8_t* traj, int64_t n_records, uint8_t* packed, void* stream);

/* Replaces Env.step_auto (envi.py:72-77; game.py:106 for every role without a network; rule_based/rule_play.py:16-23):
 * the move of the rule-based opponent.  The native step_auto is absent from the reference; its in-repo statement is
 * RuleBasedModel.choose (rule_based/utils/rule_based_model.py:43-101) over Decomposer.get_combinations
 * (rule_based/utils/decomposer.py:17-76) and cards_value (rule_based/utils/evaluator.py:10-47), which this computes
 * -- with "decomposer spec v1"  in place of the two absent native decomposition functions.
 *   ddz_auto_choose_state: for every table whose ACTOR's role bit is set in auto_roles (bit r = role r: 0 up, 1 lord,
 *     2 down) ids[t] = the canonical id of the chosen action (0 = pass), for all other tables -1.  Feed ids to
 *     ddz_step / ddz_step_slab with DDZ_STEP_IDS: rule agents move as chosen, everybody else by the engine RNG; or
 *     merge a policy's own ids into the -1 slots first.
 *   ddz_auto_choose: the same for n independent queries (server/core.py:80-87 calls choose() on a payload):
 *     hands / lasts int8[n][16] (byte 15 ignored, `last` all-zero = lead), info u8[n][4] = cards left of role 0, 1, 2
 *     (envi.py:23 `left`) and the acting role; an invalid query (no combo, role > 2, more than 20 cards) yields
 *     DDZ_AUTO_INVALID (-2: unlike -1 it is never taken for "engine RNG" by DDZ_STEP_IDS) and status bit 2 of the handle.
 *   Both entry points may be issued on any stream, also concurrently on several streams of one handle: every launch has
 *   its own work-distribution word, zeroed on its stream (nothing for the caller to initialise or re-arm).
 *   stats (may be NULL): int64[n][2] = {combinations scored, search nodes} of the FULL enumeration per table / query.
 *     With stats == NULL the search is an exact branch and bound (subtrees whose score bound is strictly
 *     below a score already reached are skipped): the same ids from about a tenth of the nodes, 3x faster.          */
int ddz_auto_choose_state(ddz_env_t* env, int auto_roles, int32_t* ids, int64_t* stats, void* stream);
int ddz_auto_choose(int device_id, const int8_t* hands, const int8_t* lasts, const uint8_t* info, int64_t n,
                    int32_t* ids, int64_t* stats, void* stream);
/* test hook: 2 * cards_value (rule_based/utils/evaluator.py:10-47) of every action id, int8[DDZ_NUM_ACTIONS] */
int ddz_debug_cards_value(int device_id, int8_t* out, void* stream);
/* test hook: the rule agent's score of one finished combination (rule_based_model.py:60-86) on its own: in int32[n][4] =
 * {2 * sum of cards_value, 2 * smallest eligible cards_value or 127 = none, actions, following | pass allowed << 1},
 * rp f64[n] = round_penalty -> value f64[n], move int32[n] (0 = pass, 7 = the eligible move, -1 = none).  The f64
 * operations round exactly as Python's (the product small_num * rp BEFORE the subtraction): tested bit for bit.     */
int ddz_debug_auto_leaf(int device_id, const int32_t* in, const double* rp, int64_t n, double* value, int32_t* move, void* stream);
/* test hook: ddz_auto_choose_state with an explicit kernel: 1 = the sequential full-enumeration walk (cross-check),
 * 2 = the lane-parallel branch-and-bound kernel as ddz_auto_choose_state runs it (tables ordered heaviest hand first),
 * 3 = the same kernel in table order.  Same ids by construction.                                                  */
int ddz_debug_auto_choose_state(ddz_env_t* env, int kernel, int auto_roles, int32_t* ids, int64_t* stats, void* stream);
/* test hook: launch geometry of a handle (tables per wavefront 1..64, 0 = keep; block-cooperative one-table-per-wave
 * form of ddz_step_slab 0 / 1, 2 = that form with every list written by one wavefront (1: leads of 5 or more scan rounds
 * -- plane-rich hands, the lord's first lead -- are written by the whole workgroup; n > 2: from n rounds on), -1 = keep; ddz_step_slab's block work list of deals + lists -- most expensive first --
 * 0 / 1, -1 = keep).  Call right after ddz_create.  Results never depend on it; the library reads no environment
 * variables.                                                                                                     */
int ddz_debug_set_geometry(ddz_env_t* env, int tables_per_wave, int slab_coop, int slab_work_list);
/* test hook: 2 (the default) = ddz_auto_choose_state's wavefronts that run out of tables help the searches still running in
 * their workgroup; 1 = additionally the predicted-heaviest decisions (one per workgroup) are searched by their whole
 * workgroup from the start ("team first": built and measured in round 4, no gain, off by default); 0 = neither.  Same ids
 * and stats in every mode (tests compare them).                                                                    */
int ddz_debug_set_auto_teams(ddz_env_t* env, int on);

/* device status word: bit0 enumerator/count mismatch, bit1 row capacity overflow,
 * bit2 invalid `last` combo, bit3 a wait of ddz_auto_choose_state's cooperating wavefronts hit its
 * hang guard (never in a working launch; the ids of that launch are not to be trusted), bit4 the sequential
 * cross-check kernel's depth guard (cannot happen: at most 20 actions), bit5 a row_index entry of ddz_q_slab_packed at or
 * beyond n_rows (not dereferenced: the count-0 row was read instead).
 * Copies 4 bytes D2H on `stream` and synchronises it.                                   */
int ddz_status(ddz_env_t* env, int32_t* status_out, void* stream);
/* the same word for the STATELESS rule-agent entry point (ddz_auto_choose has no handle): one per device, bits as above
 * (bit 3: a cooperating wait hit its hang guard -- the ids of launches since the last call are not to be trusted).
 * Copies 4 bytes D2H on `stream`, clears the word, synchronises.                                                  */
int ddz_device_status(int device_id, int32_t* status_out, void* stream);
/* hipStreamSynchronize(stream) on the device: the ONE wait per ply of the N = 1 `Env` view (doudizhu-rl_amd/envi.py), whose
 * state / list-size buffers live in pinned host memory that the kernels write directly (envi.py:63-116 reads them back
 * after every native call).  No other entry point of this library waits for the device except ddz_status.           */
int ddz_sync(int device_id, void* stream);

/* test hook: CardGroup.to_cardgroup (card.py:327-335) of arbitrary count rows int8[n][16] ->
 * out u32[n] = category | value << 8 | len << 16, or 0xFF when the row is no combo.       */
int ddz_debug_classify(int device_id, const int8_t* rows, int64_t n, uint32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
