/*
 * c4_oracle.h -- CPU ORACLE for the c4a0 self-play hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm (advait/c4a0,
 * rust/src/c4r.rs + mcts.rs + self_play.rs + types.rs + utils.rs).  It exists so that
 * the HIP product path can be checked bit-for-bit against something that follows the
 * reference line by line.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (c4a0_amd/) never does.
 *
 * PARITY PINNING (see DESIGN.md "Oracle"):
 *   - game rules, search, softmax/temperature, game loop: pinned against every
 *     known-answer test the reference holds for the path (tests/test_oracle_*.py cite
 *     each one by file:line).
 *   - libm expf/logf (what Rust f32::exp / f32::ln call): pinned by exhaustive sweep
 *     against the host glibc 2.35 libm (c4o_sweep_expf / c4o_sweep_logf).
 *   - move sampling RNG (rand 0.10.1 StdRng + WeightedIndex, chacha20 0.10.1, rand_core
 *     0.10.1): the crates' source is not in /root/reference and no reference test pins a
 *     sampled move, so every stage is pinned by a vector the crates publish in their own
 *     unit tests (tests/test_oracle_libm_rng.py): ChaCha block (RFC 7539 / eSTREAM),
 *     StdRng = ChaCha12 word order (rand `test_stdrng_construction`), the PCG32 seed
 *     expansion (rand_core `test_seed_from_u64` value-breakage constant), and
 *     UniformFloat + cumulative weights + partition_point (rand WeightedIndex
 *     `value_stability`, f32 weights under the crate's Pcg32 test generator).  Those
 *     vectors are quoted from rand 0.8/0.9 and rand_core 0.6/0.9; a value-breaking change
 *     in 0.10.1 cannot be excluded without its source (residual risk, stated in DESIGN.md).
 */
#ifndef C4_ORACLE_H
#define C4_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C4O_N_ROWS 6
#define C4O_N_COLS 7
#define C4O_BUF_LEN 84 /* 2 planes x 6 x 7, c4r.rs:47-52 */
#define C4O_MAX_MOVES 42

/* c4r.rs:13-17 */
typedef struct {
  uint64_t mask;
  uint64_t value;
} c4o_pos;

/* c4r.rs:27-32; 0 = not terminal */
enum { C4O_NOT_TERMINAL = 0, C4O_PLAYER_WIN = 1, C4O_OPPONENT_WIN = 2, C4O_DRAW = 3 };

/* error codes standing in for the reference's panics */
enum {
  C4O_OK = 0,
  C4O_ERR_NAN_IN_TREE = 1,       /* utils.rs:12  partial_cmp().unwrap() on NaN */
  C4O_ERR_DEGENERATE_POLICY = 2, /* mcts.rs:421-425 softmax max is +-inf; mcts.rs:219 WeightedIndex::new(..).unwrap() */
  C4O_ERR_ILLEGAL_MOVE = 3,      /* mcts.rs:196-200 */
  C4O_ERR_NOT_TERMINAL = 4       /* mcts.rs:277 */
};

/* ---- game rules (c4r.rs) ---- */
int c4o_make_move(const c4o_pos* p, int col, c4o_pos* out);   /* c4r.rs:58-72; 1 = ok, 0 = None */
int c4o_get(const c4o_pos* p, int row, int col);              /* c4r.rs:76-91; -1 None, 0 Opponent, 1 Player */
int c4o_ply(const c4o_pos* p);                                /* c4r.rs:95-97 */
int c4o_terminal_state(const c4o_pos* p);                     /* c4r.rs:228-238 */
int c4o_terminal_value(const c4o_pos* p, float c_ply_penalty, float* q_pen, float* q_nopen); /* c4r.rs:253-263 */
unsigned c4o_legal_mask(const c4o_pos* p);                    /* c4r.rs:266-269; bit c set = col c legal */
void c4o_flip_h(const c4o_pos* p, c4o_pos* out);              /* c4r.rs:289-299 */
void c4o_write_planes(const c4o_pos* p, float* buf84);        /* c4r.rs:378-392 */
int c4o_from_moves(const int* cols, int n, c4o_pos* out);     /* c4r.rs:315-321; 1 ok, 0 illegal */
uint64_t c4o_win_mask(int i);                                 /* c4r.rs:165-224, i in 0..69 */
/* test infrastructure: the functions above over n positions per call, and n positions reachable by legal play (c4r.rs:610-629) */
void c4o_pos_ops_batch(const uint64_t* mask, const uint64_t* value, const int32_t* col, uint64_t n, float c_ply_penalty,
                       uint64_t* out_mask, uint64_t* out_value, int32_t* out_legal, int32_t* out_term, float* out_q);
void c4o_random_positions(uint64_t n, uint64_t seed, uint64_t* out_mask, uint64_t* out_value);

/* ---- libm restatement (glibc 2.35 sysdeps/ieee754/flt-32/e_expf.c, e_logf.c) ---- */
float c4o_expf(float x);
float c4o_logf(float x);
/* compare the restatement with the host libm over bit patterns lo..hi step `stride`;
 * returns the number of mismatches, first mismatching pattern in *first_bad. */
uint64_t c4o_sweep_expf(uint32_t lo, uint32_t hi, uint32_t stride, uint32_t* first_bad);
uint64_t c4o_sweep_logf(uint32_t lo, uint32_t hi, uint32_t stride, uint32_t* first_bad);
/* host libm itself, vectorised, for checking the DEVICE port on the GPU box */
void c4o_host_expf(const float* x, float* y, size_t n);
void c4o_host_logf(const float* x, float* y, size_t n);

/* ---- policy arithmetic (mcts.rs) ---- */
int c4o_softmax7(const float* logits, float* out);                       /* mcts.rs:416-434 */
void c4o_apply_temperature(const float* policy, float t, float* out);    /* mcts.rs:439-454 */
void c4o_mask_policy(const c4o_pos* p, float* logits);                   /* c4r.rs:272-286 */

/* ---- RNG (rand 0.10.1 / chacha20 0.10.1 / rand_core 0.10.1), mcts.rs:214-222 ---- */
void c4o_seed_from_u64(uint64_t seed, uint8_t key[32]);                   /* rand_core SeedableRng::seed_from_u64 (PCG32) */
void c4o_chacha_block(const uint8_t key[32], uint64_t counter, int rounds, uint32_t out[16]);
uint32_t c4o_rng_first_u32(uint64_t seed);                                /* StdRng::seed_from_u64(seed).next_u32() */
int c4o_weighted_index(const float* w7, uint32_t u, int* out_idx);        /* WeightedIndex<f32>::new + sample */
int c4o_sample_move(uint64_t game_id, int n_moves, const float* policy, float temperature, int* out_col);

/* ---- Dirichlet root noise: BUILD EXTENSION (named by BASELINE.json, absent from the reference) ---- */
void c4o_dirichlet(uint64_t game_id, int n_moves, unsigned legal, float alpha, float* eta7);
/* rand's SliceRandom::partial_shuffle / shuffle (pybridge.rs:110-116), generic over the source of u32s (tests replay the crate's
 * vectors through it with rand_pcg::Pcg32), and on StdRng::seed_from_u64(seed) as split_train_test calls it */
typedef uint32_t (*c4o_next_u32_fn)(void* ctx);
int c4o_partial_shuffle_with(uint64_t len, uint64_t amount, c4o_next_u32_fn next, void* ctx, uint32_t* items);
int c4o_shuffle_games(uint64_t seed, uint64_t n_games, uint32_t* order);
void c4o_self_play_set_dirichlet(float alpha, float epsilon); /* for games created by c4o_self_play; (0,0) = off */

/* ---- one MCTS game (mcts.rs:27-313) ---- */
typedef struct c4o_game c4o_game;

typedef struct {
  c4o_pos pos;
  float policy[7];
  float q_penalty;
  float q_no_penalty;
} c4o_sample; /* types.rs:103-110 */

typedef struct {
  uint64_t sims;               /* on_received_policy calls (self_play.rs:272) */
  uint64_t sims_terminal_root; /* of which: leaf == root and root terminal */
  uint64_t select_levels;      /* children-array scans in select_new_leaf, every call (reference-faithful) */
  uint64_t select_levels_discarded; /* of which: the select of a job whose gate then moves or ends the game
                                     (self_play.rs:283-301): its leaf is replaced by make_move's own select */
  uint64_t backup_nodes;       /* nodes updated in backpropagate_value, excluding terminal-root sims */
  uint64_t expansions;         /* expand_leaf calls that created children */
  uint64_t nodes_created;
  uint64_t moves;
} c4o_counters;

c4o_game* c4o_game_new(const c4o_pos* start, uint64_t game_id, uint64_t player0_id, uint64_t player1_id); /* mcts.rs:48-56 */
void c4o_game_free(c4o_game* g);
void c4o_game_set_dirichlet(c4o_game* g, float alpha, float epsilon); /* extension; epsilon 0 = off */
void c4o_game_root_pos(const c4o_game* g, c4o_pos* out);
void c4o_game_leaf_pos(const c4o_game* g, c4o_pos* out);
uint64_t c4o_game_leaf_model_id(const c4o_game* g);                      /* mcts.rs:70-76 */
int c4o_game_on_received_policy(c4o_game* g, const float* logprobs7, float q_pen, float q_nopen,
                                float c_exploration, float c_ply_penalty); /* mcts.rs:83-108 */
int c4o_game_make_move(c4o_game* g, int col, float c_exploration);       /* mcts.rs:187-206 */
int c4o_game_make_random_move(c4o_game* g, float c_exploration, float temperature); /* mcts.rs:214-222 */
uint64_t c4o_game_root_visit_count(const c4o_game* g);                    /* mcts.rs:248-250 */
void c4o_game_root_policy(const c4o_game* g, float* out7);                /* mcts.rs:254-256 */
float c4o_game_root_q_penalty(const c4o_game* g);                         /* mcts.rs:260-262 */
float c4o_game_root_q_no_penalty(const c4o_game* g);                      /* mcts.rs:266-268 */
int c4o_game_n_moves(const c4o_game* g);
int c4o_game_error(const c4o_game* g);
void c4o_game_counters(const c4o_game* g, c4o_counters* out);
/* mcts.rs:271-313; writes n_moves+1 samples, returns count or -err */
int c4o_game_to_result(const c4o_game* g, float c_ply_penalty, c4o_sample* out, int cap);
/* one job of MctsThread::loop_once (self_play.rs:268-323): returns 0 = send back to NN, 1 = game over, <0 = error */
int c4o_game_step(c4o_game* g, const float* logprobs7, float q_pen, float q_nopen,
                  uint64_t n_mcts_iterations, float c_exploration, float c_ply_penalty);

/* ---- self_play (self_play.rs:39-129 scheduler, lock-step restatement) ---- */
typedef struct {
  uint64_t game_id, player0_id, player1_id;
} c4o_game_metadata; /* types.rs:37-48 */

/* EvalPosT::eval_pos (types.rs:24-26) + create_pos_batch (pybridge.rs:202-221):
 * planes = float[n][2][6][7]; must fill logprobs[n][7], q_pen[n], q_nopen[n]. return 0 ok. */
typedef int (*c4o_eval_fn)(void* ctx, uint64_t model_id, int n, const float* planes,
                           float* logprobs, float* q_pen, float* q_nopen);

typedef struct {
  uint64_t n_games;
  uint64_t n_samples;
  uint64_t nn_calls;
  uint64_t nn_positions; /* unique positions sent (pb_nn_eval, self_play.rs:221) */
  c4o_counters tree;
} c4o_selfplay_stats;

/* Plays all games to completion.  Samples of game i are written to
 * out_samples[out_offsets[i] .. out_offsets[i+1]) in reqs order (the reference's result
 * order is thread-finishing order, self_play.rs:116; compare per game_id).
 * out_samples must hold 43*n_games entries; out_offsets n_games+1.
 * n_threads: OpenMP threads for the per-game MCTS work (the reference uses ncpu-1
 * MctsThreads, self_play.rs:78); 1 = serial.  Returns 0 or a C4O_ERR code. */
int c4o_self_play(const c4o_game_metadata* reqs, uint64_t n_games, int max_nn_batch_size,
                  uint64_t n_mcts_iterations, float c_exploration, float c_ply_penalty,
                  c4o_eval_fn eval, void* eval_ctx, int n_threads,
                  c4o_sample* out_samples, uint64_t* out_offsets, c4o_selfplay_stats* stats);

/* The same job in the REFERENCE'S THREAD TOPOLOGY (self_play.rs:60-106): the calling thread is the
 * NN thread (NNThread::loop_until_close, :196-237), n_threads - 1 worker threads are the MctsThreads
 * (:268-323), games travel over two queues, network evaluation and tree work overlap.  Same samples
 * as c4o_self_play; this is what bench.py's cpu_baseline leg times. */
int c4o_self_play_async(const c4o_game_metadata* reqs, uint64_t n_games, int max_nn_batch_size,
                        uint64_t n_mcts_iterations, float c_exploration, float c_ply_penalty,
                        c4o_eval_fn eval, void* eval_ctx, int n_threads,
                        c4o_sample* out_samples, uint64_t* out_offsets, c4o_selfplay_stats* stats);

/* Timing aid (bench.py cpu_baseline): pin the NN thread to the first CPU of the affinity set and the
 * workers one per CPU over the rest, for the duration of each c4o_self_play_async call.  No effect on results. */
void c4o_set_thread_pinning(int on);

/* c4o_eval_table's ctx: n rows sorted by (mask, value); out[9 * i ..] = 7 log-probabilities, q_penalty, q_no_penalty */
typedef struct { uint64_t n; const uint64_t* mask; const uint64_t* value; const float* out; } c4o_eval_table_ctx;
int c4o_eval_table(void* ctx, uint64_t model_id, int n, const float* planes, float* lp, float* qp, float* qn);
/* built-in evaluators usable as c4o_eval_fn (ctx ignored) */
int c4o_eval_uniform(void* ctx, uint64_t model_id, int n, const float* planes,
                     float* logprobs, float* q_pen, float* q_nopen); /* self_play.rs:391-403: logits=1/7, q=0 */
int c4o_eval_zeros(void* ctx, uint64_t model_id, int n, const float* planes,
                   float* logprobs, float* q_pen, float* q_nopen);   /* pybridge_test.py:7-11: logits=0, q=0 */
int c4o_eval_hash(void* ctx, uint64_t model_id, int n, const float* planes,
                  float* logprobs, float* q_pen, float* q_nopen);    /* integer hash of the position (parity tier T1) */
/* the integer-hash evaluator on a position (shared definition with the GPU tests) */
void c4o_hash_eval_pos(uint64_t mask, uint64_t value, float* logits7, float* q_pen, float* q_nopen);

#ifdef __cplusplus
}
#endif
#endif
