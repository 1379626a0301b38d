/*
 * c4a0_hip.h -- C ABI of the MI355X-native self-play generator (libc4a0_hip.so).
 *
 * Drop-in boundary for the reference's self-play hot path.  The reference has no C header:
 * its boundary is the PyO3 function `play_games` (rust/src/pybridge.rs:20-53) which calls
 * `self_play::self_play` (rust/src/self_play.rs:39-129).  The entry points below are what a
 * Rust host would bind with `extern "C"` to replace the body of `self_play()` (INTEGRATION.md
 * shows the binding); each one cites the reference code it replaces.
 *
 * Conventions: every function returns a c4_status (0 = ok); c4_last_error_string() gives the
 * detail for the calling thread.  All pointers named *_dev are DEVICE pointers owned by the
 * caller (e.g. PyTorch tensors) and must stay valid until the session is destroyed or rebound.
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).  No function
 * synchronises the device unless its comment says so.
 */
#ifndef C4A0_HIP_H
#define C4A0_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of THIS interface: bumped whenever a signature or a struct layout below changes (version 3 added `config` in the
 * middle of c4_conv_tower_bf16's arguments, version 5 c4_session_step_head_out, version 8 the host-side record codecs c4_records_to_cbor / c4_cbor_to_records / c4_shuffle_games, version 9 c4_play_games_bf16, version 10 c4_play_games_cancel / C4_ERR_CANCELLED).  A consumer compiled against this header checks it once at start-up --
 * `if (c4_abi_version() != C4_ABI_VERSION) refuse` -- because the dynamic linker compares names, not signatures
 * (tests/abi_consumer*.c and c4a0_amd/_lib.py do).  No reference counterpart: the reference's boundary is PyO3. */
#define C4_ABI_VERSION 10

#define C4_N_COLS 7          /* rust/src/c4r.rs:45, lib.rs:28 */
#define C4_N_ROWS 6          /* rust/src/c4r.rs:44, lib.rs:29 */
#define C4_BUF_N_CHANNELS 2  /* rust/src/c4r.rs:48, lib.rs:30 */
#define C4_PLANES_LEN 84     /* rust/src/c4r.rs:52 BUF_LEN */
#define C4_MAX_SAMPLES_PER_GAME 43 /* <= 42 moves + the terminal sample, mcts.rs:271-313 */

typedef enum {
  C4_OK = 0,
  C4_ERR_BAD_ARG = 1,
  C4_ERR_HIP = 2,               /* a HIP runtime call failed */
  C4_ERR_NAN_IN_TREE = 3,       /* reference panics: utils.rs:12 (OrdF32 on NaN) */
  C4_ERR_DEGENERATE_POLICY = 4, /* reference panics: mcts.rs:421-425, mcts.rs:219 */
  C4_ERR_ARENA_OVERFLOW = 5,    /* blocks_per_slot too small for this game's tree */
  C4_ERR_NOT_BOUND = 6,         /* step before bind_io / set_games */
  C4_ERR_NO_DEVICE = 7,
  C4_ERR_ILLEGAL_MOVE = 8,      /* reference panics: mcts.rs:196-200 (sampled a full column; only possible when the
                                   root's children have no visits, i.e. n_mcts_iterations <= 1) */
  C4_ERR_CANCELLED = 9          /* c4_play_games_bf16 stopped by c4_play_games_cancel (the host's Ctrl-C) */
} c4_status;

/* types.rs:37-48 GameMetadata */
typedef struct {
  uint64_t game_id;
  uint64_t player0_id;
  uint64_t player1_id;
} c4_game_metadata;

/* One training sample (types.rs:103-110 Sample + the game it belongs to), 64 bytes.
 * meta = sample index within the game (low 16 bits) | flags << 16 (bit 0: terminal sample). */
typedef struct {
  uint64_t game_id;
  uint64_t mask;   /* c4r.rs:13-17 Pos.mask  */
  uint64_t value;  /* c4r.rs:13-17 Pos.value */
  float policy[7];
  float q_penalty;
  float q_no_penalty;
  uint32_t meta;
} c4_sample_rec;

/* Arguments of self_play() (self_play.rs:39-46) that are fixed for a session, plus sizing. */
typedef struct {
  uint32_t n_slots;           /* games resident on the GPU and advanced in lock-step */
  uint32_t blocks_per_slot;   /* tree arena per slot, in 128-byte 7-children blocks, at most 65535 (16-bit child links).
                                 0 = automatic: up to n_mcts_iterations = 1000 the worst case of a never-reclaimed arena,
                                 43*n_mcts_iterations+8; beyond that (and up to 32 200) a RECLAIMED arena of two halves of
                                 2.5*n_mcts_iterations+554 blocks each (see C4_FLAG_RECLAIM) */
  uint32_t n_mcts_iterations; /* self_play.rs:43 */
  float c_exploration;        /* self_play.rs:44 (f32, as pybridge.rs:26) */
  float c_ply_penalty;        /* self_play.rs:45 */
  uint32_t planes_dtype;      /* 0 = float32, 1 = bfloat16: element type of the NN input buffer */
  uint32_t flags;             /* C4_FLAG_* */
  int32_t device;             /* HIP device ordinal */
  uint32_t reclaim_period;    /* reclaimed arenas only: step launches between two looks at the arenas (0 = 64); tests use 1 */
} c4_config;

#define C4_FLAG_NO_MOVES 1u   /* never move: mcts.rs test helper `run_mcts` (mcts.rs:469-485) */
#define C4_FLAG_ONE_SIM_PER_STEP 2u /* exactly one simulation per game per c4_session_step.  Default: a game whose
                                       freshly selected leaf is terminal (no evaluator needed, mcts.rs:92-98) runs that
                                       simulation in the same step; samples are identical either way */

#define C4_FLAG_RECLAIM 4u    /* the tree arena is reclaimed while a game is played, as the reference frees the siblings' subtrees at
                                 every move (mcts.rs:187-206): a slot's arena is two halves; when the half in use runs short, the
                                 subtree below the current root (at most n_mcts_iterations + a few blocks) is copied compactly into
                                 the other half by a kernel of its own that follows every reclaim_period-th step launch on the
                                 session's stream (also inside HIP-graph captures).  Samples are identical with and without it; what
                                 it lifts is the limit n_mcts_iterations <= 1523 of the never-reclaimed arena, and the arena shrinks
                                 from 43 n + 8 to 2 x (2.5 n + 554) blocks per slot.  With blocks_per_slot == 0 the flag is implied
                                 for n_mcts_iterations > 1000; with an explicit blocks_per_slot that value is BOTH halves. */

#define C4_FLAG_NO_RECLAIM 8u /* keep the never-reclaimed arena where the default sizing would reclaim (n_mcts_iterations > 1000): 43 n + 8
                                 blocks per slot, refused beyond n = 1523 with the reason */

/* Device-side counters (the reference's progress bars, self_play.rs:352-381, plus the
 * roofline numerators of SURVEY 8d).  Sums over all games since set_games. */
typedef struct {
  uint64_t sims;            /* on_received_policy calls executed (self_play.rs:272) */
  uint64_t select_levels;   /* sum of S: children blocks scanned by select_new_leaf (mcts.rs:160-183) */
  uint64_t backup_nodes;    /* sum of K: nodes updated by backpropagate_value (mcts.rs:137-155) */
  uint64_t expansions;      /* sum of E: expand_leaf calls that created children (mcts.rs:114-132) */
  uint64_t moves;           /* make_random_move calls (mcts.rs:214-222) */
  uint64_t games_done;
  uint64_t ref_skipped_sims;/* terminal-root sims the reference would still run (self_play.rs:283-301) */
  uint64_t samples;
  uint64_t games_started;
  uint64_t step_kernel_ns;  /* device-clock time inside c4_session_step's kernel, summed over launches:
                               last wavefront end - first wavefront start (s_memrealtime, 10 ns ticks) */
  uint64_t step_launches;   /* launches summed in step_kernel_ns */
  uint64_t eval_cache_probes; /* leaves looked up in the evaluation cache (extension, c4_session_set_eval_cache) */
  uint64_t eval_cache_hits;   /* ... and found: simulations that needed no evaluator row */
  uint64_t reclaim_passes;    /* reclaimed arenas (C4_FLAG_RECLAIM): live subtrees copied into the other half ... */
  uint64_t reclaim_blocks;    /* ... and the 128-byte blocks those copies moved */
  uint32_t error;           /* first c4_status raised on the device, 0 = none */
  uint32_t error_slot;
} c4_counters;

typedef struct c4_session c4_session;

const char* c4_last_error_string(void);
int c4_abi_version(void); /* == C4_ABI_VERSION of the header the library was compiled with */
/* Content hash of the sources the library was compiled from (no reference counterpart: build
 * hygiene; c4a0_amd/csrc/build.py rebuilds, and c4a0_amd/_lib.py refuses, a stale library). */
const char* c4_source_hash(void);
int c4_device_count(int* out);

/* Replaces the set-up half of self_play() (self_play.rs:47-58): allocates the tree arenas,
 * slot states and counters on `cfg->device`. */
int c4_session_create(const c4_config* cfg, c4_session** out);
int c4_session_destroy(c4_session* s);
/* c4_session_destroy keeps ONE tree arena (a session's largest allocation: n_slots x blocks_per_slot x 128 bytes) per
 * process for the next session on the same device that it fits -- the driver scrubs freed device memory before reuse,
 * which a session created right after a big one was destroyed would otherwise wait for (0.6 s for 13 GB).  This gives
 * the kept arena back to the device now.  C4_ARENA_CACHE=0 in the environment disables the cache. */
int c4_trim_cached_memory(void);

/* `reqs: Vec<GameMetadata>` of self_play() (self_play.rs:41).  Copies the list to the device,
 * allocates the sample store (43 records per game), resets queue and counters.
 * start_masks/start_values (host arrays, may be NULL = empty board, self_play.rs:56) give
 * MctsGame::new_from_pos positions (mcts.rs:48) for tests.  Synchronises the stream. */
int c4_session_set_games(c4_session* s, const c4_game_metadata* reqs, uint64_t n_games,
                         const uint64_t* start_masks, const uint64_t* start_values);

/* Replaces EvalPosT / PyEvalPos / create_pos_batch (types.rs:24-34, pybridge.rs:161-221):
 * instead of a callback the evaluator's tensors are bound once.
 *   planes_dev   [n_slots][2][6][7]  written by the library (row g = leaf of slot g)
 *   logprobs_dev [n_slots][7] f32    read by the library (policy logits / log-probs)
 *   q_dev        [n_slots][2] f32    read by the library (q_penalty, q_no_penalty) */
int c4_session_bind_io(c4_session* s, void* planes_dev, const float* logprobs_dev, const float* q_dev,
                       void* stream);

/* Dirichlet root noise -- a BUILD EXTENSION named by BASELINE.json's north star; the reference has
 * none.  When epsilon > 0, the children of every search root get prior' = (1 - epsilon) * prior +
 * epsilon * eta with eta ~ Dir(alpha) over the legal columns, drawn from a private ChaCha12 stream
 * keyed by (game_id, moves played); specification = oracle/c4_oracle.c c4o_dirichlet, matched bit
 * for bit.  Off (epsilon = 0) by default and in every reference-parity run. */
int c4_session_set_dirichlet(c4_session* s, float alpha, float epsilon);

/* Multi-model games (player0_id != player1_id: tournaments, tournament.py:112-142): when bound,
 * every start/step also writes, for each slot, the id of the model that must evaluate its leaf
 * (MctsGame::leaf_model_id_to_play, mcts.rs:70-76) to leaf_models_dev [n_slots] (uint64), so the
 * caller can route rows to evaluators without leaving the device.  NULL unbinds. */
int c4_session_bind_leaf_models(c4_session* s, uint64_t* leaf_models_dev);

/* Evaluation cache -- a BUILD EXTENSION, off by default (the reference evaluates every leaf and only
 * dedups positions inside one batch, self_play.rs:203-208).  A direct-mapped table of n_entries
 * (rounded up to a power of two, 64 bytes each) in HBM keeps the evaluator's raw outputs by position;
 * a game whose freshly selected leaf is found there runs that simulation in the same launch, like a
 * terminal leaf, up to max_sims_per_step simulations per game per c4_session_step (0 = 6).  Samples
 * are unchanged provided the evaluator is a deterministic function of the position (a network in
 * inference mode is); one evaluator per session, so not together with c4_session_bind_leaf_models.
 * n_entries = 0 frees the table; c4_session_set_games empties it (new games may come with new weights).
 * Call before c4_session_start / graph capture.  Synchronises. */
int c4_session_set_eval_cache(c4_session* s, uint64_t n_entries, uint32_t max_sims_per_step);

/* Puts the first n_slots games on the slots and writes their first leaf (the start position)
 * to planes_dev: the state self_play() is in after self_play.rs:55-58. */
int c4_session_start(c4_session* s);

/* One MctsThread job (self_play.rs:268-323) for EVERY resident game: consume the evaluator
 * outputs for the current leaves (on_received_policy, mcts.rs:83-108), make a move when the
 * root has n_mcts_iterations visits (make_random_move, mcts.rs:214-222), finish and replace
 * finished games, select the next leaves and write them to planes_dev.  Asynchronous. */
int c4_session_step(c4_session* s);

/* c4_head_out_bf16 (below) + c4_session_step as ONE launch: the heads' output layers (nn.py:84-85, 98-99) are computed by
 * workgroups of 16 boards whose first two wavefronts go on as the step of those 16 games (mcts.rs:83-108, self_play.rs:268-323):
 * one launch boundary and the step's argument fetch / state-line round trip leave the per-round chain.  The outputs are also
 * written to the bound logprobs / q tensors, and every result equals the two-launch form bit for bit (same code).  Operands as
 * for c4_head_out_bf16, rows = the session's current width, stream = the session's.  Default configuration only: returns
 * C4_ERR_BAD_ARG when Dirichlet noise, the evaluation cache or per-launch timing (c4_session_set_timing) is on, or when
 * `features` is not a multiple of 1 344 -- callers then use the two entry points. */
int c4_session_step_head_out(c4_session* s, const void* hidden_policy_dev, const void* hidden_value_dev, const void* w_policy_dev,
                             const void* w_value_dev, const float* b_policy_dev, const float* b_value_dev, uint32_t features,
                             uint32_t policy_row_stride, uint32_t value_row_stride);
/* How the fused launch above shares a workgroup's 16 games among its wavefronts: 8 games per stepping wavefront (default) or 4.
 * A scheduling knob -- the games' records do not depend on it: 4 is 0.5 % faster where a second session's kernels share the chip
 * (the paired graph of c4a0_amd.session.capture_pair asks for it), 8 where the session is alone. */
int c4_session_set_step_shape(c4_session* s, uint32_t games_per_wavefront);

/* Per-launch device-clock timing of the step kernel (c4_counters.step_kernel_ns) needs a launch
 * sequence number in the kernel arguments, which a HIP-graph capture would freeze: switch it
 * off before capturing c4_session_step into a graph, on again for eager launches.  On by
 * default.  Synchronises. */
int c4_session_set_timing(c4_session* s, int enable);

/* Synchronises the stream and sums the per-wavefront counters. */
int c4_session_counters(c4_session* s, c4_counters* out);
/* Non-blocking completion probe: enqueues a copy of (games_done, error) to pinned host
 * memory; *games_done / *error hold the values of the previous probe that has landed. */
int c4_session_poll(c4_session* s, uint64_t* games_done, uint32_t* error);
/* The same probe, also reporting how many games have been taken off the request list so far. */
int c4_session_progress(c4_session* s, uint64_t* games_done, uint64_t* games_started, uint32_t* error);

/* Tail of a job: once every request has been started, finished slots stay empty while the evaluator
 * still computes rows for them.  Moves the remaining active games into the lowest slots (slot state,
 * arena, evaluator input row; which slot plays a game changes none of its samples) and narrows the
 * session to the smallest multiple of `multiple` (itself a multiple of 8) slots that holds them:
 * c4_session_step then launches that many slots and the caller evaluates only rows [0, *n_slots_now).
 * A no-op (returning the unchanged width) while requests are still queued.  Not for sessions with
 * c4_session_bind_leaf_models.  HIP graphs captured before the call carry the old width: re-capture.
 * c4_session_set_games restores the full width.  Synchronises. */
int c4_session_compact(c4_session* s, uint32_t multiple, uint32_t* n_active, uint32_t* n_slots_now);

/* GameResult list (types.rs:63-71, mcts.rs:271-313).  Two-call pattern: n_samples of every
 * game (host array of n_games uint32, 0 = unfinished), then the records of finished games
 * packed in reqs order into dst_host (capacity cap records).  Both synchronise the stream. */
/* How the session's tree arena was sized: its bytes, the blocks per slot, and the blocks per half of a reclaimed arena
 * (C4_FLAG_RECLAIM; 0 = never-reclaimed arena).  Any of the outputs may be NULL. */
int c4_session_arena(c4_session* s, uint64_t* bytes, uint32_t* blocks_per_slot, uint32_t* reclaim_half_blocks);
int c4_session_sample_counts(c4_session* s, uint32_t* counts_host, uint64_t n_games);
int c4_session_drain_samples(c4_session* s, c4_sample_rec* dst_host, uint64_t cap, uint64_t* n_written);
/* K6: packs the records of finished games, in reqs order, into the
 * caller's DEVICE buffer dst_dev (capacity cap records) so that one rank's samples are one
 * contiguous tensor for the RCCL all-gather.  *n_written = records packed.  Synchronises (the
 * per-game counts are prefix-summed on the host). */
int c4_session_pack_samples(c4_session* s, c4_sample_rec* dst_dev, uint64_t cap, uint64_t* n_written);
/* Diagnostic builds (-DC4_PHASE_STAMPS) only: per-wavefront device-clock stamps [n_waves][16] of the
 * last step launch; all zero in the product build. */
int c4_session_debug_phase_stamps(c4_session* s, uint64_t* out_host, uint64_t cap_words, uint64_t* n_words);
/* Device views for collectives (RCCL all-gather of samples): records [n_games][43], counts [n_games]. */
int c4_session_sample_store(c4_session* s, const c4_sample_rec** recs_dev, const uint32_t** counts_dev, uint64_t* n_games);

/* ---- the body of self_play() as ONE call (rust/src/self_play.rs:39-129): for a host that keeps the folded bf16 network in device
 * memory and does not want to write the schedule itself (c4a0_amd/csrc/c4_selfplay_host.hip says what the schedule is: two paired
 * sessions in one HIP graph, the output layers inside the step's launch, long graphs while slots refill and short ones in the tail,
 * narrowing, records merged in request order on the device).  Uses nothing but the entry points of this header and the HIP runtime.
 *
 * The network in the layouts the evaluator's entry points below take (what c4a0_amd/nn.py::InferenceNet holds: eval-mode BatchNorm
 * folded, the tower's weights in MFMA fragment order -- pack_tower_weights --, the first Linear of each head with its input
 * dimension permuted to the tower's [cell][channel] feature order): F = 42 * channels features; the first hidden layer of BOTH heads
 * merged into one [2F][F] matrix (policy rows first), then n_policy_hidden / n_value_hidden further [F][F] hidden layers per head,
 * then the output layers [7][F] and [2][F].  All weights bf16, all biases f32, all DEVICE pointers. */
typedef struct {
  uint32_t channels;          /* 32 or 64 (nn.py ModelConfig.conv_filter_size) */
  uint32_t n_blocks;          /* residual blocks */
  const void* tower_w0;       /* c4_conv_tower_bf16's w0_dev / w_dev / bias_dev */
  const void* tower_w;
  const float* tower_bias;
  const void* w1;             /* merged first hidden layer [2F][F], b1 [2F] */
  const float* b1;
  uint32_t n_policy_hidden;   /* hidden layers of the policy head AFTER the merged one (nn.py n_policy_layers - 2), <= 8 */
  uint32_t n_value_hidden;    /* ... of the value head (n_value_layers - 2), <= 8 */
  const void* policy_w[8];
  const float* policy_b[8];
  const void* value_w[8];
  const float* value_b[8];
  const void* policy_out_w;   /* [7][F] */
  const void* value_out_w;    /* [2][F] */
  const float* policy_out_b;  /* [7] */
  const float* value_out_b;   /* [2] */
} c4_network_bf16;

/* Everything optional (zero-initialise for the defaults; NULL = all defaults, device 0). */
typedef struct {
  int32_t device;                /* HIP device ordinal */
  uint32_t resident_games;       /* games advanced in lock-step; 0 = by job size: 4 096 / 8 192 / 16 384 (c4a0_amd/api.py default_resident_games) */
  uint32_t concurrent_sessions;  /* 0 = two paired sessions from 2 048 resident games -- except one generation (no more games than slots) of a 32-channel network --, else one; 1 or 2 to force */
  uint32_t steps_per_graph;      /* rounds per HIP-graph replay while slots are refilled; 0 = by job length (64 / 32 / 8) */
  uint32_t tail_steps_per_graph; /* ... from the first narrowing of the tail on; 0 = 16 (8 for short jobs) */
  uint32_t blocks_per_slot;      /* c4_config.blocks_per_slot */
  uint32_t flags;                /* c4_config.flags (C4_FLAG_RECLAIM / C4_FLAG_NO_RECLAIM / C4_FLAG_ONE_SIM_PER_STEP) */
  uint32_t reclaim_period;       /* c4_config.reclaim_period */
  float dirichlet_alpha;         /* EXTENSION, off while epsilon == 0: c4_session_set_dirichlet */
  float dirichlet_epsilon;
  uint64_t eval_cache_entries;   /* EXTENSION, off at 0: c4_session_set_eval_cache (split over the sessions) */
} c4_play_options;

/* Where the call's wall time went (seconds; setup_s + steady_s + tail_s + drain_s = the call) and what it chose. */
typedef struct {
  double setup_s;                /* sessions, arenas, activation buffers, streams */
  double capture_s;              /* all graph captures (the first one and those after a narrowing: inside steady_s / tail_s too) */
  double steady_s;               /* from the end of set-up (the first capture, then the first replay) until every request had been started (all slots busy) */
  double tail_s;                 /* after that, until the last game ended */
  double drain_s;                /* counters, merge, transfer of the records */
  uint64_t rounds;               /* lock-step rounds replayed */
  uint64_t rounds_until_all_started;
  uint32_t graph_captures;
  uint32_t resident_games;
  uint32_t sessions;
  uint32_t rows_at_end;          /* sum of the sessions' widths after the last narrowing */
} c4_play_phases;

/* Plays every game of reqs[0 .. n_games) to the end (self_play.rs:39-129 with the arguments of self_play.rs:39-46) and returns the
 * samples in REQUEST order: counts_host[g] = samples of game g (<= 43), records_host = their records back to back -- host memory, or
 * DEVICE memory of options->device when the records are to stay on the GPU (one rank's input to the RCCL all-gather of samples) --
 * (capacity records_cap records: 43 * n_games always suffice; a smaller buffer that turns out too small -> C4_ERR_BAD_ARG with the number
 * needed in *n_records).  totals (may be NULL) = the sessions' counters summed; phases may be NULL.  Synchronous; the records are the
 * same bytes whatever resident_games / concurrent_sessions / graph lengths are chosen (the evaluator is a function of the position,
 * a game's samples do not depend on the slot or session that plays it).  A device-side error (C4_ERR_NAN_IN_TREE, ...) is returned
 * as the status, with the slot in totals->error_slot.  Thread-safe by exclusion: calls of concurrent threads run one after another
 * (a job captures HIP graphs, which another job's set-up on the same process would break; a job fills the device anyway). */
int c4_play_games_bf16(const c4_game_metadata* reqs, uint64_t n_games, uint32_t n_mcts_iterations, float c_exploration,
                       float c_ply_penalty, const c4_network_bf16* net, const c4_play_options* options, uint32_t* counts_host,
                       c4_sample_rec* records_host, uint64_t records_cap, uint64_t* n_records, c4_counters* totals,
                       c4_play_phases* phases);
/* Asks the c4_play_games_bf16 call that is running (on whatever thread) to stop: it returns C4_ERR_CANCELLED after the graph replays
 * in flight (milliseconds), with everything given back and no records.  For a host's interrupt handling -- the reference's job is
 * stopped by killing the process; a job here can be minutes of one library call.  Has no effect when no job is running (a job
 * clears the request when it starts), and none on the c4_session_* entry points.  Callable from any thread. */
void c4_play_games_cancel(void);

/* ---- the hand-off right after the path: PlayGamesResult's wire format on the packed records (HOST functions: no device is
 * touched, they work on a machine without a GPU). ----
 * PlayGamesResult::to_cbor / __getstate__ (rust/src/pybridge.rs:73-92: `serde_cbor::to_vec(self)`; what `pickle.dump(games, f)` of
 * src/c4a0/training.py:62-63 runs every generation): games g = 0 .. n_games-1 with metadata metas[g] (types.rs:37-48) and counts[g]
 * samples each, the samples being consecutive entries of recs (pos, policy, q_penalty, q_no_penalty of types.rs:103-110; game_id
 * and meta of the records are not part of the wire format), written to dst as serde_cbor 0.11.2 writes the derive(Serialize)
 * structs: definite-length maps keyed by field name in declaration order, shortest-form unsigned integers, f32 as a half float
 * where that is lossless.  Two-call pattern: dst == NULL -> *n_written = the size of the document; then written to dst (cap >= it). */
int c4_records_to_cbor(const c4_game_metadata* metas, const uint32_t* counts, uint64_t n_games, const c4_sample_rec* recs,
                       uint64_t n_records, uint8_t* dst, uint64_t cap, uint64_t* n_written);
/* PlayGamesResult::from_cbor / __setstate__ (pybridge.rs:80-92: `serde_cbor::from_slice`), for documents in the form above (what
 * to_cbor and the reference write; floats of any width and integers are accepted where an f32 is expected).  Two-call pattern:
 * with metas == counts == recs == NULL the document is validated and counted (*n_games, *n_records); with buffers of those
 * capacities it is decoded, records getting game_id = their game's and meta = index | terminal flag << 16 like the generator's.
 * Anything else -> C4_ERR_BAD_ARG with the byte offset in c4_last_error_string() (the reference raises ValueError, pybridge.rs:254-259). */
int c4_cbor_to_records(const uint8_t* src, uint64_t len, c4_game_metadata* metas, uint32_t* counts, uint64_t cap_games,
                       c4_sample_rec* recs, uint64_t cap_records, uint64_t* n_games, uint64_t* n_records);

/* PlayGamesResult::split_train_test's permutation (pybridge.rs:110-112: `results.shuffle(&mut StdRng::seed_from_u64(seed))`, rand
 * 0.10.1): order[i] = index of the game the shuffle leaves at position i of a list of n_games (< 2^32 - 1) games.  The caller then
 * takes the first round(n_games * train_frac) games as the training set (pybridge.rs:113-119). */
int c4_shuffle_games(uint64_t seed, uint64_t n_games, uint32_t* order);

/* Root statistics of slot `slot` (MctsGame::root_policy / root_q_with_penalty /
 * root_q_no_penalty / root_visit_count, mcts.rs:248-268).  Synchronises. */
int c4_session_root_stats(c4_session* s, uint32_t slot, float policy[7], float* q_penalty,
                          float* q_no_penalty, uint64_t* visit_count, uint64_t* root_mask, uint64_t* root_value);
/* One int64 per slot identifying the leaf position waiting for the evaluator (value bits | the 7
 * column heights << 42; -1 for an idle slot), written to the DEVICE array keys_dev[n_slots] on the
 * session's stream, no synchronisation.  The callback evaluator sorts these on the device to find
 * the unique positions of a batch (NNThread::loop_once's HashSet, self_play.rs:203-208) instead of
 * copying every slot to the host. */
int c4_session_leaf_keys(c4_session* s, int64_t* keys_dev);

/* The callback evaluator's batch, built on the device (NNThread::loop_once, self_play.rs:203-208: the
 * reference collects the waiting leaves in a HashSet<(ModelID, Pos)> and evaluates each pair once).
 * For the session's resident games, on its stream, without synchronising:
 *   inverse_dev[n_slots]      row of each slot's (model, leaf position) pair in the batch, 0xFFFFFFFF for idle slots;
 *   rows_out[n_unique][2][6][7] float32 evaluator input of each unique pair (c4r.rs:378-392, pybridge.rs:202-221);
 *   models_out[n_unique]      its model id (mcts.rs:70-76; 0 without c4_session_bind_leaf_models), may be NULL;
 *   *n_unique_out             the number of rows.
 * Rows are ordered by the lowest slot holding the pair, so the batch depends on the games alone.
 * inverse_dev is device memory; rows_out (capacity n_slots rows), models_out (n_slots) and n_unique_out may be
 * device memory or PINNED HOST memory (hipHostMalloc / torch pin_memory): the kernels then write the
 * batch across PCIe themselves and the host reads it after synchronising the stream.  Anything else
 * (pageable host memory, another device's memory) is refused. */
int c4_session_unique_leaves(c4_session* s, uint32_t* inverse_dev, float* rows_out, uint64_t* models_out, uint32_t* n_unique_out);
/* The other half: answers[n_unique][9] (7 policy log-probabilities, q_penalty, q_no_penalty per row of the
 * batch above; device or pinned host memory) to the bound logprobs / q rows of every slot that asked --
 * what PyEvalPos hands back per position (pybridge.rs:161-199).  On the session's stream, no synchronisation:
 * the caller keeps `answers` untouched until the stream has passed this point. */
int c4_session_scatter_outputs(c4_session* s, const uint32_t* inverse_dev, const float* answers, uint32_t n_unique);

/* c4_session_scatter_outputs + c4_session_step as ONE launch: every game takes its evaluator outputs from row inverse_dev[slot] of
 * `answers` (as for c4_session_scatter_outputs: device or pinned host memory, 9 floats per row) inside the step kernel itself
 * (self_play.rs:222-236 hands each waiting game its result, :268-323 runs its job).  The bound logprobs / q tensors are NOT
 * updated: callers that watch them keep the two entry points.  Every configuration of the session (noise, cache, timing). */
int c4_session_step_gather(c4_session* s, const uint32_t* inverse_dev, const float* answers, uint32_t n_unique);

/* Leaf position currently waiting for the evaluator, per slot (MctsGame::leaf_pos, mcts.rs:64-66);
 * status[g] = 1 active / 0 idle, ordinal[g] = index of the slot's game in reqs.  Host arrays of
 * n_slots (any may be NULL).  Synchronises.  Used by the numpy-callback compatibility mode. */
int c4_session_leaves(c4_session* s, uint64_t* masks_host, uint64_t* values_host, uint32_t* status_host,
                      uint32_t* ordinals_host);

/* ---- element-wise device functions (SURVEY 8a kernel K1 and the arithmetic pieces), all on
 * device arrays of length n, launched on `stream`; used by the parity tests. ---- */
/* c4r.rs:58-72,228-238,253-263,266-269: for each position and column: moved position (0,0 if
 * illegal), legal mask, terminal state (0 none,1 PlayerWin,2 OpponentWin,3 Draw) and terminal values. */
int c4_pos_ops(const uint64_t* mask_dev, const uint64_t* value_dev, const int32_t* col_dev, uint64_t n,
               float c_ply_penalty, uint64_t* out_mask_dev, uint64_t* out_value_dev, uint32_t* out_legal_dev,
               uint32_t* out_terminal_dev, float* out_q_dev /* [n][2] */, void* stream);
/* c4r.rs:378-392 / pybridge.rs:202-221 */
int c4_encode_planes(const uint64_t* mask_dev, const uint64_t* value_dev, uint64_t n, uint32_t planes_dtype,
                     void* planes_dev, void* stream);
/* glibc expf/logf ports (what Rust f32::exp / f32::ln call) */
int c4_expf_logf(const float* x_dev, uint64_t n, int which /*0 expf, 1 logf*/, float* y_dev, void* stream);
/* mcts.rs:416-434 (with c4r.rs:272-286 masking when legal_dev != NULL); status per row in out_err_dev */
int c4_softmax7(const float* logits_dev, const uint32_t* legal_dev, uint64_t n, float* out_dev,
                uint32_t* out_err_dev, void* stream);
/* mcts.rs:439-454 */
int c4_apply_temperature(const float* policy_dev, const float* temperature_dev, uint64_t n, float* out_dev, void* stream);
/* extension: eta_dev[n][7] = Dir(alpha) noise for (game_id, n_moves, legal mask), see c4_session_set_dirichlet */
int c4_dirichlet(const uint64_t* game_id_dev, const uint32_t* n_moves_dev, const uint32_t* legal_dev, float alpha, uint64_t n,
                 float* eta_dev, void* stream);
/* mcts.rs:214-222 without the tree update: column sampled for (game_id, n_moves, policy, temperature);
 * out_col = -1 on DEGENERATE_POLICY.  out_u32 (may be NULL) = the RNG's first word. */
int c4_sample_move(const uint64_t* game_id_dev, const uint32_t* n_moves_dev, const float* policy_dev,
                   const float* temperature_dev, uint64_t n, int32_t* out_col_dev, uint32_t* out_u32_dev, void* stream);

/* ---- evaluator building block: the residual conv tower of ConnectFourNet (src/c4a0/nn.py:64-70,
 * 184-195; eval-mode BatchNorm folded into the second conv of each block) as one MFMA kernel.
 *   planes_dev bf16 [n_boards][2][6][7] (what c4_session_step writes with planes_dtype = 1)
 *   w0_dev / w_dev / bias_dev: MFMA-fragment-ordered weights, see c4a0_amd/nn.py::pack_tower_weights
 *   out_dev    bf16 [n_boards][42][channels]  (cell-major, channels last)
 * channels must be 32 or 64.  config: 0 = the workgroup shape chosen from n_boards (32 channels: 2 boards up to 512 boards, 4 up to
 * 1 024, 8 up to 1 280, else 16; 64 channels: 2 up to 512, 4 up to 1 024, else 8); 32 channels: 1 / 2 / 3 = 16 boards, 8 boards, 16 boards on 12 wavefronts, 4 / 5 = 4 boards on 12
 * wavefronts, 2 boards on 6 (the narrow launches of a job's tail); 64 channels: 2 / 3 = four wavefronts (one per SIMD, whole boards and all 64 output channels each)
 * with a weight ring 6 / 3 k-steps deep instead of the default eight wavefronts in channel-splitting pairs, 4 = the eight wavefronts
 * with the weights fetched once per workgroup into an LDS ring (a workgroup barrier per tap), 5 / 6 = 4 boards on eight wavefronts, 2 boards
 * on four (narrow launches), 1 = the default shape at any size -- every shape computes the same bits. */
int c4_conv_tower_bf16(const void* planes_dev, const void* w0_dev, const void* w_dev, const float* bias_dev,
                       uint32_t n_boards, uint32_t channels, uint32_t n_blocks, void* out_dev, uint32_t config, void* stream);

/* A hidden layer of a head (nn.py:75-100: Linear(F, F) + BatchNorm1d + ReLU, BN folded by the caller):
 *   y[m][n] = act(sum_k x[m][k] * w[n][k] + bias[n]),  x bf16 [m][ldx], w bf16 [n][k] (nn.Linear's layout),
 *   bias f32 [n], y bf16 [m][ldy]; relu != 0 applies max(., 0).  Hand-written MFMA kernel whose result for
 * a row depends on that row and the weights ONLY (one fixed summation order per element whatever m, the row
 * index or `config`): the evaluator is a function of the position, as the reference's is (one forward per
 * unique position, self_play.rs:203-237).  n % 192 == 0, k % 64 == 0 (42 * C features, C a multiple of 32); n, k, ldx,
 * ldy < 65 536; ldx % 8 == 0, ldy % 8 == 0 and x_dev, y_dev 16-byte aligned (rows are moved 16 bytes at a time); each operand
 * below 2 GiB.  config 0 = automatic, 1..59 = a specific tile configuration (tools/gemm_probe.py; all compute the same bits). */
int c4_linear_bf16(const void* x_dev, const void* w_dev, const float* bias_dev, void* y_dev, uint32_t m, uint32_t n,
                   uint32_t k, uint32_t ldx, uint32_t ldy, uint32_t relu, uint32_t config, void* stream);
/* The block -> tile map c4_linear_bf16 launches with for an m x n output cut into bm x bn tiles (XCD-rectangle order), computed
 * on the HOST with the kernels' own formula; no device is touched.  tiles_out[2 b] / [2 b + 1] = row / column tile of block b
 * (tiles_out may be NULL to ask for *n_blocks only).  For tests: the map must be a bijection onto the tile grid. */
int c4_linear_bf16_tile_map(uint32_t m, uint32_t n, uint32_t bm, uint32_t bn, uint32_t* tiles_out, uint32_t cap_blocks,
                            uint32_t* n_blocks);

/* The entry of `forward_numpy` (nn.py:119-130: `torch.from_numpy(x).to(device)` under autocast): float32 positions
 * [n_boards][2][6][7] -> the tower's bf16 planes [n_rows_out][2][6][7] (round to nearest even; rows n_boards .. n_rows_out - 1
 * become empty boards, so that a launch sized for a bucket of rows serves any smaller batch).  The batch may live in device
 * memory or in PINNED host memory (then the kernel reads it over PCIe: no staging copy).  batch_slot != NULL: the kernel
 * takes the batch's address AND its board count from *batch_slot (pinned host or device memory) instead of `src` /
 * `n_boards`, so that a captured launch (one HIP-graph replay per call) serves a different array every call.  The slot's
 * n_boards <= n_rows_out is the caller's to keep.  n_rows_out even; arrays and the slot 16-byte aligned. */
typedef struct c4_f32_batch {
  const float* data;
  uint32_t n_boards;
  uint32_t reserved;
} c4_f32_batch;
int c4_planes_from_f32(const c4_f32_batch* batch_slot, const float* src, uint32_t n_boards, void* planes_dev, uint32_t n_rows_out,
                       void* stream);

/* Output layers of both heads in one launch (nn.py:84-85,98-99): policy Linear(F->7) + LogSoftmax
 * and value Linear(F->2) + Tanh.  hidden_*_dev bf16 [n_boards][features] with row strides
 * *_row_stride elements (the last hidden activation of each head; they may be two column ranges
 * of one merged tensor, or the same pointer twice when a head has no hidden layer), w_* bf16
 * [7|2][features], b_* f32; writes logprobs_dev f32 [n_boards][7] and q_dev f32 [n_boards][2]
 * (the tensors bound with c4_session_bind_io).  features % 8 == 0. */
int c4_head_out_bf16(const void* hidden_policy_dev, const void* hidden_value_dev, const void* w_policy_dev,
                     const void* w_value_dev, const float* b_policy_dev, const float* b_value_dev,
                     uint32_t n_boards, uint32_t features, uint32_t policy_row_stride, uint32_t value_row_stride,
                     float* logprobs_dev, float* q_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif
