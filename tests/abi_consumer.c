/* abi_consumer.c -- a NON-Python consumer of the C ABI, written against include/c4a0_hip.h alone
 * (plain C11, no torch, no ctypes): what the Rust `extern "C"` binding of INTEGRATION.md would do in
 * place of self_play() (reference rust/src/self_play.rs:39-129, called from pybridge.rs:20-53).
 *
 *   create -> set_games -> bind raw hipMalloc'd buffers -> start -> { "evaluate", step } until poll says
 *   done -> counters -> drain_samples (two-call pattern) -> print -> destroy
 *
 * The evaluator is the constant one of the oracle's "zeros" kind: all policy outputs 0, both values 0 --
 * a hipMemset of the bound output buffers, once.  tests/test_gpu_abi_consumer.py builds this file with
 * gcc, runs it and compares every printed sample with the oracle's.
 *
 *   abi_consumer N_GAMES N_SLOTS N_MCTS_ITERATIONS [DEVICE]
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "c4a0_hip.h"

_Static_assert(sizeof(c4_sample_rec) == 64, "sample record layout");
_Static_assert(sizeof(c4_game_metadata) == 24, "request layout");

#define C4(call)                                                                          \
  do {                                                                                    \
    int rc_ = (call);                                                                     \
    if (rc_ != C4_OK) {                                                                   \
      fprintf(stderr, "%s -> status %d: %s\n", #call, rc_, c4_last_error_string());       \
      return 10 + rc_;                                                                    \
    }                                                                                     \
  } while (0)
#define HIP(call)                                                                         \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_));                        \
      return 2;                                                                           \
    }                                                                                     \
  } while (0)

static uint32_t bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s N_GAMES N_SLOTS N_MCTS_ITERATIONS [DEVICE]\n", argv[0]);
    return 1;
  }
  const uint64_t n_games = strtoull(argv[1], NULL, 10);
  const uint32_t n_slots = (uint32_t)strtoul(argv[2], NULL, 10), n_iter = (uint32_t)strtoul(argv[3], NULL, 10);
  const int device = argc > 4 ? atoi(argv[4]) : 0;

  if (c4_abi_version() != C4_ABI_VERSION) {   /* the linker compares names, not signatures */
    fprintf(stderr, "libc4a0_hip.so implements ABI %d, this host was compiled for %d\n", c4_abi_version(), C4_ABI_VERSION);
    return 4;
  }
  int n_dev = 0;
  C4(c4_device_count(&n_dev));
  if (n_dev <= device) {
    fprintf(stderr, "no HIP device %d\n", device);
    return 3;
  }
  c4_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_slots = n_slots;
  cfg.n_mcts_iterations = n_iter;
  cfg.c_exploration = 6.6f;   /* main.py:42 */
  cfg.c_ply_penalty = 0.01f;  /* main.py:43 */
  cfg.planes_dtype = 0;       /* float32 evaluator input */
  cfg.device = device;
  c4_session* s = NULL;
  C4(c4_session_create(&cfg, &s));

  c4_game_metadata* reqs = (c4_game_metadata*)calloc(n_games ? n_games : 1, sizeof *reqs);
  for (uint64_t i = 0; i < n_games; i++) reqs[i].game_id = 500 + 3 * i;   /* same model on both sides: ids 0 */
  C4(c4_session_set_games(s, reqs, n_games, NULL, NULL));

  /* the caller owns the evaluator's tensors: raw device memory here (PyTorch tensors in the Python host) */
  HIP(hipSetDevice(device));
  void *planes = NULL, *logprobs = NULL, *q = NULL;
  HIP(hipMalloc(&planes, (size_t)n_slots * C4_PLANES_LEN * sizeof(float)));
  HIP(hipMalloc(&logprobs, (size_t)n_slots * C4_N_COLS * sizeof(float)));
  HIP(hipMalloc(&q, (size_t)n_slots * 2 * sizeof(float)));
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));
  /* the whole "network": policy outputs 0, values 0, for every position, for ever */
  HIP(hipMemsetAsync(logprobs, 0, (size_t)n_slots * C4_N_COLS * sizeof(float), stream));
  HIP(hipMemsetAsync(q, 0, (size_t)n_slots * 2 * sizeof(float), stream));
  C4(c4_session_bind_io(s, planes, (const float*)logprobs, (const float*)q, (void*)stream));
  C4(c4_session_start(s));

  uint64_t done = 0, steps = 0;
  uint32_t err = 0;
  while (done < n_games && steps < 10000000ull) {
    for (int k = 0; k < 16; k++) C4(c4_session_step(s));   /* evaluator outputs are constant: nothing to launch between steps */
    steps += 16;
    HIP(hipStreamSynchronize(stream));
    C4(c4_session_poll(s, &done, &err));   /* asynchronous probe: reports the state as of an earlier call */
    if (err) break;
  }
  c4_counters c;
  C4(c4_session_counters(s, &c));          /* synchronises the stream */
  if (c.error) {
    fprintf(stderr, "device error %u in slot %u\n", c.error, c.error_slot);
    return 4;
  }
  if (c.games_done != n_games) {
    fprintf(stderr, "only %" PRIu64 " of %" PRIu64 " games finished\n", c.games_done, n_games);
    return 5;
  }
  uint64_t n = 0;
  C4(c4_session_drain_samples(s, NULL, 0, &n));            /* size query */
  c4_sample_rec* recs = (c4_sample_rec*)calloc(n ? n : 1, sizeof *recs);
  C4(c4_session_drain_samples(s, recs, n, &n));
  printf("games %" PRIu64 " sims %" PRIu64 " samples %" PRIu64 " expansions %" PRIu64 "\n", c.games_done, c.sims, c.samples, c.expansions);
  for (uint64_t i = 0; i < n; i++) {
    const c4_sample_rec* r = recs + i;
    printf("%" PRIu64 " %u %u %" PRIx64 " %" PRIx64, r->game_id, r->meta & 0xFFFFu, r->meta >> 16, r->mask, r->value);
    for (int k = 0; k < 7; k++) printf(" %08x", bits(r->policy[k]));
    printf(" %08x %08x\n", bits(r->q_penalty), bits(r->q_no_penalty));
  }
  /* a second bind on the wrong device / a foreign pointer must be refused, not crash later */
  if (c4_session_bind_io(s, NULL, (const float*)logprobs, (const float*)q, (void*)stream) == C4_OK) {
    fprintf(stderr, "bind_io accepted a null tensor\n");
    return 6;
  }
  C4(c4_session_destroy(s));
  HIP(hipFree(planes));
  HIP(hipFree(logprobs));
  HIP(hipFree(q));
  HIP(hipStreamDestroy(stream));
  free(recs);
  free(reqs);
  return 0;
}
