/* abi_consumer_nn.c -- the WHOLE hot path from plain C: self-play sessions AND the bf16 ResNet evaluator,
 * through include/c4a0_hip.h alone (no Python, no torch in this process).  What a Rust host of the reference
 * (rust/src/self_play.rs:39-129 with its NN thread, self_play.rs:196-237) would do if it kept the network on the
 * device itself instead of calling back into Python:
 *
 *   per round:  c4_conv_tower_bf16 (planes -> features)            nn.py:64-70,184-195
 *               c4_linear_bf16 x (1 + P + V) (hidden layers)        nn.py:75-100
 *               c4_head_out_bf16 (both output layers)               nn.py:84-85,98-99
 *               c4_session_step
 *
 * The weights come from a file written by tests/test_gpu_abi_consumer.py from a c4a0_amd.nn.InferenceNet
 * (BatchNorm folded, tower weights in MFMA fragment order): a header of six uint32
 * {channels, n_blocks, features, P, V, 0} (P / V = hidden layers of the policy / value head after the merged
 * first one), then blobs, each a uint64 byte count followed by the bytes, in this order:
 *   tower w0, tower w, tower bias | merged first layer w [2F][F] bf16, bias [2F] f32 |
 *   P x (w [F][F] bf16, bias [F] f32) | V x (w, bias) | policy out w [7][F] bf16, value out w [2][F] bf16,
 *   policy out bias [7] f32, value out bias [2] f32
 *
 *   abi_consumer_nn WEIGHTS N_GAMES N_SLOTS N_MCTS_ITERATIONS [native]
 * prints the samples in the format of abi_consumer.c; the test compares them with play_games(evaluator=net).
 * With "native" the whole job is ONE call, c4_play_games_bf16 (the library's own host loop: paired sessions in one HIP graph,
 * tail narrowing, records merged in request order) -- what a Rust host binds to replace the body of self_play() outright.
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "c4a0_hip.h"

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

#define MAX_HIDDEN 8

static uint32_t bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}

/* next blob of the weights file, copied to device memory */
static void* blob_to_device(FILE* f, uint64_t* n_bytes) {
  uint64_t n = 0;
  if (fread(&n, 8, 1, f) != 1) return NULL;
  void* h = malloc(n ? n : 1);
  if (n && fread(h, 1, n, f) != n) { free(h); return NULL; }
  void* d = NULL;
  if (hipMalloc(&d, n ? n : 16) != hipSuccess) { free(h); return NULL; }
  if (n && hipMemcpy(d, h, n, hipMemcpyHostToDevice) != hipSuccess) { free(h); return NULL; }
  free(h);
  if (n_bytes) *n_bytes = n;
  return d;
}

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s WEIGHTS N_GAMES N_SLOTS N_MCTS_ITERATIONS\n", argv[0]);
    return 1;
  }
  const uint64_t n_games = strtoull(argv[2], NULL, 10);
  const uint32_t G = (uint32_t)strtoul(argv[3], NULL, 10), n_iter = (uint32_t)strtoul(argv[4], NULL, 10);
  if (c4_abi_version() != C4_ABI_VERSION) {   /* the linker compares names, not signatures */
    fprintf(stderr, "libc4a0_hip.so implements ABI %d, this host was compiled for %d\n", c4_abi_version(), C4_ABI_VERSION);
    return 4;
  }
  HIP(hipSetDevice(0));

  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  uint32_t hdr[6];
  if (fread(hdr, 4, 6, f) != 6) { fprintf(stderr, "short weights file\n"); return 1; }
  const uint32_t channels = hdr[0], n_blocks = hdr[1], F = hdr[2], P = hdr[3], V = hdr[4];
  if (F != 42 * channels || P > MAX_HIDDEN || V > MAX_HIDDEN) { fprintf(stderr, "bad header\n"); return 1; }
  void *tw0, *tw, *tbias, *w1, *b1, *pw[MAX_HIDDEN], *pb[MAX_HIDDEN], *vw[MAX_HIDDEN], *vb[MAX_HIDDEN], *pow_, *vow, *pob, *vob;
#define NEXT(p) do { if (!((p) = blob_to_device(f, NULL))) { fprintf(stderr, "weights: blob " #p "\n"); return 1; } } while (0)
  NEXT(tw0); NEXT(tw); NEXT(tbias); NEXT(w1); NEXT(b1);
  for (uint32_t i = 0; i < P; i++) { NEXT(pw[i]); NEXT(pb[i]); }
  for (uint32_t i = 0; i < V; i++) { NEXT(vw[i]); NEXT(vb[i]); }
  NEXT(pow_); NEXT(vow); NEXT(pob); NEXT(vob);
  fclose(f);

  if (argc > 5 && strcmp(argv[5], "native") == 0) {
    c4_network_bf16 net;
    memset(&net, 0, sizeof net);
    net.channels = channels; net.n_blocks = n_blocks;
    net.tower_w0 = tw0; net.tower_w = tw; net.tower_bias = (const float*)tbias;
    net.w1 = w1; net.b1 = (const float*)b1;
    net.n_policy_hidden = P; net.n_value_hidden = V;
    for (uint32_t i = 0; i < P; i++) { net.policy_w[i] = pw[i]; net.policy_b[i] = (const float*)pb[i]; }
    for (uint32_t i = 0; i < V; i++) { net.value_w[i] = vw[i]; net.value_b[i] = (const float*)vb[i]; }
    net.policy_out_w = pow_; net.value_out_w = vow; net.policy_out_b = (const float*)pob; net.value_out_b = (const float*)vob;
    c4_play_options opt;
    memset(&opt, 0, sizeof opt);
    opt.resident_games = G;
    c4_game_metadata* reqs = (c4_game_metadata*)calloc(n_games ? n_games : 1, sizeof *reqs);
    for (uint64_t i = 0; i < n_games; i++) reqs[i].game_id = 900 + i;
    uint32_t* counts = (uint32_t*)calloc(n_games ? n_games : 1, sizeof *counts);
    const uint64_t cap = n_games * C4_MAX_SAMPLES_PER_GAME;
    c4_sample_rec* recs = (c4_sample_rec*)calloc(cap ? cap : 1, sizeof *recs);
    uint64_t n = 0;
    c4_counters c;
    c4_play_phases ph;
    C4(c4_play_games_bf16(reqs, n_games, n_iter, 6.6f, 0.01f, &net, &opt, counts, recs, cap, &n, &c, &ph));
    printf("games %" PRIu64 " sims %" PRIu64 " samples %" PRIu64 " expansions %" PRIu64 "\n", c.games_done, c.sims, c.samples, c.expansions);
    for (uint64_t i = 0; i < n; i++) {
      const c4_sample_rec* r = recs + i;
      printf("%" PRIu64 " %u %u %" PRIx64 " %" PRIx64, r->game_id, r->meta & 0xFFFFu, r->meta >> 16, r->mask, r->value);
      for (int k = 0; k < 7; k++) printf(" %08x", bits(r->policy[k]));
      printf(" %08x %08x\n", bits(r->q_penalty), bits(r->q_no_penalty));
    }
    fprintf(stderr, "native: %u session(s), %u resident games, %" PRIu64 " rounds, %u graph capture(s)\n", ph.sessions, ph.resident_games, ph.rounds, ph.graph_captures);
    free(recs); free(counts); free(reqs);
    return 0;
  }

  c4_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_slots = G;
  cfg.n_mcts_iterations = n_iter;
  cfg.c_exploration = 6.6f;
  cfg.c_ply_penalty = 0.01f;
  cfg.planes_dtype = 1;       /* bf16 planes: what c4_conv_tower_bf16 reads */
  cfg.device = 0;
  c4_session* s = NULL;
  C4(c4_session_create(&cfg, &s));
  c4_game_metadata* reqs = (c4_game_metadata*)calloc(n_games ? n_games : 1, sizeof *reqs);
  for (uint64_t i = 0; i < n_games; i++) reqs[i].game_id = 900 + i;
  C4(c4_session_set_games(s, reqs, n_games, NULL, NULL));

  /* activations: planes, tower features, the merged first layer's output (policy half | value half), two ping-pong
   * buffers per head for the layers after it */
  void *planes, *feat, *h1, *pbuf[2], *vbuf[2], *logprobs, *q;
  HIP(hipMalloc(&planes, (size_t)G * C4_PLANES_LEN * 2));
  HIP(hipMalloc(&feat, (size_t)G * F * 2));
  HIP(hipMalloc(&h1, (size_t)G * 2 * F * 2));
  for (int i = 0; i < 2; i++) { HIP(hipMalloc(&pbuf[i], (size_t)G * F * 2)); HIP(hipMalloc(&vbuf[i], (size_t)G * F * 2)); }
  HIP(hipMalloc(&logprobs, (size_t)G * C4_N_COLS * sizeof(float)));
  HIP(hipMalloc(&q, (size_t)G * 2 * sizeof(float)));
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));
  C4(c4_session_bind_io(s, planes, (const float*)logprobs, (const float*)q, (void*)stream));
  C4(c4_session_start(s));

  uint64_t done = 0, steps = 0;
  uint32_t err = 0;
  while (done < n_games && steps < 10000000ull) {
    for (int k = 0; k < 8; k++) {
      C4(c4_conv_tower_bf16(planes, tw0, tw, (const float*)tbias, G, channels, n_blocks, feat, 0, (void*)stream));
      C4(c4_linear_bf16(feat, w1, (const float*)b1, h1, G, 2 * F, F, F, 2 * F, 1, 0, (void*)stream));
      const void *hp = h1, *hv = (const char*)h1 + (size_t)F * 2;   /* column ranges of the merged tensor */
      uint32_t sp = 2 * F, sv = 2 * F;
      for (uint32_t i = 0; i < P; i++) {
        C4(c4_linear_bf16(hp, pw[i], (const float*)pb[i], pbuf[i & 1], G, F, F, sp, F, 1, 0, (void*)stream));
        hp = pbuf[i & 1]; sp = F;
      }
      for (uint32_t i = 0; i < V; i++) {
        C4(c4_linear_bf16(hv, vw[i], (const float*)vb[i], vbuf[i & 1], G, F, F, sv, F, 1, 0, (void*)stream));
        hv = vbuf[i & 1]; sv = F;
      }
      C4(c4_head_out_bf16(hp, hv, pow_, vow, (const float*)pob, (const float*)vob, G, F, sp, sv, (float*)logprobs, (float*)q, (void*)stream));
      C4(c4_session_step(s));
    }
    steps += 8;
    HIP(hipStreamSynchronize(stream));
    C4(c4_session_poll(s, &done, &err));
    if (err) break;
  }
  c4_counters c;
  C4(c4_session_counters(s, &c));
  if (c.error) { fprintf(stderr, "device error %u in slot %u\n", c.error, c.error_slot); return 4; }
  if (c.games_done != n_games) { fprintf(stderr, "only %" PRIu64 " of %" PRIu64 " games finished\n", c.games_done, n_games); return 5; }
  uint64_t n = 0;
  C4(c4_session_drain_samples(s, NULL, 0, &n));
  c4_sample_rec* recs = (c4_sample_rec*)calloc(n ? n : 1, sizeof *recs);
  C4(c4_session_drain_samples(s, recs, n, &n));
  printf("games %" PRIu64 " sims %" PRIu64 " samples %" PRIu64 " expansions %" PRIu64 "\n", c.games_done, c.sims, c.samples, c.expansions);
  for (uint64_t i = 0; i < n; i++) {
    const c4_sample_rec* r = recs + i;
    printf("%" PRIu64 " %u %u %" PRIx64 " %" PRIx64, r->game_id, r->meta & 0xFFFFu, r->meta >> 16, r->mask, r->value);
    for (int k = 0; k < 7; k++) printf(" %08x", bits(r->policy[k]));
    printf(" %08x %08x\n", bits(r->q_penalty), bits(r->q_no_penalty));
  }
  C4(c4_session_destroy(s));
  free(recs);
  free(reqs);
  return 0;
}
