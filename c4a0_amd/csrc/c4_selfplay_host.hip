// c4_selfplay_host.hip -- the body of self_play() as ONE native call: the host loop around the sessions and the bf16 network.
//
// The reference's `self_play()` (rust/src/self_play.rs:39-129) is compiled code: it owns the threads, the queues and the loop
// "evaluate the waiting leaves -> give every game its MCTS job" until every game is over, and returns Vec<GameResult>.  The
// session entry points of this library (c4_session_*, c4_conv_tower_bf16, c4_linear_bf16, ...) are the pieces of that loop;
// c4_play_games_bf16 below IS the loop, for a host (the reference's Rust, a C program) that holds the folded bf16 network in
// device memory and wants the whole job done without writing the schedule itself:
//
//   * the resident games split over two sessions on two streams, both sessions' rounds captured into ONE HIP graph with an
//     explicit software pipeline between them (session B's [tower, first hidden layer] starts when session A's has finished;
//     DESIGN.md 1 / c4a0_amd/session.py capture_pair measured why), or one session alone below 2 048 resident games;
//   * the heads' output layers inside the step's launch (c4_session_step_head_out) in the default configuration;
//   * 64 rounds per graph replay while slots are refilled, 16 from the first narrowing of the tail on; at most three replays in
//     flight; completion and errors from the sessions' pinned probes (no stream synchronisation per replay);
//   * tail narrowing (c4_session_compact) when at most half of a session's rows still hold a game;
//   * the finished games' records packed in REQUEST order on the device and moved to the caller's buffer in one transfer.
//
// It uses nothing but this library's own C ABI and the HIP runtime -- it is the schedule of c4a0_amd/session.py (_run_pair,
// capture_pair) and c4a0_amd/api.py (_play) in C++, and tests/test_gpu_native_host.py holds the two to the same bytes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/c4a0_hip.h"
#include "c4_host.hpp"

namespace {

using c4host::fail;

#define HIP_OK(expr)                                                                                         \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess) return fail(C4_ERR_HIP, std::string("c4_play_games_bf16: " #expr ": ") + hipGetErrorString(e_)); \
  } while (0)
#define C4_TRY(expr)                \
  do {                              \
    const int rc_ = (expr);         \
    if (rc_ != C4_OK) return rc_;   \
  } while (0)

std::atomic<uint32_t> g_cancel_requested{0};   // c4_play_games_cancel -> the running job's loop

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// One session, its stream and the evaluator's activations for its rows.
struct Part {
  c4_session* s = nullptr;
  hipStream_t stream = nullptr;
  uint32_t slots = 0, rows = 0;
  uint64_t n_games = 0;
  void *planes = nullptr, *feat = nullptr, *h1 = nullptr, *pbuf[2] = {nullptr, nullptr}, *vbuf[2] = {nullptr, nullptr};
  float *logprobs = nullptr, *q = nullptr;
  uint64_t done = 0, started = 0;
};

struct Job {
  std::vector<Part> parts;
  hipGraphExec_t exec = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_stage = nullptr, ring[4] = {nullptr, nullptr, nullptr, nullptr};
  void *merged_dev = nullptr, *offsets_dev = nullptr;
  // Waits for everything this job has launched: its streams only -- the caller's other streams on the device (a training step, a
  // copy) are none of this loop's business and are never waited for after set-up.
  hipError_t sync() {
    for (Part& p : parts)
      if (p.stream) {
        const hipError_t e = hipStreamSynchronize(p.stream);
        if (e != hipSuccess) return e;
      }
    return hipSuccess;
  }
  ~Job() {
    if (sync() != hipSuccess) (void)hipDeviceSynchronize();
    if (exec) (void)hipGraphExecDestroy(exec);
    for (Part& p : parts) {
      if (p.s) (void)c4_session_destroy(p.s);
      for (void* b : {p.planes, p.feat, p.h1, p.pbuf[0], p.pbuf[1], p.vbuf[0], p.vbuf[1], (void*)p.logprobs, (void*)p.q})
        if (b) (void)hipFree(b);
      if (p.stream) (void)hipStreamDestroy(p.stream);
    }
    for (hipEvent_t e : {ev_fork, ev_join, ev_stage, ring[0], ring[1], ring[2], ring[3]})
      if (e) (void)hipEventDestroy(e);
    if (merged_dev) (void)hipFree(merged_dev);
    if (offsets_dev) (void)hipFree(offsets_dev);
  }
};

// Tile configuration of a hidden layer when ONE session has the device to itself (c4a0_amd/nn.py InferenceNet._alone_config: the
// measurements behind every threshold are cited there); 0 = the library's automatic choice, which is also what two paired
// sessions use.  Every configuration computes the same bits.
uint32_t alone_config(uint32_t m, uint32_t n, uint32_t k, bool latency) {
  if (!latency) return 0;
  const bool wide = n > k;
  if (k >= 2048) return (m <= 1024 || wide) ? 0 : 11;
  if (m <= 1024) {
    if (wide) return m <= 384 ? 41 : (m <= 576 ? 42 : (m <= 864 ? 44 : 43));
    return m <= 512 ? 41 : 42;
  }
  if (m <= 1728) return wide ? (m <= 1152 ? 43 : 59) : 44;
  return wide ? 35 : 43;
}

// One lock-step round of one session on its stream: tower, the heads' hidden layers, then the output layers and the step
// (one launch where the session's configuration allows it).  record_stage / wait_first: the pipeline edges of the paired graph.
int launch_round(Part& p, const c4_network_bf16& net, bool latency, bool fused, hipEvent_t wait_first, hipEvent_t record_stage, bool evaluate_only = false) {
  const uint32_t F = 42u * net.channels, rows = p.rows;
  void* st = (void*)p.stream;
  if (wait_first) HIP_OK(hipStreamWaitEvent(p.stream, wait_first, 0));
  C4_TRY(c4_conv_tower_bf16(p.planes, net.tower_w0, net.tower_w, net.tower_bias, rows, net.channels, net.n_blocks, p.feat,
                            (latency && net.channels == 32 && rows > 1024 && rows <= 2048) ? 2u : 0u, st));   // (as c4a0_amd/nn.py InferenceNet.tower)
  C4_TRY(c4_linear_bf16(p.feat, net.w1, net.b1, p.h1, rows, 2 * F, F, F, 2 * F, 1, alone_config(rows, 2 * F, F, latency), st));
  if (record_stage) HIP_OK(hipEventRecord(record_stage, p.stream));
  const void *hp = p.h1, *hv = (const char*)p.h1 + (size_t)F * 2;   // column ranges of the merged first layer's output
  uint32_t sp = 2 * F, sv = 2 * F;
  for (uint32_t i = 0; i < net.n_policy_hidden; i++) {
    C4_TRY(c4_linear_bf16(hp, net.policy_w[i], net.policy_b[i], p.pbuf[i & 1], rows, F, F, sp, F, 1, alone_config(rows, F, F, latency), st));
    hp = p.pbuf[i & 1]; sp = F;
  }
  for (uint32_t i = 0; i < net.n_value_hidden; i++) {
    C4_TRY(c4_linear_bf16(hv, net.value_w[i], net.value_b[i], p.vbuf[i & 1], rows, F, F, sv, F, 1, alone_config(rows, F, F, latency), st));
    hv = p.vbuf[i & 1]; sv = F;
  }
  if (fused && !evaluate_only) return c4_session_step_head_out(p.s, hp, hv, net.policy_out_w, net.value_out_w, net.policy_out_b, net.value_out_b, F, sp, sv);
  C4_TRY(c4_head_out_bf16(hp, hv, net.policy_out_w, net.value_out_w, net.policy_out_b, net.value_out_b, rows, F, sp, sv, p.logprobs, p.q, st));
  return evaluate_only ? C4_OK : c4_session_step(p.s);
}

// `rounds` rounds of every session as one executable graph, replayed on parts[0].stream.
int capture(Job& j, const c4_network_bf16& net, uint32_t rounds, bool fused) {
  const bool paired = j.parts.size() == 2;
  const bool latency = !paired;
  if (j.exec) { (void)hipGraphExecDestroy(j.exec); j.exec = nullptr; }
  hipStream_t s0 = j.parts[0].stream;
  // outside the capture, once at this width: the evaluator on the current leaves (first use of a tile shape opts its kernel in for
  // more than 64 KB of LDS, which is not a stream operation); evaluating the leaves once more changes nothing a game sees
  for (Part& p : j.parts) C4_TRY(launch_round(p, net, latency, fused, nullptr, nullptr, true));
  HIP_OK(j.sync());
  HIP_OK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
  int rc = C4_OK;
  hipError_t he = hipSuccess;
  if (paired) {
    he = hipEventRecord(j.ev_fork, s0);
    if (he == hipSuccess) he = hipStreamWaitEvent(j.parts[1].stream, j.ev_fork, 0);   // fork: the second stream joins the capture
  }
  for (uint32_t r = 0; r < rounds && rc == C4_OK && he == hipSuccess; r++) {
    rc = launch_round(j.parts[0], net, latency, fused, nullptr, (paired && r == 0) ? j.ev_stage : nullptr);
    if (rc == C4_OK && paired) rc = launch_round(j.parts[1], net, latency, fused, r == 0 ? j.ev_stage : nullptr, nullptr);
  }
  if (paired && rc == C4_OK && he == hipSuccess) {
    he = hipEventRecord(j.ev_join, j.parts[1].stream);
    if (he == hipSuccess) he = hipStreamWaitEvent(s0, j.ev_join, 0);                  // join
  }
  hipGraph_t graph = nullptr;
  const hipError_t he_end = hipStreamEndCapture(s0, &graph);                          // always: leaves the streams usable
  if (rc != C4_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (he != hipSuccess || he_end != hipSuccess) {
    if (graph) (void)hipGraphDestroy(graph);
    return fail(C4_ERR_HIP, std::string("c4_play_games_bf16: graph capture: ") + hipGetErrorString(he != hipSuccess ? he : he_end));
  }
  he = hipGraphInstantiate(&j.exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (he != hipSuccess) { j.exec = nullptr; return fail(C4_ERR_HIP, std::string("c4_play_games_bf16: hipGraphInstantiate: ") + hipGetErrorString(he)); }
  return C4_OK;
}

// The sessions' raw sample stores (43 records per game) -> one array in REQUEST order: request g was played by session g % parts
// as its game g / parts.  One wavefront per game.
struct MergeArgs {
  const c4_sample_rec* store[2];
  const uint32_t* counts[2];
  uint32_t parts;
};
__global__ __launch_bounds__(64) void k_merge_samples(MergeArgs a, const unsigned long long* offsets, uint64_t n_games, c4_sample_rec* dst) {
  const uint64_t g = blockIdx.x;
  if (g >= n_games) return;
  const uint32_t p = (uint32_t)(g % a.parts);
  const uint64_t i = g / a.parts;
  const uint32_t n = a.counts[p][i];
  const uint4* s4 = (const uint4*)(a.store[p] + i * C4_MAX_SAMPLES_PER_GAME);
  uint4* d4 = (uint4*)(dst + offsets[g]);
  for (uint32_t k = threadIdx.x; k < n * 4u; k += 64) d4[k] = s4[k];   // a 64-byte record = 4 x 16 bytes
}

}  // namespace

extern "C" int c4_play_games_bf16(const c4_game_metadata* reqs, uint64_t n_games, uint32_t n_mcts_iterations, float c_exploration,
                                  float c_ply_penalty, const c4_network_bf16* net, const c4_play_options* opt_in, uint32_t* counts_host,
                                  c4_sample_rec* records_host, uint64_t records_cap, uint64_t* n_records, c4_counters* totals,
                                  c4_play_phases* phases) {
  if (!net || !n_records || (n_games && (!reqs || !counts_host))) return fail(C4_ERR_BAD_ARG, "c4_play_games_bf16: null argument");
  c4_play_options opt{};
  if (opt_in) opt = *opt_in;
  *n_records = 0;
  if (totals) std::memset(totals, 0, sizeof *totals);
  if (phases) std::memset(phases, 0, sizeof *phases);
  if (n_games == 0) return C4_OK;
  if (net->channels != 32 && net->channels != 64) return fail(C4_ERR_BAD_ARG, "c4_play_games_bf16: the evaluator's kernels take 32 or 64 channels");
  if (!net->w1 || !net->b1) return fail(C4_ERR_BAD_ARG, "c4_play_games_bf16: both heads need a hidden layer (the merged first layer w1 / b1)");
  if (net->n_policy_hidden > 8 || net->n_value_hidden > 8) return fail(C4_ERR_BAD_ARG, "c4_play_games_bf16: at most 8 further hidden layers per head");
  // One job at a time: a job captures HIP graphs, and a capture does not tolerate what another job's set-up does meanwhile (allocations,
  // memsets and transfers on the legacy stream: "operation would make the legacy stream depend on a capturing stream") -- two threads of
  // a host calling at once used to fail that way.  A job fills the device anyway; the second caller waits here.
  static std::mutex one_job;
  std::lock_guard<std::mutex> hold(one_job);
  g_cancel_requested.store(0, std::memory_order_relaxed);   // a request is for the job that was running when it was made
  const double t0 = now_s();
  c4host::DeviceGuard guard(opt.device);
  if (guard.error() != hipSuccess) return fail(C4_ERR_HIP, std::string("c4_play_games_bf16: hipSetDevice: ") + hipGetErrorString(guard.error()));

  // ---- how many games are resident, in how many sessions, how many rounds per graph: c4a0_amd/api.py's rules (measured there)
  uint64_t resident = opt.resident_games;
  if (resident == 0) {
    resident = 4096;
    while (resident < 16384 && n_games >= 8 * resident) resident *= 2;
    const uint64_t n = std::max<uint32_t>(1u, n_mcts_iterations);
    const uint64_t per_slot = 128ull * (n <= 1000 ? 43 * n + 8 : 2 * (5 * n / 2 + 554));
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
      while (resident > 4096 && resident * per_slot > free_b / 4) resident /= 2;
  }
  resident = std::min<uint64_t>(resident, n_games);
  // two paired sessions from 2 048 resident games -- unless the job is one generation (nothing is ever refilled: it is all tail, and
  // below 2 048 rows one chain's round is shorter than two paired chains'; c4a0_amd/api.py _play has the measurements)
  // (measured with the 32-channel network; the 64-channel one, five times the arithmetic per row, keeps the pair)
  const bool one_generation = n_games <= resident && net->channels <= 32;
  uint32_t n_parts = opt.concurrent_sessions ? opt.concurrent_sessions : ((resident >= 2048 && !one_generation) ? 2u : 1u);
  if (n_parts > 2) return fail(C4_ERR_BAD_ARG, "c4_play_games_bf16: one session, or two paired ones");
  n_parts = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_parts, resident));
  const bool extensions = opt.dirichlet_epsilon > 0.0f || opt.eval_cache_entries != 0;
  const bool fused = !extensions;                       // what c4_session_step_head_out accepts (per-launch timing is switched off below)
  const uint64_t est_rounds = ((n_games + resident - 1) / resident) * 15ull * std::max<uint32_t>(1u, n_mcts_iterations);
  uint32_t steady = opt.steps_per_graph, tail = opt.tail_steps_per_graph;
  if (steady == 0) steady = n_parts == 2 ? (est_rounds >= 4000 ? 64u : (est_rounds >= 1500 ? 32u : 8u)) : (est_rounds >= 1500 ? 32u : 8u);
  if (tail == 0) tail = (n_parts == 1 && est_rounds >= 10000) ? steady : (steady >= 32 ? 16u : 8u);   // a long one-session job keeps its long graphs

  Job j;
  j.parts.resize(n_parts);
  const uint32_t F = 42u * net->channels;
  for (uint32_t p = 0; p < n_parts; p++) {
    Part& part = j.parts[p];
    part.n_games = (n_games + n_parts - 1 - p) / n_parts;                           // requests p, p + parts, ...
    part.slots = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(part.n_games, (resident + n_parts - 1 - p) / n_parts));
    part.rows = part.slots;
    c4_config cfg{};
    cfg.n_slots = part.slots;
    cfg.blocks_per_slot = opt.blocks_per_slot;
    cfg.n_mcts_iterations = n_mcts_iterations;
    cfg.c_exploration = c_exploration;
    cfg.c_ply_penalty = c_ply_penalty;
    cfg.planes_dtype = 1;
    cfg.flags = opt.flags;
    cfg.device = opt.device;
    cfg.reclaim_period = opt.reclaim_period;
    C4_TRY(c4_session_create(&cfg, &part.s));
    std::vector<c4_game_metadata> mine(part.n_games);
    for (uint64_t i = 0; i < part.n_games; i++) mine[i] = reqs[i * n_parts + p];
    C4_TRY(c4_session_set_games(part.s, mine.data(), part.n_games, nullptr, nullptr));
    if (opt.dirichlet_epsilon > 0.0f) C4_TRY(c4_session_set_dirichlet(part.s, opt.dirichlet_alpha, opt.dirichlet_epsilon));
    if (opt.eval_cache_entries) C4_TRY(c4_session_set_eval_cache(part.s, std::max<uint64_t>(1024, opt.eval_cache_entries / n_parts), 0));
#ifdef C4_DIAG_VARIANTS   // diagnostic build only (build.py --diag): the chip PARTITIONED between the two sessions by CU masks (measured, not adopted)
    static const int cu_mask_mode = [] { const char* e = getenv("C4_PAIR_CU_MASK"); return e ? atoi(e) : 0; }();
    if (cu_mask_mode && n_parts == 2) {
      // the queue's CU mask enumerates compute units round-robin over the 8 XCDs (bit i: XCD i % 8, CU i / 8 of it)
      uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (uint32_t cu = 0; cu < 256; cu++) {
        const bool mine = cu_mask_mode == 1 ? ((cu / 8) < 16) == (p == 0)      // 1: sixteen CUs of EVERY XCD per session
                                            : ((cu % 8) < 4) == (p == 0);      // 2: four whole XCDs (and their L2s) per session
        if (mine) mask[cu / 32] |= 1u << (cu % 32);
      }
      HIP_OK(hipExtStreamCreateWithCUMask(&part.stream, 8, mask));
    } else
#endif
    HIP_OK(hipStreamCreateWithFlags(&part.stream, hipStreamNonBlocking));
    const size_t rows = part.slots;
    HIP_OK(hipMalloc(&part.planes, rows * C4_PLANES_LEN * 2));
    HIP_OK(hipMalloc(&part.feat, rows * F * 2));
    HIP_OK(hipMalloc(&part.h1, rows * 2 * F * 2));
    for (int b = 0; b < 2; b++) {
      HIP_OK(hipMalloc(&part.pbuf[b], rows * F * 2));
      HIP_OK(hipMalloc(&part.vbuf[b], rows * F * 2));
    }
    HIP_OK(hipMalloc((void**)&part.logprobs, rows * C4_N_COLS * sizeof(float)));
    HIP_OK(hipMalloc((void**)&part.q, rows * 2 * sizeof(float)));
    HIP_OK(hipMemsetAsync(part.logprobs, 0, rows * C4_N_COLS * sizeof(float), part.stream));
    HIP_OK(hipMemsetAsync(part.q, 0, rows * 2 * sizeof(float), part.stream));
    C4_TRY(c4_session_bind_io(part.s, part.planes, part.logprobs, part.q, (void*)part.stream));
    C4_TRY(c4_session_set_timing(part.s, 0));            // the launch sequence number of per-launch timing would be frozen in a graph
    if (n_parts == 2) C4_TRY(c4_session_set_step_shape(part.s, 4));
    C4_TRY(c4_session_start(part.s));
  }
  for (hipEvent_t* e : {&j.ev_fork, &j.ev_join, &j.ev_stage, &j.ring[0], &j.ring[1], &j.ring[2], &j.ring[3]})
    HIP_OK(hipEventCreateWithFlags(e, hipEventDisableTiming));
  HIP_OK(hipDeviceSynchronize());
  const double t_setup = now_s();

  uint32_t per_graph = steady, captures = 0;
  double capture_s = 0.0;
  auto recapture = [&](uint32_t rounds) -> int {
    const double tc = now_s();
    const int rc = capture(j, *net, rounds, fused);
    capture_s += now_s() - tc;
    captures++;
    return rc;
  };
  C4_TRY(recapture(per_graph));
  double t_all_started = 0.0;
  uint64_t steps = 0, steps_all_started = 0, since_check = 0, launched = 0;
  hipStream_t s0 = j.parts[0].stream;
  for (;;) {
#ifdef C4_DIAG_VARIANTS   // eager launches instead of the graph (a graph's forked branch runs on an internal stream: no CU mask there)
    static const int eager = [] { const char* e = getenv("C4_PAIR_EAGER"); return e ? atoi(e) : 0; }();
    if (eager) {
      for (uint32_t r = 0; r < per_graph; r++)
        for (Part& p : j.parts) C4_TRY(launch_round(p, *net, j.parts.size() == 1, fused, nullptr, nullptr));
      if (j.parts.size() == 2) {    // the replay's join: the event ring lives on s0
        HIP_OK(hipEventRecord(j.ev_join, j.parts[1].stream));
        HIP_OK(hipStreamWaitEvent(s0, j.ev_join, 0));
      }
    } else
#endif
    HIP_OK(hipGraphLaunch(j.exec, s0));
    HIP_OK(hipEventRecord(j.ring[launched & 3], s0));
    launched++;
    steps += per_graph;
    since_check += per_graph;
    if (launched > 2) HIP_OK(hipEventSynchronize(j.ring[(launched - 3) & 3]));   // bounded run-ahead: at most three replays in flight
    bool done_all = true, started_all = true;
    for (Part& p : j.parts) {
      uint32_t err = 0;
      C4_TRY(c4_session_progress(p.s, &p.done, &p.started, &err));
      if (err) {
        (void)j.sync();
        c4_counters c{};
        (void)c4_session_counters(p.s, &c);
        if (totals) { totals->error = c.error; totals->error_slot = c.error_slot; }
        return fail((int)(c.error ? c.error : err), "c4_play_games_bf16: raised on the device by slot " + std::to_string(c.error_slot));
      }
      done_all = done_all && p.done >= p.n_games;
      started_all = started_all && p.started >= p.n_games;
    }
    if (started_all && t_all_started == 0.0) { t_all_started = now_s(); steps_all_started = steps; }
    if (done_all) break;
    if (g_cancel_requested.load(std::memory_order_relaxed)) {
      (void)j.sync();
      return fail(C4_ERR_CANCELLED, "c4_play_games_bf16: stopped by c4_play_games_cancel after " + std::to_string(steps) + " rounds");
    }
    if (since_check >= 64 && started_all) {
      since_check = 0;
      // tail: a session narrows when at most half of its rows still hold a game (decided from the pinned probes: the evaluator is a
      // function of the position, so WHEN a session narrows changes no sample); both decisions first, then the device is drained --
      // both sessions' kernels are nodes of one graph -- and only then are games moved
      bool any = false;
      std::vector<bool> wants(j.parts.size(), false);
      for (size_t k = 0; k < j.parts.size(); k++) {
        const Part& p = j.parts[k];
        wants[k] = p.rows > 256 && (p.n_games - p.done) <= p.rows / 2;
        any = any || wants[k];
      }
      if (any) {
        HIP_OK(j.sync());
        bool changed = false;
        for (size_t k = 0; k < j.parts.size(); k++) {
          if (!wants[k]) continue;
          uint32_t active = 0, rows_now = 0;
          C4_TRY(c4_session_compact(j.parts[k].s, 256, &active, &rows_now));
          changed = changed || rows_now != j.parts[k].rows;
          j.parts[k].rows = rows_now;
        }
        if (changed) {
          per_graph = tail;
          C4_TRY(recapture(per_graph));
          launched = 0;
        }
      }
    }
  }
  HIP_OK(j.sync());
  const double t_drain = now_s();

  // ---- counters, then the records: per-game counts to the host, offsets in request order back, one merge kernel, one transfer
  c4_counters sum{};
  for (Part& p : j.parts) {
    c4_counters c{};
    C4_TRY(c4_session_counters(p.s, &c));
    uint64_t* dst = &sum.sims;
    const uint64_t* src = &c.sims;
    for (size_t k = 0; k < offsetof(c4_counters, error) / sizeof(uint64_t); k++) dst[k] += src[k];
    if (c.error && !sum.error) { sum.error = c.error; sum.error_slot = c.error_slot; }
  }
  if (totals) *totals = sum;
  if (sum.error) return fail((int)sum.error, "c4_play_games_bf16: raised on the device by slot " + std::to_string(sum.error_slot));
  MergeArgs margs{};
  margs.parts = n_parts;
  std::vector<uint32_t> part_counts;
  for (uint32_t p = 0; p < n_parts; p++) {
    part_counts.resize(j.parts[p].n_games);
    C4_TRY(c4_session_sample_counts(j.parts[p].s, part_counts.data(), j.parts[p].n_games));
    for (uint64_t i = 0; i < j.parts[p].n_games; i++) counts_host[i * n_parts + p] = part_counts[i];
    uint64_t ng = 0;
    C4_TRY(c4_session_sample_store(j.parts[p].s, &margs.store[p], &margs.counts[p], &ng));
  }
  std::vector<unsigned long long> offsets(n_games);
  unsigned long long total = 0;
  for (uint64_t g = 0; g < n_games; g++) { offsets[g] = total; total += counts_host[g]; }
  *n_records = total;
  if (total > records_cap || (total && !records_host))
    return fail(C4_ERR_BAD_ARG, "c4_play_games_bf16: " + std::to_string(total) + " records, room for " + std::to_string(records_cap) +
                                " (43 per game always suffice)");
  if (total) {
    HIP_OK(hipMalloc(&j.offsets_dev, n_games * sizeof(unsigned long long)));
    HIP_OK(hipMalloc(&j.merged_dev, total * sizeof(c4_sample_rec)));
    HIP_OK(hipMemcpyAsync(j.offsets_dev, offsets.data(), n_games * sizeof(unsigned long long), hipMemcpyHostToDevice, s0));
    hipLaunchKernelGGL(k_merge_samples, dim3((unsigned)n_games), dim3(64), 0, s0, margs, (const unsigned long long*)j.offsets_dev, n_games,
                       (c4_sample_rec*)j.merged_dev);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(records_host, j.merged_dev, total * sizeof(c4_sample_rec), hipMemcpyDefault, s0));   // host memory, or a device buffer (the sample all-gather's input)
    HIP_OK(hipStreamSynchronize(s0));
  }
  const double t_end = now_s();
  if (phases) {
    phases->setup_s = t_setup - t0;
    phases->capture_s = capture_s;
    phases->steady_s = (t_all_started ? t_all_started : t_drain) - t_setup;   // (the first capture included: the phases add up to the call)
    phases->tail_s = t_drain - (t_all_started ? t_all_started : t_drain);
    phases->drain_s = t_end - t_drain;
    phases->rounds = steps;
    phases->rounds_until_all_started = steps_all_started;
    phases->graph_captures = captures;
    phases->resident_games = (uint32_t)resident;
    phases->sessions = n_parts;
    phases->rows_at_end = 0;
    for (const Part& p : j.parts) phases->rows_at_end += p.rows;
  }
  return C4_OK;
}

extern "C" void c4_play_games_cancel(void) { g_cancel_requested.store(1, std::memory_order_relaxed); }
