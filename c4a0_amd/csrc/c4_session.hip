// c4_session.hip -- GPU-resident batched MCTS self-play for Connect Four on MI355X (gfx950),
// behind the C ABI of include/c4a0_hip.h.
//
// Replaces the reference's CPU loop (rust/src/self_play.rs + mcts.rs + c4r.rs): G games stay
// resident in HBM and advance in lock-step, one MCTS simulation per game per `c4_session_step`
// (plus a second one in the same launch when the leaf just selected is terminal and so needs no
// evaluator).
//
// Mapping to the hardware
//   * one game  <-> one 8-lane group of a wave64 (lanes 0..6 = the 7 children of a node,
//     lane 7 = bookkeeping); 8 games per wavefront, one wavefront per workgroup so that a
//     launch of G games spreads over G/8 workgroups (G = 4096 -> 512 workgroups on 256 CUs).
//   * tree node storage = "children blocks": the 7 children of an expanded node live in ONE
//     128-byte, 128-byte-aligned block {n, q_penalty, q_no_penalty, prior} x 7 + seven 16-bit
//     child links, so one select level is one cache line read by one 16-byte load per lane, the
//     child link needed for the next level arrives with it, and a backup touches one sector.
//   * no parent links: select records the root->leaf path in the slot state and the next
//     step's backup walks that list with independent loads (no pointer chase).
//   * positions are not stored in nodes: select replays make_move from the root position.
//   * arenas are bump-allocated per slot and never compacted: 288 GB of HBM holds the worst
//     case (43 * n_mcts_iterations blocks per game) for thousands of games.
//
// Bit-exactness: f32 arithmetic follows the reference operation by operation; this file is
// compiled with -ffp-contract=off (see build.py) and uses the glibc expf/logf ports.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/c4a0_hip.h"
#include "c4_device.hpp"
#include "c4_head_out.hpp"
#include "c4_host.hpp"
#include "c4_timeline.hpp"

#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------------------------------
// HBM layout
// ------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) Entry {
  uint32_t n;     // visit_count            (mcts.rs:335)
  float q_pen;    // q_sum_penalty          (mcts.rs:336)
  float q_nopen;  // q_sum_no_penalty       (mcts.rs:337)
  float prior;    // initial_policy_value   (mcts.rs:338)
};
struct __attribute__((aligned(16))) Tail {
  uint16_t child[7];  // per column: block holding that child's own children, 0 = not expanded (mcts.rs:339)
  uint16_t legal;     // legal-move mask of the parent position (informational)
};
// The 7 children of an expanded node: ONE 128-byte line.  Lane c < 7 of a game's lane group loads
// entry c, lane 7 the tail, in a single 16-byte-per-lane instruction; a backup touches one entry
// (one 32-byte sector).  16-bit child links bound an arena to 65 535 blocks per slot.
struct __attribute__((aligned(128))) Block {
  Entry e[7];
  Tail t;
};
constexpr uint32_t kMaxBlocksPerSlot = 65535;

constexpr uint32_t kMaxPath = 43;   // root + at most 42 moves below it
constexpr uint32_t kHotPath = 16;   // path levels kept in the slot's hot line
// A resident game's state.  Everything a simulation needs is ONE 128-byte line, read by the game's
// 8 lanes with one 16-byte-per-lane instruction at the start of the step kernel and written back
// with one at the end (lane k owns dwords 4k..4k+3):
//   lane 0: root position          lane 1: leaf position (waiting for the evaluator)
//   lane 2: game id, ordinal, root visit count
//   lane 3: state word, arena words, root ref, precomputed move RNG word
//   lanes 4..7: the recorded path, TRANSPOSED: lane 4 + j holds levels j, j + 4, j + 8, j + 12, so
//               the backup of level d runs on lane 4 + (d & 3) and the first four levels update in parallel
// The second line holds path levels 16..42 (deep searches only).
struct __attribute__((aligned(256))) Slot {
  uint64_t root_mask, root_value;  // MctsGame::root position
  uint64_t leaf_mask, leaf_value;  // MctsGame::leaf position
  uint64_t game_id;
  uint32_t ordinal;     // index into reqs / the sample store
  uint32_t root_n;      // mirror of the root entry's visit count
  uint32_t state;       // status[0:8] (0 idle, 1 active, >1 = c4_status error) | depth[8:16] (path[depth] = the leaf's entry)
                        // | n_moves[16:24] | terminal_state of the leaf [24:26] | rng_for[26:32] (n_moves + 1 rng_word is for; 0 = none)
  uint32_t arena;       // n_blocks[0:16] (bump pointer) | root_block[16:32] (the root's children block, 0 = not expanded)
  uint32_t root_ref;    // (block << 3 | column) of the root's own entry
  uint32_t rng_word;    // first ChaCha12 word for the NEXT move (mcts.rs:215-216), precomputed off the critical path
  uint32_t path[kHotPath];        // entry refs of levels 0..15, transposed: path[4 * j + i] = level j + 4 * i
  uint32_t path_deep[kMaxPath - kHotPath];   // levels 16..42
  uint32_t pad_[32 - (kMaxPath - kHotPath)];
};
static_assert(sizeof(Slot) == 256 && offsetof(Slot, path) == 64 && offsetof(Slot, path_deep) == 128, "slot state: one hot line + the deep path");
C4_DEV constexpr uint32_t slot_state(uint32_t status, uint32_t depth, uint32_t n_moves, uint32_t term, uint32_t rng_for) {
  return status | (depth << 8) | (n_moves << 16) | (term << 24) | (rng_for << 26);
}
C4_DEV uint32_t slot_status(uint32_t state) { return state & 0xFFu; }
static_assert(sizeof(Block) == 128 && sizeof(Entry) == 16 && sizeof(Tail) == 16 && sizeof(c4_sample_rec) == 64, "layout");

enum : uint32_t { kIdle = 0, kActive = 1 };
constexpr uint32_t kWavesPerTimingHelper = 1024;   // stamps one timing helper workgroup reduces

// Diagnostic build only (-DC4_PHASE_STAMPS, tools/phase_profile.py): per-wavefront device-clock
// stamps at the phase boundaries of the step kernel.  `force` makes the stamp wait for the values
// the phase produced.  Never compiled into libc4a0_hip.so.
#ifdef C4_PHASE_STAMPS
#define C4_STAMP(i, force)                                                                          \
  do {                                                                                              \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                       \
    if (lane == 0) p.phase[(size_t)wave_index * 16 + (i)] = __builtin_amdgcn_s_memrealtime() + ((force) & 0);  \
  } while (0)
// stamp by whichever lane is active first (second trip of the simulation loop: lane 0's game may have left it)
#define C4_STAMP_ANY(i)                                                                             \
  do {                                                                                              \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                       \
    if (lane == (uint32_t)(__ffsll((long long)__ballot(1)) - 1)) p.phase[(size_t)wave_index * 16 + (i)] = __builtin_amdgcn_s_memrealtime();  \
  } while (0)
#else
#define C4_STAMP(i, force) do { } while (0)
#define C4_STAMP_ANY(i) do { } while (0)
#endif
enum : int { CTR_SIMS = 0, CTR_S, CTR_K, CTR_E, CTR_MOVES, CTR_DONE, CTR_SKIPPED, CTR_SAMPLES, CTR_PROBES, CTR_HITS, CTR_N = 16 };

struct Globals {             // one small device struct of cross-wave words
  unsigned long long queue_head;   // next game ordinal to start
  unsigned long long games_done;
  uint32_t error;            // first error status
  uint32_t error_slot;
};

struct Params {
  Slot* slots;
  Block* blocks;
  unsigned long long* wave_ctr;  // [n_waves][CTR_N]
  unsigned long long* stamps;    // [2][n_waves][2] start/end device clock of each wavefront, by launch parity
  unsigned long long* clock_acc; // [0] sum of (last end - first start) over launches, [1] launches summed, [2..4] the timing helpers' scratch
  uint32_t n_waves;
  uint32_t seq;                  // launch sequence number
  unsigned long long* phase;     // diagnostic build: [n_waves][16] phase stamps of the last launch
  uint64_t* leaf_models;         // optional [n_slots]: model id that must evaluate each slot's leaf (mcts.rs:70-76)
  Globals* glob;
  const c4_game_metadata* reqs;
  const uint64_t* start_mask;    // may be null
  const uint64_t* start_value;
  c4_sample_rec* samples;        // [n_games][43]
  uint32_t* sample_counts;       // [n_games]
  void* planes;
  const float* logprobs;
  const float* q;
  unsigned long long n_games;
  uint32_t n_slots;
  uint32_t blocks_per_slot;
  uint32_t n_iter;
  float c_exploration;
  float c_ply_penalty;
  uint32_t flags;
  float dir_alpha, dir_eps;       // Dirichlet root noise (extension); dir_eps == 0 disables
  uint2* cache;                   // optional evaluation cache (extension): [cache_mask + 1] entries of 64 bytes, 8 x uint2
  uint32_t cache_mask;
  uint32_t max_sims;              // simulations one game may run in one launch (terminal / cached leaves need no evaluator)
  const float* ln_tab;            // ln_tab[k] = c4_logf((float)k), k < n_ln: the parent-visit term of uct_value (mcts.rs:379)
  uint32_t n_ln;
  uint32_t half_blocks;           // reclaimed arenas (C4_FLAG_RECLAIM): blocks per half, blocks_per_slot = 2 x this; 0 = never-reclaimed arena
  unsigned long long* reclaim_ctr;   // [2] passes, blocks copied (k_arena_reclaim)
};

// ------------------------------------------------------------------------------------------
// Evaluation cache (extension, off by default): evaluator outputs keyed by position.
// Entry = 16 dwords: [0..1] mask, [2..3] value, [4..10] the 7 policy outputs, [11..12] the 2 values,
// [13] seal, [14..15] 0.  Lane `sub` of a game's group owns dwords 2 sub, 2 sub + 1, so an entry is
// read and written by ONE 8-byte-per-lane instruction.  The seal makes the XOR of all 16 dwords a
// constant: an empty (zeroed) or torn entry never validates.  Direct-mapped, always overwritten.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kCacheMagic = 0xC4A0C4A0u;
C4_DEV uint32_t cache_index(uint64_t mask, uint64_t value, uint32_t cache_mask) {
  const uint64_t h = (mask * 0x9E3779B97F4A7C15ull) ^ (value * 0xC2B2AE3D27D4EB4Full);
  return (uint32_t)(h >> 24) & cache_mask;
}

// ------------------------------------------------------------------------------------------
// 8-lane group helpers
// ------------------------------------------------------------------------------------------
C4_DEV uint32_t shfl_u32(uint32_t v, int src_lane) { return (uint32_t)__shfl((int)v, src_lane, 64); }
C4_DEV float shfl_f32(float v, int src_lane) { return __shfl(v, src_lane, 64); }

// Exchanges inside an 8-lane group without the LDS crossbar (DPP): lane ^ 1, lane ^ 2, and 7 - lane.
// After the first two every lane of a quad holds the quad's combination, so the mirror step pairs
// the two quads: three steps reduce a group for any commutative, associative combination.
template <int kStep>
C4_DEV uint32_t grp_xchg(uint32_t v) {
  static_assert(kStep >= 0 && kStep < 3, "three butterfly steps");
  constexpr int ctrl = kStep == 0 ? 0xB1 /* quad_perm [1,0,3,2] */ : (kStep == 1 ? 0x4E /* quad_perm [2,3,0,1] */ : 0x141 /* row_half_mirror */);
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xF, 0xF, true);
}
template <int kStep>
C4_DEV float grp_xchg(float v) { return __uint_as_float(grp_xchg<kStep>(__float_as_uint(v))); }

// lane `sub` of a group loads its 16 bytes of block `blk`: entry `sub` (sub < 7) or the tail (sub == 7)
C4_DEV uint4 load_block_lane(const Block* blocks, uint32_t blk, uint32_t sub) {
  return reinterpret_cast<const uint4*>(blocks + blk)[sub];
}
// child link of column `col` out of the tail held by lane 7 of the group (col is group-uniform)
C4_DEV uint32_t child_link(const uint4& raw, uint32_t col, int gbase) {
  const uint32_t w = col >> 1;
  const uint32_t mine = w == 0 ? raw.x : (w == 1 ? raw.y : (w == 2 ? raw.z : raw.w));
  return (shfl_u32(mine, gbase + 7) >> (16u * (col & 1u))) & 0xFFFFu;
}

// Evaluation cache, lane-group side (entry layout at kCacheMagic).  `logit` = policy output `sub` on
// lanes 0..6; q_pen / q_nopen group-uniform.
C4_DEV void cache_store(uint2* cache, uint32_t cache_mask, uint64_t mask, uint64_t value, float logit, float q_pen,
                        float q_nopen, uint32_t sub, int gbase) {
  const uint32_t lb = __float_as_uint(logit);
  const int s2 = 2 * ((int)sub - 2);
  uint32_t a = shfl_u32(lb, gbase + (s2 < 0 ? 0 : (s2 > 6 ? 6 : s2)));
  uint32_t b = shfl_u32(lb, gbase + (s2 + 1 < 0 ? 0 : (s2 + 1 > 6 ? 6 : s2 + 1)));
  if (sub == 0) { a = (uint32_t)mask; b = (uint32_t)(mask >> 32); }
  if (sub == 1) { a = (uint32_t)value; b = (uint32_t)(value >> 32); }
  if (sub == 5) b = __float_as_uint(q_pen);
  if (sub == 6) { a = __float_as_uint(q_nopen); b = 0u; }
  if (sub == 7) { a = 0u; b = 0u; }
  uint32_t x = a ^ b;
  x ^= grp_xchg<0>(x); x ^= grp_xchg<1>(x); x ^= grp_xchg<2>(x);
  if (sub == 6) b = x ^ kCacheMagic;                               // the 16 dwords now XOR to the constant
  cache[(size_t)cache_index(mask, value, cache_mask) * 8 + sub] = make_uint2(a, b);
}
// true (for the whole group) when a valid entry for (mask, value) is present; then the outputs are set
C4_DEV bool cache_lookup(const uint2* cache, uint32_t cache_mask, uint64_t mask, uint64_t value, float& logit, float& q_pen,
                         float& q_nopen, uint32_t sub, int gbase) {
  const uint2 w = cache[(size_t)cache_index(mask, value, cache_mask) * 8 + sub];
  uint32_t x = w.x ^ w.y;
  x ^= grp_xchg<0>(x); x ^= grp_xchg<1>(x); x ^= grp_xchg<2>(x);
  const uint64_t key = ((uint64_t)w.y << 32) | w.x;
  const bool kok = sub == 0 ? key == mask : (sub == 1 ? key == value : true);
  const bool hit = x == kCacheMagic && ((__ballot(kok) >> gbase) & 0xFFull) == 0xFFull;
  if (hit) {
    const int src = gbase + (int)((4 + sub) >> 1);                   // dword 4 + sub lives in lane (4 + sub) / 2
    const uint32_t ux = shfl_u32(w.x, src), uy = shfl_u32(w.y, src);
    logit = __uint_as_float(((4 + sub) & 1u) ? uy : ux);
    q_pen = __uint_as_float(shfl_u32(w.y, gbase + 5));
    q_nopen = __uint_as_float(shfl_u32(w.x, gbase + 6));
  }
  return hit;
}

// Level `level` of the path a game's lanes hold (pv of lane 4 + j = levels j, j + 4, j + 8, j + 12),
// for every lane of the group; levels beyond the hot line come from the slot's second line.
C4_DEV uint32_t path_level(const uint4& pv, const Slot* st, uint32_t level, int gbase) {
  const uint32_t c = (level >> 2) & 3u;
  const uint32_t mine = c == 0 ? pv.x : (c == 1 ? pv.y : (c == 2 ? pv.z : pv.w));
  const uint32_t hotv = shfl_u32(mine, gbase + 4 + (int)(level & 3u));
  return level < kHotPath ? hotv : st->path_deep[level - kHotPath];
}

// ln(visit count) of uct_value (mcts.rs:379): visit counts are small integers, so the glibc logf port's
// result is read from a table filled by that same port at session creation (L1-resident: 4 bytes per
// count); counts beyond the table (reference KATs with tens of thousands of iterations) compute it.
C4_DEV float ln_visits(const Params& p, uint32_t n) {
  return n < p.n_ln ? p.ln_tab[n] : c4::c4_logf((float)n);
}

// select_new_leaf (mcts.rs:160-183) for one game on its 8 lanes: from the root down to the first
// unexpanded node, replaying make_move, recording the path (entry refs) in the lanes as the slot's
// hot line keeps it: lane 4 + j holds levels j, j + 4, j + 8, j + 12 in pv[0..3] (deeper levels go
// straight to the slot's second line).  Returns 0 or C4_ERR_NAN_IN_TREE.
C4_DEV uint32_t select_leaf(const Params& p, const Block* blocks, Slot* st, uint64_t rmask, uint64_t rvalue, uint32_t root_block,
                            uint32_t root_ref, uint32_t root_n, float c_exploration, uint32_t sub, int gbase,
                            uint64_t& leaf_mask, uint64_t& leaf_value, uint32_t& depth, uint32_t& leaf_ref,
                            uint4& pv, uint32_t& levels) {
  uint64_t m = rmask, v = rvalue;
  uint32_t blk = root_block, d = 0, last_ref = root_ref;
  float ln_np = ln_visits(p, root_n);                     // ln(parent visits) of the level being scored
  uint32_t nan_seen = 0;
  pv.x = (sub == 4) ? root_ref : pv.x;                    // level 0 of the path = the root's own entry
  while (blk != 0 && d + 1 < kMaxPath) {
    const uint4 ce = load_block_lane(blocks, blk, sub);   // {n, q_pen, q_nopen, prior} | tail
    // off the dependent chain: ln of THIS lane's child's visit count -- the parent term of the next
    // level if that child wins (lane 7 holds the tail, not an entry)
    const float my_ln = ln_visits(p, sub < 7 ? ce.x : 0u);
    const uint32_t legal = c4::legal_mask(m);
    const bool ok = sub < 7 && ((legal >> sub) & 1u);
    // uct_value (mcts.rs:359-388), computed on every lane without a branch (a lane without a legal
    // move computes on whatever it loaded and its score is never looked at); ln(1) == 0 makes every
    // first-level score -0+0
    const float nf = (float)ce.x + 1.0f;
    const float qv = __uint_as_float(ce.y) / nf;
    float ex = ln_np / nf;
    ex = __builtin_sqrtf(ex);
    ex = ex * (__uint_as_float(ce.w) + 1e-8f);
    const float cx = c_exploration * ex;
    const float score = -qv + cx;
    // max_by_key keeps the LAST maximum (mcts.rs:165-173); a NaN among two or more candidates panics
    // (utils.rs:12).  The walk finishes the level either way and reports after the loop: an errored
    // game's state is never used again.
    nan_seen |= (ok && (score != score)) ? (uint32_t)(__popc(legal) >= 2) : 0u;
    // Argmax as the maximum of a 64-bit key: high word = the score mapped monotonically onto
    // unsigned integers (-0 first folded into +0: the two compare equal as floats), low word =
    // column + 1, so equal scores go to the LAST column and a lane without a legal move (key 0)
    // never wins.  Three DPP exchanges, one 64-bit compare each.
    const uint32_t sbits = __float_as_uint(score + 0.0f);
    const uint32_t ord = sbits ^ ((uint32_t)((int32_t)sbits >> 31) | 0x80000000u);
    unsigned long long key = ok ? (((unsigned long long)ord << 32) | (sub + 1u)) : 0ull;
#define C4_KEYMAX_STEP(K)                                                                                          \
    {                                                                                                          \
      const unsigned long long o = ((unsigned long long)grp_xchg<K>((uint32_t)(key >> 32)) << 32) | grp_xchg<K>((uint32_t)key); \
      key = o > key ? o : key;                                                                                 \
    }
    C4_KEYMAX_STEP(0) C4_KEYMAX_STEP(1) C4_KEYMAX_STEP(2)
#undef C4_KEYMAX_STEP
    const uint32_t best = (uint32_t)key - 1u;
    // the winner's ln (next level's parent term) and its child link (out of the tail, lane 7)
    ln_np = shfl_f32(my_ln, gbase + (int)best);
    const uint32_t w = best >> 1;
    const uint32_t tw = w == 0 ? ce.x : (w == 1 ? ce.y : (w == 2 ? ce.z : ce.w));
    const uint32_t next_blk = (shfl_u32(tw, gbase + 7) >> (16u * (best & 1u))) & 0xFFFFu;
    c4::make_move(m, v, best);
    d += 1;
    last_ref = (blk << 3) | best;
    {                                                      // the lanes keep the path: level d on lane 4 + (d & 3)
      const bool mine = sub == 4u + (d & 3u);
      const uint32_t comp = d >> 2;
      pv.x = (mine && comp == 0) ? last_ref : pv.x;
      pv.y = (mine && comp == 1) ? last_ref : pv.y;
      pv.z = (mine && comp == 2) ? last_ref : pv.z;
      pv.w = (mine && comp == 3) ? last_ref : pv.w;
    }
    if (d >= kHotPath && sub == 0) st->path_deep[d - kHotPath] = last_ref;   // rare: beyond what the lanes hold
    blk = next_blk;
    levels += 1;
  }
  leaf_mask = m; leaf_value = v; depth = d; leaf_ref = last_ref;
  return ((__ballot(nan_seen != 0) >> gbase) & 0xFFull) ? (uint32_t)C4_ERR_NAN_IN_TREE : 0u;
}

C4_DEV void raise_error(const Params& p, Slot* st, uint32_t g, uint32_t code) {
  st->state = (st->state & ~0xFFu) | code;
  if (atomicCAS(&p.glob->error, 0u, code) == 0u) p.glob->error_slot = g;
}

template <typename PlaneT>
C4_DEV void store_plane(void* base, size_t idx, uint32_t bit);
template <>
C4_DEV void store_plane<float>(void* base, size_t idx, uint32_t bit) {
  ((float*)base)[idx] = bit ? 1.0f : 0.0f;
}
template <>
C4_DEV void store_plane<uint16_t>(void* base, size_t idx, uint32_t bit) {
  ((uint16_t*)base)[idx] = bit ? (uint16_t)0x3F80 : (uint16_t)0;  // bf16 1.0 / 0.0
}

// A game's leaf as the evaluator's input row (c4r.rs:378-392): 84 elements, plane 0 = the bits of
// `value`, plane 1 = the opponent's pieces, in bit order.  bf16: the 84-bit string value | opp << 42 is
// cut into bytes, lane `sub` expands byte `sub` (and, lanes 0..2, the bits 64..83) into eight bf16 0/1
// and writes them with one 16-byte store -- two store instructions per game instead of eleven.
template <typename PlaneT>
C4_DEV void encode_leaf(void* planes, uint32_t g, uint64_t mask, uint64_t value, uint32_t sub);
template <>
C4_DEV void encode_leaf<float>(void* planes, uint32_t g, uint64_t mask, uint64_t value, uint32_t sub) {
  for (uint32_t e = sub; e < C4_PLANES_LEN; e += 8) store_plane<float>(planes, (size_t)g * C4_PLANES_LEN + e, c4::plane_bit(mask, value, e));
}
C4_DEV uint4 bf16_bits8(uint32_t byte) {   // bit j of `byte` -> bf16 1.0 / 0.0 in element j
  uint4 r;
  r.x = ((byte >> 0) & 1u) * 0x3F80u + ((byte >> 1) & 1u) * 0x3F800000u;
  r.y = ((byte >> 2) & 1u) * 0x3F80u + ((byte >> 3) & 1u) * 0x3F800000u;
  r.z = ((byte >> 4) & 1u) * 0x3F80u + ((byte >> 5) & 1u) * 0x3F800000u;
  r.w = ((byte >> 6) & 1u) * 0x3F80u + ((byte >> 7) & 1u) * 0x3F800000u;
  return r;
}
template <>
C4_DEV void encode_leaf<uint16_t>(void* planes, uint32_t g, uint64_t mask, uint64_t value, uint32_t sub) {
  const uint64_t opp = mask & ~value;
  const uint64_t lo = value | (opp << 42);          // elements 0..63
  const uint32_t hi = (uint32_t)(opp >> 22);        // elements 64..83
  uint16_t* row = (uint16_t*)planes + (size_t)g * C4_PLANES_LEN;
  const uint4 a = bf16_bits8((uint32_t)(lo >> (8u * sub)) & 0xFFu);
  // rows are 168 bytes apart: 8-byte aligned, so the 16 bytes go out as two 8-byte stores
  reinterpret_cast<uint2*>(row + 8 * sub)[0] = make_uint2(a.x, a.y);
  reinterpret_cast<uint2*>(row + 8 * sub)[1] = make_uint2(a.z, a.w);
  if (sub < 3) {
    const uint4 b = bf16_bits8((hi >> (8u * sub)) & 0xFFu);
    reinterpret_cast<uint2*>(row + 64 + 8 * sub)[0] = make_uint2(b.x, b.y);
    if (sub < 2) reinterpret_cast<uint2*>(row + 64 + 8 * sub)[1] = make_uint2(b.z, b.w);
  }
}

// MctsGame::leaf_model_id_to_play (mcts.rs:70-76): player 0's model on even plies, player 1's on odd.
C4_DEV void publish_leaf_model(const Params& p, uint32_t g, unsigned long long ordinal, uint64_t leaf_mask) {
  if (p.leaf_models) {
    const c4_game_metadata md = p.reqs[ordinal];
    p.leaf_models[g] = (__popcll(leaf_mask) & 1) ? md.player1_id : md.player0_id;
  }
}

// Put game `ordinal` on a slot: MctsGame::new_from_pos (mcts.rs:48-56).  Called by all 8 lanes.
C4_DEV void reset_slot(const Params& p, Slot* st, Block* blocks, uint32_t sub, unsigned long long ordinal) {
  const uint64_t m = p.start_mask ? p.start_mask[ordinal] : 0ull;
  const uint64_t v = p.start_value ? p.start_value[ordinal] : 0ull;
  // block 0 holds only the root's own entry (prior 1.0, mcts.rs:49) in column 0
  reinterpret_cast<uint4*>(blocks)[sub] = make_uint4(0, 0, 0, sub == 0 ? __float_as_uint(1.0f) : 0u);
  if (sub == 0) {
    st->root_mask = m; st->root_value = v;
    st->leaf_mask = m; st->leaf_value = v;
    st->game_id = p.reqs[ordinal].game_id;
    st->ordinal = (uint32_t)ordinal;
    st->root_n = 0;
    // the start position may be anything, terminal included: full terminal_state
    st->state = slot_state(kActive, 0, 0, c4::terminal_state(m, v), 0);
    st->arena = 1;            // n_blocks = 1, root not expanded
    st->root_ref = 0;
    st->path[0] = 0;
  }
}

// ------------------------------------------------------------------------------------------
// K0: start -- the state after self_play.rs:55-58 (every slot holds a fresh game whose leaf is
// its start position) with the first leaves written to the evaluator's input tensor.
// ------------------------------------------------------------------------------------------
template <typename PlaneT>
__global__ __launch_bounds__(64) void c4_start_kernel(Params p) {
  const uint32_t sub = threadIdx.x & 7;
  const uint32_t g = blockIdx.x * (blockDim.x >> 3) + (threadIdx.x >> 3);
  if (g >= p.n_slots) return;
  Slot* st = p.slots + g;
  Block* blocks = p.blocks + (size_t)g * p.blocks_per_slot;
  uint64_t m = 0, v = 0;
  if (g < p.n_games) {
    reset_slot(p, st, blocks, sub, g);
    m = p.start_mask ? p.start_mask[g] : 0ull;
    v = p.start_value ? p.start_value[g] : 0ull;
    if (sub == 0) publish_leaf_model(p, g, g, m);
  } else if (sub == 0) {
    st->state = kIdle;
    st->ordinal = 0xFFFFFFFFu;
  }
  for (uint32_t e = sub; e < C4_PLANES_LEN; e += 8) store_plane<PlaneT>(p.planes, (size_t)g * C4_PLANES_LEN + e, c4::plane_bit(m, v, e));
}

// ------------------------------------------------------------------------------------------
// K2-K5 fused: one MCTS simulation for every resident game.
//   expand (mcts.rs:114-132, softmax mcts.rs:416-434)  ->  backup (mcts.rs:137-155)
//   -> gate / move / finish / refill (self_play.rs:283-308, mcts.rs:187-222, 271-313)
//   -> select (mcts.rs:160-183)  ->  encode the new leaf (c4r.rs:378-392)
// ------------------------------------------------------------------------------------------
// The extra workgroups a timed launch carries (they own no games; one per kWavesPerTimingHelper
// wavefronts): duration of the PREVIOUS launch = last wavefront end - first wavefront start (its stamps are
// complete: kernel boundary).  Each helper reduces its chunk, the last one to finish adds the launch to the
// totals.  No game's critical path carries any of this.
C4_DEV void timing_helper(const Params& p, uint32_t lane) {
  if (p.seq <= 1) return;
  const uint32_t h = blockIdx.x - p.n_waves, n_helpers = gridDim.x - p.n_waves;
  const unsigned long long* prev = p.stamps + (size_t)((p.seq - 1) & 1) * p.n_waves * 2;
  const uint32_t w_end = (h + 1) * kWavesPerTimingHelper < p.n_waves ? (h + 1) * kWavesPerTimingHelper : p.n_waves;
  unsigned long long lo = ~0ull, hi = 0ull;
#pragma unroll 8
  for (uint32_t w = h * kWavesPerTimingHelper + lane; w < w_end; w += 64) {
    const ulonglong2 ab = reinterpret_cast<const ulonglong2*>(prev)[w];
    lo = ab.x < lo ? ab.x : lo;
    hi = ab.y > hi ? ab.y : hi;
  }
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long ol = ((unsigned long long)shfl_u32((uint32_t)(lo >> 32), (int)(lane ^ off)) << 32) | shfl_u32((uint32_t)lo, (int)(lane ^ off));
    const unsigned long long oh = ((unsigned long long)shfl_u32((uint32_t)(hi >> 32), (int)(lane ^ off)) << 32) | shfl_u32((uint32_t)hi, (int)(lane ^ off));
    lo = ol < lo ? ol : lo;
    hi = oh > hi ? oh : hi;
  }
  if (lane == 0) {
    unsigned long long* tmp = p.clock_acc + 2;   // [2] running min start, [3] running max end, [4] helpers done
    atomicMin(&tmp[0], lo);
    atomicMax(&tmp[1], hi);
    __threadfence();
    if (atomicAdd(&tmp[2], 1ull) == n_helpers - 1) {
      __threadfence();
      const unsigned long long first = atomicExch(&tmp[0], ~0ull), last = atomicExch(&tmp[1], 0ull);
      atomicExch(&tmp[2], 0ull);
      if (last > first) { p.clock_acc[0] += last - first; p.clock_acc[1] += 1; }
    }
  }
}

// Wavefronts per SIMD the default instantiation is compiled for.  Measured (tools/tree_roofline.py, 65 536 /
// 131 072 games per launch): 3 (137 registers) 37.7 / 71.4 us, 4 (127 registers, no spill) 35.1 / 64.4 us,
// 5 (96 registers, spills) 46.2 / 93.6 us; at 2 048 games all three take 10.3 us.  The extension
// instantiations keep what they need.
#ifndef C4_STEP_WAVES
#define C4_STEP_WAVES 4
#endif
// The step of the 8 games of ONE wavefront (the body of c4_step_kernel, and of the fused output + step kernel below).
// `wave_index` = the wavefront's row in the per-wavefront arrays (games 8 wave_index .. + 7), `hot` = this lane's 16 bytes of its
// game's state line, `nn_logit` / `nn_q` = the evaluator's outputs for the game (lane sub < 7: logit sub; q_penalty / q_no_penalty
// on even / odd lanes), `t_start` = the wavefront's start stamp (timed launches).
template <typename PlaneT, bool NOISE, bool CACHE>
C4_DEV void step_body(const Params& p, const uint32_t wave_index, const uint32_t lane, const uint32_t n_slots, Slot* st, const uint4 hot,
                      const float nn_logit, const float nn_q, const unsigned long long t_start, const uint32_t g) {
  // g = this lane group's game (>= n_slots: none).  The stand-alone kernels give a wavefront the 8 games 8 wave_index .. + 7; the fused
  // output + step kernel may give it fewer (c4_session_set_step_shape), its other lane groups idle.
  const uint32_t sub = lane & 7;
  const int gbase = (int)(lane & ~7u);
  uint32_t c_sims = 0, c_S = 0, c_K = 0, c_E = 0, c_moves = 0, c_done = 0, c_skipped = 0, c_samples = 0;   // this launch only
  uint32_t c_probes = 0, c_hits = 0;
  // header words to every lane of the group (lane 3: state, arena, root ref, rng word)
  const uint32_t state0 = shfl_u32(hot.x, gbase + 3);
  bool active = (g < n_slots) && (slot_status(state0) == kActive);
  uint4 line = hot;          // what goes back to the slot at the end
  bool store_line = false;
  // move RNG precompute (see the end of the kernel): what this game will need at its next move
  bool pre_need = false;
  uint32_t pre_n_moves = 0;
  unsigned long long pre_game_id = 0;

  if (active) {
    Block* blocks = p.blocks + (size_t)g * p.blocks_per_slot;

#define C4_BCAST64(lo, hi, src) (((uint64_t)shfl_u32(hi, gbase + (src)) << 32) | shfl_u32(lo, gbase + (src)))
    uint64_t rmask = C4_BCAST64(hot.x, hot.y, 0), rvalue = C4_BCAST64(hot.z, hot.w, 0);
    uint64_t leaf_mask = C4_BCAST64(hot.x, hot.y, 1), leaf_value = C4_BCAST64(hot.z, hot.w, 1);
    unsigned long long game_id = C4_BCAST64(hot.x, hot.y, 2);
#undef C4_BCAST64
    uint32_t ordinal = shfl_u32(hot.z, gbase + 2);
    const uint32_t arena0 = shfl_u32(hot.y, gbase + 3);
    uint32_t root_ref = shfl_u32(hot.z, gbase + 3);
    uint32_t rng_word = shfl_u32(hot.w, gbase + 3);
    uint32_t depth = (state0 >> 8) & 0xFFu;
    uint32_t n_moves = (state0 >> 16) & 0xFFu;
    uint32_t term = (state0 >> 24) & 3u;          // terminal_state of the waiting leaf, computed when it was selected
    uint32_t rng_for = state0 >> 26;
    uint32_t n_blocks = arena0 & 0xFFFFu;
    uint32_t root_block = arena0 >> 16;
    // the recorded path as the line keeps it: lane 4 + j holds levels j, j + 4, j + 8, j + 12
    uint4 pv = hot;
    uint32_t leaf_ref = path_level(pv, st, depth, gbase);   // the waiting leaf's own entry
    bool fresh = false;                  // the slot took a new game in this launch
    uint32_t err = 0;
    uint32_t root_n = 0;
    // One simulation per game per launch, plus a second one when the leaf just selected is terminal:
    // such a leaf needs no evaluator row (mcts.rs:92-98 ignores the network for it), so its value is
    // backed up here and now instead of idling through an evaluator pass.  The order of a game's
    // simulations and every value in them is unchanged; C4_FLAG_NO_MOVES / C4_FLAG_ONE_SIM_PER_STEP
    // keep exactly one.
    // (p.max_sims: 1 under those flags, else 2 -- measured: a third trip costs more launch time than
    // it saves rows -- or more with the evaluation cache, whose hits need no evaluator either.)
    const uint32_t max_sims = p.max_sims;
    // this trip's evaluator outputs: the network's for the first trip, a cached entry's after a hit
    float cur_logit = nn_logit;
    float cur_qp = shfl_f32(nn_q, gbase), cur_qn = shfl_f32(nn_q, gbase + 1);

    C4_STAMP(1, depth + n_blocks + pv.x + (uint32_t)leaf_mask);
#define C4_STAMP_TRIP1(i, force) do { if (sim == 0) C4_STAMP(i, force); } while (0)
#pragma clang loop unroll(disable)
    for (uint32_t sim = 0; sim < max_sims; sim++) {
      // ---------------- on_received_policy: terminal value or expansion -------------------
      float v_pen, v_nopen;
      if (term) {
        c4::terminal_value(term, leaf_mask, p.c_ply_penalty, v_pen, v_nopen);  // NN output ignored (mcts.rs:92-98)
      } else {
        // first simulation: the network's outputs; a later one gets here only on an evaluation-cache hit
        const uint32_t legal = c4::legal_mask(leaf_mask);
        const bool is_legal = sub < 7 && ((legal >> sub) & 1u);
        float logit = __uint_as_float(0xff800000u);                             // mask_policy, c4r.rs:272-286
        if (is_legal) logit = cur_logit;
        float mx = logit;                                                       // f32::max fold (NaN-ignoring)
        mx = c4::rust_max(mx, grp_xchg<0>(mx));
        mx = c4::rust_max(mx, grp_xchg<1>(mx));
        mx = c4::rust_max(mx, grp_xchg<2>(mx));
        if (__builtin_isinf(mx)) err = C4_ERR_DEGENERATE_POLICY;                // mcts.rs:421-425
        const float ex = c4::c4_expf(logit - mx);
        float sum = 0.0f;                                                       // left-to-right, mcts.rs:432
        for (int i = 0; i < 7; i++) sum = sum + shfl_f32(ex, gbase + i);
        float prior = ex / sum;
        if (NOISE && p.dir_eps > 0.0f && depth == 0) {
          // extension: the root is expanded only now -> its children start with noisy priors
          float eta[7];
          c4::dirichlet_noise(game_id, n_moves, legal, p.dir_alpha, eta);
          float mine = eta[0];
          for (int c = 1; c < 7; c++) mine = (sub == (uint32_t)c) ? eta[c] : mine;
          if (is_legal) {
            const float keep = (1.0f - p.dir_eps) * prior;
            const float add = p.dir_eps * mine;
            prior = keep + add;
          }
        }
        const uint32_t nb = n_blocks;
        // (a reclaimed arena is two halves: the bump pointer of the lower one stops at the boundary; half_blocks == 0 otherwise, never a block number)
        if (nb >= p.blocks_per_slot || nb == p.half_blocks) err = err ? err : C4_ERR_ARENA_OVERFLOW;
        if (!err) {
          // Node::new (mcts.rs:345-355) for the 7 children; lane 7 writes the tail (no links yet)
          reinterpret_cast<uint4*>(blocks + nb)[sub] =
              sub < 7 ? make_uint4(0u, 0u, 0u, __float_as_uint(prior)) : make_uint4(0u, 0u, 0u, legal << 16);
          if (sub == 0) blocks[leaf_ref >> 3].t.child[leaf_ref & 7] = (uint16_t)nb;   // leaf.children = Some(..)
          if (depth == 0) root_block = nb;
          n_blocks = nb + 1;
          c_E += 1;
        }
        v_pen = cur_qp;
        v_nopen = cur_qn;
        if (CACHE && sim == 0)     // extension: remember what the evaluator said about this position
          cache_store(p.cache, p.cache_mask, leaf_mask, leaf_value, nn_logit, cur_qp, cur_qn, sub, gbase);
      }
      if (err) break;

      C4_STAMP_TRIP1(2, n_blocks);
      // ---------------- backpropagate_value: leaf -> root along the recorded path ----------
      root_n = 0;
      if (sub >= 4) {   // level d on lane 4 + (d & 3): the first four levels update in parallel
        for (uint32_t d = sub - 4; d <= depth; d += 4) {
          const uint32_t c = d >> 2;
          const uint32_t ref = c == 0 ? pv.x : (c == 1 ? pv.y : (c == 2 ? pv.z : (c == 3 ? pv.w : st->path_deep[d - kHotPath])));
          Entry* e = &blocks[ref >> 3].e[ref & 7];
          const bool odd = ((depth - d) & 1u) != 0;                             // value negated per step up
          const uint32_t n1 = e->n + 1;
          const float q1 = e->q_pen + (odd ? -v_pen : v_pen);
          const float q2 = e->q_nopen + (odd ? -v_nopen : v_nopen);
          e->n = n1;
          e->q_pen = q1;
          e->q_nopen = q2;
          if (d == 0) root_n = n1;
        }
      }
      root_n = shfl_u32(root_n, gbase + 4);
      c_sims += 1;
      c_K += depth + 1;
      // The stores above are read back below through other lanes of THIS wavefront.  A wavefront's
      // vector-memory instructions reach the cache in program order, so a later load of the same
      // address returns the stored bytes without waiting for the store's acknowledgement: only the
      // compiler must not reorder them (a wavefront-scope fence emits no instruction).
      C4_STAMP_TRIP1(3, root_n);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      C4_STAMP_TRIP1(4, root_n);

      // ---------------- gate: self_play.rs:283-308 ------------------------------------------
      bool finished = false;
      if (root_n >= p.n_iter && !(p.flags & C4_FLAG_NO_MOVES)) {
        const size_t rec0 = (size_t)ordinal * C4_MAX_SAMPLES_PER_GAME;
        uint32_t rterm = c4::terminal_state(rmask, rvalue);  // non-zero only for a terminal START position
        uint32_t retained = p.n_iter;
        if (!rterm) {
          // root_policy (mcts.rs:396-412): child visit counts / their sum
          const uint4 re = load_block_lane(blocks, root_block, sub);
          const float cnt = (sub < 7) ? (float)re.x : 0.0f;
          float w[7];
          float csum = 0.0f;
          for (int i = 0; i < 7; i++) { w[i] = shfl_f32(cnt, gbase + i); csum = csum + w[i]; }
          float pol[7];
          for (int i = 0; i < 7; i++) pol[i] = (csum == 0.0f) ? (1.0f / 7.0f) : (w[i] / csum);
          // make_random_move (mcts.rs:214-222); the 7 columns' logf/expf and the 4 ChaCha columns
          // run on the game's own lanes instead of 8 redundant copies
          const float temperature = c4::temperature_for_ply((uint32_t)__popcll(rmask));
          C4_STAMP_TRIP1(9, (uint32_t)pol[0]);
          float tp[7];
          c4::apply_temperature_group(pol, temperature, tp, sub, gbase);
          C4_STAMP_TRIP1(10, (uint32_t)tp[0]);
          const uint64_t seed = game_id * (uint64_t)(42 + n_moves);
          // the word was normally computed in an earlier, uncontended step (end of this kernel)
          const uint32_t u32 = (!fresh && rng_for == n_moves + 1) ? rng_word : c4::rng_first_u32_group(seed, sub, gbase);
          const int col = c4::weighted_index(tp, u32);
          C4_STAMP_TRIP1(11, (uint32_t)col);
          if (col < 0) {
            err = C4_ERR_DEGENERATE_POLICY;
          } else if (!((c4::legal_mask(rmask) >> col) & 1u)) {
            err = C4_ERR_ILLEGAL_MOVE;                                        // mcts.rs:196-200 expect()
          } else {
            // make_move (mcts.rs:187-206): record (root position, untempered policy), re-root
            c4_sample_rec* rec = p.samples + rec0 + n_moves;
            if (sub < 7) rec->policy[sub] = pol[sub];
            if (sub == 7) {
              rec->game_id = game_id; rec->mask = rmask; rec->value = rvalue; rec->meta = n_moves;
            }
            retained = shfl_u32(re.x, gbase + col);
            const uint32_t child_blk = child_link(re, (uint32_t)col, gbase);
            root_ref = (root_block << 3) | (uint32_t)col;
            root_block = child_blk;
            root_n = retained;
            c4::make_move(rmask, rvalue, (uint32_t)col);
            n_moves += 1;
            c_moves += 1;
            rterm = c4::terminal_after_move(rmask, rvalue);   // the position moved from was not terminal
            if (NOISE && p.dir_eps > 0.0f && !rterm && root_block != 0) {
              // extension: the new root keeps its subtree; fresh noise goes into its children's priors
              const uint32_t nlegal = c4::legal_mask(rmask);
              float eta[7];
              c4::dirichlet_noise(game_id, n_moves, nlegal, p.dir_alpha, eta);
              float mine = eta[0];
              for (int c = 1; c < 7; c++) mine = (sub == (uint32_t)c) ? eta[c] : mine;
              if (sub < 7 && ((nlegal >> sub) & 1u)) {
                Entry* ce = &blocks[root_block].e[sub];
                const float keep = (1.0f - p.dir_eps) * ce->prior;
                const float add = p.dir_eps * mine;
                ce->prior = keep + add;
              }
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
          }
        }
        if (!err && rterm) {
          // Game over (self_play.rs:302-308).  After a move INTO a terminal position the
          // reference keeps evaluating that root until it has n visits (self_play.rs:283-301);
          // those sims cannot change the samples (mcts.rs:271-313), so the game is closed now
          // and the skipped sims are counted.
          c_skipped += (p.n_iter > retained) ? (p.n_iter - retained) : 0;
          float tq_pen, tq_nopen;
          c4::terminal_value(rterm, rmask, p.c_ply_penalty, tq_pen, tq_nopen);
          // to_result (mcts.rs:271-313): sample i gets +q iff (M - i) is even
          for (uint32_t i = sub; i < n_moves; i += 8) {
            const bool neg = ((n_moves - i) & 1u) != 0;
            p.samples[rec0 + i].q_penalty = neg ? -tq_pen : tq_pen;
            p.samples[rec0 + i].q_no_penalty = neg ? -tq_nopen : tq_nopen;
          }
          c4_sample_rec* tr = p.samples + rec0 + n_moves;
          if (sub < 7) tr->policy[sub] = 1.0f / 7.0f;                         // UNIFORM_POLICY, mcts.rs:45
          if (sub == 7) {
            tr->game_id = game_id; tr->mask = rmask; tr->value = rvalue;
            tr->q_penalty = tq_pen; tr->q_no_penalty = tq_nopen; tr->meta = n_moves | (1u << 16);
            p.sample_counts[ordinal] = n_moves + 1;
          }
          c_done += 1;
          c_samples += n_moves + 1;
          finished = true;
        }
      }
      // Did any game of this wavefront move in this launch?  Asked here, where every game that entered the
      // trip is still in it (a wavefront's 8 games run in lock-step: one game's extra work is everybody's).
      const bool wave_moved = __ballot(c_moves != 0) != 0ull;
      if (err) break;

      C4_STAMP_TRIP1(12, root_n);
      if (finished) {
        // replace the finished game by the next one of the request list (keeps the batch full)
        unsigned long long next = 0;
        if (sub == 0) {
          atomicAdd(&p.glob->games_done, 1ull);
          next = atomicAdd(&p.glob->queue_head, 1ull);
        }
        next = ((unsigned long long)shfl_u32((uint32_t)(next >> 32), gbase) << 32) | shfl_u32((uint32_t)next, gbase);
        if (next < p.n_games) {
          reset_slot(p, st, blocks, sub, next);
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
          rmask = p.start_mask ? p.start_mask[next] : 0ull;
          rvalue = p.start_value ? p.start_value[next] : 0ull;
          root_ref = 0; root_block = 0; root_n = 0; n_blocks = 1; n_moves = 0;
          ordinal = (uint32_t)next;
          game_id = p.reqs[next].game_id;
          fresh = true;
        } else {
          active = false;
          if (sub == 0) { st->state = kIdle; st->ordinal = 0xFFFFFFFFu; }
          break;
        }
      }
      C4_STAMP_TRIP1(5, root_n);
      // ---------------- select_new_leaf (mcts.rs:160-183) -------------------------------
      err = select_leaf(p, blocks, st, rmask, rvalue, root_block, root_ref, root_n, p.c_exploration, sub, gbase,
                        leaf_mask, leaf_value, depth, leaf_ref, pv, c_S);
      if (err) break;
      C4_STAMP_TRIP1(6, depth);
      if (sim == 1) C4_STAMP_ANY(13);
      // terminal_state of the new leaf (kept for the simulation that consumes it): below the root it was
      // reached by a move from a non-terminal node; the root itself may be an arbitrary start position
      term = depth == 0 ? c4::terminal_state(leaf_mask, leaf_value) : c4::terminal_after_move(leaf_mask, leaf_value);
      // a terminal leaf needs no evaluator, nor does one whose evaluation is in the cache: run that
      // simulation now, while trips remain
      if (sim + 1 < max_sims) {
        // A launch lasts as long as its slowest wavefront.  A wavefront with a MOVING game has already
        // paid the longest path of the step (root policy, temperature, sampling, sample record, re-root:
        // +1.7-2.7 us), and a second, terminal-leaf simulation costs +2.0 us more: such a wavefront leaves
        // its terminal leaves to the next launch (an evaluator row is then wasted on them, as the reference
        // wastes one on every terminal leaf, mcts.rs:92-98; samples are identical either way).  Measured:
        // 10.36 -> 9.61 us per 2 048-game launch alone, 11.5 -> 10.6 beside the evaluator
        // (profiles/r03_step_second_trip_ab.txt).  The evaluation-cache build keeps all its trips.
        bool again = term != 0 && (CACHE || !wave_moved);
        if (CACHE && !again) {
          c_probes += 1;
          again = cache_lookup(p.cache, p.cache_mask, leaf_mask, leaf_value, cur_logit, cur_qp, cur_qn, sub, gbase);
          c_hits += again ? 1 : 0;
        }
        if (again) {
          if (depth >= kHotPath) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // deeper levels are re-read from the slot's second line
          continue;
        }
      }
      break;
    }

#undef C4_STAMP_TRIP1
    if (err) {
      if (sub == 0) raise_error(p, st, g, err);
    } else if (active) {
      if (sub == 0) publish_leaf_model(p, g, ordinal, leaf_mask);
      pre_need = fresh || (rng_for != n_moves + 1);   // after a move / refill the stored word is stale
      pre_n_moves = n_moves;
      pre_game_id = game_id;
      // ---------------- leaf -> evaluator input (c4r.rs:378-392) ----------------------
      encode_leaf<PlaneT>(p.planes, g, leaf_mask, leaf_value, sub);
      // ---------------- the game's state goes back as one line: lane k owns dwords 4k..4k+3 ----
      line = pv;                                                    // lanes 4..7: the recorded path
      if (sub == 0) line = make_uint4((uint32_t)rmask, (uint32_t)(rmask >> 32), (uint32_t)rvalue, (uint32_t)(rvalue >> 32));
      if (sub == 1) line = make_uint4((uint32_t)leaf_mask, (uint32_t)(leaf_mask >> 32), (uint32_t)leaf_value, (uint32_t)(leaf_value >> 32));
      if (sub == 2) line = make_uint4((uint32_t)game_id, (uint32_t)(game_id >> 32), ordinal, root_n);
      if (sub == 3) line = make_uint4(slot_state(kActive, depth, n_moves, term, fresh ? 0u : rng_for), n_blocks | (root_block << 16), root_ref, rng_word);
      store_line = true;
    }
  }

  C4_STAMP(7, 0);
  // ---------------- per-wavefront counters (one row per wavefront: no contention) ----------
  // lane `sub` of each game adds that game's counter number `sub` to the wavefront's row: one
  // no-return atomic instruction for the whole wave (nobody waits for it; rows have one writer wave)
  {
    uint32_t add = c_sims;
    add = sub == CTR_S ? c_S : add;
    add = sub == CTR_K ? c_K : add;
    add = sub == CTR_E ? c_E : add;
    add = sub == CTR_MOVES ? c_moves : add;
    add = sub == CTR_DONE ? c_done : add;
    add = sub == CTR_SKIPPED ? c_skipped : add;
    add = sub == CTR_SAMPLES ? c_samples : add;
    if (add) atomicAdd(&p.wave_ctr[(size_t)wave_index * CTR_N + sub], (unsigned long long)add);
    const uint32_t add2 = sub == 0 ? c_probes : (sub == 1 ? c_hits : 0u);
    if (CACHE && add2) atomicAdd(&p.wave_ctr[(size_t)wave_index * CTR_N + CTR_PROBES + sub], (unsigned long long)add2);
  }
  // ---------------- move RNG, off the critical path ------------------------------------------
  // The launch lasts as long as its slowest wavefront, and that is one with a MOVING game.  The
  // ChaCha12 word a game will need at its next move depends only on (game_id, moves played), so it
  // is computed here, in a step where no game of this wavefront moved, and kept in the slot.
  if (__ballot(c_moves != 0) == 0ull && pre_need) {
    const uint32_t w = c4::rng_first_u32_group(pre_game_id * (uint64_t)(42 + pre_n_moves), sub, gbase);
    if (sub == 3) { line.w = w; line.x = (line.x & 0x03FFFFFFu) | ((pre_n_moves + 1u) << 26); }
  }
  if (store_line) reinterpret_cast<uint4*>(st)[sub] = line;
  C4_STAMP(8, 0);
  if (lane == 0 && p.seq) {
    unsigned long long* my = p.stamps + ((size_t)(p.seq & 1) * p.n_waves + wave_index) * 2;
    my[0] = t_start;
    my[1] = __builtin_amdgcn_s_memrealtime();
  }
}

template <typename PlaneT, bool NOISE, bool CACHE>
__global__ __launch_bounds__(64, (NOISE || CACHE) ? 1 : C4_STEP_WAVES) void c4_step_kernel(
    // What the head of every wavefront's dependent chain needs, as leading SCALAR arguments (copies of p's fields): with
    // -mllvm -amdgpu-kernarg-preload-count (build.py) they are in SGPRs when the wavefront starts, so the state line and
    // the evaluator's outputs are requested at once instead of behind a scalar-load round trip to the kernarg segment.
    Slot* __restrict__ a_slots, const float* __restrict__ a_logprobs, const float* __restrict__ a_q, uint32_t a_n_waves, uint32_t a_n_slots,
    Params p) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t sub = lane & 7;
  const uint32_t wave_index = blockIdx.x;
  const uint32_t g = wave_index * 8 + (lane >> 3);
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();   // 100 MHz device clock
  C4_STAMP(0, 0);
#ifdef C4_PHASE_STAMPS
  if (lane == 0) for (int i = 9; i < 16; i++) p.phase[(size_t)wave_index * 16 + i] = 0;
#endif
  if (blockIdx.x >= a_n_waves) { timing_helper(p, lane); return; }

  const uint32_t gs = g < a_n_slots ? g : 0;
  Slot* st = a_slots + gs;
  // The game's state: ONE 128-byte line, 16 bytes per lane in one instruction; the evaluator's
  // outputs for this game travel in the same round trip.
  const uint4 hot = reinterpret_cast<const uint4*>(st)[sub];
  const float nn_logit = a_logprobs[(size_t)gs * 7 + (sub < 7 ? sub : 6)];
  const float nn_q = a_q[(size_t)gs * 2 + (sub & 1)];
  // ... and they must LEAVE together: without this fence hipcc sinks the two evaluator loads into the
  // `if (active)` below, i.e. behind the wait for the state line -- a second, serial memory round trip (plus the
  // scalar loads of the two pointers) at the head of every wavefront's chain (round 3, found in the ISA).
  __builtin_amdgcn_sched_barrier(0);
  step_body<PlaneT, NOISE, CACHE>(p, wave_index, lane, a_n_slots, st, hot, nn_logit, nn_q, t_start, g);
}

// c4_session_scatter_outputs + c4_session_step as ONE launch (callback mode, c4_session_step_gather): a game takes its evaluator
// outputs from row inverse[g] of the callback's answers (7 log-probabilities, q_penalty, q_no_penalty per row; pinned host memory
// or device memory) instead of from the bound tensors -- one launch and one boundary less per callback round trip.  The bound
// logprobs / q tensors are NOT updated by this form (callers that watch them use the two entry points).
template <typename PlaneT, bool NOISE, bool CACHE>
__global__ __launch_bounds__(64, (NOISE || CACHE) ? 1 : C4_STEP_WAVES) void c4_step_gather_kernel(
    Slot* __restrict__ a_slots, const float* __restrict__ a_answers, const uint32_t* __restrict__ a_inverse, uint32_t a_n_unique, uint32_t a_n_waves,
    uint32_t a_n_slots, Params p) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t sub = lane & 7;
  const uint32_t wave_index = blockIdx.x;
  const uint32_t g = wave_index * 8 + (lane >> 3);
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  C4_STAMP(0, 0);
  if (blockIdx.x >= a_n_waves) { timing_helper(p, lane); return; }
  const uint32_t gs = g < a_n_slots ? g : 0;
  Slot* st = a_slots + gs;
  const uint4 hot = reinterpret_cast<const uint4*>(st)[sub];
  const uint32_t row = a_inverse[gs];
  float nn_logit = 0.f, nn_q = 0.f;
  if (row < a_n_unique) {                                           // (idle slots have no row: their values are never used)
    nn_logit = a_answers[(size_t)row * 9 + (sub < 7 ? sub : 6)];
    nn_q = a_answers[(size_t)row * 9 + 7 + (sub & 1)];
  }
  step_body<PlaneT, NOISE, CACHE>(p, wave_index, lane, a_n_slots, st, hot, nn_logit, nn_q, t_start, g);
}

// ------------------------------------------------------------------------------------------
// The heads' output layers AND the step in one launch (c4_session_step_head_out).  A workgroup of the output kernel owns
// 16 boards -- the games of exactly two step wavefronts -- so nothing crosses a workgroup: wavefronts 0 and 1 request
// their games' state lines at the very start (the round trip is over long before the outputs exist), all six compute the
// outputs (c4_head_out.hpp: the same code as the stand-alone kernel, the same bits; the outputs still go to the bound
// logprobs / q tensors for whoever watches them), and wavefronts 0 and 1 go on as the step of their 8 games each, taking
// the logits from LDS.  One launch boundary, the step kernel's argument fetch and both of its head-of-chain round trips
// leave a session's per-round chain.  Default configuration only (no Dirichlet noise, no evaluation cache, no per-launch
// timing): everything else keeps the two launches.
// ------------------------------------------------------------------------------------------
// (151 registers: two of this kernel's wavefronts on a SIMD take 304 of its 512, and the launch shares a compute unit with a
// 4-wavefront hidden-layer GEMM workgroup of the OTHER session -- 208 registers per SIMD, 120 KB + 15 KB of LDS -- instead of
// queueing for a free one; not with the 8-wavefront form's 2 x 112.)
#ifdef C4_OUT_STEP_WAVES_PER_EU   // diagnostic (build_variant.py ... -DC4_OUT_STEP_WAVES_PER_EU=4): 128 registers, fits beside any GEMM workgroup; spills 116 bytes per lane
#define C4_OUT_STEP_ATTR __attribute__((amdgpu_waves_per_eu(C4_OUT_STEP_WAVES_PER_EU, C4_OUT_STEP_WAVES_PER_EU)))
#else
#define C4_OUT_STEP_ATTR
#endif
// kGpw: games per stepping wavefront -- 8 (wavefronts 0 and 1 step the workgroup's 16 games) or 4 (wavefronts 0-3 step four games
// each, their other four lane groups idle; c4_session_set_step_shape).  A wavefront runs the UNION of its games' control flow (the
// deepest descent, a mover's temperature + sampling, a second simulation behind a terminal leaf, one after the other), and the
// workgroup's six wavefronts are there anyway.  Measured (profiles/r05_out_step_gpw.txt): beside a second session's kernels
// (BASELINE config 2, the paired graph) 4 is +0.5 % (six of six alternating pairs), for a session alone (the reference's default job)
// -0.4 %: the paired driver asks for 4, everybody else gets 8.  Which wavefront steps a game changes nothing the game records.
// (Counter rows: two wavefronts add to one row, by atomics as before.)
template <typename PlaneT, uint32_t kGpw>
__global__ __launch_bounds__(64 * c4ho::kHeadWaves, 1) C4_OUT_STEP_ATTR void c4_out_step_kernel(
    const uint4* __restrict__ hp, const uint4* __restrict__ hv, const uint4* __restrict__ wp, const uint4* __restrict__ wv,
    const float* __restrict__ bp, const float* __restrict__ bv, Slot* __restrict__ a_slots, uint32_t a_n_slots, uint32_t f8, uint32_t sp8, uint32_t sv8,
    Params p) {
  __shared__ c4ho::Shared sh;
  C4_TL_BEGIN();
  constexpr uint32_t kStepWaves = 16 / kGpw;
  static_assert(kGpw == 8 || kGpw == 4, "16 games per workgroup on two or four stepping wavefronts");
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 7, grp = lane >> 3;
  const uint32_t b = wave * kGpw + grp;                                 // the lane group's board among the workgroup's 16
  const bool mine = wave < kStepWaves && grp < kGpw;
  const uint32_t g = mine ? blockIdx.x * 16 + b : a_n_slots;            // (>= n_slots: this lane group steps nothing)
  const uint32_t wave_index = blockIdx.x * 2 + wave / (kStepWaves / 2);  // the counters' row: as many rows as the stand-alone kernel has wavefronts
  const uint32_t gs = g < a_n_slots ? g : 0;
  Slot* st = a_slots + gs;
  uint4 hot = make_uint4(0, 0, 0, 0);
  if (wave < kStepWaves) hot = reinterpret_cast<const uint4*>(st)[sub]; // in flight under the output layers
  c4ho::head_out_block<16>(sh, hp, hv, wp, wv, bp, bv, a_n_slots, f8, sp8, sv8, const_cast<float*>(p.logprobs), const_cast<float*>(p.q), blockIdx.x);
  __syncthreads();
  if (wave >= kStepWaves || wave_index >= p.n_waves) return;
  const float nn_logit = sh.res[b & 15][sub < 7 ? sub : 6];
  const float nn_q = sh.res[b & 15][7 + (sub & 1)];
  step_body<PlaneT, false, false>(p, wave_index, lane, a_n_slots, st, hot, nn_logit, nn_q, 0ull, g);
  C4_TL_END(3, a_slots);
}

// ------------------------------------------------------------------------------------------
// Element-wise kernels (SURVEY 8a K1 and the arithmetic pieces) for the parity tests
// ------------------------------------------------------------------------------------------
__global__ void k_pos_ops(const uint64_t* mask, const uint64_t* value, const int32_t* col, uint64_t n, float c_ply,
                          uint64_t* om, uint64_t* ov, uint32_t* olegal, uint32_t* oterm, float* oq) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t m = mask[i], v = value[i];
  const uint32_t legal = c4::legal_mask(m);
  const uint32_t t = c4::terminal_state(m, v);
  float a = 0.0f, b = 0.0f;
  if (t) c4::terminal_value(t, m, c_ply, a, b);
  olegal[i] = legal;
  oterm[i] = t;
  oq[2 * i] = a;
  oq[2 * i + 1] = b;
  const int32_t c = col[i];
  if (c >= 0 && c < 7 && ((legal >> c) & 1u)) {
    c4::make_move(m, v, (uint32_t)c);
    om[i] = m; ov[i] = v;
  } else {
    om[i] = 0; ov[i] = 0;  // make_move returns None (c4r.rs:71)
  }
}

template <typename PlaneT>
__global__ void k_encode(const uint64_t* mask, const uint64_t* value, uint64_t n, void* planes) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * C4_PLANES_LEN) return;
  const uint64_t g = i / C4_PLANES_LEN;
  const uint32_t e = (uint32_t)(i % C4_PLANES_LEN);
  store_plane<PlaneT>(planes, i, c4::plane_bit(mask[g], value[g], e));
}

// ln_tab[k] = c4_logf((float)k): the same port the kernel would otherwise run per tree level
__global__ void k_ln_table(float* tab, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tab[i] = c4::c4_logf((float)i);
}

__global__ void k_expf_logf(const float* x, uint64_t n, int which, float* y) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  y[i] = which ? c4::c4_logf(x[i]) : c4::c4_expf(x[i]);
}

__global__ void k_softmax7(const float* logits, const uint32_t* legal, uint64_t n, float* out, uint32_t* err) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float l[7], o[7];
  for (int c = 0; c < 7; c++) {
    l[c] = logits[7 * i + c];
    if (legal && !((legal[i] >> c) & 1u)) l[c] = __uint_as_float(0xff800000u);
  }
  const bool ok = c4::softmax7(l, o);
  err[i] = ok ? 0u : (uint32_t)C4_ERR_DEGENERATE_POLICY;
  for (int c = 0; c < 7; c++) out[7 * i + c] = ok ? o[c] : 0.0f;
}

__global__ void k_temperature(const float* policy, const float* t, uint64_t n, float* out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float p[7], o[7];
  for (int c = 0; c < 7; c++) p[c] = policy[7 * i + c];
  c4::apply_temperature(p, t[i], o);
  for (int c = 0; c < 7; c++) out[7 * i + c] = o[c];
}

__global__ void k_sample_move(const uint64_t* game_id, const uint32_t* n_moves, const float* policy, const float* t,
                              uint64_t n, int32_t* out_col, uint32_t* out_u32) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float p[7], o[7];
  for (int c = 0; c < 7; c++) p[c] = policy[7 * i + c];
  c4::apply_temperature(p, t[i], o);
  const uint32_t u = c4::rng_first_u32(game_id[i] * (uint64_t)(42 + n_moves[i]));
  out_col[i] = c4::weighted_index(o, u);
  if (out_u32) out_u32[i] = u;
}

__global__ void k_dirichlet(const uint64_t* game_id, const uint32_t* n_moves, const uint32_t* legal, float alpha, uint64_t n, float* eta) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float e[7];
  c4::dirichlet_noise(game_id[i], n_moves[i], legal[i], alpha, e);
  for (int c = 0; c < 7; c++) eta[7 * i + c] = e[c];
}

// ------------------------------------------------------------------------------------------
// Tail compaction: once the request queue is empty, finished slots stay empty and the evaluator
// would keep computing rows for them.  k_compact_plan pairs every active slot beyond the first A
// slots (A = number of active games) with an idle slot below A; k_compact_move copies the game
// (slot state, the used part of its arena, its evaluator input row) across.  All references inside
// a game's state are arena-relative, so nothing needs patching.
// ------------------------------------------------------------------------------------------
struct CompactPlan {
  uint32_t n_active;
  uint32_t n_pairs;
};
__global__ __launch_bounds__(1024) void k_compact_plan(const Slot* slots, uint32_t n_slots, CompactPlan* plan, uint2* pairs) {
  __shared__ uint32_t s_active, s_holes, s_movers;
  if (threadIdx.x == 0) { s_active = 0; s_holes = 0; s_movers = 0; }
  __syncthreads();
  uint32_t mine = 0;
  for (uint32_t g = threadIdx.x; g < n_slots; g += blockDim.x) mine += slot_status(slots[g].state) == kActive ? 1u : 0u;
  atomicAdd(&s_active, mine);
  __syncthreads();
  const uint32_t A = s_active;
  // any bijection between holes (< A, idle) and movers (>= A, active) will do: which slot plays a
  // game changes nothing a game records
  for (uint32_t g = threadIdx.x; g < n_slots; g += blockDim.x) {
    const bool act = slot_status(slots[g].state) == kActive;
    if (g < A && !act) pairs[atomicAdd(&s_holes, 1u)].y = g;
    if (g >= A && act) pairs[atomicAdd(&s_movers, 1u)].x = g;
  }
  __syncthreads();
  if (threadIdx.x == 0) { plan->n_active = A; plan->n_pairs = s_movers; }
}

template <typename PlaneT>
__global__ __launch_bounds__(256) void k_compact_move(Params p, const CompactPlan* plan, const uint2* pairs) {
  const uint32_t k = blockIdx.x;
  if (k >= plan->n_pairs) return;
  const uint32_t src = pairs[k].x, dst = pairs[k].y;
  const Slot* ss = p.slots + src;
  const uint32_t n_blocks = ss->arena & 0xFFFFu;
  const uint4* sb = reinterpret_cast<const uint4*>(p.blocks + (size_t)src * p.blocks_per_slot);
  uint4* db = reinterpret_cast<uint4*>(p.blocks + (size_t)dst * p.blocks_per_slot);
  for (uint32_t i = threadIdx.x; i < n_blocks * 8u; i += blockDim.x) db[i] = sb[i];
  PlaneT* pl = reinterpret_cast<PlaneT*>(p.planes);
  for (uint32_t e = threadIdx.x; e < C4_PLANES_LEN; e += blockDim.x) pl[(size_t)dst * C4_PLANES_LEN + e] = pl[(size_t)src * C4_PLANES_LEN + e];
  if (threadIdx.x < 16) reinterpret_cast<uint4*>(p.slots + dst)[threadIdx.x] = reinterpret_cast<const uint4*>(ss)[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) { p.slots[src].state = kIdle; p.slots[src].ordinal = 0xFFFFFFFFu; }
}

// ------------------------------------------------------------------------------------------
// Reclaimed arenas (C4_FLAG_RECLAIM).  The reference drops the siblings' subtrees at every move (mcts.rs:187-206: the new
// root's Rc is the only one left); the never-reclaimed arena keeps them, which is what bounds n_mcts_iterations by the 16-bit
// child links (43 n + 8 <= 65 535 blocks).  Here a slot's arena is two halves of half_blocks blocks.  A game allocates in one of
// them; when that half runs short, this kernel -- launched behind every reclaim_period-th step launch, between two steps, when
// every game waits for the evaluator with a recorded path -- copies what is still reachable into the other half, compactly:
//   new[0] = the block that holds the root's OWN entry (level 0 of every backup),  new[1] = the root's children block,
//   then breadth first: the workgroup walks the copied blocks in order; a block's children are copied to the next free
//   places and its links patched.  An old block's forwarding address goes into its tail's `legal` half-word (written by
//   expansion, read by nobody), from which the recorded path (entry refs = block << 3 | column) is translated at the end.
// Reachable blocks number at most (visits of the root + simulations of one launch + 2) <= half_blocks, so the copy always fits.
// One workgroup of 1 024 threads per slot (128 lane groups of 8: a block is one 16-byte load per lane, as everywhere);
// slots with room left return after one load.  Which block a node sits in changes nothing a game records.
// ------------------------------------------------------------------------------------------
constexpr int kReclaimThreads = 1024;
constexpr uint32_t kReclaimSlotsPerGroup = 16;            // slots one workgroup looks at (a look is one 8-byte load; few slots ever need more)
C4_DEV void reclaim_slot(const Params& p, uint32_t g, uint32_t min_free, uint32_t& s_tail, uint32_t& s_overflow);
__global__ __launch_bounds__(kReclaimThreads) void k_arena_reclaim(Params p, uint32_t min_free) {
  __shared__ uint32_t s_tail, s_overflow;
  // one workgroup per 16 slots, one after the other (the test is workgroup-uniform): a launch that finds nothing to do is ~100
  // workgroups that return after 16 loads, not one 1 024-thread workgroup per slot
  for (uint32_t k = 0; k < kReclaimSlotsPerGroup; k++) {
    const uint32_t g = blockIdx.x * kReclaimSlotsPerGroup + k;
    if (g >= p.n_slots) return;
    reclaim_slot(p, g, min_free, s_tail, s_overflow);
    __syncthreads();
  }
}
C4_DEV void reclaim_slot(const Params& p, const uint32_t g, const uint32_t min_free, uint32_t& s_tail, uint32_t& s_overflow) {
  Slot* st = p.slots + g;
  const uint32_t state = st->state, arena = st->arena;
  if (slot_status(state) != kActive) return;
  const uint32_t H = p.half_blocks;
  const uint32_t n_blocks = arena & 0xFFFFu, root_block = arena >> 16;
  const uint32_t cur = n_blocks > H ? H : 0u;             // the half in use (the upper one never holds fewer than H + 1)
  if (H - (n_blocks - cur) >= min_free) return;           // room until the next look
  Block* blocks = p.blocks + (size_t)g * p.blocks_per_slot;
  const uint32_t dst = cur ? 0u : H;
  const uint32_t tid = threadIdx.x, sub = tid & 7u, grp = tid >> 3;
  const int gbase = (int)(tid & 63u & ~7u);
  const uint32_t root_ref = st->root_ref, depth = (state >> 8) & 0xFFu;
  if (grp == 0) reinterpret_cast<uint4*>(blocks + dst)[sub] = load_block_lane(blocks, root_ref >> 3, sub);
  if (grp == 1 && root_block) {
    reinterpret_cast<uint4*>(blocks + dst + 1)[sub] = load_block_lane(blocks, root_block, sub);
    if (sub == 7) blocks[root_block].t.legal = (uint16_t)(dst + 1);
  }
  if (tid == 0) { s_tail = dst + (root_block ? 2u : 1u); s_overflow = 0u; }
  __syncthreads();
  uint32_t lo = dst + 1, hi = s_tail;
  while (lo < hi) {                                       // one level of the tree per trip
    for (uint32_t i = lo + grp; i < hi; i += kReclaimThreads / 8) {
      const uint4 tl = sub == 7 ? load_block_lane(blocks, i, 7) : make_uint4(0, 0, 0, 0);   // the copied block's links are still OLD block numbers
      uint32_t link[7], cnt = 0;
#pragma unroll
      for (int k = 0; k < 7; k++) {
        const uint32_t w = k >> 1;
        const uint32_t word = w == 0 ? tl.x : (w == 1 ? tl.y : (w == 2 ? tl.z : tl.w));
        link[k] = (shfl_u32(word, gbase + 7) >> (16u * (k & 1u))) & 0xFFFFu;
        cnt += link[k] ? 1u : 0u;
      }
      if (cnt == 0) continue;                             // (group-uniform)
      uint32_t base = sub == 0 ? atomicAdd(&s_tail, cnt) : 0u;
      base = shfl_u32(base, gbase);
      // The host sizes the halves so that the live subtree always fits (reclaim_half_min); should that ever not hold (a damaged
      // tree, a configuration changed under a running game), nothing is written past the destination half -- an upward copy
      // would run into the NEXT slot's arena, a downward one into the half still being read -- and the slot raises the error the
      // step kernel raises for a full arena, with its root, links and path left as they were.
      if (base + cnt > dst + H) {                         // (group-uniform)
        if (sub == 0) s_overflow = 1u;
        continue;
      }
      uint4 v[7];
#pragma unroll
      for (int k = 0; k < 7; k++) if (link[k]) v[k] = load_block_lane(blocks, link[k], sub);
      uint32_t r = 0, nl[7];
#pragma unroll
      for (int k = 0; k < 7; k++) {
        nl[k] = 0;
        if (link[k]) {
          nl[k] = base + r++;
          reinterpret_cast<uint4*>(blocks + nl[k])[sub] = v[k];
          if (sub == 7) blocks[link[k]].t.legal = (uint16_t)nl[k];        // forwarding address (after the load above: one wavefront, program order)
        }
      }
      if (sub == 7) {
        uint4 t2 = tl;
        t2.x = nl[0] | (nl[1] << 16); t2.y = nl[2] | (nl[3] << 16); t2.z = nl[4] | (nl[5] << 16); t2.w = nl[6] | (tl.w & 0xFFFF0000u);
        reinterpret_cast<uint4*>(blocks + i)[7] = t2;
      }
    }
    __syncthreads();                                      // this level's copies (and forwarding addresses) are visible to the workgroup
    lo = hi; hi = s_tail;
    if (s_overflow) hi = lo;                              // (workgroup-uniform: read between two barriers) stop walking
    __syncthreads();
  }
  if (s_overflow) {
    if (tid == 0) raise_error(p, st, g, C4_ERR_ARENA_OVERFLOW);
    return;
  }
  // the recorded path: level L sits at path[4 (L & 3) + (L >> 2)] (L < 16) or path_deep[L - 16]; level 0 is the root's own entry
  if (tid <= depth && tid < kMaxPath) {
    uint32_t* slot_word = tid < kHotPath ? &st->path[4 * (tid & 3u) + (tid >> 2)] : &st->path_deep[tid - kHotPath];
    const uint32_t ref = *slot_word;
    *slot_word = tid == 0 ? ((dst << 3) | (root_ref & 7u)) : (((uint32_t)blocks[ref >> 3].t.legal << 3) | (ref & 7u));
  }
  if (tid == 0) {
    st->root_ref = (dst << 3) | (root_ref & 7u);
    st->arena = hi | ((root_block ? dst + 1u : 0u) << 16);
    atomicAdd(&p.reclaim_ctr[0], 1ull);
    atomicAdd(&p.reclaim_ctr[1], (unsigned long long)(hi - dst));
  }
}

// Exclusive prefix sum of the per-game sample counts = where each game's records start in the packed
// array.  One 1024-thread workgroup walks the list with a running carry (n_games is a few 10^4..10^6).
__global__ __launch_bounds__(1024) void k_sample_offsets(const uint32_t* counts, unsigned long long n_games,
                                                         unsigned long long* offsets, unsigned long long* total) {
  __shared__ unsigned long long wave_sum[16];
  __shared__ unsigned long long carry;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (unsigned long long base = 0; base < n_games; base += 1024) {
    const unsigned long long i = base + tid;
    const unsigned long long v = i < n_games ? counts[i] : 0ull;
    unsigned long long x = v;                                   // inclusive scan inside the wavefront
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long y = ((unsigned long long)__shfl_up((int)(x >> 32), off, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)x, off, 64);
      if ((int)lane >= off) x += y;
    }
    if (lane == 63) wave_sum[wave] = x;
    __syncthreads();
    unsigned long long before = carry;                           // sums of the wavefronts before this one
    for (uint32_t w = 0; w < wave; w++) before += wave_sum[w];
    if (i < n_games) offsets[i] = before + x - v;
    __syncthreads();
    if (tid == 1023) carry = before + x;
    __syncthreads();
  }
  if (tid == 0) *total = carry;
}

// Leaf keys for the callback evaluator's batching (NNThread::loop_once, self_play.rs:203-208: unique
// (model, leaf position) pairs).  A position is its `value` bits (42) plus the 7 column heights
// (3 bits each: the stones of a column stack from the bottom, so the heights determine `mask`):
// 63 bits, one non-negative int64 per resident game; idle slots get -1.
C4_DEV long long leaf_key_of(const Slot* st) {
  if (slot_status(st->state) != kActive) return -1;
  const uint64_t m = st->leaf_mask, v = st->leaf_value;
  uint64_t heights = 0;
  for (uint32_t c = 0; c < 7; c++) heights |= (uint64_t)__popcll(m & (c4::kCol0 << c)) << (3 * c);
  return (long long)(v | (heights << 42));
}

__global__ void k_leaf_keys(const Slot* slots, uint32_t n_slots, long long* keys) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n_slots) keys[g] = leaf_key_of(slots + g);
}

// ---- the callback evaluator's batch on the device (NNThread::loop_once, self_play.rs:203-208) ----
// The reference collects the waiting leaves in a HashSet<(model, Pos)> and evaluates each pair once.
// Three small launches do the same for all resident games: (1) every slot enters an open-addressed
// table of SLOT INDICES (a cell's pair is its slot's pair; slots with one pair meet in one cell and
// keep the lowest index), (2) one workgroup ranks the representatives in slot order -- the batch's
// row order depends on nothing but the games -- and maps every slot to its row, (3) the
// representatives write their rows of the evaluator input ([2, 6, 7] float32, c4r.rs:378-392)
// wherever the caller asked: normally pinned host memory, so the batch crosses PCIe once, written
// by the kernel, and the host learns its size from one pinned word.
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr uint32_t kRepFlag = 0x80000000u;

__global__ void k_unique_insert(const Slot* slots, const uint64_t* leaf_models, uint32_t n_slots, uint32_t* tab,
                                uint32_t tab_mask, uint32_t* cell) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_slots) return;
  const long long key = leaf_key_of(slots + g);
  if (key < 0) { cell[g] = kNoSlot; return; }
  const uint64_t model = leaf_models ? leaf_models[g] : 0ull;
  uint64_t x = ((uint64_t)key ^ (model * 0x9E3779B97F4A7C15ull)) * 0xD6E8FEB86659FD93ull;   // any mix does: only the
  uint32_t h = (uint32_t)(x >> 32) & tab_mask;                                              // probe order depends on it
  for (;;) {
    const uint32_t cur = atomicCAS(&tab[h], kNoSlot, g);
    if (cur == kNoSlot) break;                                     // first of its pair: this cell is the pair's
    if (leaf_key_of(slots + cur) == key && (!leaf_models || leaf_models[cur] == model)) {
      atomicMin(&tab[h], g);                                       // same pair: the lowest slot represents it
      break;
    }
    h = (h + 1) & tab_mask;                                        // another pair's cell (cells never change pair)
  }
  cell[g] = h;
}

__global__ __launch_bounds__(1024) void k_unique_rank(const uint32_t* tab, const uint32_t* cell, uint32_t n_slots,
                                                      uint32_t* row_of, uint32_t* inverse, uint32_t* n_unique_dev,
                                                      uint32_t* n_unique_out) {
  __shared__ uint32_t wave_sum[16];
  __shared__ uint32_t carry;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_slots; base += 1024) {
    const uint32_t g = base + tid;
    const bool rep = g < n_slots && cell[g] != kNoSlot && tab[cell[g]] == g;
    const unsigned long long b = __ballot(rep);
    if (lane == 0) wave_sum[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t before = carry;
    for (uint32_t w = 0; w < wave; w++) before += wave_sum[w];
    if (g < n_slots) row_of[g] = rep ? ((before + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))) | kRepFlag) : 0u;
    __syncthreads();
    if (tid == 1023) carry = before + (uint32_t)__popcll(b);
    __syncthreads();
  }
  __threadfence_block();
  for (uint32_t g = tid; g < n_slots; g += 1024)
    inverse[g] = cell[g] == kNoSlot ? kNoSlot : (row_of[tab[cell[g]]] & ~kRepFlag);
  if (tid == 0) { *n_unique_dev = carry; *n_unique_out = carry; }
}

// one wavefront per slot; also hands the table back empty (nothing reads it here)
__global__ __launch_bounds__(256) void k_unique_emit(const Slot* slots, const uint64_t* leaf_models, uint32_t n_slots,
                                                     uint32_t* tab, const uint32_t* cell, const uint32_t* row_of,
                                                     float* rows_out, uint64_t* models_out) {
  const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (g >= n_slots) return;
  const uint32_t c = cell[g];
  if (c == kNoSlot) return;
  if (lane == 0) tab[c] = kNoSlot;
  const uint32_t r = row_of[g];
  if (!(r & kRepFlag)) return;
  const uint32_t row = r & ~kRepFlag;
  const uint64_t m = slots[g].leaf_mask, v = slots[g].leaf_value;
  for (uint32_t e = lane; e < C4_PLANES_LEN; e += 64) rows_out[(size_t)row * C4_PLANES_LEN + e] = c4::plane_bit(m, v, e) ? 1.0f : 0.0f;
  if (models_out && lane == 0) models_out[row] = leaf_models ? leaf_models[g] : 0ull;
}

// the evaluator's answers back to every slot that asked: answers[row] = 7 log-probabilities, q_penalty, q_no_penalty
__global__ void k_unique_scatter(const uint32_t* inverse, const float* answers, uint32_t n_slots, uint32_t n_unique,
                                 float* logprobs, float* q) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t g = i / 9, e = i % 9;
  if (g >= n_slots) return;
  const uint32_t row = inverse[g];
  if (row >= n_unique) return;                                     // idle slot
  const float a = answers[(size_t)row * 9 + e];
  if (e < 7) logprobs[(size_t)g * 7 + e] = a; else q[(size_t)g * 2 + (e - 7)] = a;
}

// K6: pack finished games' records contiguously (one wavefront per game, 4 records per pass)
__global__ __launch_bounds__(64) void k_pack_samples(const c4_sample_rec* src, const uint32_t* counts,
                                                     const unsigned long long* offsets, uint64_t n_games, c4_sample_rec* dst) {
  const uint64_t game = blockIdx.x;
  if (game >= n_games) return;
  const uint32_t n = counts[game];
  const uint4* s4 = (const uint4*)(src + game * C4_MAX_SAMPLES_PER_GAME);
  uint4* d4 = (uint4*)(dst + offsets[game]);
  for (uint32_t i = threadIdx.x; i < n * 4u; i += 64) d4[i] = s4[i];  // 64-byte record = 4 x 16 bytes
}

// ------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------
thread_local std::string g_last_error;
using c4host::fail;

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return fail(C4_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                  \
  } while (0)

}  // namespace

int c4host::fail(int code, const std::string& msg) {
  g_last_error = msg;
  // HIP keeps a failed call's code as the thread's "last error" until somebody reads it; the launch checks of this library read it
  // (hipGetLastError after every launch), so a failed hipMalloc of ONE call -- device memory full -- used to surface as "out of
  // memory" in the next call's first launch, long after the memory was there again.  The failure is reported here: take it off.
  if (code == C4_ERR_HIP) (void)hipGetLastError();
  return code;
}

// run the rest of the entry point on the session's device; the caller's current device is restored on return
#define C4_ON_DEVICE(dev)                   \
  c4host::DeviceGuard _device_guard(dev);   \
  HIP_TRY(_device_guard.error())

struct c4_session {
  c4_config cfg{};
  Params p{};
  hipStream_t stream = nullptr;
  uint32_t lanes_per_game = 8;   // c4_step_kernel's mapping (a 4-lane variant was built and measured: tools/experiments/)
  uint32_t n_waves = 0;       // wavefronts a step launches now (shrinks with c4_session_compact)
  uint32_t out_step_gpw = 8;  // games per stepping wavefront of the fused output + step launch (c4_session_set_step_shape)
  uint32_t n_waves_cap = 0;   // as created: size of the per-wavefront arrays
  uint32_t seq = 0;
  bool timing = true;
  bool bound = false, have_games = false;
  uint64_t n_games = 0;
  c4_game_metadata* reqs_dev = nullptr;
  uint64_t* start_mask_dev = nullptr;
  uint64_t* start_value_dev = nullptr;
  // pinned probe buffer for c4_session_poll
  Globals* probe_host = nullptr;
  hipEvent_t probe_event = nullptr;
  bool probe_pending = false;
  uint64_t probe_done = 0;
  uint64_t probe_started = 0;
  uint32_t probe_error = 0;
  CompactPlan* plan_dev = nullptr;   // tail compaction scratch
  uint2* pairs_dev = nullptr;
  float* ln_tab_dev = nullptr;                 // ln(visit count) table of select (Params::ln_tab)
  unsigned long long* offsets_dev = nullptr;   // pack_samples: [n_games] record offsets + [1] total, sized by set_games
  unsigned long long* total_host = nullptr;    // pinned
  // c4_session_unique_leaves scratch (first use): table of slot indices, each slot's cell, its row, the count
  size_t arena_bytes = 0;                      // of p.blocks (kept for the next session when this one is destroyed)
  uint32_t* uniq_tab = nullptr;
  uint32_t uniq_tab_mask = 0;
  uint32_t* uniq_cell = nullptr;
  uint32_t* uniq_row = nullptr;
  uint32_t* uniq_count = nullptr;
  // reclaimed arenas (C4_FLAG_RECLAIM): step launches since the last look at the arenas, and the capture they were counted in
  uint32_t reclaim_period = 0;
  uint32_t reclaim_count = 0;
  unsigned long long reclaim_capture_id = 0;
};

C4_TL_SETTER(c4_debug_timeline_session)

extern "C" {

const char* c4_last_error_string(void) { return g_last_error.c_str(); }
int c4_abi_version(void) { return C4_ABI_VERSION; }

#ifndef C4_SOURCE_HASH
#define C4_SOURCE_HASH "unknown"
#endif
// content hash of the sources this library was compiled from (c4a0_amd/csrc/build.py); the marker
// prefix lets the build script read it from the file's bytes without loading the library
const char* c4_source_hash(void) {
  static const char marked[] = "c4a0-src-hash:" C4_SOURCE_HASH;
  return marked + 14;
}

int c4_device_count(int* out) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) n = 0;
  if (out) *out = n;
  return C4_OK;
}

// The tree arena is by far a session's largest allocation (13 GB for the reference's default job: 1 700 slots x
// (43 x 1 400 + 8) blocks) and the driver scrubs freed device memory before it hands it out again: a session created
// right after one of that size was destroyed waited 0.6 s in hipMalloc (tools/whole_job_phases.py).  One freed arena
// per process is therefore kept for the next session on the same device that fits it (no more than twice as big as
// needed); nothing in it is ever read before it is written (blocks are bump-allocated per slot, c4_start_kernel
// writes each slot's root).  c4_trim_cached_memory() gives it back; C4_ARENA_CACHE=0 switches the cache off.
namespace {
struct ArenaCache { void* ptr = nullptr; size_t bytes = 0; int device = -1; };
ArenaCache g_arena_cache;
std::mutex g_arena_mutex;
bool arena_cache_enabled() {
  static const bool on = [] { const char* e = getenv("C4_ARENA_CACHE"); return !(e && e[0] == '0'); }();
  return on;
}
// Gives the kept arena back to its device (the caller holds no lock).  Returns true if there was one.
bool arena_drop_cached() {
  ArenaCache old;
  {
    std::lock_guard<std::mutex> lock(g_arena_mutex);
    old = g_arena_cache;
    g_arena_cache = ArenaCache{};
  }
  if (!old.ptr) return false;
  c4host::DeviceGuard guard(old.device);
  (void)hipFree(old.ptr);
  return true;
}
hipError_t arena_acquire(int device, size_t bytes, void** out) {
  {
    std::lock_guard<std::mutex> lock(g_arena_mutex);
    ArenaCache& c = g_arena_cache;
    if (c.ptr && c.device == device && c.bytes >= bytes && c.bytes / 2 <= bytes) {
      *out = c.ptr;
      c = ArenaCache{};
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e == hipErrorOutOfMemory && arena_drop_cached()) {   // a kept arena that does not fit this request must not cause its failure
    (void)hipGetLastError();
    e = hipMalloc(out, bytes);
  }
  return e;
}
void arena_release(int device, void* ptr, size_t bytes) {
  if (!ptr) return;
  void* drop = ptr;
  if (arena_cache_enabled()) {
    std::lock_guard<std::mutex> lock(g_arena_mutex);
    ArenaCache& c = g_arena_cache;
    if (!c.ptr || c.bytes < bytes) {       // keep the bigger one
      drop = c.ptr;
      if (drop && c.device != device) { c4host::DeviceGuard guard(c.device); (void)hipFree(drop); drop = nullptr; }
      c = ArenaCache{ptr, bytes, device};
    }
  }
  if (drop) (void)hipFree(drop);
}
}  // namespace

// ---- reclaimed arenas (C4_FLAG_RECLAIM, k_arena_reclaim) ----
constexpr uint32_t kReclaimPeriod = 64;          // step launches between two looks at the arenas
constexpr uint32_t kReclaimAuto = 1000;          // blocks_per_slot == 0: reclaim above this many iterations per move
constexpr uint32_t kReclaimMaxSims = 8;          // simulations one game may run per launch in a reclaimed arena (evaluation cache)
// A half is compacted when fewer than this many blocks are free in it.  Between two looks a game takes at most
// max_sims blocks per step launch, and two looks are at most 2 x period launches apart (an eager step sequence that runs
// into a graph replay, or the other way round: each form alone keeps the period, see maybe_reclaim).
static uint32_t reclaim_min_free(uint32_t period, uint32_t max_sims) { return 2u * period * max_sims + 16u; }
// What a half must hold at the very least: the live subtree right after a compaction (<= n + max_sims + 2 blocks, + slack) and
// twice the trigger above, so that a freshly compacted half is not at its next trigger already.
static uint64_t reclaim_half_min(uint32_t n_iter, uint32_t period, uint32_t max_sims) {
  return (uint64_t)n_iter + max_sims + 8u + 2ull * reclaim_min_free(period, max_sims);
}
static bool reclaim_mode(const c4_config* cfg) {
  return (cfg->flags & C4_FLAG_RECLAIM) != 0 ||
         (cfg->blocks_per_slot == 0 && cfg->n_mcts_iterations > kReclaimAuto && !(cfg->flags & (C4_FLAG_NO_MOVES | C4_FLAG_NO_RECLAIM)));
}

// Called behind every step launch of a reclaimed session: every `period`-th launch is followed by k_arena_reclaim on the same
// stream.  Launches are counted per capture while the stream is being captured into a HIP graph (the count restarts with the
// capture, so EVERY graph carries a look behind its first step and every `period` steps after it: replays never run longer
// than min(period, graph length) steps without one), and continuously otherwise.
static int maybe_reclaim(c4_session* s) {
  if (!s->p.half_blocks) return C4_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo(s->stream, &cs, &id) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; id = 0; }
  if (cs != hipStreamCaptureStatusActive) id = 0;
  if (id != s->reclaim_capture_id) { s->reclaim_capture_id = id; s->reclaim_count = 0; }
  if (s->reclaim_count++ % s->reclaim_period != 0) return C4_OK;
  hipLaunchKernelGGL(k_arena_reclaim, dim3((s->p.n_slots + kReclaimSlotsPerGroup - 1) / kReclaimSlotsPerGroup), dim3(kReclaimThreads), 0, s->stream, s->p,
                     reclaim_min_free(s->reclaim_period, s->p.max_sims));
  HIP_TRY(hipGetLastError());
  return C4_OK;
}

static hipError_t reset_clock_acc(unsigned long long* acc_dev) {
  const unsigned long long init[5] = {0ull, 0ull, ~0ull, 0ull, 0ull};   // totals; running min start, max end, helpers done
  return hipMemcpy(acc_dev, init, sizeof init, hipMemcpyHostToDevice);
}

static int session_create_on_device(const c4_config* cfg, c4_session* s) {
  uint64_t bps = cfg->blocks_per_slot;
  const bool reclaim = reclaim_mode(cfg);
  if (reclaim) {
    s->reclaim_period = cfg->reclaim_period ? cfg->reclaim_period : kReclaimPeriod;
    if (bps == 0) {
      // automatic: 1.5 n blocks beyond the minimum, i.e. a compaction every second or third move of a game
      // (1 700 slots at n = 1 400: 1.8 GB where the never-reclaimed arena took 13 GB)
      uint64_t half = reclaim_half_min(cfg->n_mcts_iterations, s->reclaim_period, 2) + 3ull * cfg->n_mcts_iterations / 2;
      bps = 2 * (half > kMaxBlocksPerSlot / 2 ? kMaxBlocksPerSlot / 2 : half);
    }
    s->p.half_blocks = (uint32_t)(bps / 2);
    bps = 2ull * s->p.half_blocks;
  } else {
    if (bps == 0) bps = 43ull * (cfg->n_mcts_iterations ? cfg->n_mcts_iterations : 1) + 8;
    if (bps < 2) bps = 2;
  }
  s->cfg.blocks_per_slot = (uint32_t)bps;
  const size_t n = cfg->n_slots;
  const uint32_t games_per_wave = 64u / s->lanes_per_game;
  s->n_waves = (uint32_t)((n + games_per_wave - 1) / games_per_wave);
  s->n_waves_cap = (uint32_t)((n + 7) / 8);   // per-wavefront arrays are sized for the finer kernel
  Params& p = s->p;
  p.n_slots = cfg->n_slots;
  p.blocks_per_slot = (uint32_t)bps;
  p.n_iter = cfg->n_mcts_iterations;
  p.c_exploration = cfg->c_exploration;
  p.c_ply_penalty = cfg->c_ply_penalty;
  p.flags = cfg->flags;
  p.max_sims = (cfg->flags & (C4_FLAG_NO_MOVES | C4_FLAG_ONE_SIM_PER_STEP)) ? 1u : 2u;
  // visit counts stay below n_mcts_iterations + 1 in self-play (the gate); the table covers them with
  // room to spare, larger counts (C4_FLAG_NO_MOVES runs) take the computed path
  uint32_t n_ln = cfg->n_mcts_iterations + 64u;
  n_ln = n_ln < 1024u ? 1024u : (n_ln > 65536u ? 65536u : n_ln);
  hipError_t e;
  // one row of phase stamps per wavefront plus one per timing-helper workgroup (one helper per kWavesPerTimingHelper wavefronts)
  const size_t phase_rows = (size_t)s->n_waves_cap + (s->n_waves_cap + kWavesPerTimingHelper - 1) / kWavesPerTimingHelper + 1;
  if ((e = hipMalloc(&p.slots, n * sizeof(Slot))) != hipSuccess ||
      (e = arena_acquire(cfg->device, s->arena_bytes = n * bps * sizeof(Block), (void**)&p.blocks)) != hipSuccess ||
      (e = hipMalloc(&p.wave_ctr, (size_t)s->n_waves_cap * CTR_N * sizeof(unsigned long long))) != hipSuccess ||
      (e = hipMalloc(&p.glob, sizeof(Globals))) != hipSuccess ||
      (e = hipMalloc(&p.stamps, (size_t)s->n_waves_cap * 4 * sizeof(unsigned long long))) != hipSuccess ||
      (e = hipMalloc(&p.clock_acc, 5 * sizeof(unsigned long long))) != hipSuccess ||
      (e = hipMalloc(&p.phase, phase_rows * 16 * sizeof(unsigned long long))) != hipSuccess ||   // + the timing helper workgroups (diagnostic builds stamp them too)
      (e = hipMalloc(&s->ln_tab_dev, (size_t)n_ln * sizeof(float))) != hipSuccess ||
      (e = hipMalloc(&p.reclaim_ctr, 2 * sizeof(unsigned long long))) != hipSuccess ||
      (e = hipHostMalloc(&s->probe_host, sizeof(Globals))) != hipSuccess ||
      (e = hipEventCreateWithFlags(&s->probe_event, hipEventDisableTiming)) != hipSuccess)
    return fail(C4_ERR_HIP, std::string("allocating session (") + std::to_string((n * bps * sizeof(Block)) >> 20) +
                                " MiB of tree arena): " + hipGetErrorString(e));
  p.n_waves = s->n_waves;
  p.ln_tab = s->ln_tab_dev;
  p.n_ln = n_ln;
  hipLaunchKernelGGL(k_ln_table, dim3((n_ln + 255) / 256), dim3(256), 0, nullptr, s->ln_tab_dev, n_ln);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(p.stamps, 0, (size_t)s->n_waves_cap * 4 * sizeof(unsigned long long)));
  HIP_TRY(reset_clock_acc(p.clock_acc));
  HIP_TRY(hipMemset(p.phase, 0, phase_rows * 16 * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(p.slots, 0, n * sizeof(Slot)));
  HIP_TRY(hipMemset(p.glob, 0, sizeof(Globals)));
  HIP_TRY(hipMemset(p.reclaim_ctr, 0, 2 * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(p.wave_ctr, 0, (size_t)s->n_waves_cap * CTR_N * sizeof(unsigned long long)));
  memset(s->probe_host, 0, sizeof(Globals));
  return C4_OK;
}

int c4_session_create(const c4_config* cfg, c4_session** out) {
  if (!cfg || !out) return fail(C4_ERR_BAD_ARG, "null argument");
  *out = nullptr;
  if (cfg->n_slots == 0) return fail(C4_ERR_BAD_ARG, "n_slots must be > 0");
  if (cfg->planes_dtype > 1) return fail(C4_ERR_BAD_ARG, "planes_dtype must be 0 (f32) or 1 (bf16)");
  if (cfg->blocks_per_slot > kMaxBlocksPerSlot) return fail(C4_ERR_BAD_ARG, "blocks_per_slot is limited to 65535 (16-bit child links)");
  // A game's arena is never reclaimed while it is played: every expansion of the whole game takes one
  // block, at most n_mcts_iterations per move and 42 moves, so 43 n + 8 blocks always suffice.  That
  // provable bound fits the 16-bit child links up to n = 1523 (the reference's own sweep tops out at
  // 1 500, src/c4a0/main.py:176).  Beyond it a long game could overflow its arena in the middle of a
  // job (C4_ERR_ARENA_OVERFLOW loses the whole call), so the default sizing is refused here, with the
  // reason; a caller who knows its games are short may still pass blocks_per_slot explicitly.
  // (From round 5 the default sizing switches to a RECLAIMED arena above 1 000 iterations, see C4_FLAG_RECLAIM: the live subtree is at
  // most n + a few blocks, so two halves of 2.5 n + 554 blocks (reclaim_half_min + 1.5 n of slack) serve any game and the links stay 16 bits wide up to n = 32 213.)
  if (reclaim_mode(cfg)) {
    const uint32_t period = cfg->reclaim_period ? cfg->reclaim_period : kReclaimPeriod;
    const uint64_t need = reclaim_half_min(cfg->n_mcts_iterations, period, 2);
    const uint64_t half = cfg->blocks_per_slot ? cfg->blocks_per_slot / 2 : kMaxBlocksPerSlot / 2;
    if (cfg->flags & C4_FLAG_NO_MOVES) return fail(C4_ERR_BAD_ARG, "C4_FLAG_RECLAIM: a search that never moves never frees anything (C4_FLAG_NO_MOVES)");
    if (cfg->flags & C4_FLAG_NO_RECLAIM) return fail(C4_ERR_BAD_ARG, "C4_FLAG_RECLAIM and C4_FLAG_NO_RECLAIM exclude each other");
    if (period > 4096) return fail(C4_ERR_BAD_ARG, "reclaim_period is limited to 4096 step launches");
    if (half < need)
      return fail(C4_ERR_BAD_ARG, "reclaimed arena too small: each half must hold the live subtree (n_mcts_iterations + simulations per launch + 8 blocks) and the blocks of 4 x "
                                  "reclaim_period step launches (" + std::to_string(need) + " blocks per half, i.e. blocks_per_slot >= " + std::to_string(2 * need) +
                                  "; a half is at most 32 767 blocks: n_mcts_iterations <= " + std::to_string(kMaxBlocksPerSlot / 2 - (need - cfg->n_mcts_iterations)) + ")");
  } else if (cfg->blocks_per_slot == 0 && 43ull * cfg->n_mcts_iterations + 8 > kMaxBlocksPerSlot)
    return fail(C4_ERR_BAD_ARG, "n_mcts_iterations > 1523 is not supported with a never-reclaimed arena (C4_FLAG_NO_MOVES / C4_FLAG_NO_RECLAIM): a game's tree arena is limited to "
                                "65535 blocks (16-bit child links; up to 43 n + 8 blocks per game). "
                                "Pass blocks_per_slot explicitly to accept C4_ERR_ARENA_OVERFLOW on long searches");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(C4_ERR_NO_DEVICE, "no HIP device visible");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(C4_ERR_BAD_ARG, "device ordinal out of range");
  C4_ON_DEVICE(cfg->device);
  c4_session* s = new c4_session();
  s->cfg = *cfg;
  const int rc = session_create_on_device(cfg, s);
  if (rc != C4_OK) {   // every failure path releases what was allocated (the message survives destroy)
    const std::string msg = g_last_error;
    c4_session_destroy(s);
    return fail(rc, msg);
  }
  *out = s;
  return C4_OK;
}

int c4_session_destroy(c4_session* s) {
  if (!s) return C4_OK;
  c4host::DeviceGuard guard(s->cfg.device);
  if (s->stream) (void)hipStreamSynchronize(s->stream); else (void)hipDeviceSynchronize();
  (void)hipFree(s->p.slots); arena_release(s->cfg.device, s->p.blocks, s->arena_bytes); (void)hipFree(s->p.wave_ctr); (void)hipFree(s->p.glob); (void)hipFree(s->p.stamps); (void)hipFree(s->p.clock_acc); (void)hipFree(s->p.phase); (void)hipFree(s->p.reclaim_ctr);
  (void)hipFree(s->p.samples); (void)hipFree(s->p.sample_counts); (void)hipFree(s->p.cache);
  (void)hipFree(s->plan_dev); (void)hipFree(s->pairs_dev); (void)hipFree(s->offsets_dev); (void)hipFree(s->ln_tab_dev);
  if (s->total_host) (void)hipHostFree(s->total_host);
  (void)hipFree(s->uniq_tab); (void)hipFree(s->uniq_cell); (void)hipFree(s->uniq_row); (void)hipFree(s->uniq_count);
  (void)hipFree(s->reqs_dev); (void)hipFree(s->start_mask_dev); (void)hipFree(s->start_value_dev);
  if (s->probe_host) (void)hipHostFree(s->probe_host);
  if (s->probe_event) (void)hipEventDestroy(s->probe_event);
  delete s;
  return C4_OK;
}

int c4_trim_cached_memory(void) {
  ArenaCache c;
  {
    std::lock_guard<std::mutex> lock(g_arena_mutex);
    c = g_arena_cache;
    g_arena_cache = ArenaCache{};
  }
  if (c.ptr) {
    c4host::DeviceGuard guard(c.device);
    HIP_TRY(hipFree(c.ptr));
  }
  return C4_OK;
}

int c4_session_set_games(c4_session* s, const c4_game_metadata* reqs, uint64_t n_games,
                         const uint64_t* start_masks, const uint64_t* start_values) {
  if (!s || (!reqs && n_games)) return fail(C4_ERR_BAD_ARG, "null argument");
  if ((start_masks == nullptr) != (start_values == nullptr)) return fail(C4_ERR_BAD_ARG, "start_masks and start_values go together");
  if (n_games >= (1ull << 32) - 1) return fail(C4_ERR_BAD_ARG, "too many games for one session");
  C4_ON_DEVICE(s->cfg.device);
  HIP_TRY(hipStreamSynchronize(s->stream));
  (void)hipFree(s->reqs_dev); (void)hipFree(s->start_mask_dev); (void)hipFree(s->start_value_dev);
  (void)hipFree(s->p.samples); (void)hipFree(s->p.sample_counts); (void)hipFree(s->offsets_dev);
  s->reqs_dev = nullptr; s->start_mask_dev = s->start_value_dev = nullptr;
  s->p.samples = nullptr; s->p.sample_counts = nullptr; s->offsets_dev = nullptr;
  const size_t ng = n_games ? n_games : 1;
  HIP_TRY(hipMalloc(&s->offsets_dev, (ng + 1) * sizeof(unsigned long long)));
  if (!s->total_host) HIP_TRY(hipHostMalloc(&s->total_host, sizeof(unsigned long long)));
  HIP_TRY(hipMalloc(&s->reqs_dev, ng * sizeof(c4_game_metadata)));
  HIP_TRY(hipMalloc(&s->p.samples, ng * C4_MAX_SAMPLES_PER_GAME * sizeof(c4_sample_rec)));
  HIP_TRY(hipMalloc(&s->p.sample_counts, ng * sizeof(uint32_t)));
  HIP_TRY(hipMemset(s->p.sample_counts, 0, ng * sizeof(uint32_t)));
  if (n_games) HIP_TRY(hipMemcpy(s->reqs_dev, reqs, n_games * sizeof(c4_game_metadata), hipMemcpyHostToDevice));
  if (start_masks && n_games) {
    HIP_TRY(hipMalloc(&s->start_mask_dev, n_games * 8));
    HIP_TRY(hipMalloc(&s->start_value_dev, n_games * 8));
    HIP_TRY(hipMemcpy(s->start_mask_dev, start_masks, n_games * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->start_value_dev, start_values, n_games * 8, hipMemcpyHostToDevice));
  }
  s->p.n_slots = s->cfg.n_slots;   // a compacted session goes back to its full width
  s->n_waves = s->p.n_waves = (s->cfg.n_slots + 64u / s->lanes_per_game - 1) / (64u / s->lanes_per_game);
  Globals g0{};
  g0.queue_head = n_games < s->cfg.n_slots ? n_games : s->cfg.n_slots;
  HIP_TRY(hipMemcpy(s->p.glob, &g0, sizeof g0, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(s->p.wave_ctr, 0, (size_t)s->n_waves_cap * CTR_N * sizeof(unsigned long long)));
  s->p.reqs = s->reqs_dev;
  s->p.start_mask = s->start_mask_dev;
  s->p.start_value = s->start_value_dev;
  s->p.n_games = n_games;
  s->n_games = n_games;
  s->have_games = true;
  HIP_TRY(hipMemset(s->p.stamps, 0, (size_t)s->n_waves_cap * 4 * sizeof(unsigned long long)));
  HIP_TRY(reset_clock_acc(s->p.clock_acc));
  // a new list of games may come with new evaluator weights: forget the old evaluations
  if (s->p.cache) HIP_TRY(hipMemset(s->p.cache, 0, ((size_t)s->p.cache_mask + 1) * 64));
  s->seq = 0;
  s->reclaim_count = 0; s->reclaim_capture_id = 0;
  HIP_TRY(hipMemset(s->p.reclaim_ctr, 0, 2 * sizeof(unsigned long long)));
  s->probe_pending = false; s->probe_done = 0; s->probe_error = 0;
  return C4_OK;
}

int c4_session_bind_io(c4_session* s, void* planes_dev, const float* logprobs_dev, const float* q_dev, void* stream) {
  if (!s || !planes_dev || !logprobs_dev || !q_dev) return fail(C4_ERR_BAD_ARG, "null argument");
  // the tensors and the stream must live on the session's device: a pointer of another device (or of
  // the host) would only fault later, inside a kernel, far from its cause
  C4_ON_DEVICE(s->cfg.device);
  const struct { const void* ptr; const char* name; } io[3] = {{planes_dev, "planes_dev"}, {logprobs_dev, "logprobs_dev"}, {q_dev, "q_dev"}};
  for (const auto& t : io) {
    hipPointerAttribute_t attr{};
    const hipError_t e = hipPointerGetAttributes(&attr, t.ptr);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(C4_ERR_BAD_ARG, std::string("c4_session_bind_io: ") + t.name + " is not a device pointer known to HIP (" + hipGetErrorString(e) + ")");
    }
    if (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged)
      return fail(C4_ERR_BAD_ARG, std::string("c4_session_bind_io: ") + t.name + " is not device memory");
    if (attr.type == hipMemoryTypeDevice && attr.device != s->cfg.device)
      return fail(C4_ERR_BAD_ARG, std::string("c4_session_bind_io: ") + t.name + " lives on device " + std::to_string(attr.device) +
                                      ", the session on device " + std::to_string(s->cfg.device));
  }
  if (stream != nullptr) {
    int sdev = -1;
    if (hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess && sdev >= 0 && sdev != s->cfg.device)
      return fail(C4_ERR_BAD_ARG, "c4_session_bind_io: the stream belongs to device " + std::to_string(sdev) + ", the session to device " +
                                      std::to_string(s->cfg.device));
  }
  s->p.planes = planes_dev;
  s->p.logprobs = logprobs_dev;
  s->p.q = q_dev;
  s->stream = (hipStream_t)stream;
  s->bound = true;
  return C4_OK;
}

int c4_session_set_dirichlet(c4_session* s, float alpha, float epsilon) {
  if (!s) return fail(C4_ERR_BAD_ARG, "null session");
  if (!(epsilon >= 0.0f && epsilon <= 1.0f) || (epsilon > 0.0f && !(alpha > 0.0f))) return fail(C4_ERR_BAD_ARG, "need 0 <= epsilon <= 1 and alpha > 0");
  s->p.dir_alpha = alpha;
  s->p.dir_eps = epsilon;
  return C4_OK;
}

int c4_session_bind_leaf_models(c4_session* s, uint64_t* leaf_models_dev) {
  if (!s) return fail(C4_ERR_BAD_ARG, "null session");
  if (leaf_models_dev && s->p.cache) return fail(C4_ERR_BAD_ARG, "the evaluation cache holds ONE evaluator's outputs: not with multi-model games");
  s->p.leaf_models = leaf_models_dev;
  return C4_OK;
}

int c4_session_set_eval_cache(c4_session* s, uint64_t n_entries, uint32_t max_sims_per_step) {
  if (!s) return fail(C4_ERR_BAD_ARG, "null session");
  if (n_entries && s->p.leaf_models) return fail(C4_ERR_BAD_ARG, "the evaluation cache holds ONE evaluator's outputs: not with multi-model games");
  if (n_entries > (1ull << 31)) return fail(C4_ERR_BAD_ARG, "at most 2^31 cache entries");
  C4_ON_DEVICE(s->cfg.device);
  HIP_TRY(hipStreamSynchronize(s->stream));
  (void)hipFree(s->p.cache);
  s->p.cache = nullptr;
  s->p.cache_mask = 0;
  const bool single = (s->cfg.flags & (C4_FLAG_NO_MOVES | C4_FLAG_ONE_SIM_PER_STEP)) != 0;
  s->p.max_sims = single ? 1u : 2u;
  if (n_entries == 0) return C4_OK;
  uint64_t n = 1024;
  while (n < n_entries) n <<= 1;
  HIP_TRY(hipMalloc(&s->p.cache, n * 64));
  HIP_TRY(hipMemset(s->p.cache, 0, n * 64));   // an all-zero entry does not validate (its dwords XOR to 0, not to the seal constant)
  s->p.cache_mask = (uint32_t)(n - 1);
  if (!single) s->p.max_sims = max_sims_per_step ? (max_sims_per_step > 64 ? 64u : max_sims_per_step) : 6u;   // measured at BASELINE config 2: 4 -> 43.3 k, 6 -> 45.5 k, 8 -> 44.8 k games/s
  if (s->p.half_blocks) {   // a reclaimed half holds the blocks of 4 x reclaim_period launches: as many simulations per launch as that allows
    if (s->p.max_sims > kReclaimMaxSims) s->p.max_sims = kReclaimMaxSims;
    while (s->p.max_sims > 2 && reclaim_half_min(s->p.n_iter, s->reclaim_period, s->p.max_sims) > s->p.half_blocks) s->p.max_sims--;
    // more simulations per launch = a larger margin to keep free: the next step launch is followed by a look at the arenas with
    // the new margin, whatever the count since the last one (a slot left with just over the OLD margin could otherwise
    // take reclaim_period launches of the new width before the next look)
    s->reclaim_count = 0;
  }
  return C4_OK;
}

int c4_session_start(c4_session* s) {
  if (!s) return fail(C4_ERR_BAD_ARG, "null session");
  if (!s->bound || !s->have_games) return fail(C4_ERR_NOT_BOUND, "bind_io and set_games must precede start");
  C4_ON_DEVICE(s->cfg.device);
  if (s->cfg.planes_dtype == 0)
    hipLaunchKernelGGL(c4_start_kernel<float>, dim3((s->p.n_slots + 7) / 8), dim3(64), 0, s->stream, s->p);
  else
    hipLaunchKernelGGL(c4_start_kernel<uint16_t>, dim3((s->p.n_slots + 7) / 8), dim3(64), 0, s->stream, s->p);
  HIP_TRY(hipGetLastError());
  return C4_OK;
}

static int device_view(const void* ptr, int device, const char* what, void** out);   // defined with the callback-mode entry points below

static int launch_step(c4_session* s, const uint32_t* inverse, const float* answers, uint32_t n_unique) {
  if (!s) return fail(C4_ERR_BAD_ARG, "null session");
  if (!s->bound || !s->have_games) return fail(C4_ERR_NOT_BOUND, "bind_io and set_games must precede step");
  C4_ON_DEVICE(s->cfg.device);
  // launch sequence number for the device-clock stamps; frozen at 0 (= no per-launch timing) when
  // timing is off, which is what a launch captured into a HIP graph needs (arguments are baked in)
  s->p.seq = s->timing ? ++s->seq : 0;
  // the Dirichlet-noise and evaluation-cache extensions are separate instantiations: the default
  // kernel carries none of their registers or scratch
  const bool noise = s->p.dir_eps > 0.0f, cache = s->p.cache != nullptr;
#ifdef C4_DIAG_VARIANTS
  // C4_STEP_LDS_BYTES (diagnostic build only, tools/occupancy_probe.sh): unused dynamic LDS per workgroup caps the
  // wavefronts a CU holds (160 KB / bytes) without touching the code: how the launch time scales with occupancy
  static const unsigned lds_pad = [] { const char* e = getenv("C4_STEP_LDS_BYTES"); return e ? (unsigned)atoi(e) : 0u; }();
#else
  constexpr unsigned lds_pad = 0;
#endif
  // a timed launch (seq != 0) carries extra workgroups that fold the previous launch's stamps
  const uint32_t helpers = s->p.seq ? (s->n_waves + kWavesPerTimingHelper - 1) / kWavesPerTimingHelper : 0u;
  auto launch = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(s->n_waves + helpers), dim3(64), lds_pad, s->stream, s->p.slots, s->p.logprobs, s->p.q, s->p.n_waves, s->p.n_slots, s->p);
  };
  auto launch_gather = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(s->n_waves + helpers), dim3(64), lds_pad, s->stream, s->p.slots, answers, inverse, n_unique, s->p.n_waves, s->p.n_slots, s->p);
  };
  const bool f32 = s->cfg.planes_dtype == 0;
#define C4_LAUNCH_STEP(KERNEL, LAUNCH)                                                                               \
  do {                                                                                                               \
    if (f32) {                                                                                                       \
      if (noise) { if (cache) LAUNCH(KERNEL<float, true, true>); else LAUNCH(KERNEL<float, true, false>); }          \
      else       { if (cache) LAUNCH(KERNEL<float, false, true>); else LAUNCH(KERNEL<float, false, false>); }        \
    } else {                                                                                                         \
      if (noise) { if (cache) LAUNCH(KERNEL<uint16_t, true, true>); else LAUNCH(KERNEL<uint16_t, true, false>); }    \
      else       { if (cache) LAUNCH(KERNEL<uint16_t, false, true>); else LAUNCH(KERNEL<uint16_t, false, false>); }  \
    }                                                                                                                \
  } while (0)
  if (inverse) C4_LAUNCH_STEP(c4_step_gather_kernel, launch_gather); else C4_LAUNCH_STEP(c4_step_kernel, launch);
#undef C4_LAUNCH_STEP
  HIP_TRY(hipGetLastError());
  return maybe_reclaim(s);
}

int c4_session_step(c4_session* s) { return launch_step(s, nullptr, nullptr, 0); }

int c4_session_step_gather(c4_session* s, const uint32_t* inverse_dev, const float* answers, uint32_t n_unique) {
  if (!s || !inverse_dev || (!answers && n_unique)) return fail(C4_ERR_BAD_ARG, "null argument");
  if (!s->bound) return fail(C4_ERR_NOT_BOUND, "c4_session_step_gather: bind_io first");
  void *inv = nullptr, *ans = nullptr;
  {
    C4_ON_DEVICE(s->cfg.device);
    if (int rc = device_view(inverse_dev, s->cfg.device, "c4_session_step_gather: inverse_dev", &inv)) return rc;
    if (answers) if (int rc = device_view(answers, s->cfg.device, "c4_session_step_gather: answers", &ans)) return rc;
  }
  return launch_step(s, (const uint32_t*)inv, (const float*)ans, n_unique);
}

int c4_session_step_head_out(c4_session* s, const void* hidden_policy_dev, const void* hidden_value_dev, const void* w_policy_dev,
                             const void* w_value_dev, const float* b_policy_dev, const float* b_value_dev, uint32_t features,
                             uint32_t policy_row_stride, uint32_t value_row_stride) {
  if (!s) return fail(C4_ERR_BAD_ARG, "null session");
  if (!s->bound || !s->have_games) return fail(C4_ERR_NOT_BOUND, "bind_io and set_games must precede step");
  if (!hidden_policy_dev || !hidden_value_dev || !w_policy_dev || !w_value_dev || !b_policy_dev || !b_value_dev)
    return fail(C4_ERR_BAD_ARG, "c4_session_step_head_out: null argument");
  if (features % 8 != 0 || policy_row_stride % 8 != 0 || value_row_stride % 8 != 0 || (features / 8) % (4 * c4ho::kHeadSteps * c4ho::kHeadWaves) != 0)
    return fail(C4_ERR_BAD_ARG, "c4_session_step_head_out: features must be a multiple of 1 344 (42 x 32 channels) and the row strides of 8 elements");
  if (s->p.dir_eps > 0.0f || s->p.cache != nullptr || s->timing)
    return fail(C4_ERR_BAD_ARG, "c4_session_step_head_out: the fused launch exists for the default configuration only (no Dirichlet noise, no evaluation "
                                "cache, per-launch timing off): call c4_head_out_bf16 and c4_session_step");
  C4_ON_DEVICE(s->cfg.device);
  s->p.seq = 0;
  const uint32_t groups = (s->p.n_slots + 15) / 16;
  auto launch = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(groups), dim3(64 * c4ho::kHeadWaves), 0, s->stream, (const uint4*)hidden_policy_dev, (const uint4*)hidden_value_dev,
                       (const uint4*)w_policy_dev, (const uint4*)w_value_dev, b_policy_dev, b_value_dev, s->p.slots, s->p.n_slots, features / 8,
                       policy_row_stride / 8, value_row_stride / 8, s->p);
  };
  if (s->out_step_gpw == 4) {
    if (s->cfg.planes_dtype == 0) launch(c4_out_step_kernel<float, 4>); else launch(c4_out_step_kernel<uint16_t, 4>);
  } else {
    if (s->cfg.planes_dtype == 0) launch(c4_out_step_kernel<float, 8>); else launch(c4_out_step_kernel<uint16_t, 8>);
  }
  HIP_TRY(hipGetLastError());
  return maybe_reclaim(s);
}

// Games per stepping wavefront of the fused output + step launch: 8 (default) or 4 (see c4_out_step_kernel).  A scheduling knob: the
// games' records do not depend on it.
int c4_session_set_step_shape(c4_session* s, uint32_t games_per_wavefront) {
  if (!s) return fail(C4_ERR_BAD_ARG, "null session");
  if (games_per_wavefront != 4 && games_per_wavefront != 8) return fail(C4_ERR_BAD_ARG, "c4_session_set_step_shape: games_per_wavefront must be 4 or 8");
  s->out_step_gpw = games_per_wavefront;
  return C4_OK;
}

int c4_session_set_timing(c4_session* s, int enable) {
  if (!s) return fail(C4_ERR_BAD_ARG, "null session");
  C4_ON_DEVICE(s->cfg.device);
  HIP_TRY(hipStreamSynchronize(s->stream));
  // fold nothing across the switch: restart the stamp buffers
  HIP_TRY(hipMemset(s->p.stamps, 0, (size_t)s->n_waves_cap * 4 * sizeof(unsigned long long)));
  if (enable && !s->timing) s->seq = 0;
  s->timing = enable != 0;
  return C4_OK;
}

int c4_session_counters(c4_session* s, c4_counters* out) {
  if (!s || !out) return fail(C4_ERR_BAD_ARG, "null argument");
  C4_ON_DEVICE(s->cfg.device);
  HIP_TRY(hipStreamSynchronize(s->stream));
  std::vector<unsigned long long> h((size_t)s->n_waves_cap * CTR_N);   // waves retired by a compaction keep their counts
  HIP_TRY(hipMemcpy(h.data(), s->p.wave_ctr, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  Globals g{};
  HIP_TRY(hipMemcpy(&g, s->p.glob, sizeof g, hipMemcpyDeviceToHost));
  unsigned long long sum[CTR_N] = {0};
  for (size_t w = 0; w < s->n_waves_cap; w++)
    for (int k = 0; k < CTR_N; k++) sum[k] += h[w * CTR_N + k];
  memset(out, 0, sizeof *out);
  out->sims = sum[CTR_SIMS]; out->select_levels = sum[CTR_S]; out->backup_nodes = sum[CTR_K];
  out->expansions = sum[CTR_E]; out->moves = sum[CTR_MOVES]; out->games_done = sum[CTR_DONE];
  out->ref_skipped_sims = sum[CTR_SKIPPED]; out->samples = sum[CTR_SAMPLES];
  out->eval_cache_probes = sum[CTR_PROBES]; out->eval_cache_hits = sum[CTR_HITS];
  out->games_started = g.queue_head < s->n_games ? g.queue_head : s->n_games;
  out->error = g.error; out->error_slot = g.error_slot;
  unsigned long long rc2[2] = {0, 0};
  HIP_TRY(hipMemcpy(rc2, s->p.reclaim_ctr, sizeof rc2, hipMemcpyDeviceToHost));
  out->reclaim_passes = rc2[0]; out->reclaim_blocks = rc2[1];
  // device-clock time of the step kernel: launches already folded in by the following launch,
  // plus the last one from its raw stamps
  unsigned long long acc[2] = {0, 0};
  HIP_TRY(hipMemcpy(acc, s->p.clock_acc, sizeof acc, hipMemcpyDeviceToHost));
  if (s->seq > 0) {
    std::vector<unsigned long long> st((size_t)s->n_waves * 2);
    HIP_TRY(hipMemcpy(st.data(), s->p.stamps + (size_t)(s->seq & 1) * s->n_waves * 2, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long lo = ~0ull, hi = 0ull;
    for (size_t w = 0; w < s->n_waves; w++) { if (st[2 * w] < lo) lo = st[2 * w]; if (st[2 * w + 1] > hi) hi = st[2 * w + 1]; }
    if (hi > lo) { acc[0] += hi - lo; acc[1] += 1; }
  }
  out->step_kernel_ns = acc[0] * 10ull;  // s_memrealtime ticks at 100 MHz
  out->step_launches = acc[1];
  return C4_OK;
}

int c4_session_poll(c4_session* s, uint64_t* games_done, uint32_t* error) {
  if (!s) return fail(C4_ERR_BAD_ARG, "null session");
  C4_ON_DEVICE(s->cfg.device);
  if (s->probe_pending && hipEventQuery(s->probe_event) == hipSuccess) {
    s->probe_done = s->probe_host->games_done;
    s->probe_started = s->probe_host->queue_head < s->n_games ? s->probe_host->queue_head : s->n_games;
    s->probe_error = s->probe_host->error;
    s->probe_pending = false;
  }
  if (!s->probe_pending) {
    HIP_TRY(hipMemcpyAsync(s->probe_host, s->p.glob, sizeof(Globals), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipEventRecord(s->probe_event, s->stream));
    s->probe_pending = true;
  }
  if (games_done) *games_done = s->probe_done;
  if (error) *error = s->probe_error;
  return C4_OK;
}

int c4_session_progress(c4_session* s, uint64_t* games_done, uint64_t* games_started, uint32_t* error) {
  const int rc = c4_session_poll(s, games_done, error);
  if (rc == C4_OK && games_started) *games_started = s->probe_started;
  return rc;
}

int c4_session_compact(c4_session* s, uint32_t multiple, uint32_t* n_active, uint32_t* n_slots_now) {
  if (!s || !s->bound || !s->have_games) return fail(C4_ERR_BAD_ARG, "compact needs a bound session with games");
  if (s->p.leaf_models) return fail(C4_ERR_BAD_ARG, "compaction does not move the per-slot model ids of multi-model sessions");
  if (multiple == 0 || multiple % 8) return fail(C4_ERR_BAD_ARG, "multiple must be a positive multiple of 8");
  C4_ON_DEVICE(s->cfg.device);
  HIP_TRY(hipStreamSynchronize(s->stream));
  Globals g{};
  HIP_TRY(hipMemcpy(&g, s->p.glob, sizeof g, hipMemcpyDeviceToHost));
  if (n_slots_now) *n_slots_now = s->p.n_slots;
  if (g.queue_head < s->n_games) {   // slots are still being refilled: nothing to gain
    if (n_active) *n_active = s->p.n_slots;
    return C4_OK;
  }
  if (!s->plan_dev) {
    HIP_TRY(hipMalloc(&s->plan_dev, sizeof(CompactPlan)));
    HIP_TRY(hipMalloc(&s->pairs_dev, (size_t)s->cfg.n_slots * sizeof(uint2)));
  }
  hipLaunchKernelGGL(k_compact_plan, dim3(1), dim3(1024), 0, s->stream, s->p.slots, s->p.n_slots, s->plan_dev, s->pairs_dev);
  const uint32_t max_pairs = s->p.n_slots / 2 + 1;
  if (s->cfg.planes_dtype == 0) hipLaunchKernelGGL(k_compact_move<float>, dim3(max_pairs), dim3(256), 0, s->stream, s->p, s->plan_dev, s->pairs_dev);
  else hipLaunchKernelGGL(k_compact_move<uint16_t>, dim3(max_pairs), dim3(256), 0, s->stream, s->p, s->plan_dev, s->pairs_dev);
  HIP_TRY(hipGetLastError());
  CompactPlan plan{};
  HIP_TRY(hipMemcpyAsync(&plan, s->plan_dev, sizeof plan, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  uint32_t want = ((plan.n_active + multiple - 1) / multiple) * multiple;
  if (want < multiple) want = multiple;
  if (want < s->p.n_slots) {
    s->p.n_slots = want;
    s->n_waves = (want + 64u / s->lanes_per_game - 1) / (64u / s->lanes_per_game);
    s->p.n_waves = s->n_waves;
    HIP_TRY(hipMemset(s->p.stamps, 0, (size_t)s->n_waves_cap * 4 * sizeof(unsigned long long)));   // the stamp stride changed
  }
  if (n_active) *n_active = plan.n_active;
  if (n_slots_now) *n_slots_now = s->p.n_slots;
  return C4_OK;
}

int c4_session_arena(c4_session* s, uint64_t* bytes, uint32_t* blocks_per_slot, uint32_t* reclaim_half_blocks) {
  if (!s) return fail(C4_ERR_BAD_ARG, "null session");
  if (bytes) *bytes = s->arena_bytes;
  if (blocks_per_slot) *blocks_per_slot = s->p.blocks_per_slot;
  if (reclaim_half_blocks) *reclaim_half_blocks = s->p.half_blocks;
  return C4_OK;
}

int c4_session_sample_counts(c4_session* s, uint32_t* counts_host, uint64_t n_games) {
  if (!s || !counts_host) return fail(C4_ERR_BAD_ARG, "null argument");
  if (n_games != s->n_games) return fail(C4_ERR_BAD_ARG, "n_games does not match set_games");
  C4_ON_DEVICE(s->cfg.device);
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (n_games) HIP_TRY(hipMemcpy(counts_host, s->p.sample_counts, n_games * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return C4_OK;
}

int c4_session_drain_samples(c4_session* s, c4_sample_rec* dst_host, uint64_t cap, uint64_t* n_written) {
  if (!s || !n_written) return fail(C4_ERR_BAD_ARG, "null argument");
  if (!s->have_games) return fail(C4_ERR_NOT_BOUND, "set_games must precede drain_samples");
  // The records are packed ON THE DEVICE (prefix sum + K6, as for the collective) and come back in ONE
  // transfer straight into the caller's buffer: no host-side staging of the 43-record-per-game store.
  uint64_t total = 0;
  int rc = c4_session_pack_samples(s, nullptr, 0, &total);   // size query: offsets + total (synchronises the stream)
  if (rc != C4_OK) return rc;
  *n_written = total;
  if (!dst_host || total == 0) return C4_OK;
  if (cap < total) return fail(C4_ERR_BAD_ARG, "destination too small");
  C4_ON_DEVICE(s->cfg.device);
  c4_sample_rec* tmp = nullptr;
  HIP_TRY(hipMalloc(&tmp, total * sizeof(c4_sample_rec)));
  rc = c4_session_pack_samples(s, tmp, total, &total);
  if (rc == C4_OK) {
    const hipError_t e = hipMemcpy(dst_host, tmp, total * sizeof(c4_sample_rec), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(C4_ERR_HIP, std::string("drain_samples: copying the packed records: ") + hipGetErrorString(e));
  }
  (void)hipFree(tmp);
  return rc;
}

int c4_session_pack_samples(c4_session* s, c4_sample_rec* dst_dev, uint64_t cap, uint64_t* n_written) {
  if (!s || !n_written) return fail(C4_ERR_BAD_ARG, "null argument");
  if (!s->have_games) return fail(C4_ERR_NOT_BOUND, "set_games must precede pack_samples");
  C4_ON_DEVICE(s->cfg.device);
  // record offsets by a device prefix sum into the session's persistent buffer; only the total comes back
  unsigned long long* total_dev = s->offsets_dev + (s->n_games ? s->n_games : 1);
  hipLaunchKernelGGL(k_sample_offsets, dim3(1), dim3(1024), 0, s->stream, s->p.sample_counts, (unsigned long long)s->n_games,
                     s->offsets_dev, total_dev);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(s->total_host, total_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  const uint64_t total = *s->total_host;
  *n_written = total;
  if (!dst_dev || total == 0) return C4_OK;  // size query
  if (cap < total) return fail(C4_ERR_BAD_ARG, "destination too small");
  hipLaunchKernelGGL(k_pack_samples, dim3((unsigned)s->n_games), dim3(64), 0, s->stream, s->p.samples, s->p.sample_counts,
                     s->offsets_dev, s->n_games, dst_dev);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s->stream));
  return C4_OK;
}

// diagnostic builds only: raw phase stamps [n_waves][16] of the last launch (zeros otherwise)
int c4_session_debug_phase_stamps(c4_session* s, uint64_t* out_host, uint64_t cap_words, uint64_t* n_words) {
  if (!s || !n_words) return fail(C4_ERR_BAD_ARG, "null argument");
  *n_words = (uint64_t)s->n_waves * 16;
  if (!out_host) return C4_OK;
  if (cap_words < *n_words) return fail(C4_ERR_BAD_ARG, "destination too small");
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipMemcpy(out_host, s->p.phase, *n_words * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return C4_OK;
}

int c4_session_sample_store(c4_session* s, const c4_sample_rec** recs_dev, const uint32_t** counts_dev, uint64_t* n_games) {
  if (!s) return fail(C4_ERR_BAD_ARG, "null session");
  if (recs_dev) *recs_dev = s->p.samples;
  if (counts_dev) *counts_dev = s->p.sample_counts;
  if (n_games) *n_games = s->n_games;
  return C4_OK;
}

int c4_session_root_stats(c4_session* s, uint32_t slot, float policy[7], float* q_penalty, float* q_no_penalty,
                          uint64_t* visit_count, uint64_t* root_mask, uint64_t* root_value) {
  if (!s || slot >= s->cfg.n_slots) return fail(C4_ERR_BAD_ARG, "bad slot");
  C4_ON_DEVICE(s->cfg.device);
  HIP_TRY(hipStreamSynchronize(s->stream));
  Slot st;
  HIP_TRY(hipMemcpy(&st, s->p.slots + slot, sizeof st, hipMemcpyDeviceToHost));
  const size_t base = (size_t)slot * s->cfg.blocks_per_slot;
  Block rb;
  HIP_TRY(hipMemcpy(&rb, s->p.blocks + base + (st.root_ref >> 3), sizeof rb, hipMemcpyDeviceToHost));
  const Entry& re = rb.e[st.root_ref & 7];
  // mcts.rs:359-367: q_sum / (visit_count as f32 + 1.0)
  const float nf = (float)re.n + 1.0f;
  if (q_penalty) *q_penalty = re.q_pen / nf;
  if (q_no_penalty) *q_no_penalty = re.q_nopen / nf;
  if (visit_count) *visit_count = re.n;
  if (root_mask) *root_mask = st.root_mask;
  if (root_value) *root_value = st.root_value;
  if (policy) {
    // mcts.rs:396-412
    float cnt[7] = {0, 0, 0, 0, 0, 0, 0}, sum = 0.0f;
    const uint32_t root_block = st.arena >> 16;
    if (root_block) {
      Block cb;
      HIP_TRY(hipMemcpy(&cb, s->p.blocks + base + root_block, sizeof cb, hipMemcpyDeviceToHost));
      for (int c = 0; c < 7; c++) cnt[c] = (float)cb.e[c].n;
    }
    for (int c = 0; c < 7; c++) sum = sum + cnt[c];
    for (int c = 0; c < 7; c++) policy[c] = (sum == 0.0f) ? (1.0f / 7.0f) : cnt[c] / sum;
  }
  return C4_OK;
}

int c4_session_leaf_keys(c4_session* s, int64_t* keys_dev) {
  if (!s || !keys_dev) return fail(C4_ERR_BAD_ARG, "null argument");
  C4_ON_DEVICE(s->cfg.device);
  hipLaunchKernelGGL(k_leaf_keys, dim3((s->cfg.n_slots + 255) / 256), dim3(256), 0, s->stream, s->p.slots, s->cfg.n_slots, (long long*)keys_dev);
  HIP_TRY(hipGetLastError());
  return C4_OK;
}

// A pointer kernels of device `device` may use for `ptr`: device memory of that device as it is, pinned host
// memory through its device mapping; anything else (pageable host memory, another device) is refused here,
// where the message can say so, instead of faulting inside a kernel.
static int device_view(const void* ptr, int device, const char* what, void** out) {
  hipPointerAttribute_t attr{};
  const hipError_t e = hipPointerGetAttributes(&attr, ptr);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(C4_ERR_BAD_ARG, std::string(what) + " is neither device memory nor pinned host memory (" + hipGetErrorString(e) + ")");
  }
  if (attr.type == hipMemoryTypeHost) {
    if (!attr.devicePointer) return fail(C4_ERR_BAD_ARG, std::string(what) + ": pinned host memory without a device mapping");
    *out = attr.devicePointer;
    return C4_OK;
  }
  if (attr.type == hipMemoryTypeManaged || (attr.type == hipMemoryTypeDevice && attr.device == device)) { *out = const_cast<void*>(ptr); return C4_OK; }
  return fail(C4_ERR_BAD_ARG, std::string(what) + " is neither memory of device " + std::to_string(device) + " nor pinned host memory");
}

int c4_session_unique_leaves(c4_session* s, uint32_t* inverse_dev, float* rows_out, uint64_t* models_out, uint32_t* n_unique_out) {
  if (!s || !inverse_dev || !rows_out || !n_unique_out) return fail(C4_ERR_BAD_ARG, "null argument");
  if (!s->bound || !s->have_games) return fail(C4_ERR_BAD_ARG, "c4_session_unique_leaves: bind_io and set_games first");
  C4_ON_DEVICE(s->cfg.device);
  const uint32_t n = s->cfg.n_slots;
  void *inv = nullptr, *rows = nullptr, *models = nullptr, *count = nullptr;
  if (int rc = device_view(inverse_dev, s->cfg.device, "c4_session_unique_leaves: inverse_dev", &inv)) return rc;
  if (int rc = device_view(rows_out, s->cfg.device, "c4_session_unique_leaves: rows_out", &rows)) return rc;
  if (models_out) if (int rc = device_view(models_out, s->cfg.device, "c4_session_unique_leaves: models_out", &models)) return rc;
  if (int rc = device_view(n_unique_out, s->cfg.device, "c4_session_unique_leaves: n_unique_out", &count)) return rc;
  if (!s->uniq_tab) {
    uint32_t cells = 64;
    while (cells < 2 * n) cells <<= 1;                             // at most half full: short probe runs
    // all four or none: the session's fields are set only when every allocation succeeded (a half-made table with
    // mask 0 would send the next call's kernels through null pointers)
    uint32_t *tab = nullptr, *cell = nullptr, *row = nullptr, *cnt = nullptr;
    hipError_t e = hipMalloc(&tab, (size_t)cells * 4);
    if (e == hipSuccess) e = hipMalloc(&cell, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc(&row, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc(&cnt, 4);
    if (e == hipSuccess) e = hipMemsetAsync(tab, 0xFF, (size_t)cells * 4, s->stream);   // empty; k_unique_emit keeps it so
    if (e != hipSuccess) {
      (void)hipFree(tab); (void)hipFree(cell); (void)hipFree(row); (void)hipFree(cnt);
      return fail(C4_ERR_HIP, std::string("c4_session_unique_leaves: ") + hipGetErrorString(e));
    }
    s->uniq_tab = tab; s->uniq_cell = cell; s->uniq_row = row; s->uniq_count = cnt;
    s->uniq_tab_mask = cells - 1;
  }
  hipLaunchKernelGGL(k_unique_insert, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->p.slots, s->p.leaf_models, n, s->uniq_tab,
                     s->uniq_tab_mask, s->uniq_cell);
  hipLaunchKernelGGL(k_unique_rank, dim3(1), dim3(1024), 0, s->stream, s->uniq_tab, s->uniq_cell, n, s->uniq_row, (uint32_t*)inv,
                     s->uniq_count, (uint32_t*)count);
  hipLaunchKernelGGL(k_unique_emit, dim3((n + 3) / 4), dim3(256), 0, s->stream, s->p.slots, s->p.leaf_models, n, s->uniq_tab,
                     s->uniq_cell, s->uniq_row, (float*)rows, (uint64_t*)models);
  HIP_TRY(hipGetLastError());
  return C4_OK;
}

int c4_session_scatter_outputs(c4_session* s, const uint32_t* inverse_dev, const float* answers, uint32_t n_unique) {
  if (!s || !inverse_dev || (!answers && n_unique)) return fail(C4_ERR_BAD_ARG, "null argument");
  if (!s->bound) return fail(C4_ERR_BAD_ARG, "c4_session_scatter_outputs: bind_io first");
  if (n_unique == 0) return C4_OK;
  C4_ON_DEVICE(s->cfg.device);
  void *inv = nullptr, *ans = nullptr;
  if (int rc = device_view(inverse_dev, s->cfg.device, "c4_session_scatter_outputs: inverse_dev", &inv)) return rc;
  if (int rc = device_view(answers, s->cfg.device, "c4_session_scatter_outputs: answers", &ans)) return rc;
  const uint32_t n = s->cfg.n_slots;
  hipLaunchKernelGGL(k_unique_scatter, dim3((n * 9 + 255) / 256), dim3(256), 0, s->stream, (const uint32_t*)inv, (const float*)ans, n,
                     n_unique, const_cast<float*>(s->p.logprobs), const_cast<float*>(s->p.q));
  HIP_TRY(hipGetLastError());
  return C4_OK;
}

int c4_session_leaves(c4_session* s, uint64_t* masks_host, uint64_t* values_host, uint32_t* status_host,
                      uint32_t* ordinals_host) {
  if (!s) return fail(C4_ERR_BAD_ARG, "null session");
  C4_ON_DEVICE(s->cfg.device);
  HIP_TRY(hipStreamSynchronize(s->stream));
  std::vector<Slot> h(s->cfg.n_slots);
  HIP_TRY(hipMemcpy(h.data(), s->p.slots, h.size() * sizeof(Slot), hipMemcpyDeviceToHost));
  for (uint32_t g = 0; g < s->cfg.n_slots; g++) {
    if (masks_host) masks_host[g] = h[g].leaf_mask;
    if (values_host) values_host[g] = h[g].leaf_value;
    if (status_host) status_host[g] = h[g].state & 0xFFu;
    if (ordinals_host) ordinals_host[g] = h[g].ordinal;
  }
  return C4_OK;
}

// ---- element-wise entry points ----
// They run on the device their stream belongs to (the caller's current device for the null stream) and
// leave the caller's current device as they found it, like the session entry points.
static inline dim3 grid_for(uint64_t n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }
#define C4_ON_STREAM_DEVICE(stream) C4_ON_DEVICE(c4host::stream_device((hipStream_t)(stream)))

int c4_pos_ops(const uint64_t* mask_dev, const uint64_t* value_dev, const int32_t* col_dev, uint64_t n, float c_ply_penalty,
               uint64_t* out_mask_dev, uint64_t* out_value_dev, uint32_t* out_legal_dev, uint32_t* out_terminal_dev,
               float* out_q_dev, void* stream) {
  if (n == 0) return C4_OK;
  C4_ON_STREAM_DEVICE(stream);
  hipLaunchKernelGGL(k_pos_ops, grid_for(n), dim3(256), 0, (hipStream_t)stream, mask_dev, value_dev, col_dev, n, c_ply_penalty,
                     out_mask_dev, out_value_dev, out_legal_dev, out_terminal_dev, out_q_dev);
  HIP_TRY(hipGetLastError());
  return C4_OK;
}

int c4_encode_planes(const uint64_t* mask_dev, const uint64_t* value_dev, uint64_t n, uint32_t planes_dtype, void* planes_dev, void* stream) {
  if (n == 0) return C4_OK;
  C4_ON_STREAM_DEVICE(stream);
  if (planes_dtype == 0)
    hipLaunchKernelGGL(k_encode<float>, grid_for(n * C4_PLANES_LEN), dim3(256), 0, (hipStream_t)stream, mask_dev, value_dev, n, planes_dev);
  else if (planes_dtype == 1)
    hipLaunchKernelGGL(k_encode<uint16_t>, grid_for(n * C4_PLANES_LEN), dim3(256), 0, (hipStream_t)stream, mask_dev, value_dev, n, planes_dev);
  else
    return fail(C4_ERR_BAD_ARG, "planes_dtype must be 0 or 1");
  HIP_TRY(hipGetLastError());
  return C4_OK;
}

int c4_expf_logf(const float* x_dev, uint64_t n, int which, float* y_dev, void* stream) {
  if (n == 0) return C4_OK;
  C4_ON_STREAM_DEVICE(stream);
  hipLaunchKernelGGL(k_expf_logf, grid_for(n), dim3(256), 0, (hipStream_t)stream, x_dev, n, which, y_dev);
  HIP_TRY(hipGetLastError());
  return C4_OK;
}

int c4_softmax7(const float* logits_dev, const uint32_t* legal_dev, uint64_t n, float* out_dev, uint32_t* out_err_dev, void* stream) {
  if (n == 0) return C4_OK;
  C4_ON_STREAM_DEVICE(stream);
  hipLaunchKernelGGL(k_softmax7, grid_for(n), dim3(256), 0, (hipStream_t)stream, logits_dev, legal_dev, n, out_dev, out_err_dev);
  HIP_TRY(hipGetLastError());
  return C4_OK;
}

int c4_apply_temperature(const float* policy_dev, const float* temperature_dev, uint64_t n, float* out_dev, void* stream) {
  if (n == 0) return C4_OK;
  C4_ON_STREAM_DEVICE(stream);
  hipLaunchKernelGGL(k_temperature, grid_for(n), dim3(256), 0, (hipStream_t)stream, policy_dev, temperature_dev, n, out_dev);
  HIP_TRY(hipGetLastError());
  return C4_OK;
}

int c4_dirichlet(const uint64_t* game_id_dev, const uint32_t* n_moves_dev, const uint32_t* legal_dev, float alpha, uint64_t n,
                 float* eta_dev, void* stream) {
  if (n == 0) return C4_OK;
  C4_ON_STREAM_DEVICE(stream);
  hipLaunchKernelGGL(k_dirichlet, grid_for(n, 64), dim3(64), 0, (hipStream_t)stream, game_id_dev, n_moves_dev, legal_dev, alpha, n, eta_dev);
  HIP_TRY(hipGetLastError());
  return C4_OK;
}

int c4_sample_move(const uint64_t* game_id_dev, const uint32_t* n_moves_dev, const float* policy_dev, const float* temperature_dev,
                   uint64_t n, int32_t* out_col_dev, uint32_t* out_u32_dev, void* stream) {
  if (n == 0) return C4_OK;
  C4_ON_STREAM_DEVICE(stream);
  hipLaunchKernelGGL(k_sample_move, grid_for(n), dim3(256), 0, (hipStream_t)stream, game_id_dev, n_moves_dev, policy_dev,
                     temperature_dev, n, out_col_dev, out_u32_dev);
  HIP_TRY(hipGetLastError());
  return C4_OK;
}

}  // extern "C"
