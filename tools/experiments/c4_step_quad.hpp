// c4_step_quad.hpp -- the step kernel with FOUR lanes per game (16 games per wavefront).
//
// Included by c4_session.hip inside its anonymous namespace (it uses that file's Params, Slot, Block,
// counters and helpers).  Same algorithm, same arithmetic, same results as the 8-lanes-per-game kernel
// (c4_step_kernel); what changes is the mapping:
//   * a game is a QUAD of lanes, so every cross-lane step is a DPP quad_perm (no LDS crossbar at all);
//   * lane q owns two of the eight 16-byte slots of a children block: entries 2q and 2q + 1 (lane 3:
//     entry 6 and the tail), i.e. two UCT scores per lane and a 2-step argmax;
//   * the game's state line is read as header piece q + path piece q (lane q owns path levels q, q + 4,
//     q + 8, q + 12, so four backup levels update in parallel);
//   * the per-game scalar work (bitboards, terminal tests, gate, bookkeeping) -- most of the
//     instruction stream -- is issued once per 16 games instead of once per 8.
// Why: the 8-lane kernel is instruction-issue bound at large launches.
#pragma once

// ---- quad helpers: quad_perm DPP ------------------------------------------------------------
template <int K>
C4_DEV uint32_t qb(uint32_t v) {   // lane K of the quad -> all four lanes
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xF, 0xF, true);
}
template <int K>
C4_DEV float qb(float v) { return __uint_as_float(qb<K>(__float_as_uint(v))); }
template <int K>
C4_DEV uint64_t qb64(uint32_t lo, uint32_t hi) { return ((uint64_t)qb<K>(hi) << 32) | qb<K>(lo); }
// lane `src` (quad-uniform, run-time) of the quad -> all four lanes
C4_DEV uint32_t qfrom(uint32_t v, uint32_t src) {
  const uint32_t a = qb<0>(v), b = qb<1>(v), c = qb<2>(v), d = qb<3>(v);
  return src == 0 ? a : (src == 1 ? b : (src == 2 ? c : d));
}
C4_DEV float qfrom(float v, uint32_t src) { return __uint_as_float(qfrom(__float_as_uint(v), src)); }
// element c (0..6) of a 7-vector whose lane q holds elements 2q (v0) and 2q + 1 (v1) -> all four lanes
C4_DEV float q_elem(float v0, float v1, uint32_t c) { return qfrom((c & 1u) ? v1 : v0, c >> 1); }
C4_DEV uint32_t q_elem(uint32_t v0, uint32_t v1, uint32_t c) { return qfrom((c & 1u) ? v1 : v0, c >> 1); }
// all seven elements to every lane
C4_DEV void q_gather7(float v0, float v1, float* out) {
  out[0] = qb<0>(v0); out[1] = qb<0>(v1); out[2] = qb<1>(v0); out[3] = qb<1>(v1);
  out[4] = qb<2>(v0); out[5] = qb<2>(v1); out[6] = qb<3>(v0);
}
// sum of the seven elements in index order, starting from 0.0f (mcts.rs:432 `iter().sum()`): lane 0
// adds its two, hands the partial sum to lane 1, ... ; every lane returns the total
C4_DEV float q_sum7_in_order(float v0, float v1) {
  float t = (0.0f + v0) + v1;
  float s = qb<0>(t);
  t = (s + v0) + v1;
  s = qb<1>(t);
  t = (s + v0) + v1;
  s = qb<2>(t);
  t = s + v0;
  return qb<3>(t);
}

// Level `level` of the path (lane q holds levels q, q + 4, q + 8, q + 12 in pv) for every lane.
C4_DEV uint32_t q_path_level(const uint4& pv, const Slot* st, uint32_t level) {
  const uint32_t c = (level >> 2) & 3u;
  const uint32_t mine = c == 0 ? pv.x : (c == 1 ? pv.y : (c == 2 ? pv.z : pv.w));
  const uint32_t hotv = qfrom(mine, level & 3u);
  return level < kHotPath ? hotv : st->path_deep[level - kHotPath];
}

// mcts.rs:439-454 on a quad: lane q evaluates columns 2q and 2q + 1; sums run left to right exactly
// as the scalar version.  Every lane returns the full out[7].
C4_DEV void q_apply_temperature(const float* p, float t, float* out, uint32_t q) {
  bool all_eq = true;
  for (int i = 0; i < 7; i++) all_eq = all_eq && (p[i] == p[0]);
  if (t == 1.0f || all_eq) {
    for (int i = 0; i < 7; i++) out[i] = p[i];
    return;
  }
  if (t == 0.0f) {
    float mx = __uint_as_float(0xff800000u), s = 0.0f;
    for (int i = 0; i < 7; i++) mx = c4::rust_max(mx, p[i]);
    for (int i = 0; i < 7; i++) { out[i] = (p[i] == mx) ? 1.0f : 0.0f; s = s + out[i]; }
    for (int i = 0; i < 7; i++) out[i] = out[i] / s;
    return;
  }
  float m0 = p[0], m1 = p[1];
  for (int k = 1; k < 4; k++) { m0 = (q == (uint32_t)k) ? p[2 * k] : m0; m1 = (q == (uint32_t)k) ? p[k < 3 ? 2 * k + 1 : 6] : m1; }
  const float pl0 = c4::c4_logf(m0) / t, pl1 = c4::c4_logf(m1) / t;
  const float e0 = c4::c4_expf(pl0), e1 = (q < 3) ? c4::c4_expf(pl1) : 0.0f;
  // s = 0 + e[0] + ... + e[6] in order (lane 3 contributes its first element only)
  float tt = (0.0f + e0) + e1;
  float s = qb<0>(tt);
  tt = (s + e0) + e1;
  s = qb<1>(tt);
  tt = (s + e0) + e1;
  s = qb<2>(tt);
  tt = s + e0;
  s = qb<3>(tt);
  const float lse = c4::c4_logf(s);
  float v0 = c4::c4_expf(pl0 - lse), v1 = c4::c4_expf(pl1 - lse);
  v0 = v0 < 0.0f ? 0.0f : v0; v0 = v0 > 1.0f ? 1.0f : v0;
  v1 = v1 < 0.0f ? 0.0f : v1; v1 = v1 > 1.0f ? 1.0f : v1;
  q_gather7(v0, v1, out);
}

// leaf -> evaluator input row (c4r.rs:378-392): the 84-bit string value | opp << 42 cut into 8-element
// chunks; lane q expands chunks q, q + 4 and q + 8 (chunk 10 is half a chunk) into bf16 0/1
typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
template <typename PlaneT>
C4_DEV void q_encode_leaf(void* planes, uint32_t g, uint64_t mask, uint64_t value, uint32_t q);
template <>
C4_DEV void q_encode_leaf<float>(void* planes, uint32_t g, uint64_t mask, uint64_t value, uint32_t q) {
  for (uint32_t e = q; e < C4_PLANES_LEN; e += 4) store_plane<float>(planes, (size_t)g * C4_PLANES_LEN + e, c4::plane_bit(mask, value, e));
}
template <>
C4_DEV void q_encode_leaf<uint16_t>(void* planes, uint32_t g, uint64_t mask, uint64_t value, uint32_t q) {
  const uint64_t opp = mask & ~value;
  const uint64_t lo = value | (opp << 42);          // elements 0..63
  const uint32_t hi = (uint32_t)(opp >> 22);        // elements 64..83
  uint16_t* row = (uint16_t*)planes + (size_t)g * C4_PLANES_LEN;   // rows are 168 bytes apart: 8-byte aligned
  const uint4 a = bf16_bits8((uint32_t)(lo >> (8u * q)) & 0xFFu);
  const uint4 b = bf16_bits8((uint32_t)(lo >> (8u * q + 32u)) & 0xFFu);
  *reinterpret_cast<u32x4_a8*>(row + 8 * q) = u32x4_a8{a.x, a.y, a.z, a.w};
  *reinterpret_cast<u32x4_a8*>(row + 32 + 8 * q) = u32x4_a8{b.x, b.y, b.z, b.w};
  if (q < 3) {
    const uint4 c = bf16_bits8((hi >> (8u * q)) & 0xFFu);
    if (q < 2) *reinterpret_cast<u32x4_a8*>(row + 64 + 8 * q) = u32x4_a8{c.x, c.y, c.z, c.w};
    else *reinterpret_cast<uint2*>(row + 80) = make_uint2(c.x, c.y);
  }
}

// Evaluation cache on a quad.  Entry = 16 dwords, lane q owns dwords 4q..4q+3 (one 16-byte access per lane):
// [0..1] mask, [2..3] value | [4..7] logits 0..3 | [8..10] logits 4..6, [11] q_penalty | [12] q_no_penalty, [13] seal, 0, 0.
// The seal makes the XOR of all 16 dwords a constant: an empty or torn entry never validates.
C4_DEV void q_cache_store(uint4* cache, uint32_t cache_mask, uint64_t mask, uint64_t value, float l0, float l1, float q_pen,
                          float q_nopen, uint32_t q) {
  float lg[7];
  q_gather7(l0, l1, lg);
  uint4 w;
  if (q == 0) w = make_uint4((uint32_t)mask, (uint32_t)(mask >> 32), (uint32_t)value, (uint32_t)(value >> 32));
  else if (q == 1) w = make_uint4(__float_as_uint(lg[0]), __float_as_uint(lg[1]), __float_as_uint(lg[2]), __float_as_uint(lg[3]));
  else if (q == 2) w = make_uint4(__float_as_uint(lg[4]), __float_as_uint(lg[5]), __float_as_uint(lg[6]), __float_as_uint(q_pen));
  else w = make_uint4(__float_as_uint(q_nopen), 0u, 0u, 0u);
  uint32_t x = w.x ^ w.y ^ w.z ^ w.w;
  x ^= grp_xchg<0>(x); x ^= grp_xchg<1>(x);
  if (q == 3) w.y = x ^ kCacheMagic;                                   // the 16 dwords now XOR to the constant
  cache[(size_t)cache_index(mask, value, cache_mask) * 4 + q] = w;
}
C4_DEV bool q_cache_lookup(const uint4* cache, uint32_t cache_mask, uint64_t mask, uint64_t value, float& l0, float& l1,
                           float& q_pen, float& q_nopen, uint32_t q, int qbase) {
  const uint4 w = cache[(size_t)cache_index(mask, value, cache_mask) * 4 + q];
  uint32_t x = w.x ^ w.y ^ w.z ^ w.w;
  x ^= grp_xchg<0>(x); x ^= grp_xchg<1>(x);
  const bool kok = q != 0 || (w.x == (uint32_t)mask && w.y == (uint32_t)(mask >> 32) && w.z == (uint32_t)value && w.w == (uint32_t)(value >> 32));
  const bool hit = x == kCacheMagic && ((__ballot(kok) >> qbase) & 0xFull) == 0xFull;
  if (hit) {
    // logits 0..3 live in lane 1, 4..6 in lane 2: lane q wants logits 2q and 2q + 1
    const uint32_t a0 = qb<1>(w.x), a1 = qb<1>(w.y), a2 = qb<1>(w.z), a3 = qb<1>(w.w);
    const uint32_t b0 = qb<2>(w.x), b1 = qb<2>(w.y), b2 = qb<2>(w.z), b3 = qb<2>(w.w);
    l0 = __uint_as_float(q == 0 ? a0 : (q == 1 ? a2 : (q == 2 ? b0 : b2)));
    l1 = __uint_as_float(q == 0 ? a1 : (q == 1 ? a3 : b1));
    q_pen = __uint_as_float(b3);
    q_nopen = __uint_as_float(qb<3>(w.x));
  }
  return hit;
}

// select_new_leaf (mcts.rs:160-183) for one game on its quad.  Returns 0 or C4_ERR_NAN_IN_TREE.
C4_DEV uint32_t q_select_leaf(const Params& p, const Block* blocks, Slot* st, uint64_t rmask, uint64_t rvalue, uint32_t root_block,
                              uint32_t root_ref, uint32_t root_n, float c_exploration, uint32_t q, int qbase,
                              uint64_t& leaf_mask, uint64_t& leaf_value, uint32_t& depth, uint32_t& leaf_ref,
                              uint4& pv, uint32_t& levels) {
  uint64_t m = rmask, v = rvalue;
  uint32_t blk = root_block, d = 0, last_ref = root_ref;
  float ln_np = ln_visits(p, root_n);                     // ln(parent visits) of the level being scored
  uint32_t nan_seen = 0;
  pv.x = (q == 0) ? root_ref : pv.x;                      // level 0 of the path = the root's own entry
  while (blk != 0 && d + 1 < kMaxPath) {
    const uint4* bl = reinterpret_cast<const uint4*>(blocks + blk);
    const uint4 e0 = bl[2 * q], e1 = bl[2 * q + 1];       // entries 2q, 2q+1 (lane 3: entry 6, tail)
    // off the dependent chain: ln of this lane's children's visit counts (the next level's parent term if one of them wins)
    const float ln0 = ln_visits(p, e0.x), ln1 = ln_visits(p, q < 3 ? e1.x : 0u);
    const uint32_t legal = c4::legal_mask(m);
    const bool ok0 = (legal >> (2u * q)) & 1u, ok1 = q < 3 && ((legal >> (2u * q + 1u)) & 1u);
    // uct_value (mcts.rs:359-388) for both, without a branch
    const float nf0 = (float)e0.x + 1.0f, nf1 = (float)e1.x + 1.0f;
    const float qv0 = __uint_as_float(e0.y) / nf0, qv1 = __uint_as_float(e1.y) / nf1;
    float ex0 = ln_np / nf0, ex1 = ln_np / nf1;
    ex0 = __builtin_sqrtf(ex0); ex1 = __builtin_sqrtf(ex1);
    ex0 = ex0 * (__uint_as_float(e0.w) + 1e-8f); ex1 = ex1 * (__uint_as_float(e1.w) + 1e-8f);
    const float cx0 = c_exploration * ex0, cx1 = c_exploration * ex1;
    const float s0 = -qv0 + cx0, s1 = -qv1 + cx1;
    // a NaN among two or more candidates panics (utils.rs:12): reported after the walk
    nan_seen |= ((ok0 && (s0 != s0)) || (ok1 && (s1 != s1))) ? (uint32_t)(__popc(legal) >= 2) : 0u;
    // argmax = maximum of (score mapped monotonically to unsigned, column + 1): the LAST maximum wins
    // (mcts.rs:165-173); the winner's ln travels with the key
    const uint32_t b0 = __float_as_uint(s0 + 0.0f), b1 = __float_as_uint(s1 + 0.0f);
    const uint32_t o0 = b0 ^ ((uint32_t)((int32_t)b0 >> 31) | 0x80000000u), o1 = b1 ^ ((uint32_t)((int32_t)b1 >> 31) | 0x80000000u);
    const unsigned long long k0 = ok0 ? (((unsigned long long)o0 << 32) | (2u * q + 1u)) : 0ull;
    const unsigned long long k1 = ok1 ? (((unsigned long long)o1 << 32) | (2u * q + 2u)) : 0ull;
    unsigned long long key = k1 > k0 ? k1 : k0;
    float kln = k1 > k0 ? ln1 : ln0;
#define C4_QKEY_STEP(K)                                                                                             \
    {                                                                                                              \
      const unsigned long long o = ((unsigned long long)grp_xchg<K>((uint32_t)(key >> 32)) << 32) | grp_xchg<K>((uint32_t)key); \
      const float oln = grp_xchg<K>(kln);                                                                          \
      const bool take = o > key;                                                                                   \
      key = take ? o : key; kln = take ? oln : kln;                                                                \
    }
    C4_QKEY_STEP(0) C4_QKEY_STEP(1)
#undef C4_QKEY_STEP
    const uint32_t best = (uint32_t)key - 1u;
    ln_np = kln;
    // the winner's child link out of the tail (lane 3's second slot)
    const uint32_t w = best >> 1;
    const uint32_t tw = w == 0 ? e1.x : (w == 1 ? e1.y : (w == 2 ? e1.z : e1.w));
    const uint32_t next_blk = (qb<3>(tw) >> (16u * (best & 1u))) & 0xFFFFu;
    c4::make_move(m, v, best);
    d += 1;
    last_ref = (blk << 3) | best;
    {                                                      // the lanes keep the path: level d on lane d & 3
      const bool mine = q == (d & 3u);
      const uint32_t comp = d >> 2;
      pv.x = (mine && comp == 0) ? last_ref : pv.x;
      pv.y = (mine && comp == 1) ? last_ref : pv.y;
      pv.z = (mine && comp == 2) ? last_ref : pv.z;
      pv.w = (mine && comp == 3) ? last_ref : pv.w;
    }
    if (d >= kHotPath && q == 0) st->path_deep[d - kHotPath] = last_ref;   // rare: beyond what the lanes hold
    blk = next_blk;
    levels += 1;
  }
  leaf_mask = m; leaf_value = v; depth = d; leaf_ref = last_ref;
  return ((__ballot(nan_seen != 0) >> qbase) & 0xFull) ? (uint32_t)C4_ERR_NAN_IN_TREE : 0u;
}

// block 0 of a slot's arena for a fresh game: only the root's own entry (prior 1.0, mcts.rs:49) in column 0
C4_DEV void q_reset_arena(Block* blocks, uint32_t q) {
  uint4* b = reinterpret_cast<uint4*>(blocks);
  b[2 * q] = make_uint4(0, 0, 0, q == 0 ? __float_as_uint(1.0f) : 0u);
  b[2 * q + 1] = make_uint4(0, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------
// K2-K5 fused, four lanes per game: one MCTS simulation for every resident game (see c4_step_kernel
// for the phase-by-phase commentary; the phases and their order are identical).
// ------------------------------------------------------------------------------------------
template <typename PlaneT, bool NOISE, bool CACHE>
__global__ __launch_bounds__(64, (NOISE || CACHE) ? 1 : C4_STEP_WAVES) void c4_step_kernel_quad(Params p) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t q = lane & 3;
  const int qbase = (int)(lane & ~3u);
  const uint32_t g = blockIdx.x * 16u + (lane >> 2);

  uint32_t c_sims = 0, c_S = 0, c_K = 0, c_E = 0, c_moves = 0, c_done = 0, c_skipped = 0, c_samples = 0;   // this launch only
  uint32_t c_probes = 0, c_hits = 0;
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();   // 100 MHz device clock
  if (blockIdx.x >= p.n_waves) { timing_helper(p, lane); return; }

  const uint32_t gs = g < p.n_slots ? g : 0;
  Slot* st = p.slots + gs;
  // the game's state line: header piece q and path piece q; the evaluator's outputs travel with them
  const uint4 hdr = reinterpret_cast<const uint4*>(st)[q];
  const uint4 pth = reinterpret_cast<const uint4*>(st)[4 + q];
  const float nn_l0 = p.logprobs[(size_t)gs * 7 + 2 * q];
  const float nn_l1 = p.logprobs[(size_t)gs * 7 + (q < 3 ? 2 * q + 1 : 6)];
  const float nn_q = p.q[(size_t)gs * 2 + (q & 1)];
  const uint32_t state0 = qb<3>(hdr.x);
  bool active = (g < p.n_slots) && (slot_status(state0) == kActive);
  uint4 line_h = hdr, line_p = pth;     // what goes back to the slot at the end
  bool store_line = false;
  bool pre_need = false;                // move RNG precompute (end of the kernel)
  uint32_t pre_n_moves = 0;
  unsigned long long pre_game_id = 0;

  if (active) {
    Block* blocks = p.blocks + (size_t)g * p.blocks_per_slot;
    uint64_t rmask = qb64<0>(hdr.x, hdr.y), rvalue = qb64<0>(hdr.z, hdr.w);
    uint64_t leaf_mask = qb64<1>(hdr.x, hdr.y), leaf_value = qb64<1>(hdr.z, hdr.w);
    unsigned long long game_id = qb64<2>(hdr.x, hdr.y);
    uint32_t ordinal = qb<2>(hdr.z);
    const uint32_t arena0 = qb<3>(hdr.y);
    uint32_t root_ref = qb<3>(hdr.z);
    uint32_t rng_word = qb<3>(hdr.w);
    uint32_t depth = (state0 >> 8) & 0xFFu;
    uint32_t n_moves = (state0 >> 16) & 0xFFu;
    uint32_t term = (state0 >> 24) & 3u;          // terminal_state of the waiting leaf, computed when it was selected
    const uint32_t rng_for = state0 >> 26;
    uint32_t n_blocks = arena0 & 0xFFFFu;
    uint32_t root_block = arena0 >> 16;
    uint4 pv = pth;                               // path levels q, q + 4, q + 8, q + 12
    uint32_t leaf_ref = q_path_level(pv, st, depth);
    bool fresh = false;                           // the slot took a new game in this launch
    uint32_t err = 0;
    uint32_t root_n = 0;
    const uint32_t max_sims = p.max_sims;
    float cur_l0 = nn_l0, cur_l1 = nn_l1;         // this trip's evaluator outputs (a cached entry's after a hit)
    float cur_qp = qb<0>(nn_q), cur_qn = qb<1>(nn_q);

#pragma clang loop unroll(disable)
    for (uint32_t sim = 0; sim < max_sims; sim++) {
      // ---------------- on_received_policy: terminal value or expansion -------------------
      float v_pen, v_nopen;
      if (term) {
        c4::terminal_value(term, leaf_mask, p.c_ply_penalty, v_pen, v_nopen);  // NN output ignored (mcts.rs:92-98)
      } else {
        const uint32_t legal = c4::legal_mask(leaf_mask);
        const bool ok0 = (legal >> (2u * q)) & 1u, ok1 = q < 3 && ((legal >> (2u * q + 1u)) & 1u);
        const float ninf = __uint_as_float(0xff800000u);
        const float lg0 = ok0 ? cur_l0 : ninf, lg1 = ok1 ? cur_l1 : ninf;        // mask_policy, c4r.rs:272-286
        float mx = c4::rust_max(lg0, lg1);                                        // f32::max fold (NaN-ignoring)
        mx = c4::rust_max(mx, grp_xchg<0>(mx));
        mx = c4::rust_max(mx, grp_xchg<1>(mx));
        if (__builtin_isinf(mx)) err = C4_ERR_DEGENERATE_POLICY;                // mcts.rs:421-425
        const float ex0 = c4::c4_expf(lg0 - mx);
        const float ex1 = q < 3 ? c4::c4_expf(lg1 - mx) : 0.0f;
        const float sum = q_sum7_in_order(ex0, ex1);                              // left-to-right, mcts.rs:432
        float prior0 = ex0 / sum, prior1 = ex1 / sum;
        if (NOISE && p.dir_eps > 0.0f && depth == 0) {
          // extension: the root is expanded only now -> its children start with noisy priors
          float eta[7];
          c4::dirichlet_noise(game_id, n_moves, legal, p.dir_alpha, eta);
          float m0 = eta[0], m1 = eta[1];
          for (int k = 1; k < 4; k++) { m0 = (q == (uint32_t)k) ? eta[2 * k] : m0; m1 = (q == (uint32_t)k) ? eta[k < 3 ? 2 * k + 1 : 6] : m1; }
          if (ok0) { const float keep = (1.0f - p.dir_eps) * prior0; const float add = p.dir_eps * m0; prior0 = keep + add; }
          if (ok1) { const float keep = (1.0f - p.dir_eps) * prior1; const float add = p.dir_eps * m1; prior1 = keep + add; }
        }
        const uint32_t nb = n_blocks;
        if (nb >= p.blocks_per_slot) err = err ? err : C4_ERR_ARENA_OVERFLOW;
        if (!err) {
          // Node::new (mcts.rs:345-355) for the 7 children; lane 3's second slot is the tail (no links yet)
          uint4* nbp = reinterpret_cast<uint4*>(blocks + nb);
          nbp[2 * q] = make_uint4(0u, 0u, 0u, __float_as_uint(prior0));
          nbp[2 * q + 1] = q < 3 ? make_uint4(0u, 0u, 0u, __float_as_uint(prior1)) : make_uint4(0u, 0u, 0u, legal << 16);
          if (q == 0) blocks[leaf_ref >> 3].t.child[leaf_ref & 7] = (uint16_t)nb;   // leaf.children = Some(..)
          if (depth == 0) root_block = nb;
          n_blocks = nb + 1;
          c_E += 1;
        }
        v_pen = cur_qp;
        v_nopen = cur_qn;
        if (CACHE && sim == 0)     // extension: remember what the evaluator said about this position
          q_cache_store(reinterpret_cast<uint4*>(p.cache), p.cache_mask, leaf_mask, leaf_value, nn_l0, nn_l1, cur_qp, cur_qn, q);
      }
      if (err) break;

      // ---------------- backpropagate_value: level d on lane d & 3 ----------
      root_n = 0;
      for (uint32_t d = q; d <= depth; d += 4) {
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
      root_n = qb<0>(root_n);
      c_sims += 1;
      c_K += depth + 1;
      // stores above are read back below through other lanes of THIS wavefront: program order suffices
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

      // ---------------- gate: self_play.rs:283-308 ------------------------------------------
      bool finished = false;
      if (root_n >= p.n_iter && !(p.flags & C4_FLAG_NO_MOVES)) {
        const size_t rec0 = (size_t)ordinal * C4_MAX_SAMPLES_PER_GAME;
        uint32_t rterm = c4::terminal_state(rmask, rvalue);  // non-zero only for a terminal START position
        uint32_t retained = p.n_iter;
        if (!rterm) {
          // root_policy (mcts.rs:396-412): child visit counts / their sum
          const uint4* rb = reinterpret_cast<const uint4*>(blocks + root_block);
          const uint4 re0 = rb[2 * q], re1 = rb[2 * q + 1];
          float w[7];
          q_gather7((float)re0.x, q < 3 ? (float)re1.x : 0.0f, w);
          float csum = 0.0f;
          for (int i = 0; i < 7; i++) csum = csum + w[i];
          float pol[7];
          for (int i = 0; i < 7; i++) pol[i] = (csum == 0.0f) ? (1.0f / 7.0f) : (w[i] / csum);
          // make_random_move (mcts.rs:214-222)
          const float temperature = c4::temperature_for_ply((uint32_t)__popcll(rmask));
          float tp[7];
          q_apply_temperature(pol, temperature, tp, q);
          const uint64_t seed = game_id * (uint64_t)(42 + n_moves);
          // the word was normally computed in an earlier, uncontended step (end of this kernel)
          const uint32_t u32 = (!fresh && rng_for == n_moves + 1) ? rng_word : c4::rng_first_u32_group(seed, q, qbase);
          const int col = c4::weighted_index(tp, u32);
          if (col < 0) {
            err = C4_ERR_DEGENERATE_POLICY;
          } else if (!((c4::legal_mask(rmask) >> col) & 1u)) {
            err = C4_ERR_ILLEGAL_MOVE;                                        // mcts.rs:196-200 expect()
          } else {
            // make_move (mcts.rs:187-206): record (root position, untempered policy), re-root.  The
            // 64-byte record is one 16-byte store per lane; its q fields are written when the game ends.
            uint4 rw;
            if (q == 0) rw = make_uint4((uint32_t)game_id, (uint32_t)(game_id >> 32), (uint32_t)rmask, (uint32_t)(rmask >> 32));
            else if (q == 1) rw = make_uint4((uint32_t)rvalue, (uint32_t)(rvalue >> 32), __float_as_uint(pol[0]), __float_as_uint(pol[1]));
            else if (q == 2) rw = make_uint4(__float_as_uint(pol[2]), __float_as_uint(pol[3]), __float_as_uint(pol[4]), __float_as_uint(pol[5]));
            else rw = make_uint4(__float_as_uint(pol[6]), 0u, 0u, n_moves);
            reinterpret_cast<uint4*>(p.samples + rec0 + n_moves)[q] = rw;
            retained = q_elem(re0.x, re1.x, (uint32_t)col);
            const uint32_t wi = (uint32_t)col >> 1;
            const uint32_t tw = wi == 0 ? re1.x : (wi == 1 ? re1.y : (wi == 2 ? re1.z : re1.w));
            const uint32_t child_blk = (qb<3>(tw) >> (16u * ((uint32_t)col & 1u))) & 0xFFFFu;
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
              float m0 = eta[0], m1 = eta[1];
              for (int k = 1; k < 4; k++) { m0 = (q == (uint32_t)k) ? eta[2 * k] : m0; m1 = (q == (uint32_t)k) ? eta[k < 3 ? 2 * k + 1 : 6] : m1; }
              if ((nlegal >> (2u * q)) & 1u) {
                Entry* ce = &blocks[root_block].e[2 * q];
                const float keep = (1.0f - p.dir_eps) * ce->prior;
                const float add = p.dir_eps * m0;
                ce->prior = keep + add;
              }
              if (q < 3 && ((nlegal >> (2u * q + 1u)) & 1u)) {
                Entry* ce = &blocks[root_block].e[2 * q + 1];
                const float keep = (1.0f - p.dir_eps) * ce->prior;
                const float add = p.dir_eps * m1;
                ce->prior = keep + add;
              }
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
          }
        }
        if (!err && rterm) {
          // Game over (self_play.rs:302-308); see c4_step_kernel for the skipped-simulation accounting
          c_skipped += (p.n_iter > retained) ? (p.n_iter - retained) : 0;
          float tq_pen, tq_nopen;
          c4::terminal_value(rterm, rmask, p.c_ply_penalty, tq_pen, tq_nopen);
          // to_result (mcts.rs:271-313): sample i gets +q iff (M - i) is even
          for (uint32_t i = q; i < n_moves; i += 4) {
            const bool neg = ((n_moves - i) & 1u) != 0;
            p.samples[rec0 + i].q_penalty = neg ? -tq_pen : tq_pen;
            p.samples[rec0 + i].q_no_penalty = neg ? -tq_nopen : tq_nopen;
          }
          const float u7 = 1.0f / 7.0f;                                       // UNIFORM_POLICY, mcts.rs:45
          uint4 tw4;
          if (q == 0) tw4 = make_uint4((uint32_t)game_id, (uint32_t)(game_id >> 32), (uint32_t)rmask, (uint32_t)(rmask >> 32));
          else if (q == 1) tw4 = make_uint4((uint32_t)rvalue, (uint32_t)(rvalue >> 32), __float_as_uint(u7), __float_as_uint(u7));
          else if (q == 2) tw4 = make_uint4(__float_as_uint(u7), __float_as_uint(u7), __float_as_uint(u7), __float_as_uint(u7));
          else tw4 = make_uint4(__float_as_uint(u7), __float_as_uint(tq_pen), __float_as_uint(tq_nopen), n_moves | (1u << 16));
          reinterpret_cast<uint4*>(p.samples + rec0 + n_moves)[q] = tw4;
          if (q == 0) p.sample_counts[ordinal] = n_moves + 1;
          c_done += 1;
          c_samples += n_moves + 1;
          finished = true;
        }
      }
      if (err) break;

      if (finished) {
        // replace the finished game by the next one of the request list (keeps the batch full)
        unsigned long long next = 0;
        if (q == 0) {
          atomicAdd(&p.glob->games_done, 1ull);
          next = atomicAdd(&p.glob->queue_head, 1ull);
        }
        next = qb64<0>((uint32_t)next, (uint32_t)(next >> 32));
        if (next < p.n_games) {
          q_reset_arena(blocks, q);                       // MctsGame::new_from_pos (mcts.rs:48-56); the state goes out with the final store
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
          rmask = p.start_mask ? p.start_mask[next] : 0ull;
          rvalue = p.start_value ? p.start_value[next] : 0ull;
          root_ref = 0; root_block = 0; root_n = 0; n_blocks = 1; n_moves = 0;
          ordinal = (uint32_t)next;
          game_id = p.reqs[next].game_id;
          fresh = true;
        } else {
          active = false;
          if (q == 0) { st->state = kIdle; st->ordinal = 0xFFFFFFFFu; }
          break;
        }
      }
      // ---------------- select_new_leaf (mcts.rs:160-183) -------------------------------
      err = q_select_leaf(p, blocks, st, rmask, rvalue, root_block, root_ref, root_n, p.c_exploration, q, qbase,
                          leaf_mask, leaf_value, depth, leaf_ref, pv, c_S);
      if (err) break;
      // terminal_state of the new leaf (kept for the simulation that consumes it)
      term = depth == 0 ? c4::terminal_state(leaf_mask, leaf_value) : c4::terminal_after_move(leaf_mask, leaf_value);
      if (sim + 1 < max_sims) {
        bool again = term != 0;
        if (CACHE && !again) {
          c_probes += 1;
          again = q_cache_lookup(reinterpret_cast<const uint4*>(p.cache), p.cache_mask, leaf_mask, leaf_value, cur_l0, cur_l1, cur_qp, cur_qn, q, qbase);
          c_hits += again ? 1 : 0;
        }
        if (again) {
          if (depth >= kHotPath) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // deeper levels are re-read from the slot's second line
          continue;
        }
      }
      break;
    }

    if (err) {
      if (q == 0) raise_error(p, st, g, err);
    } else if (active) {
      if (q == 0) publish_leaf_model(p, g, ordinal, leaf_mask);
      pre_need = fresh || (rng_for != n_moves + 1);   // after a move / refill the stored word is stale
      pre_n_moves = n_moves;
      pre_game_id = game_id;
      q_encode_leaf<PlaneT>(p.planes, g, leaf_mask, leaf_value, q);
      // the game's state goes back: header piece q, path piece q
      line_p = pv;
      if (q == 0) line_h = make_uint4((uint32_t)rmask, (uint32_t)(rmask >> 32), (uint32_t)rvalue, (uint32_t)(rvalue >> 32));
      else if (q == 1) line_h = make_uint4((uint32_t)leaf_mask, (uint32_t)(leaf_mask >> 32), (uint32_t)leaf_value, (uint32_t)(leaf_value >> 32));
      else if (q == 2) line_h = make_uint4((uint32_t)game_id, (uint32_t)(game_id >> 32), ordinal, root_n);
      else line_h = make_uint4(slot_state(kActive, depth, n_moves, term, fresh ? 0u : rng_for), n_blocks | (root_block << 16), root_ref, rng_word);
      store_line = true;
    }
  }

  // ---------------- per-wavefront counters: lane q of a game adds counters q and q + 4 ----------
  {
    const uint32_t a0 = q == 0 ? c_sims : (q == 1 ? c_S : (q == 2 ? c_K : c_E));
    const uint32_t a1 = q == 0 ? c_moves : (q == 1 ? c_done : (q == 2 ? c_skipped : c_samples));
    if (a0) atomicAdd(&p.wave_ctr[(size_t)blockIdx.x * CTR_N + q], (unsigned long long)a0);
    if (a1) atomicAdd(&p.wave_ctr[(size_t)blockIdx.x * CTR_N + 4 + q], (unsigned long long)a1);
    const uint32_t a2 = q == 0 ? c_probes : (q == 1 ? c_hits : 0u);
    if (CACHE && a2) atomicAdd(&p.wave_ctr[(size_t)blockIdx.x * CTR_N + CTR_PROBES + q], (unsigned long long)a2);
  }
  // ---------------- move RNG, off the critical path (see c4_step_kernel) -----------------------
  if (__ballot(c_moves != 0) == 0ull && pre_need) {
    const uint32_t w = c4::rng_first_u32_group(pre_game_id * (uint64_t)(42 + pre_n_moves), q, qbase);
    if (q == 3) { line_h.w = w; line_h.x = (line_h.x & 0x03FFFFFFu) | ((pre_n_moves + 1u) << 26); }
  }
  if (store_line) {
    reinterpret_cast<uint4*>(st)[q] = line_h;
    reinterpret_cast<uint4*>(st)[4 + q] = line_p;
  }
  if (lane == 0 && p.seq) {
    unsigned long long* my = p.stamps + ((size_t)(p.seq & 1) * p.n_waves + blockIdx.x) * 2;
    my[0] = t_start;
    my[1] = __builtin_amdgcn_s_memrealtime();
  }
}
