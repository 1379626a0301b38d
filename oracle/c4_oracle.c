/*
 * c4_oracle.c -- CPU ORACLE (test infrastructure, not product code).  See c4_oracle.h.
 *
 * Plain-C restatement of the reference self-play hot path.  Every function cites the
 * reference lines it follows (paths relative to the reference repo root).
 * Build: see oracle/Makefile (-O2 -ffp-contract=off -mfma: no implicit FMA anywhere;
 * the one explicit fma() below is the one glibc's FMA ifunc variant performs).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE   /* pthread_setaffinity_np / CPU_SET for the timing runs' thread pinning */
#endif
#include "c4_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * Game rules -- rust/src/c4r.rs
 * ---------------------------------------------------------------------------------------- */

/* c4r.rs:119-122  bit index = row * N_COLS + col, row 0 = bottom */
static inline uint64_t idx_mask(int row, int col) { return (uint64_t)1 << (row * C4O_N_COLS + col); }

/* c4r.rs:165-224  WIN_MASKS in the reference's generation order:
 * horizontal (row-major), vertical (col-major), diagonal up-right from rows 0..2, diagonal
 * down-right from rows 3..5. */
static uint64_t g_win_masks[69];
static int g_win_masks_ready = 0;

static void build_win_masks(void) {
  int index = 0;
  for (int row = 0; row < C4O_N_ROWS; row++)
    for (int col = 0; col <= C4O_N_COLS - 4; col++)
      g_win_masks[index++] = idx_mask(row, col) | idx_mask(row, col + 1) | idx_mask(row, col + 2) | idx_mask(row, col + 3);
  for (int col = 0; col < C4O_N_COLS; col++)
    for (int row = 0; row <= C4O_N_ROWS - 4; row++)
      g_win_masks[index++] = idx_mask(row, col) | idx_mask(row + 1, col) | idx_mask(row + 2, col) | idx_mask(row + 3, col);
  for (int row = 0; row <= C4O_N_ROWS - 4; row++)
    for (int col = 0; col <= C4O_N_COLS - 4; col++)
      g_win_masks[index++] = idx_mask(row, col) | idx_mask(row + 1, col + 1) | idx_mask(row + 2, col + 2) | idx_mask(row + 3, col + 3);
  for (int row = 3; row < C4O_N_ROWS; row++)
    for (int col = 0; col <= C4O_N_COLS - 4; col++)
      g_win_masks[index++] = idx_mask(row, col) | idx_mask(row - 1, col + 1) | idx_mask(row - 2, col + 2) | idx_mask(row - 3, col + 3);
  g_win_masks_ready = (index == 69);
}

__attribute__((constructor)) static void c4o_init(void) { build_win_masks(); }

uint64_t c4o_win_mask(int i) { return (i >= 0 && i < 69) ? g_win_masks[i] : 0; }

/* c4r.rs:58-72  lowest empty row of `col` gets the mover's piece, then the colours are
 * inverted (c4r.rs:125-129) so `value` is again "pieces of the side to move". */
int c4o_make_move(const c4o_pos* p, int col, c4o_pos* out) {
  if (col < 0 || col >= C4O_N_COLS) return 0; /* reference bound is col > 7 (off by one); callers pass 0..6 */
  for (int row = 0; row < C4O_N_ROWS; row++) {
    uint64_t idx = idx_mask(row, col);
    if ((idx & p->mask) == 0) {
      uint64_t mask = p->mask | idx;
      uint64_t value = p->value | idx;
      value = ~value & mask; /* invert() */
      out->mask = mask;
      out->value = value;
      return 1;
    }
  }
  return 0;
}

/* c4r.rs:76-91 */
int c4o_get(const c4o_pos* p, int row, int col) {
  if (col < 0 || col >= C4O_N_COLS || row < 0 || row >= C4O_N_ROWS) return -1;
  uint64_t idx = idx_mask(row, col);
  if ((p->mask & idx) == 0) return -1;
  return (p->value & idx) ? 1 : 0;
}

/* c4r.rs:95-97 */
int c4o_ply(const c4o_pos* p) { return __builtin_popcountll(p->mask); }

/* c4r.rs:241-249 */
static int is_terminal_for_player(uint64_t mask, uint64_t value) {
  uint64_t player_tokens = mask & value;
  for (int i = 0; i < 69; i++)
    if (__builtin_popcountll(player_tokens & g_win_masks[i]) == 4) return 1;
  return 0;
}

/* c4r.rs:228-238  order: PlayerWin, OpponentWin, Draw */
int c4o_terminal_state(const c4o_pos* p) {
  if (is_terminal_for_player(p->mask, p->value)) return C4O_PLAYER_WIN;
  if (is_terminal_for_player(p->mask, ~p->value & p->mask)) return C4O_OPPONENT_WIN;
  if (c4o_ply(p) == C4O_N_COLS * C4O_N_ROWS) return C4O_DRAW;
  return C4O_NOT_TERMINAL;
}

/* c4r.rs:253-263 */
int c4o_terminal_value(const c4o_pos* p, float c_ply_penalty, float* q_pen, float* q_nopen) {
  float ply_penalty_magnitude = c_ply_penalty * (float)c4o_ply(p);
  int t = c4o_terminal_state(p);
  switch (t) {
    case C4O_PLAYER_WIN: *q_pen = 1.0f - ply_penalty_magnitude; *q_nopen = 1.0f; break;
    case C4O_OPPONENT_WIN: *q_pen = -1.0f + ply_penalty_magnitude; *q_nopen = -1.0f; break;
    case C4O_DRAW: *q_pen = 0.0f; *q_nopen = 0.0f; break;
    default: break;
  }
  return t;
}

/* c4r.rs:266-269 */
unsigned c4o_legal_mask(const c4o_pos* p) {
  unsigned m = 0;
  for (int col = 0; col < C4O_N_COLS; col++)
    if (c4o_get(p, C4O_N_ROWS - 1, col) < 0) m |= 1u << col;
  return m;
}

/* Test infrastructure: the rule functions above over n positions in one C call (tests/test_gpu_elementwise.py compares a
 * million positions; a Python loop over ctypes calls took minutes).  Each output is what the single-position function returns:
 * legal mask, terminal state + values (values written only for terminal positions, as c4o_terminal_value does), and the
 * position after `col[i]` (0, 0 where make_move returns None, c4r.rs:58-72). */
void c4o_pos_ops_batch(const uint64_t* mask, const uint64_t* value, const int32_t* col, uint64_t n, float c_ply_penalty,
                       uint64_t* out_mask, uint64_t* out_value, int32_t* out_legal, int32_t* out_term, float* out_q) {
  for (uint64_t i = 0; i < n; i++) {
    c4o_pos p = {mask[i], value[i]}, nx = {0, 0};
    out_legal[i] = (int32_t)c4o_legal_mask(&p);
    float a = 0.0f, b = 0.0f;
    out_term[i] = c4o_terminal_value(&p, c_ply_penalty, &a, &b);
    out_q[2 * i] = a; out_q[2 * i + 1] = b;
    if (!c4o_make_move(&p, col[i], &nx)) { nx.mask = 0; nx.value = 0; }
    out_mask[i] = nx.mask; out_value[i] = nx.value;
  }
}

/* Test infrastructure: the reference's `random_pos` proptest strategy (c4r.rs:610-629) for n positions -- from the empty
 * board play up to k < 60 random columns, skipping illegal ones, stopping at a terminal position; every prefix is emitted
 * (each is a reachable position).  The generator is a splitmix64 stream of `seed`: no claim about the reference's RNG, only
 * that every position is reachable by legal play. */
static uint64_t c4o_splitmix64(uint64_t* s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
void c4o_random_positions(uint64_t n, uint64_t seed, uint64_t* out_mask, uint64_t* out_value) {
  uint64_t s = seed, k = 0;
  while (k < n) {
    c4o_pos pos = {0, 0};
    int len = (int)(c4o_splitmix64(&s) % 60);
    for (int j = 0; j < len && k < n; j++) {
      if (c4o_terminal_state(&pos) != C4O_NOT_TERMINAL) break;
      int mov = (int)(c4o_splitmix64(&s) % 7);
      c4o_pos nx;
      if (((c4o_legal_mask(&pos) >> mov) & 1u) && c4o_make_move(&pos, mov, &nx)) pos = nx;
      out_mask[k] = pos.mask; out_value[k] = pos.value; k++;
    }
  }
}

/* c4r.rs:272-286 */
void c4o_mask_policy(const c4o_pos* p, float* logits) {
  unsigned legal = c4o_legal_mask(p);
  for (int c = 0; c < C4O_N_COLS; c++)
    if (!((legal >> c) & 1)) logits[c] = -INFINITY;
}

/* c4r.rs:289-299 */
void c4o_flip_h(const c4o_pos* p, c4o_pos* out) {
  c4o_pos r = {0, 0};
  for (int row = 0; row < C4O_N_ROWS; row++)
    for (int col = 0; col < C4O_N_COLS; col++) {
      int piece = c4o_get(p, row, col);
      if (piece >= 0) {
        uint64_t idx = idx_mask(row, C4O_N_COLS - 1 - col);
        r.mask |= idx;
        if (piece == 1) r.value |= idx;
      }
    }
  *out = r;
}

/* c4r.rs:378-392  plane 0 = side to move, plane 1 = opponent, row-major, row 0 = bottom */
void c4o_write_planes(const c4o_pos* p, float* buf) {
  for (int player = 0; player < 2; player++)
    for (int row = 0; row < C4O_N_ROWS; row++)
      for (int col = 0; col < C4O_N_COLS; col++) {
        int idx = player * 42 + row * C4O_N_COLS + col;
        int cell = c4o_get(p, row, col);
        buf[idx] = ((cell == 1 && player == 0) || (cell == 0 && player == 1)) ? 1.0f : 0.0f;
      }
}

/* c4r.rs:315-321 */
int c4o_from_moves(const int* cols, int n, c4o_pos* out) {
  c4o_pos pos = {0, 0};
  for (int i = 0; i < n; i++) {
    c4o_pos nx;
    if (!c4o_make_move(&pos, cols[i], &nx)) return 0;
    pos = nx;
  }
  *out = pos;
  return 1;
}

/* ------------------------------------------------------------------------------------------
 * libm restatement.  Rust's f32::exp / f32::ln (mcts.rs:379,430,451-453) lower to the
 * platform libm's expf/logf.  Third-party dependency: glibc 2.35 (the image's libm.so.6),
 * sysdeps/ieee754/flt-32/e_expf.c, e_logf.c, e_exp2f_data.c, e_logf_data.c.  The
 * constants below were checked against the bytes of /lib/x86_64-linux-gnu/libm.so.6.
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

static const uint64_t EXP2F_TAB[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51,
    0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
    0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
    0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
    0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
    0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};

float c4o_expf(float x) {
  const double InvLn2N = 0x1.71547652b82fep+5; /* N/ln2, N = 32 */
  const double SHIFT = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
  double xd = (double)x;
  uint32_t abstop = (f2u(x) >> 20) & 0x7ff;
  if (abstop >= (f2u(88.0f) >> 20)) {
    /* |x| >= 88 or x is nan */
    if (f2u(x) == f2u(-INFINITY)) return 0.0f;
    if (abstop >= (f2u(INFINITY) >> 20)) return x + x;
    if (x > 0x1.62e42ep6f) return INFINITY;  /* __math_oflowf(0) */
    if (x < -0x1.9fe368p6f) return 0.0f;     /* __math_uflowf(0) */
    if (x < -0x1.9d1d9ep6f) {                /* __math_may_uflowf(0) = 0x1.4p-75f * 0x1.4p-75f */
      volatile float t = 0x1.4p-75f;
      return t * 0x1.4p-75f;
    }
  }
  double z = InvLn2N * xd;
  double kd = z + SHIFT;
  uint64_t ki = d2u(kd);
  kd -= SHIFT;
  /* glibc computes r = z - kd; its FMA ifunc variant (selected on every FMA-capable x86)
   * contracts this with z = InvLn2N * xd into one fused op.  SURVEY Appendix A.2. */
  double r = __builtin_fma(InvLn2N, xd, -kd);
  uint64_t t = EXP2F_TAB[ki % 32];
  t += ki << (52 - 5);
  double s = u2d(t);
  double zz = C0 * r + C1;
  double r2 = r * r;
  double y = C2 * r + 1.0;
  y = zz * r2 + y;
  y = y * s;
  return (float)y;
}

static const struct { double invc, logc; } LOGF_TAB[16] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2},
};

float c4o_logf(float x) {
  const double Ln2 = 0x1.62e42fefa39efp-1;
  const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  uint32_t ix = f2u(x);
  if (ix == 0x3f800000) return 0.0f;
  if (ix - 0x00800000 >= 0x7f800000 - 0x00800000) {
    /* x < 0x1p-126 or inf or nan */
    if (ix * 2 == 0) return -INFINITY;                       /* __math_divzerof(1) */
    if (ix == 0x7f800000) return x;                          /* log(inf) == inf */
    if ((ix & 0x80000000) || ix * 2 >= 0xff000000) return (x - x) / 0.0f; /* __math_invalidf: NaN */
    ix = f2u(x * 0x1p23f); /* subnormal: normalise */
    ix -= 23u << 23;
  }
  uint32_t tmp = ix - 0x3f330000;
  int i = (tmp >> 19) % 16;
  int k = (int32_t)tmp >> 23; /* arithmetic shift */
  uint32_t iz = ix - (tmp & 0xff800000);
  double invc = LOGF_TAB[i].invc, logc = LOGF_TAB[i].logc;
  double z = (double)u2f(iz);
  double r = z * invc - 1.0;
  double y0 = logc + (double)k * Ln2;
  double r2 = r * r;
  double y = A1 * r + A2;
  y = A0 * r2 + y;
  y = y * r2 + (y0 + r);
  return (float)y;
}

static int same_f32(float a, float b) {
  if (a != a && b != b) return 1; /* NaN payloads are not part of the contract */
  return f2u(a) == f2u(b);
}

uint64_t c4o_sweep_expf(uint32_t lo, uint32_t hi, uint32_t stride, uint32_t* first_bad) {
  uint64_t bad = 0;
  if (stride == 0) stride = 1;
  for (uint64_t u = lo; u <= hi; u += stride) {
    float x = u2f((uint32_t)u);
    if (!same_f32(c4o_expf(x), expf(x))) {
      if (bad == 0 && first_bad) *first_bad = (uint32_t)u;
      bad++;
    }
  }
  return bad;
}

uint64_t c4o_sweep_logf(uint32_t lo, uint32_t hi, uint32_t stride, uint32_t* first_bad) {
  uint64_t bad = 0;
  if (stride == 0) stride = 1;
  for (uint64_t u = lo; u <= hi; u += stride) {
    float x = u2f((uint32_t)u);
    if (!same_f32(c4o_logf(x), logf(x))) {
      if (bad == 0 && first_bad) *first_bad = (uint32_t)u;
      bad++;
    }
  }
  return bad;
}

void c4o_host_expf(const float* x, float* y, size_t n) {
  for (size_t i = 0; i < n; i++) y[i] = expf(x[i]);
}
void c4o_host_logf(const float* x, float* y, size_t n) {
  for (size_t i = 0; i < n; i++) y[i] = logf(x[i]);
}

/* ------------------------------------------------------------------------------------------
 * Policy arithmetic -- rust/src/mcts.rs
 * ---------------------------------------------------------------------------------------- */

/* Rust f32::max: if one argument is NaN the other is returned. */
static inline float rust_f32_max(float a, float b) {
  if (a != a) return b;
  if (b != b) return a;
  return a > b ? a : b;
}

/* mcts.rs:416-434 */
int c4o_softmax7(const float* logits, float* out) {
  float max = -INFINITY;
  for (int i = 0; i < 7; i++) max = rust_f32_max(max, logits[i]);
  if (isinf(max)) return C4O_ERR_DEGENERATE_POLICY; /* reference panics */
  float exps[7];
  for (int i = 0; i < 7; i++) exps[i] = c4o_expf(logits[i] - max);
  float sum = 0.0f;
  for (int i = 0; i < 7; i++) sum = sum + exps[i];
  for (int i = 0; i < 7; i++) out[i] = exps[i] / sum;
  return C4O_OK;
}

/* mcts.rs:439-454 */
void c4o_apply_temperature(const float* policy, float t, float* out) {
  int all_eq = 1;
  for (int i = 0; i < 7; i++)
    if (!(policy[i] == policy[0])) all_eq = 0;
  if (t == 1.0f || all_eq) {
    for (int i = 0; i < 7; i++) out[i] = policy[i];
    return;
  } else if (t == 0.0f) {
    float max = -INFINITY;
    for (int i = 0; i < 7; i++) max = rust_f32_max(max, policy[i]);
    float ret[7], sum = 0.0f;
    for (int i = 0; i < 7; i++) ret[i] = (policy[i] == max) ? 1.0f : 0.0f;
    for (int i = 0; i < 7; i++) sum = sum + ret[i];
    for (int i = 0; i < 7; i++) out[i] = ret[i] / sum;
    return;
  }
  float pl[7], sum = 0.0f;
  for (int i = 0; i < 7; i++) pl[i] = c4o_logf(policy[i]) / t;
  for (int i = 0; i < 7; i++) sum = sum + c4o_expf(pl[i]);
  float lse = c4o_logf(sum);
  for (int i = 0; i < 7; i++) {
    float v = c4o_expf(pl[i] - lse);
    /* f32::clamp(0.0, 1.0): NaN stays NaN */
    if (v < 0.0f) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    out[i] = v;
  }
}

/* ------------------------------------------------------------------------------------------
 * RNG -- third-party crates rand 0.10.1, rand_core 0.10.1, chacha20 0.10.1
 * (rust/Cargo.lock:1585-1593, 1621-1622, 269-277).  Source not in /root/reference:
 * restated from the crates' published algorithm and pinned stage by stage by the crates' own
 * published unit-test vectors (see the header; tests/test_oracle_libm_rng.py).
 * Call site: mcts.rs:214-222.
 * ---------------------------------------------------------------------------------------- */

/* rand_core SeedableRng::seed_from_u64: PCG32 (XSH RR 64/32) stream fills the 32-byte seed */
void c4o_seed_from_u64(uint64_t state, uint8_t key[32]) {
  const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
  for (int i = 0; i < 8; i++) {
    state = state * MUL + INC;
    uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
    uint32_t rot = (uint32_t)(state >> 59);
    uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    key[4 * i + 0] = (uint8_t)x;
    key[4 * i + 1] = (uint8_t)(x >> 8);
    key[4 * i + 2] = (uint8_t)(x >> 16);
    key[4 * i + 3] = (uint8_t)(x >> 24);
  }
}

static inline uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
#define C4O_QR(a, b, c, d) \
  a += b; d ^= a; d = rotl32(d, 16); \
  c += d; b ^= c; b = rotl32(b, 12); \
  a += b; d ^= a; d = rotl32(d, 8);  \
  c += d; b ^= c; b = rotl32(b, 7);

/* ChaCha block, 64-bit counter in words 12-13, 64-bit stream id (0) in words 14-15 */
void c4o_chacha_block(const uint8_t key[32], uint64_t counter, int rounds, uint32_t out[16]) {
  uint32_t s[16], x[16];
  s[0] = 0x61707865; s[1] = 0x3320646e; s[2] = 0x79622d32; s[3] = 0x6b206574;
  for (int i = 0; i < 8; i++)
    s[4 + i] = (uint32_t)key[4 * i] | ((uint32_t)key[4 * i + 1] << 8) | ((uint32_t)key[4 * i + 2] << 16) | ((uint32_t)key[4 * i + 3] << 24);
  s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = 0; s[15] = 0;
  memcpy(x, s, sizeof x);
  for (int r = 0; r < rounds; r += 2) {
    C4O_QR(x[0], x[4], x[8], x[12]) C4O_QR(x[1], x[5], x[9], x[13])
    C4O_QR(x[2], x[6], x[10], x[14]) C4O_QR(x[3], x[7], x[11], x[15])
    C4O_QR(x[0], x[5], x[10], x[15]) C4O_QR(x[1], x[6], x[11], x[12])
    C4O_QR(x[2], x[7], x[8], x[13]) C4O_QR(x[3], x[4], x[9], x[14])
  }
  for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

/* StdRng (= ChaCha12) seeded from a u64, first 32-bit output = word 0 of block 0 */
uint32_t c4o_rng_first_u32(uint64_t seed) {
  uint8_t key[32];
  uint32_t blk[16];
  c4o_seed_from_u64(seed, key);
  c4o_chacha_block(key, 0, 12, blk);
  return blk[0];
}

/* rand::distr::weighted::WeightedIndex<f32>::new(w).sample(rng) given the rng's next_u32:
 *   cumulative[i] = w0+..+wi for i < 6, total = w0+..+w6 (f32, left to right);
 *   UniformFloat<f32>::new(0, total): scale = total, decremented while
 *   scale * (1 - eps) + 0 >= total;  sample: u01 = bits(0x3f800000 | u >> 9) - 1;
 *   x = u01 * scale + 0;  index = partition_point(cumulative, |w| w <= x). */
int c4o_weighted_index(const float* w, uint32_t u, int* out_idx) {
  float cum[6];
  float total = w[0];
  if (!(total >= 0.0f)) return C4O_ERR_DEGENERATE_POLICY; /* InvalidWeight */
  for (int i = 1; i < 7; i++) {
    if (!(w[i] >= 0.0f)) return C4O_ERR_DEGENERATE_POLICY;
    cum[i - 1] = total;
    total = total + w[i];
  }
  if (total == 0.0f) return C4O_ERR_DEGENERATE_POLICY;   /* InsufficientNonZero */
  if (!isfinite(total)) return C4O_ERR_DEGENERATE_POLICY; /* Uniform::new NonFinite */
  const float low = 0.0f, high = total;
  const float max_rand = 1.0f - 0x1p-23f; /* 1 - f32::EPSILON */
  float scale = high - low;
  for (;;) {
    float t = scale * max_rand;
    t = t + low;
    if (!(t >= high)) break;
    scale = u2f(f2u(scale) - 1);
  }
  float value1_2 = u2f(0x3f800000u | (u >> 9));
  float value0_1 = value1_2 - 1.0f;
  float x = value0_1 * scale;
  x = x + low;
  int idx = 0;
  while (idx < 6 && cum[idx] <= x) idx++; /* cumulative weights are non-decreasing */
  *out_idx = idx;
  return C4O_OK;
}

/* mcts.rs:214-222 without the tree update */
int c4o_sample_move(uint64_t game_id, int n_moves, const float* policy, float temperature, int* out_col) {
  uint64_t seed = game_id * (uint64_t)(C4O_N_ROWS * C4O_N_COLS + n_moves); /* wrapping (release build) */
  float tempered[7];
  c4o_apply_temperature(policy, temperature, tempered);
  return c4o_weighted_index(tempered, c4o_rng_first_u32(seed), out_col);
}

/* ------------------------------------------------------------------------------------------
 * Dirichlet root noise -- a BUILD EXTENSION named by BASELINE.json's north star; the reference has
 * no such code, so there is nothing to be faithful to: this is the
 * specification, and the HIP kernel must match it bit for bit.  Off unless epsilon > 0.
 *
 *   eta = Dir(alpha) over the legal columns of the root, from a private ChaCha12 stream keyed by
 *   PCG32-expand((game_id * (42 + n_moves)) ^ "Dirichle"), words consumed in order;
 *   prior'_c = (1 - eps) * prior_c + eps * eta_c for legal c (f32, separate roundings).
 *   Gamma(a): a == 1: -ln(u);  a > 1: Marsaglia-Tsang with polar-method normals;  a < 1:
 *   Gamma(a + 1) * exp(ln(u) / a).  u = ((next_u32 >> 8) + 0.5) * 2^-24 in (0, 1).
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint8_t key[32]; uint64_t counter; uint32_t buf[16]; int idx; } c4o_stream;

static void stream_init(c4o_stream* st, uint64_t seed) {
  c4o_seed_from_u64(seed, st->key);
  st->counter = 0;
  st->idx = 16;
}
static uint32_t stream_u32(c4o_stream* st) {
  if (st->idx == 16) { c4o_chacha_block(st->key, st->counter++, 12, st->buf); st->idx = 0; }
  return st->buf[st->idx++];
}
static float stream_u01(c4o_stream* st) { return ((float)(stream_u32(st) >> 8) + 0.5f) * 0x1p-24f; }

static float stream_normal(c4o_stream* st) { /* Marsaglia polar method, second variate discarded */
  for (;;) {
    float a = 2.0f * stream_u01(st) - 1.0f;
    float b = 2.0f * stream_u01(st) - 1.0f;
    float s = a * a;
    float bb = b * b;
    s = s + bb;
    if (s > 0.0f && s < 1.0f) {
      float t = -2.0f * c4o_logf(s);
      t = t / s;
      return a * sqrtf(t);
    }
  }
}

static float stream_gamma(c4o_stream* st, float alpha) {
  if (alpha == 1.0f) return -c4o_logf(stream_u01(st));
  float boost = 1.0f;
  float a = alpha;
  if (alpha < 1.0f) {
    float lu = c4o_logf(stream_u01(st));
    boost = c4o_expf(lu / alpha);
    a = alpha + 1.0f;
  }
  const float d = a - (1.0f / 3.0f);
  const float c = 1.0f / sqrtf(9.0f * d);
  for (;;) {
    float x = stream_normal(st);
    float v = 1.0f + c * x;
    if (v <= 0.0f) continue;
    v = v * v * v;
    float u = stream_u01(st);
    float lhs = c4o_logf(u);
    float x2 = x * x;
    float rhs = 0.5f * x2;
    rhs = rhs + d;
    float dv = d * v;
    rhs = rhs - dv;
    float dlv = d * c4o_logf(v);
    rhs = rhs + dlv;
    if (lhs < rhs) return (d * v) * boost;
  }
}

void c4o_dirichlet(uint64_t game_id, int n_moves, unsigned legal, float alpha, float* eta7) {
  c4o_stream st;
  stream_init(&st, (game_id * (uint64_t)(42 + n_moves)) ^ 0x4469726963686C65ull);
  float g[7], sum = 0.0f;
  for (int c = 0; c < 7; c++) {
    g[c] = ((legal >> c) & 1) ? stream_gamma(&st, alpha) : 0.0f;
    sum = sum + g[c];
  }
  /* every Gamma draw underflowed (tiny alpha): fall back to the uniform point of the simplex */
  const int degenerate = !(sum > 0.0f) || isinf(sum);
  const float uniform = 1.0f / (float)__builtin_popcount(legal & 0x7Fu);
  for (int c = 0; c < 7; c++) eta7[c] = ((legal >> c) & 1) ? (degenerate ? uniform : g[c] / sum) : 0.0f;
}

/* ------------------------------------------------------------------------------------------
 * `results.shuffle(&mut rng)` of PlayGamesResult::split_train_test (rust/src/pybridge.rs:110-116):
 * rand's SliceRandom::shuffle (third-party, rand 0.10.1, source not in /root/reference; restated from
 * the crate's published algorithm, rand 0.9 src/seq/slice.rs + seq/increasing_uniform.rs +
 * distr/uniform_int.rs, and pinned by the crate's own `value_stability_slice` vectors in
 * tests/test_oracle_libm_rng.py):
 *   shuffle = partial_shuffle(len) (nothing for len <= 1); partial_shuffle(amount): m = len - amount,
 *   for i in m..len { swap(i, chooser.next_index()) } -- Durstenfeld from the bottom index UP, the index
 *   for position i drawn from [0, i].  The chooser (IncreasingUniform, slices shorter than 2^32 - 1)
 *   draws ONE u32 for a whole run of positions: for the next bound n + 1 it takes the longest product
 *   (n+1)(n+2)...(n+k) that fits a u32, draws chunk = random_range(..product), and hands out
 *   chunk % (n+1), then chunk /= (n+1), ... the last position of the run gets what is left of chunk.
 *   random_range(..bound) on u32 = Canon's method with one bias-reducing retry: (hi, lo) = wide product
 *   of next_u32() and bound; if lo > bound.wrapping_neg() { hi += carry of lo + high word of a second
 *   next_u32() * bound }.
 * ---------------------------------------------------------------------------------------- */
static uint32_t range_u32_below(c4o_next_u32_fn next, void* ctx, uint32_t bound) {
  uint64_t m = (uint64_t)next(ctx) * bound;
  uint32_t result = (uint32_t)(m >> 32), lo = (uint32_t)m;
  if (lo > (uint32_t)(0u - bound)) {
    uint32_t new_hi = (uint32_t)(((uint64_t)next(ctx) * bound) >> 32);
    if ((uint64_t)lo + new_hi > 0xFFFFFFFFull) result += 1;
  }
  return result;
}

int c4o_partial_shuffle_with(uint64_t len, uint64_t amount, c4o_next_u32_fn next, void* ctx, uint32_t* items) {
  if (len >= 0xFFFFFFFFull) return 1; /* the crate switches to per-index random_range on usize there; not restated */
  const uint64_t m = amount >= len ? 0 : len - amount;
  uint32_t n = (uint32_t)m, chunk = 0;
  unsigned chunk_remaining = n == 0 ? 1 : 0;
  for (uint64_t i = m; i < len; i++) {
    const uint32_t next_n = n + 1;
    unsigned next_remaining;
    if (chunk_remaining >= 1) {
      next_remaining = chunk_remaining - 1;
    } else {
      uint32_t product = next_n, current = next_n + 1;
      for (;;) {
        uint64_t p = (uint64_t)product * current;
        if (p > 0xFFFFFFFFull) break;
        product = (uint32_t)p;
        current++;
      }
      chunk = range_u32_below(next, ctx, product);
      next_remaining = (current - next_n) - 1;
    }
    uint32_t index;
    if (next_remaining == 0) {
      index = chunk;
    } else {
      index = chunk % next_n;
      chunk /= next_n;
    }
    chunk_remaining = next_remaining;
    n = next_n;
    uint32_t t = items[i]; items[i] = items[index]; items[index] = t;
  }
  return 0;
}

static uint32_t stream_next_cb(void* ctx) { return stream_u32((c4o_stream*)ctx); }

/* order[i] = index of the game that results.shuffle(&mut StdRng::seed_from_u64(seed)) leaves at position i */
int c4o_shuffle_games(uint64_t seed, uint64_t n_games, uint32_t* order) {
  for (uint64_t i = 0; i < n_games; i++) order[i] = (uint32_t)i;
  if (n_games <= 1) return 0;
  c4o_stream st;
  stream_init(&st, seed);
  return c4o_partial_shuffle_with(n_games, n_games, stream_next_cb, &st, order);
}

/* ------------------------------------------------------------------------------------------
 * MCTS game -- rust/src/mcts.rs:27-413.  Rc<RefCell<Node>> graph restated as an arena of
 * nodes addressed by index; a dead Weak parent link is parent == -1.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  c4o_pos pos;
  int32_t parent;
  int32_t child[7]; /* -1 = None (illegal move) */
  int32_t has_children;
  uint64_t visit_count;
  float q_sum_penalty;
  float q_sum_no_penalty;
  float initial_policy_value;
} c4o_node;

struct c4o_game {
  uint64_t game_id, player0_id, player1_id;
  c4o_node* nodes;
  int n_nodes, cap;
  int root, leaf;
  int n_moves;
  c4o_pos mv_pos[C4O_MAX_MOVES + 1];
  float mv_policy[C4O_MAX_MOVES + 1][7];
  int mv_col[C4O_MAX_MOVES + 1];
  int error;
  float dir_alpha, dir_eps; /* Dirichlet root noise (extension); eps == 0 disables */
  uint64_t last_select_levels;
  c4o_counters ctr;
};

static const float UNIFORM_P = 1.0f / 7.0f; /* mcts.rs:45 */
static const float NODE_EPS = 1e-8f;         /* mcts.rs:343 */

static int node_alloc(c4o_game* g, const c4o_pos* pos, int parent, float prior) {
  if (g->n_nodes == g->cap) {
    g->cap = g->cap ? g->cap * 2 : 64;
    g->nodes = (c4o_node*)realloc(g->nodes, (size_t)g->cap * sizeof(c4o_node));
  }
  c4o_node* nd = &g->nodes[g->n_nodes];
  nd->pos = *pos;
  nd->parent = parent;
  for (int i = 0; i < 7; i++) nd->child[i] = -1;
  nd->has_children = 0;
  nd->visit_count = 0;
  nd->q_sum_penalty = 0.0f;
  nd->q_sum_no_penalty = 0.0f;
  nd->initial_policy_value = prior;
  g->ctr.nodes_created++;
  return g->n_nodes++;
}

/* mcts.rs:48-56 */
c4o_game* c4o_game_new(const c4o_pos* start, uint64_t game_id, uint64_t p0, uint64_t p1) {
  c4o_game* g = (c4o_game*)calloc(1, sizeof(c4o_game));
  g->game_id = game_id;
  g->player0_id = p0;
  g->player1_id = p1;
  g->root = node_alloc(g, start, -1, 1.0f);
  g->leaf = g->root;
  return g;
}

void c4o_game_set_dirichlet(c4o_game* g, float alpha, float epsilon) {
  g->dir_alpha = alpha;
  g->dir_eps = epsilon;
}

void c4o_game_free(c4o_game* g) {
  if (!g) return;
  free(g->nodes);
  free(g);
}

void c4o_game_root_pos(const c4o_game* g, c4o_pos* out) { *out = g->nodes[g->root].pos; }
void c4o_game_leaf_pos(const c4o_game* g, c4o_pos* out) { *out = g->nodes[g->leaf].pos; }

/* mcts.rs:70-76 */
uint64_t c4o_game_leaf_model_id(const c4o_game* g) {
  return (c4o_ply(&g->nodes[g->leaf].pos) % 2 == 0) ? g->player0_id : g->player1_id;
}

/* mcts.rs:359-361 */
static float node_q_with_penalty(const c4o_node* n) { return n->q_sum_penalty / ((float)n->visit_count + 1.0f); }
/* mcts.rs:365-367 */
static float node_q_no_penalty(const c4o_node* n) { return n->q_sum_no_penalty / ((float)n->visit_count + 1.0f); }

/* mcts.rs:372-381 */
static float node_exploration_value(const c4o_game* g, const c4o_node* n) {
  float parent_visit_count = (n->parent >= 0) ? (float)g->nodes[n->parent].visit_count : (float)n->visit_count;
  float e = c4o_logf(parent_visit_count) / ((float)n->visit_count + 1.0f);
  e = sqrtf(e);
  return e * (n->initial_policy_value + NODE_EPS);
}

/* mcts.rs:386-388 */
static float node_uct_value(const c4o_game* g, const c4o_node* n, float c_exploration) {
  float q = node_q_with_penalty(n);
  float ex = c_exploration * node_exploration_value(g, n);
  return -q + ex;
}

/* Extension: mix Dirichlet noise into the priors of the root's children (see c4o_dirichlet). */
static void apply_root_noise(c4o_game* g) {
  if (!(g->dir_eps > 0.0f)) return;
  c4o_node* root = &g->nodes[g->root];
  if (!root->has_children) return;
  float eta[7];
  c4o_dirichlet(g->game_id, g->n_moves, c4o_legal_mask(&root->pos), g->dir_alpha, eta);
  for (int m = 0; m < 7; m++) {
    int c = root->child[m];
    if (c < 0) continue;
    float keep = (1.0f - g->dir_eps) * g->nodes[c].initial_policy_value;
    float add = g->dir_eps * eta[m];
    g->nodes[c].initial_policy_value = keep + add;
  }
}

/* mcts.rs:114-132 */
static void expand_leaf(c4o_game* g, const float* policy_probs) {
  c4o_pos leaf_pos = g->nodes[g->leaf].pos;
  if (c4o_terminal_state(&leaf_pos) != C4O_NOT_TERMINAL) return;
  unsigned legal = c4o_legal_mask(&leaf_pos);
  for (int m = 0; m < 7; m++) {
    if ((legal >> m) & 1) {
      c4o_pos child_pos;
      c4o_make_move(&leaf_pos, m, &child_pos);
      int c = node_alloc(g, &child_pos, g->leaf, policy_probs[m]);
      g->nodes[g->leaf].child[m] = c;
    } else {
      g->nodes[g->leaf].child[m] = -1;
    }
  }
  g->nodes[g->leaf].has_children = 1;
  g->ctr.expansions++;
}

/* mcts.rs:137-155 */
static void backpropagate_value(c4o_game* g, float q_penalty, float q_no_penalty, int count) {
  int idx = g->leaf;
  for (;;) {
    c4o_node* n = &g->nodes[idx];
    n->visit_count += 1;
    n->q_sum_penalty += q_penalty;
    n->q_sum_no_penalty += q_no_penalty;
    q_penalty = -q_penalty;
    q_no_penalty = -q_no_penalty;
    if (count) g->ctr.backup_nodes++;
    if (n->parent >= 0) idx = n->parent; else break;
  }
}

/* mcts.rs:160-183  max_by_key keeps the LAST maximum; OrdF32 (utils.rs:5-14) panics on NaN
 * as soon as two keys are compared. */
static void select_new_leaf(c4o_game* g, float c_exploration) {
  int idx = g->root;
  g->last_select_levels = 0;
  for (;;) {
    const c4o_node* n = &g->nodes[idx];
    if (!n->has_children) break;
    int best = -1;
    float best_score = 0.0f;
    for (int m = 0; m < 7; m++) {
      int c = n->child[m];
      if (c < 0) continue;
      float score = node_uct_value(g, &g->nodes[c], c_exploration);
      if (best < 0) {
        best = c;
        best_score = score;
      } else {
        if (score != score || best_score != best_score) { g->error = C4O_ERR_NAN_IN_TREE; g->leaf = idx; return; }
        if (score >= best_score) { best = c; best_score = score; }
      }
    }
    if (best < 0) break; /* children Some([None;7]) cannot occur: expand needs a non-terminal leaf */
    g->ctr.select_levels++;
    g->last_select_levels++;
    idx = best;
  }
  g->leaf = idx;
}

/* mcts.rs:83-108 */
int c4o_game_on_received_policy(c4o_game* g, const float* logprobs_in, float q_penalty, float q_no_penalty,
                                float c_exploration, float c_ply_penalty) {
  if (g->error) return g->error;
  c4o_pos leaf_pos = g->nodes[g->leaf].pos;
  float tq_pen, tq_nopen;
  g->ctr.sims++;
  if (c4o_terminal_value(&leaf_pos, c_ply_penalty, &tq_pen, &tq_nopen) != C4O_NOT_TERMINAL) {
    int is_root = (g->leaf == g->root);
    if (is_root) g->ctr.sims_terminal_root++;
    backpropagate_value(g, tq_pen, tq_nopen, !is_root);
    select_new_leaf(g, c_exploration);
  } else {
    float logits[7], probs[7];
    memcpy(logits, logprobs_in, sizeof logits);
    c4o_mask_policy(&leaf_pos, logits);
    int e = c4o_softmax7(logits, probs);
    if (e) { g->error = e; return e; }
    expand_leaf(g, probs);
    if (g->leaf == g->root) apply_root_noise(g); /* extension: a root expanded only now */
    backpropagate_value(g, q_penalty, q_no_penalty, 1);
    select_new_leaf(g, c_exploration);
  }
  return g->error;
}

/* mcts.rs:396-412 */
static void node_policy(const c4o_game* g, const c4o_node* n, float* out) {
  if (n->has_children) {
    float counts[7], sum = 0.0f;
    for (int m = 0; m < 7; m++) counts[m] = (n->child[m] >= 0) ? (float)g->nodes[n->child[m]].visit_count : 0.0f;
    for (int m = 0; m < 7; m++) sum = sum + counts[m];
    if (sum == 0.0f) {
      for (int m = 0; m < 7; m++) out[m] = UNIFORM_P;
    } else {
      for (int m = 0; m < 7; m++) out[m] = counts[m] / sum;
    }
  } else {
    for (int m = 0; m < 7; m++) out[m] = UNIFORM_P;
  }
}

uint64_t c4o_game_root_visit_count(const c4o_game* g) { return g->nodes[g->root].visit_count; }
void c4o_game_root_policy(const c4o_game* g, float* out7) { node_policy(g, &g->nodes[g->root], out7); }
float c4o_game_root_q_penalty(const c4o_game* g) { return node_q_with_penalty(&g->nodes[g->root]); }
float c4o_game_root_q_no_penalty(const c4o_game* g) { return node_q_no_penalty(&g->nodes[g->root]); }
int c4o_game_n_moves(const c4o_game* g) { return g->n_moves; }
int c4o_game_error(const c4o_game* g) { return g->error; }
void c4o_game_counters(const c4o_game* g, c4o_counters* out) { *out = g->ctr; }

/* Dropping the old root (mcts.rs:194-202) frees every node outside the chosen child's
 * subtree; restated as a copy of that subtree into a fresh arena.  Node ORDER inside the
 * arena is irrelevant to the algorithm. */
static void reroot_compact(c4o_game* g, int new_root) {
  c4o_node* old = g->nodes;
  int old_n = g->n_nodes;
  int* map = (int*)malloc((size_t)old_n * sizeof(int));
  int* queue = (int*)malloc((size_t)old_n * sizeof(int));
  int qh = 0, qt = 0;
  queue[qt++] = new_root;
  map[new_root] = 0;
  while (qh < qt) {
    int o = queue[qh++];
    if (old[o].has_children)
      for (int m = 0; m < 7; m++) {
        int c = old[o].child[m];
        if (c >= 0) { map[c] = qt; queue[qt++] = c; }
      }
  }
  int cap = 64;
  while (cap < qt * 2) cap *= 2;
  c4o_node* nn = (c4o_node*)malloc((size_t)cap * sizeof(c4o_node));
  for (int i = 0; i < qt; i++) {
    c4o_node nd = old[queue[i]];
    nd.parent = (i == 0) ? -1 : map[nd.parent]; /* the new root's Weak parent no longer upgrades */
    for (int m = 0; m < 7; m++)
      if (nd.child[m] >= 0) nd.child[m] = map[nd.child[m]];
    nn[i] = nd;
  }
  free(old);
  free(map);
  free(queue);
  g->nodes = nn;
  g->n_nodes = qt;
  g->cap = cap;
  g->root = 0;
  g->leaf = 0;
}

/* mcts.rs:187-206 */
int c4o_game_make_move(c4o_game* g, int m, float c_exploration) {
  if (g->error) return g->error;
  c4o_node* root = &g->nodes[g->root];
  if (m < 0 || m >= 7 || !root->has_children || root->child[m] < 0 || g->n_moves >= C4O_MAX_MOVES) {
    g->error = C4O_ERR_ILLEGAL_MOVE;
    return g->error;
  }
  g->mv_pos[g->n_moves] = root->pos;
  node_policy(g, root, g->mv_policy[g->n_moves]);
  g->mv_col[g->n_moves] = m;
  g->n_moves++;
  g->ctr.moves++;
  reroot_compact(g, root->child[m]);
  apply_root_noise(g); /* extension: the new root starts its search with fresh noise */
  select_new_leaf(g, c_exploration);
  return g->error;
}

/* mcts.rs:214-222 */
int c4o_game_make_random_move(c4o_game* g, float c_exploration, float temperature) {
  if (g->error) return g->error;
  float policy[7];
  int mov = 0;
  node_policy(g, &g->nodes[g->root], policy);
  int e = c4o_sample_move(g->game_id, g->n_moves, policy, temperature, &mov);
  if (e) { g->error = e; return e; }
  return c4o_game_make_move(g, mov, c_exploration);
}

/* mcts.rs:271-313 */
int c4o_game_to_result(const c4o_game* g, float c_ply_penalty, c4o_sample* out, int cap) {
  float q_penalty, q_no_penalty;
  const c4o_pos* root_pos = &g->nodes[g->root].pos;
  if (c4o_terminal_value(root_pos, c_ply_penalty, &q_penalty, &q_no_penalty) == C4O_NOT_TERMINAL) return -C4O_ERR_NOT_TERMINAL;
  if (cap < g->n_moves + 1) return -C4O_ERR_ILLEGAL_MOVE;
  /* cycle [(q, q'), (-q, -q')], skipping one when the number of moves is odd */
  int phase = (g->n_moves % 2 == 1) ? 1 : 0;
  for (int i = 0; i < g->n_moves; i++) {
    out[i].pos = g->mv_pos[i];
    memcpy(out[i].policy, g->mv_policy[i], sizeof out[i].policy);
    if (((i + phase) & 1) == 0) {
      out[i].q_penalty = q_penalty;
      out[i].q_no_penalty = q_no_penalty;
    } else {
      out[i].q_penalty = -q_penalty;
      out[i].q_no_penalty = -q_no_penalty;
    }
  }
  out[g->n_moves].pos = *root_pos;
  for (int m = 0; m < 7; m++) out[g->n_moves].policy[m] = UNIFORM_P;
  out[g->n_moves].q_penalty = q_penalty;
  out[g->n_moves].q_no_penalty = q_no_penalty;
  return g->n_moves + 1;
}

/* self_play.rs:268-323  one MctsJob::Job */
int c4o_game_step(c4o_game* g, const float* logprobs7, float q_pen, float q_nopen,
                  uint64_t n_mcts_iterations, float c_exploration, float c_ply_penalty) {
  int e = c4o_game_on_received_policy(g, logprobs7, q_pen, q_nopen, c_exploration, c_ply_penalty);
  if (e) return -e;
  if (c4o_game_root_visit_count(g) < n_mcts_iterations) return 0; /* self_play.rs:283-286 */
  g->ctr.select_levels_discarded += g->last_select_levels;
  c4o_pos root_pos = g->nodes[g->root].pos;
  if (c4o_terminal_state(&root_pos) == C4O_NOT_TERMINAL) {
    int ply = c4o_ply(&root_pos); /* self_play.rs:294-299 */
    float temperature = (ply < 4) ? 4.0f : (ply < 8) ? 2.0f : 1.0f;
    e = c4o_game_make_random_move(g, c_exploration, temperature);
    if (e) return -e;
    return 0;
  }
  return 1; /* self_play.rs:302-308 */
}

/* ------------------------------------------------------------------------------------------
 * Evaluators
 * ---------------------------------------------------------------------------------------- */
int c4o_eval_uniform(void* ctx, uint64_t model_id, int n, const float* planes, float* lp, float* qp, float* qn) {
  (void)ctx; (void)model_id; (void)planes;
  for (int i = 0; i < n; i++) {
    for (int c = 0; c < 7; c++) lp[7 * i + c] = UNIFORM_P;
    qp[i] = 0.0f;
    qn[i] = 0.0f;
  }
  return 0;
}

int c4o_eval_zeros(void* ctx, uint64_t model_id, int n, const float* planes, float* lp, float* qp, float* qn) {
  (void)ctx; (void)model_id; (void)planes;
  memset(lp, 0, sizeof(float) * 7 * (size_t)n);
  memset(qp, 0, sizeof(float) * (size_t)n);
  memset(qn, 0, sizeof(float) * (size_t)n);
  return 0;
}

/* Integer-hash evaluator: a pure function of (mask, value) in integer arithmetic whose
 * outputs are dyadic rationals exactly representable in f32, so CPU and GPU evaluators agree
 * bit for bit (parity tier T1).  Not part of the reference; a test fake like UniformEvalPos. */
void c4o_hash_eval_pos(uint64_t mask, uint64_t value, float* logits7, float* q_pen, float* q_nopen) {
  int64_t v0 = (int64_t)(value & 0x1FFFFF), v1 = (int64_t)(value >> 21);
  int64_t m0 = (int64_t)(mask & 0x1FFFFF), m1 = (int64_t)(mask >> 21);
  int64_t h = (v0 * 1000003 + v1 * 998244353 + m0 * 19260817 + m1 * 1000000007) % 2147483647;
  for (int c = 0; c < 7; c++) {
    int64_t hc = (h * (2 * c + 3) + 7919 * c) % 1000003;
    logits7[c] = (float)((hc & 63) - 32) / 8.0f;
  }
  *q_pen = (float)(((h >> 5) & 255) - 128) / 128.0f;
  *q_nopen = (float)(((h >> 13) & 255) - 128) / 128.0f;
}

static void planes_to_pos(const float* pl, c4o_pos* out) {
  uint64_t value = 0, opp = 0;
  for (int i = 0; i < 42; i++) {
    if (pl[i] != 0.0f) value |= (uint64_t)1 << i;
    if (pl[42 + i] != 0.0f) opp |= (uint64_t)1 << i;
  }
  out->value = value;
  out->mask = value | opp;
}

int c4o_eval_hash(void* ctx, uint64_t model_id, int n, const float* planes, float* lp, float* qp, float* qn) {
  (void)ctx; (void)model_id;
  for (int i = 0; i < n; i++) {
    c4o_pos p;
    planes_to_pos(planes + (size_t)84 * i, &p);
    c4o_hash_eval_pos(p.mask, p.value, lp + 7 * i, qp + i, qn + i);
  }
  return 0;
}

/* Table evaluator (parity tier T3 at full size): answers a position with what ANOTHER evaluator said for it -- the rows
 * (mask, value) -> (7 log-probabilities, q_penalty, q_no_penalty) a device run logged, sorted by (mask, value), binary search.
 * A position the table does not hold must be a terminal one (the device never shows its evaluator a terminal leaf it can
 * resolve in the launch that selected it; the reference asks and ignores the answer, mcts.rs:92-98): zeros.  Anything else is
 * a divergence between the two searches: error 1 (c4o_self_play fails).  Not part of the reference; a test fake. */
int c4o_eval_table(void* ctx, uint64_t model_id, int n, const float* planes, float* lp, float* qp, float* qn) {
  (void)model_id;
  const c4o_eval_table_ctx* t = (const c4o_eval_table_ctx*)ctx;
  for (int i = 0; i < n; i++) {
    c4o_pos p;
    planes_to_pos(planes + (size_t)84 * i, &p);
    uint64_t lo = 0, hi = t->n;
    while (lo < hi) {
      const uint64_t mid = lo + (hi - lo) / 2;
      if (t->mask[mid] < p.mask || (t->mask[mid] == p.mask && t->value[mid] < p.value)) lo = mid + 1; else hi = mid;
    }
    if (lo < t->n && t->mask[lo] == p.mask && t->value[lo] == p.value) {
      memcpy(lp + 7 * i, t->out + 9 * lo, 7 * sizeof(float));
      qp[i] = t->out[9 * lo + 7];
      qn[i] = t->out[9 * lo + 8];
    } else if (c4o_terminal_state(&p) != C4O_NOT_TERMINAL) {
      memset(lp + 7 * i, 0, 7 * sizeof(float));
      qp[i] = qn[i] = 0.0f;
    } else {
      return 1;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * self_play -- rust/src/self_play.rs:39-129.  The reference runs one NNThread (196-237) and
 * ncpu-1 MctsThreads (268-323) exchanging games over channels; which games share an NN batch
 * depends on thread timing and HashSet order, but each game's trajectory depends only on the
 * evaluator's answer for its own leaf.  Restated as lock-step ticks: one NNThread::loop_once
 * over all pending games, then every answered game runs one MctsThread job (in parallel).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  uint64_t model;
  c4o_pos pos;
  int uid;
  int used;
} c4o_slot;

static inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

static float g_sp_dir_alpha = 0.0f, g_sp_dir_eps = 0.0f;
void c4o_self_play_set_dirichlet(float alpha, float epsilon) {
  g_sp_dir_alpha = alpha;
  g_sp_dir_eps = epsilon;
}

int c4o_self_play(const c4o_game_metadata* reqs, uint64_t n_games, int max_nn_batch_size,
                  uint64_t n_mcts_iterations, float c_exploration, float c_ply_penalty,
                  c4o_eval_fn eval, void* eval_ctx, int n_threads,
                  c4o_sample* out_samples, uint64_t* out_offsets, c4o_selfplay_stats* stats) {
  int rc = C4O_OK;
  if (max_nn_batch_size < 1) max_nn_batch_size = 1;
  if (n_threads < 1) n_threads = 1;
  c4o_selfplay_stats st;
  memset(&st, 0, sizeof st);
  st.n_games = n_games;

  c4o_game** games = (c4o_game**)calloc(n_games ? n_games : 1, sizeof(c4o_game*));
  int* n_out = (int*)calloc(n_games ? n_games : 1, sizeof(int));
  c4o_sample* tmp_samples = (c4o_sample*)malloc(sizeof(c4o_sample) * 43 * (n_games ? n_games : 1));
  uint64_t* pending = (uint64_t*)malloc(sizeof(uint64_t) * (n_games ? n_games : 1));
  uint64_t n_pending = n_games;
  c4o_pos start = {0, 0};
  for (uint64_t i = 0; i < n_games; i++) {
    games[i] = c4o_game_new(&start, reqs[i].game_id, reqs[i].player0_id, reqs[i].player1_id); /* self_play.rs:55-58 */
    c4o_game_set_dirichlet(games[i], g_sp_dir_alpha, g_sp_dir_eps);
    pending[i] = i;
  }

  size_t tab_cap = 16;
  while (tab_cap < 2 * (size_t)n_games + 2) tab_cap *= 2;
  c4o_slot* tab = (c4o_slot*)malloc(sizeof(c4o_slot) * tab_cap);
  int* game_uid = (int*)malloc(sizeof(int) * (n_games ? n_games : 1));
  uint64_t* umodel = (uint64_t*)malloc(sizeof(uint64_t) * (n_games ? n_games : 1));
  c4o_pos* upos = (c4o_pos*)malloc(sizeof(c4o_pos) * (n_games ? n_games : 1));
  int* ubatch = (int*)malloc(sizeof(int) * (n_games ? n_games : 1));
  float* planes = (float*)malloc(sizeof(float) * 84 * (size_t)max_nn_batch_size);
  float* lp = (float*)malloc(sizeof(float) * 7 * (size_t)max_nn_batch_size);
  float* qp = (float*)malloc(sizeof(float) * (size_t)max_nn_batch_size);
  float* qn = (float*)malloc(sizeof(float) * (size_t)max_nn_batch_size);
  int* status = (int*)malloc(sizeof(int) * (n_games ? n_games : 1));

  while (n_pending > 0 && rc == C4O_OK) {
    /* NNThread::loop_once, self_play.rs:196-237: unique (model, leaf position) pairs */
    memset(tab, 0, sizeof(c4o_slot) * tab_cap);
    int n_unique = 0;
    for (uint64_t k = 0; k < n_pending; k++) {
      uint64_t gi = pending[k];
      uint64_t model = c4o_game_leaf_model_id(games[gi]);
      c4o_pos lpz;
      c4o_game_leaf_pos(games[gi], &lpz);
      size_t h = (size_t)(mix64(lpz.mask * 0x9E3779B97F4A7C15ull ^ mix64(lpz.value ^ model * 0xD6E8FEB86659FD93ull))) & (tab_cap - 1);
      for (;;) {
        if (!tab[h].used) {
          tab[h].used = 1; tab[h].model = model; tab[h].pos = lpz; tab[h].uid = n_unique;
          umodel[n_unique] = model; upos[n_unique] = lpz;
          n_unique++;
          break;
        }
        if (tab[h].model == model && tab[h].pos.mask == lpz.mask && tab[h].pos.value == lpz.value) break;
        h = (h + 1) & (tab_cap - 1);
      }
      game_uid[gi] = tab[h].uid;
    }
    /* model with the most unique positions; BTreeMap order + max_by_key => ties go to the
     * largest model id (self_play.rs:211-215) */
    uint64_t best_model = 0, best_count = 0;
    int have = 0;
    {
      uint64_t mids[64], mcnt[64];
      int n_models = 0, overflow = 0;
      for (int u = 0; u < n_unique; u++) {
        int w = 0;
        while (w < n_models && mids[w] != umodel[u]) w++;
        if (w == n_models) {
          if (n_models == 64) { overflow = 1; break; }
          mids[n_models] = umodel[u]; mcnt[n_models] = 0; n_models++;
        }
        mcnt[w]++;
      }
      if (overflow) { rc = C4O_ERR_ILLEGAL_MOVE; break; } /* oracle limit: 64 distinct models per tick */
      for (int w = 0; w < n_models; w++)
        if (!have || mcnt[w] > best_count || (mcnt[w] == best_count && mids[w] > best_model)) {
          best_model = mids[w]; best_count = mcnt[w]; have = 1;
        }
    }
    int nb = 0;
    for (int u = 0; u < n_unique; u++) {
      if (umodel[u] == best_model && nb < max_nn_batch_size) {
        ubatch[u] = nb;
        c4o_write_planes(&upos[u], planes + (size_t)84 * nb); /* pybridge.rs:202-221 */
        nb++;
      } else {
        ubatch[u] = -1;
      }
    }
    st.nn_calls++;
    st.nn_positions += (uint64_t)nb;
    if (eval(eval_ctx, best_model, nb, planes, lp, qp, qn) != 0) { rc = C4O_ERR_DEGENERATE_POLICY; break; }

    /* MctsThread::loop_once for every answered game (self_play.rs:225-236, 268-323) */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads) if (n_threads > 1 && n_pending >= 512)
#endif
    for (uint64_t k = 0; k < n_pending; k++) {
      uint64_t gi = pending[k];
      int b = ubatch[game_uid[gi]];
      if (b < 0) { status[gi] = 0; continue; }
      status[gi] = c4o_game_step(games[gi], lp + 7 * b, qp[b], qn[b], n_mcts_iterations, c_exploration, c_ply_penalty);
      if (status[gi] == 1) {
        n_out[gi] = c4o_game_to_result(games[gi], c_ply_penalty, tmp_samples + 43 * gi, 43);
        c4o_counters c;
        c4o_game_counters(games[gi], &c);
        /* keep the counters, drop the tree */
        free(games[gi]->nodes);
        games[gi]->nodes = NULL;
        games[gi]->n_nodes = games[gi]->cap = 0;
        games[gi]->ctr = c;
      }
    }
    uint64_t w = 0;
    for (uint64_t k = 0; k < n_pending; k++) {
      uint64_t gi = pending[k];
      if (status[gi] < 0) { rc = -status[gi]; }
      if (status[gi] == 0) pending[w++] = gi;
    }
    n_pending = w;
  }

  uint64_t off = 0;
  for (uint64_t i = 0; i < n_games; i++) {
    out_offsets[i] = off;
    if (rc == C4O_OK && n_out[i] > 0) {
      memcpy(out_samples + off, tmp_samples + 43 * i, sizeof(c4o_sample) * (size_t)n_out[i]);
      off += (uint64_t)n_out[i];
    }
    c4o_counters c;
    c4o_game_counters(games[i], &c);
    st.tree.sims += c.sims; st.tree.sims_terminal_root += c.sims_terminal_root;
    st.tree.select_levels += c.select_levels; st.tree.select_levels_discarded += c.select_levels_discarded;
    st.tree.backup_nodes += c.backup_nodes;
    st.tree.expansions += c.expansions; st.tree.nodes_created += c.nodes_created; st.tree.moves += c.moves;
    c4o_game_free(games[i]);
  }
  out_offsets[n_games] = off;
  st.n_samples = off;
  if (stats) *stats = st;

  free(games); free(n_out); free(tmp_samples); free(pending); free(tab); free(game_uid);
  free(umodel); free(upos); free(ubatch); free(planes); free(lp); free(qp); free(qn); free(status);
  return rc;
}

/* ------------------------------------------------------------------------------------------
 * self_play in the REFERENCE'S THREAD TOPOLOGY -- rust/src/self_play.rs:39-129: one NN thread
 * (NNThread, :196-237) and n_threads - 1 MCTS worker threads (MctsThread, :268-323) exchanging
 * games over two queues (crossbeam channels there, mutex + condvar rings here), so that network
 * evaluation and tree work overlap exactly as in the reference.  The NN thread is the CALLING
 * thread (the evaluator may be a Python callback).  Every game's samples equal c4o_self_play's:
 * only the batching differs, and a game's trajectory depends on the answers for its own leaves
 * alone.  Used by bench.py's cpu_baseline leg and checked against c4o_self_play in the tests.
 * ---------------------------------------------------------------------------------------- */
#include <pthread.h>
#include <sched.h>
#include <time.h>

/* Timing aid for bench.py's cpu_baseline leg (no effect on any result): with pinning on, the NN thread
 * (the caller) keeps the first CPU of the process's affinity set to itself and the MCTS workers are
 * spread one per CPU over the others, so that 15 spinning workers cannot preempt the thread that
 * feeds them (the source of a +-25 % run-to-run spread). */
static int g_pin_threads = 0;
void c4o_set_thread_pinning(int on) { g_pin_threads = on; }

typedef struct {
  uint64_t game;      /* index into games[]; UINT64_MAX = MctsJob::PoisonPill (self_play.rs:317-322) */
  float lp[7], qp, qn;
} c4o_job;

/* Bounded multi-producer multi-consumer queue (D. Vyukov's array queue: one sequence number per
 * cell), standing in for crossbeam_channel::bounded (self_play.rs:51-53): lock-free, a blocked
 * receiver spins briefly, then yields, then sleeps -- as crossbeam's Backoff/park does. */
typedef struct { uint64_t seq; c4o_job job; } c4o_cell;
typedef struct {
  c4o_cell* cells; uint64_t mask;
  uint64_t enq __attribute__((aligned(64)));
  uint64_t deq __attribute__((aligned(64)));
} c4o_queue;

static void queue_init(c4o_queue* q, uint64_t min_cap) {
  uint64_t cap = 2;
  while (cap < min_cap) cap <<= 1;
  q->cells = (c4o_cell*)malloc(sizeof(c4o_cell) * cap);
  q->mask = cap - 1;
  for (uint64_t i = 0; i < cap; i++) q->cells[i].seq = i;
  q->enq = q->deq = 0;
}

static int queue_push(c4o_queue* q, const c4o_job* j) {
  uint64_t pos = __atomic_load_n(&q->enq, __ATOMIC_RELAXED);
  for (;;) {
    c4o_cell* c = &q->cells[pos & q->mask];
    const int64_t dif = (int64_t)(__atomic_load_n(&c->seq, __ATOMIC_ACQUIRE) - pos);
    if (dif == 0) {
      if (__atomic_compare_exchange_n(&q->enq, &pos, pos + 1, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
        c->job = *j;
        __atomic_store_n(&c->seq, pos + 1, __ATOMIC_RELEASE);
        return 1;
      }
    } else if (dif < 0) {
      return 0; /* full: cannot happen, every game is in at most one queue and the rings hold them all */
    } else {
      pos = __atomic_load_n(&q->enq, __ATOMIC_RELAXED);
    }
  }
}

static int queue_try_pop(c4o_queue* q, c4o_job* out) {
  uint64_t pos = __atomic_load_n(&q->deq, __ATOMIC_RELAXED);
  for (;;) {
    c4o_cell* c = &q->cells[pos & q->mask];
    const int64_t dif = (int64_t)(__atomic_load_n(&c->seq, __ATOMIC_ACQUIRE) - (pos + 1));
    if (dif == 0) {
      if (__atomic_compare_exchange_n(&q->deq, &pos, pos + 1, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
        *out = c->job;
        __atomic_store_n(&c->seq, pos + q->mask + 1, __ATOMIC_RELEASE);
        return 1;
      }
    } else if (dif < 0) {
      return 0; /* empty */
    } else {
      pos = __atomic_load_n(&q->deq, __ATOMIC_RELAXED);
    }
  }
}

static inline void backoff(unsigned* n) {
  if (*n < 64) { __builtin_ia32_pause(); }
  else if (*n < 256) { sched_yield(); }
  else { struct timespec ts = {0, 20000}; nanosleep(&ts, NULL); }
  (*n)++;
}

typedef struct {
  c4o_queue nn_queue;   /* games waiting for the network (self_play.rs:51); only .game is used */
  c4o_queue mcts_queue; /* evaluated games waiting for a worker (self_play.rs:52) */
  int nn_closed;
  c4o_game** games; int* n_out; c4o_sample* tmp_samples;
  uint64_t n_remaining; int n_workers; int error;
  uint64_t n_iter; float c_exploration, c_ply_penalty;
} c4o_async;

/* The rings hold every game at once, so a push can only find its cell "full" while the consumer
 * that emptied it a lap ago has claimed it but not yet released it (preempted between its CAS and
 * its sequence store): wait for it -- dropping the item would lose a game. */
static void queue_push_wait(c4o_queue* q, const c4o_job* j) {
  unsigned spins = 0;
  while (!queue_push(q, j)) backoff(&spins);
}

static void async_push_nn(c4o_async* a, uint64_t gi) {
  c4o_job j;
  memset(&j, 0, sizeof j);
  j.game = gi;
  queue_push_wait(&a->nn_queue, &j);
}

static void async_push_jobs(c4o_async* a, const c4o_job* j, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) queue_push_wait(&a->mcts_queue, &j[i]);
}

/* MctsThread::loop_until_close (self_play.rs:268-323) */
static void* async_worker(void* arg) {
  c4o_async* a = (c4o_async*)arg;
  for (;;) {
    c4o_job j;
    unsigned spins = 0;
    while (!queue_try_pop(&a->mcts_queue, &j)) backoff(&spins);
    if (j.game == UINT64_MAX) break;
    c4o_game* g = a->games[j.game];
    int st = c4o_game_step(g, j.lp, j.qp, j.qn, a->n_iter, a->c_exploration, a->c_ply_penalty);
    if (st == 0) { async_push_nn(a, j.game); continue; }
    if (st == 1) {
      a->n_out[j.game] = c4o_game_to_result(g, a->c_ply_penalty, a->tmp_samples + 43 * j.game, 43);
      c4o_counters c;
      c4o_game_counters(g, &c);
      free(g->nodes);
      g->nodes = NULL;
      g->n_nodes = g->cap = 0;
      g->ctr = c;
    } else {
      __atomic_store_n(&a->error, -st, __ATOMIC_RELAXED);
    }
    /* game over (or failed): the thread that retires the last game poisons the others and lets the
     * NN thread see its queue close (self_play.rs:302-312, 30-38) */
    if (__atomic_sub_fetch(&a->n_remaining, 1, __ATOMIC_ACQ_REL) == 0) {
      c4o_job pill;
      memset(&pill, 0, sizeof pill);
      pill.game = UINT64_MAX;
      for (int w = 0; w < a->n_workers - 1; w++) async_push_jobs(a, &pill, 1);
      __atomic_store_n(&a->nn_closed, 1, __ATOMIC_RELEASE);
      break;
    }
  }
  return NULL;
}

int c4o_self_play_async(const c4o_game_metadata* reqs, uint64_t n_games, int max_nn_batch_size,
                        uint64_t n_mcts_iterations, float c_exploration, float c_ply_penalty,
                        c4o_eval_fn eval, void* eval_ctx, int n_threads,
                        c4o_sample* out_samples, uint64_t* out_offsets, c4o_selfplay_stats* stats) {
  if (max_nn_batch_size < 1) max_nn_batch_size = 1;
  if (n_threads < 2) n_threads = 2; /* the NN thread + at least one worker (self_play.rs:78) */
  c4o_selfplay_stats st;
  memset(&st, 0, sizeof st);
  st.n_games = n_games;
  const size_t ng = n_games ? n_games : 1;
  c4o_async a;
  memset(&a, 0, sizeof a);
  a.n_workers = n_threads - 1;
  queue_init(&a.nn_queue, ng + 1);
  queue_init(&a.mcts_queue, ng + (size_t)n_threads + 1);
  a.games = (c4o_game**)calloc(ng, sizeof(c4o_game*));
  a.n_out = (int*)calloc(ng, sizeof(int));
  a.tmp_samples = (c4o_sample*)malloc(sizeof(c4o_sample) * 43 * ng);
  a.n_remaining = n_games;
  a.n_iter = n_mcts_iterations; a.c_exploration = c_exploration; a.c_ply_penalty = c_ply_penalty;
  c4o_pos start = {0, 0};
  for (uint64_t i = 0; i < n_games; i++) {
    a.games[i] = c4o_game_new(&start, reqs[i].game_id, reqs[i].player0_id, reqs[i].player1_id);
    c4o_game_set_dirichlet(a.games[i], g_sp_dir_alpha, g_sp_dir_eps);
    async_push_nn(&a, i); /* self_play.rs:55-58 */
  }
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)a.n_workers);
  int started = 0;
  if (n_games > 0)
    for (; started < a.n_workers; started++)
      if (pthread_create(&th[started], NULL, async_worker, &a) != 0) break;
  cpu_set_t old_set;
  int pinned = 0;
  if (g_pin_threads && started > 0 && sched_getaffinity(0, sizeof old_set, &old_set) == 0 && CPU_COUNT(&old_set) >= 2) {
    int cpus[CPU_SETSIZE], n_cpus = 0;
    for (int c = 0; c < CPU_SETSIZE; c++)
      if (CPU_ISSET(c, &old_set)) cpus[n_cpus++] = c;
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(cpus[0], &one);
    pinned = pthread_setaffinity_np(pthread_self(), sizeof one, &one) == 0;
    for (int w = 0; pinned && w < started; w++) {
      CPU_ZERO(&one);
      CPU_SET(cpus[1 + w % (n_cpus - 1)], &one);
      (void)pthread_setaffinity_np(th[w], sizeof one, &one);
    }
  }
  int rc = (n_games > 0 && started < a.n_workers) ? C4O_ERR_ILLEGAL_MOVE : C4O_OK;

  /* NNThread::loop_until_close (self_play.rs:196-237) on the calling thread */
  uint64_t* pend = (uint64_t*)malloc(sizeof(uint64_t) * ng);
  uint64_t n_pend = 0;
  size_t tab_cap = 16;
  while (tab_cap < 2 * ng + 2) tab_cap *= 2;
  c4o_slot* tab = (c4o_slot*)malloc(sizeof(c4o_slot) * tab_cap);
  int* game_uid = (int*)malloc(sizeof(int) * ng);
  uint64_t* umodel = (uint64_t*)malloc(sizeof(uint64_t) * ng);
  c4o_pos* upos = (c4o_pos*)malloc(sizeof(c4o_pos) * ng);
  int* ubatch = (int*)malloc(sizeof(int) * ng);
  float* planes = (float*)malloc(sizeof(float) * 84 * (size_t)max_nn_batch_size);
  float* lp = (float*)malloc(sizeof(float) * 7 * (size_t)max_nn_batch_size);
  float* qp = (float*)malloc(sizeof(float) * (size_t)max_nn_batch_size);
  float* qn = (float*)malloc(sizeof(float) * (size_t)max_nn_batch_size);
  c4o_job* out_jobs = (c4o_job*)malloc(sizeof(c4o_job) * ng);
  int closed = 0;
  while (rc == C4O_OK && n_games > 0 && (!closed || n_pend > 0)) {
    /* drain_queue (self_play.rs:175-193): block while nothing is pending, then take everything that is there */
    {
      c4o_job j;
      unsigned spins = 0;
      for (;;) {
        closed = __atomic_load_n(&a.nn_closed, __ATOMIC_ACQUIRE);
        while (queue_try_pop(&a.nn_queue, &j)) pend[n_pend++] = j.game;
        if (n_pend > 0 || closed || __atomic_load_n(&a.error, __ATOMIC_RELAXED)) break;
        backoff(&spins);
      }
    }
    if (__atomic_load_n(&a.error, __ATOMIC_RELAXED)) break;
    if (n_pend == 0) continue;
    /* unique (model, leaf position) pairs of the pending games */
    memset(tab, 0, sizeof(c4o_slot) * tab_cap);
    int n_unique = 0;
    for (uint64_t k = 0; k < n_pend; k++) {
      uint64_t gi = pend[k];
      uint64_t model = c4o_game_leaf_model_id(a.games[gi]);
      c4o_pos lpz;
      c4o_game_leaf_pos(a.games[gi], &lpz);
      size_t h = (size_t)(mix64(lpz.mask * 0x9E3779B97F4A7C15ull ^ mix64(lpz.value ^ model * 0xD6E8FEB86659FD93ull))) & (tab_cap - 1);
      for (;;) {
        if (!tab[h].used) {
          tab[h].used = 1; tab[h].model = model; tab[h].pos = lpz; tab[h].uid = n_unique;
          umodel[n_unique] = model; upos[n_unique] = lpz;
          n_unique++;
          break;
        }
        if (tab[h].model == model && tab[h].pos.mask == lpz.mask && tab[h].pos.value == lpz.value) break;
        h = (h + 1) & (tab_cap - 1);
      }
      game_uid[gi] = tab[h].uid;
    }
    /* the model with the most unique positions, ties to the largest id (self_play.rs:211-215) */
    uint64_t best_model = 0, best_count = 0;
    {
      uint64_t mids[64], mcnt[64];
      int n_models = 0, overflow = 0, have = 0;
      for (int u = 0; u < n_unique; u++) {
        int w = 0;
        while (w < n_models && mids[w] != umodel[u]) w++;
        if (w == n_models) {
          if (n_models == 64) { overflow = 1; break; }
          mids[n_models] = umodel[u]; mcnt[n_models] = 0; n_models++;
        }
        mcnt[w]++;
      }
      if (overflow) { rc = C4O_ERR_ILLEGAL_MOVE; break; }
      for (int w = 0; w < n_models; w++)
        if (!have || mcnt[w] > best_count || (mcnt[w] == best_count && mids[w] > best_model)) {
          best_model = mids[w]; best_count = mcnt[w]; have = 1;
        }
    }
    int nb = 0;
    for (int u = 0; u < n_unique; u++) {
      if (umodel[u] == best_model && nb < max_nn_batch_size) {
        ubatch[u] = nb;
        c4o_write_planes(&upos[u], planes + (size_t)84 * nb);
        nb++;
      } else {
        ubatch[u] = -1;
      }
    }
    st.nn_calls++;
    st.nn_positions += (uint64_t)nb;
    if (eval(eval_ctx, best_model, nb, planes, lp, qp, qn) != 0) { rc = C4O_ERR_DEGENERATE_POLICY; break; }
    /* answered games go to the workers in one batch, the others stay pending (self_play.rs:225-236) */
    uint64_t n_jobs = 0, w = 0;
    for (uint64_t k = 0; k < n_pend; k++) {
      uint64_t gi = pend[k];
      int b = ubatch[game_uid[gi]];
      if (b < 0) { pend[w++] = gi; continue; }
      c4o_job* j = &out_jobs[n_jobs++];
      j->game = gi;
      memcpy(j->lp, lp + 7 * b, sizeof j->lp);
      j->qp = qp[b]; j->qn = qn[b];
    }
    n_pend = w;
    async_push_jobs(&a, out_jobs, n_jobs);
  }
  if (rc == C4O_OK && a.error) rc = a.error;
  if (rc != C4O_OK || a.n_remaining > 0) {
    /* error path: release the workers still waiting for jobs */
    c4o_job pill;
    memset(&pill, 0, sizeof pill);
    pill.game = UINT64_MAX;
    for (int w = 0; w < started; w++) async_push_jobs(&a, &pill, 1);
  }
  for (int w = 0; w < started; w++) pthread_join(th[w], NULL);
  if (pinned) (void)pthread_setaffinity_np(pthread_self(), sizeof old_set, &old_set);

  uint64_t off = 0;
  for (uint64_t i = 0; i < n_games; i++) {
    out_offsets[i] = off;
    if (rc == C4O_OK && a.n_out[i] > 0) {
      memcpy(out_samples + off, a.tmp_samples + 43 * i, sizeof(c4o_sample) * (size_t)a.n_out[i]);
      off += (uint64_t)a.n_out[i];
    }
    c4o_counters c;
    c4o_game_counters(a.games[i], &c);
    st.tree.sims += c.sims; st.tree.sims_terminal_root += c.sims_terminal_root;
    st.tree.select_levels += c.select_levels; st.tree.select_levels_discarded += c.select_levels_discarded;
    st.tree.backup_nodes += c.backup_nodes;
    st.tree.expansions += c.expansions; st.tree.nodes_created += c.nodes_created; st.tree.moves += c.moves;
    c4o_game_free(a.games[i]);
  }
  out_offsets[n_games] = off;
  st.n_samples = off;
  if (stats) *stats = st;
  free(th); free(pend); free(tab); free(game_uid); free(umodel); free(upos); free(ubatch);
  free(planes); free(lp); free(qp); free(qn); free(out_jobs);
  free(a.nn_queue.cells); free(a.mcts_queue.cells); free(a.games); free(a.n_out); free(a.tmp_samples);
  return rc;
}
