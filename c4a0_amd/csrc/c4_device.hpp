// c4_device.hpp -- device-side building blocks of the self-play generator (gfx950).
//
// Bitboard rules, the glibc expf/logf ports, ChaCha12 move sampling and the policy
// arithmetic, written once as inline device functions and used by the fused step kernel and by
// the element-wise parity kernels.  Floating point here must reproduce the reference's f32
// results exactly, so this translation unit is compiled with -ffp-contract=off and every
// fused operation is written out (__builtin_fma).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#define C4_DEV __device__ __forceinline__

namespace c4 {

// ------------------------------------------------------------------------------------------
// Bitboard (reference rust/src/c4r.rs).  bit = row*7 + col, row 0 = bottom; `value` = pieces
// of the side to move (c4r.rs:13-24,119-122).
// ------------------------------------------------------------------------------------------
constexpr uint64_t kBoard = (1ull << 42) - 1;
constexpr uint64_t kCol0 = 0x810204081ull;  // bits 0,7,14,21,28,35
// start columns 0..3 of every row: a 4-run to the right must not wrap into the next row
constexpr uint64_t kStart03 = (kCol0 | (kCol0 << 1) | (kCol0 << 2) | (kCol0 << 3));

// c4r.rs:266-269: column c is legal iff the top cell (row 5) is empty.
C4_DEV uint32_t legal_mask(uint64_t mask) { return (uint32_t)(~mask >> 35) & 0x7Fu; }

// c4r.rs:58-72 + invert c4r.rs:125-129.  Pieces stack from the bottom, so the lowest empty row
// of a column is its piece count.  Caller guarantees the column is legal.
C4_DEV void make_move(uint64_t& mask, uint64_t& value, uint32_t col) {
  uint32_t h = __popcll(mask & (kCol0 << col));
  uint64_t bit = 1ull << (7u * h + col);
  mask |= bit;
  value = ~(value | bit) & mask;
}

// Four in a row among the bits of x (c4r.rs:165-224,241-249; any method, the result is boolean).
// Per direction with stride s: bit i of the result = x[i] & x[i+s] & x[i+2s] & x[i+3s], folded pairwise (pairs first,
// then pairs of pairs: two shifts per direction instead of three); the start-column mask keeps the horizontal and the
// two diagonal lines inside their rows.
C4_DEV bool has_four(uint64_t x) {
  const uint64_t h2 = x & (x >> 1), v2 = x & (x >> 7), a2 = x & (x >> 8);
  const uint64_t t = x >> 3, b2 = t & (t >> 6);                      // anti-diagonal: x[i+3], x[i+9], x[i+15], x[i+21]
  const uint64_t h = h2 & (h2 >> 2), v = v2 & (v2 >> 14), d1 = a2 & (a2 >> 16), d2 = b2 & (b2 >> 12);
  return (((h | d1 | d2) & kStart03) | v) != 0;
}

// c4r.rs:228-238: 0 none, 1 PlayerWin, 2 OpponentWin, 3 Draw -- in that order.
C4_DEV uint32_t terminal_state(uint64_t mask, uint64_t value) {
  if (has_four(mask & value)) return 1;
  if (has_four(mask & ~value)) return 2;
  if (__popcll(mask) == 42) return 3;
  return 0;
}

// terminal_state of a position reached by a legal move FROM A NON-TERMINAL POSITION: no four existed
// before the move and the mover's new piece belongs to the opponent of the side now to move, so
// PlayerWin is impossible and only the opponent's stones need the four-in-a-row test.
C4_DEV uint32_t terminal_after_move(uint64_t mask, uint64_t value) {
  if (has_four(mask & ~value)) return 2;
  if (__popcll(mask) == 42) return 3;
  return 0;
}

// c4r.rs:253-263
C4_DEV void terminal_value(uint32_t t, uint64_t mask, float c_ply_penalty, float& q_pen, float& q_nopen) {
  float mag = c_ply_penalty * (float)__popcll(mask);
  if (t == 1) { q_pen = 1.0f - mag; q_nopen = 1.0f; }
  else if (t == 2) { q_pen = -1.0f + mag; q_nopen = -1.0f; }
  else { q_pen = 0.0f; q_nopen = 0.0f; }
}

// c4r.rs:378-392: element e of the [2][6][7] planes: plane 0 = bits of value, plane 1 = bits of
// the opponent's pieces, in bit order.
C4_DEV uint32_t plane_bit(uint64_t mask, uint64_t value, uint32_t e) {
  return e < 42 ? (uint32_t)(value >> e) & 1u : (uint32_t)((mask & ~value) >> (e - 42)) & 1u;
}

// ------------------------------------------------------------------------------------------
// glibc 2.35 expf / logf (sysdeps/ieee754/flt-32/e_expf.c, e_logf.c): Rust's f32::exp / f32::ln
// (mcts.rs:379,430,451-453) call the platform libm, which is not correctly rounded, so the
// algorithm itself is reproduced in f64.  Checked on the GPU against the host libm.
// ------------------------------------------------------------------------------------------
__device__ const uint64_t kExp2fTab[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51,
    0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
    0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
    0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
    0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
    0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};

__device__ const double kLogfTab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2},
};

C4_DEV float c4_expf(float x) {
  const double InvLn2N = 0x1.71547652b82fep+5, SHIFT = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
  const uint32_t ix = __float_as_uint(x);
  const uint32_t abstop = (ix >> 20) & 0x7ff;
  if (abstop >= 0x42b) {  // |x| >= 88 or NaN   (top12(88.0f) = 0x42b)
    if (ix == 0xff800000u) return 0.0f;
    if (abstop >= 0x7f8) return x + x;
    if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);
    if (x < -0x1.9fe368p6f) return 0.0f;
    if (x < -0x1.9d1d9ep6f) return __uint_as_float(0x00000001u);  // 0x1.4p-75f squared rounds to 2^-149
  }
  const double xd = (double)x;
  const double z = InvLn2N * xd;
  double kd = z + SHIFT;
  const uint64_t ki = (uint64_t)__double_as_longlong(kd);
  kd -= SHIFT;
  const double r = __builtin_fma(InvLn2N, xd, -kd);  // the fusion glibc's FMA variant performs
  const uint64_t t = kExp2fTab[ki & 31] + (ki << 47);
  const double s = __longlong_as_double((long long)t);
  const double zz = C0 * r + C1;
  const double r2 = r * r;
  double y = C2 * r + 1.0;
  y = zz * r2 + y;
  y = y * s;
  return (float)y;
}

C4_DEV float c4_logf(float x) {
  const double Ln2 = 0x1.62e42fefa39efp-1;
  const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  uint32_t ix = __float_as_uint(x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (ix * 2 == 0) return __uint_as_float(0xff800000u);  // log(0) = -inf
    if (ix == 0x7f800000u) return x;
    if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return __uint_as_float(0x7fc00000u);
    ix = __float_as_uint(x * 0x1p23f);  // subnormal
    ix -= 23u << 23;
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (tmp >> 19) & 15;
  const int k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & 0xff800000u);
  const double invc = kLogfTab[i][0], logc = kLogfTab[i][1];
  const double z = (double)__uint_as_float(iz);
  const double r = z * invc - 1.0;
  const double y0 = logc + (double)k * Ln2;
  const double r2 = r * r;
  double y = A1 * r + A2;
  y = A0 * r2 + y;
  y = y * r2 + (y0 + r);
  return (float)y;
}

// Rust f32::max: NaN-ignoring.
C4_DEV float rust_max(float a, float b) {
  if (a != a) return b;
  if (b != b) return a;
  return a > b ? a : b;
}

// mcts.rs:416-434 on one thread.  Returns false where the reference panics.
C4_DEV bool softmax7(const float* logits, float* out) {
  float mx = __uint_as_float(0xff800000u);
  for (int i = 0; i < 7; i++) mx = rust_max(mx, logits[i]);
  if (__builtin_isinf(mx)) return false;
  float e[7], s = 0.0f;
  for (int i = 0; i < 7; i++) e[i] = c4_expf(logits[i] - mx);
  for (int i = 0; i < 7; i++) s = s + e[i];
  for (int i = 0; i < 7; i++) out[i] = e[i] / s;
  return true;
}

// mcts.rs:439-454
C4_DEV void apply_temperature(const float* p, float t, float* out) {
  bool all_eq = true;
  for (int i = 0; i < 7; i++) all_eq = all_eq && (p[i] == p[0]);
  if (t == 1.0f || all_eq) {
    for (int i = 0; i < 7; i++) out[i] = p[i];
    return;
  }
  if (t == 0.0f) {
    float mx = __uint_as_float(0xff800000u), s = 0.0f;
    for (int i = 0; i < 7; i++) mx = rust_max(mx, p[i]);
    for (int i = 0; i < 7; i++) { out[i] = (p[i] == mx) ? 1.0f : 0.0f; s = s + out[i]; }
    for (int i = 0; i < 7; i++) out[i] = out[i] / s;
    return;
  }
  float pl[7], s = 0.0f;
  for (int i = 0; i < 7; i++) pl[i] = c4_logf(p[i]) / t;
  for (int i = 0; i < 7; i++) s = s + c4_expf(pl[i]);
  const float lse = c4_logf(s);
  for (int i = 0; i < 7; i++) {
    float v = c4_expf(pl[i] - lse);
    if (v < 0.0f) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    out[i] = v;
  }
}

// ------------------------------------------------------------------------------------------
// Move sampling (mcts.rs:214-222): StdRng::seed_from_u64 + WeightedIndex<f32>, i.e. rand 0.10.1 /
// rand_core 0.10.1 (PCG32 seed expansion) / chacha20 0.10.1 (ChaCha12).  Restated from the
// crates' documented algorithm (their source is not part of the reference checkout); the oracle
// twin of each stage reproduces the crates' own published test vectors (tests/test_oracle_libm_rng.py).
// ------------------------------------------------------------------------------------------
C4_DEV uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

#define C4_QR(a, b, c, d)                 \
  a += b; d ^= a; d = rotl32(d, 16);      \
  c += d; b ^= c; b = rotl32(b, 12);      \
  a += b; d ^= a; d = rotl32(d, 8);       \
  c += d; b ^= c; b = rotl32(b, 7);

// First 32-bit word of ChaCha12 keyed by the PCG32 expansion of `seed`.
C4_DEV uint32_t rng_first_u32(uint64_t state) {
  const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
  uint32_t k[8];
  for (int i = 0; i < 8; i++) {
    state = state * MUL + INC;
    uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
    uint32_t rot = (uint32_t)(state >> 59);
    k[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
  }
  uint32_t x0 = 0x61707865, x1 = 0x3320646e, x2 = 0x79622d32, x3 = 0x6b206574;
  uint32_t x4 = k[0], x5 = k[1], x6 = k[2], x7 = k[3], x8 = k[4], x9 = k[5], x10 = k[6], x11 = k[7];
  uint32_t x12 = 0, x13 = 0, x14 = 0, x15 = 0;
  for (int r = 0; r < 6; r++) {
    C4_QR(x0, x4, x8, x12) C4_QR(x1, x5, x9, x13) C4_QR(x2, x6, x10, x14) C4_QR(x3, x7, x11, x15)
    C4_QR(x0, x5, x10, x15) C4_QR(x1, x6, x11, x12) C4_QR(x2, x7, x8, x13) C4_QR(x3, x4, x9, x14)
  }
  return x0 + 0x61707865u;
}

// WeightedIndex<f32>::new(w).sample() given the generator's next_u32.  -1 where new() fails.
C4_DEV int weighted_index(const float* w, uint32_t u) {
  float cum[6];
  float total = w[0];
  if (!(total >= 0.0f)) return -1;
  for (int i = 1; i < 7; i++) {
    if (!(w[i] >= 0.0f)) return -1;
    cum[i - 1] = total;
    total = total + w[i];
  }
  if (total == 0.0f) return -1;
  if (!(__builtin_fabsf(total) < __uint_as_float(0x7f800000u))) return -1;
  const float max_rand = 1.0f - 0x1p-23f;
  float scale = total;  // high - low with low = 0
  for (;;) {
    float t = scale * max_rand;
    t = t + 0.0f;
    if (!(t >= total)) break;
    scale = __uint_as_float(__float_as_uint(scale) - 1);
  }
  const float u01 = __uint_as_float(0x3f800000u | (u >> 9)) - 1.0f;
  float x = u01 * scale;
  x = x + 0.0f;
  int idx = 0;
  while (idx < 6 && cum[idx] <= x) idx++;
  return idx;
}

// ------------------------------------------------------------------------------------------
// 8-lane-group versions used by the step kernel's move phase: same arithmetic, spread over the
// lanes of one game (`sub` = lane within the group, `gbase` = first lane of the group).
// ------------------------------------------------------------------------------------------
C4_DEV float grp_shfl(float v, int src) { return __shfl(v, src, 64); }
C4_DEV uint32_t grp_shfl(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }

// mcts.rs:439-454: lane c < 7 evaluates column c; sums run left to right over broadcast values,
// exactly as the scalar version.  Every lane returns the full out[7].
C4_DEV void apply_temperature_group(const float* p, float t, float* out, uint32_t sub, int gbase) {
  bool all_eq = true;
  for (int i = 0; i < 7; i++) all_eq = all_eq && (p[i] == p[0]);
  if (t == 1.0f || all_eq || t == 0.0f) {  // the cheap branches stay scalar
    apply_temperature(p, t, out);
    return;
  }
  const float mine = p[sub < 7 ? sub : 6];
  const float pl = c4_logf(mine) / t;
  const float ex = c4_expf(pl);
  float s = 0.0f;
  for (int i = 0; i < 7; i++) s = s + grp_shfl(ex, gbase + i);
  const float lse = c4_logf(s);
  float v = c4_expf(pl - lse);
  if (v < 0.0f) v = 0.0f;
  if (v > 1.0f) v = 1.0f;
  for (int i = 0; i < 7; i++) out[i] = grp_shfl(v, gbase + i);
}

// rng_first_u32 with the ChaCha state spread over lanes 0..3 of the group: lane i holds column i
// (x_i, x_{4+i}, x_{8+i}, x_{12+i}); a column round is lane-local, a diagonal round rotates rows
// b, c, d by 1, 2, 3 lanes.  Lanes 4..7 compute along harmlessly.
C4_DEV uint32_t rng_first_u32_group(uint64_t state, uint32_t sub, int gbase) {
  const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
  const uint32_t i = sub & 3;
  uint32_t kb = 0, kc = 0;  // key words k[i] (row b) and k[4+i] (row c)
  for (uint32_t w = 0; w < 8; w++) {
    state = state * MUL + INC;
    const uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
    const uint32_t rot = (uint32_t)(state >> 59);
    const uint32_t k = (xs >> rot) | (xs << ((32 - rot) & 31));
    kb = (w == i) ? k : kb;
    kc = (w == 4 + i) ? k : kc;
  }
  const uint32_t c0 = i == 0 ? 0x61707865u : (i == 1 ? 0x3320646eu : (i == 2 ? 0x79622d32u : 0x6b206574u));
  uint32_t a = c0, b = kb, c = kc, d = 0;
  const int l1 = gbase + (int)((i + 1) & 3), l2 = gbase + (int)((i + 2) & 3), l3 = gbase + (int)((i + 3) & 3);
  for (int r = 0; r < 6; r++) {
    C4_QR(a, b, c, d)
    b = grp_shfl(b, l1); c = grp_shfl(c, l2); d = grp_shfl(d, l3);   // diagonals: (x0,x5,x10,x15) ...
    C4_QR(a, b, c, d)
    b = grp_shfl(b, l3); c = grp_shfl(c, l2); d = grp_shfl(d, l1);   // back to columns
  }
  return grp_shfl(a, gbase) + 0x61707865u;  // word 0 = column 0's `a` + its constant
}

// ------------------------------------------------------------------------------------------
// Dirichlet root noise: BUILD EXTENSION (named by BASELINE.json's north star, absent from the
// reference).  Specification = oracle/c4_oracle.c `c4o_dirichlet`; this must match it bit for bit.
// Kept out of line: it is a rare path and must not cost the step kernel registers.
// ------------------------------------------------------------------------------------------
struct NoiseStream {
  uint32_t key[8];
  uint32_t buf[16];
  uint64_t counter;
  int idx;
};

__device__ __attribute__((noinline)) void noise_refill(NoiseStream& st) {
  uint32_t x[16];
  x[0] = 0x61707865; x[1] = 0x3320646e; x[2] = 0x79622d32; x[3] = 0x6b206574;
  for (int i = 0; i < 8; i++) x[4 + i] = st.key[i];
  x[12] = (uint32_t)st.counter; x[13] = (uint32_t)(st.counter >> 32); x[14] = 0; x[15] = 0;
  uint32_t s0[16];
  for (int i = 0; i < 16; i++) s0[i] = x[i];
  for (int r = 0; r < 6; r++) {
    C4_QR(x[0], x[4], x[8], x[12]) C4_QR(x[1], x[5], x[9], x[13]) C4_QR(x[2], x[6], x[10], x[14]) C4_QR(x[3], x[7], x[11], x[15])
    C4_QR(x[0], x[5], x[10], x[15]) C4_QR(x[1], x[6], x[11], x[12]) C4_QR(x[2], x[7], x[8], x[13]) C4_QR(x[3], x[4], x[9], x[14])
  }
  for (int i = 0; i < 16; i++) st.buf[i] = x[i] + s0[i];
  st.counter += 1;
  st.idx = 0;
}

C4_DEV float noise_u01(NoiseStream& st) {
  if (st.idx == 16) noise_refill(st);
  const uint32_t u = st.buf[st.idx++];
  return ((float)(u >> 8) + 0.5f) * 0x1p-24f;
}

C4_DEV float noise_normal(NoiseStream& st) {  // Marsaglia polar method, second variate discarded
  for (;;) {
    const float a = 2.0f * noise_u01(st) - 1.0f;
    const float b = 2.0f * noise_u01(st) - 1.0f;
    float s = a * a;
    const float bb = b * b;
    s = s + bb;
    if (s > 0.0f && s < 1.0f) {
      float t = -2.0f * c4_logf(s);
      t = t / s;
      return a * __builtin_sqrtf(t);
    }
  }
}

C4_DEV float noise_gamma(NoiseStream& st, float alpha) {
  if (alpha == 1.0f) return -c4_logf(noise_u01(st));
  float boost = 1.0f, a = alpha;
  if (alpha < 1.0f) {
    const float lu = c4_logf(noise_u01(st));
    boost = c4_expf(lu / alpha);
    a = alpha + 1.0f;
  }
  const float d = a - (1.0f / 3.0f);
  const float c = 1.0f / __builtin_sqrtf(9.0f * d);
  for (;;) {
    const float x = noise_normal(st);
    float v = 1.0f + c * x;
    if (v <= 0.0f) continue;
    v = v * v * v;
    const float u = noise_u01(st);
    const float lhs = c4_logf(u);
    const float x2 = x * x;
    float rhs = 0.5f * x2;
    rhs = rhs + d;
    const float dv = d * v;
    rhs = rhs - dv;
    const float dlv = d * c4_logf(v);
    rhs = rhs + dlv;
    if (lhs < rhs) return (d * v) * boost;
  }
}

// eta[7] = Dir(alpha) over the legal columns (0 elsewhere) for (game_id, n_moves)
__device__ __attribute__((noinline)) void dirichlet_noise(uint64_t game_id, uint32_t n_moves, uint32_t legal, float alpha, float* eta) {
  NoiseStream st;
  uint64_t state = (game_id * (uint64_t)(42 + n_moves)) ^ 0x4469726963686C65ull;
  const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
  for (int i = 0; i < 8; i++) {
    state = state * MUL + INC;
    const uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
    const uint32_t rot = (uint32_t)(state >> 59);
    st.key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
  }
  st.counter = 0;
  st.idx = 16;
  float g[7], sum = 0.0f;
  for (int c = 0; c < 7; c++) {
    g[c] = ((legal >> c) & 1u) ? noise_gamma(st, alpha) : 0.0f;
    sum = sum + g[c];
  }
  // every Gamma draw underflowed (tiny alpha): fall back to the uniform point of the simplex
  const bool degenerate = !(sum > 0.0f) || __builtin_isinf(sum);
  const float uniform = 1.0f / (float)__popc(legal & 0x7Fu);
  for (int c = 0; c < 7; c++) eta[c] = ((legal >> c) & 1u) ? (degenerate ? uniform : g[c] / sum) : 0.0f;
}

// self_play.rs:294-299
C4_DEV float temperature_for_ply(uint32_t ply) { return ply < 4 ? 4.0f : (ply < 8 ? 2.0f : 1.0f); }

}  // namespace c4
