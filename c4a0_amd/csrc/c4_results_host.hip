// c4_results_host.hip -- what PlayGamesResult does with finished games, on the packed 64-byte sample records (HOST code only:
// no kernel, no device call; the functions work on a machine without a GPU): the CBOR wire format and the train / test permutation.
//
// Replaces `serde_cbor::to_vec(self)` / `serde_cbor::from_slice(cbor)` of PlayGamesResult::to_cbor / from_cbor
// (reference rust/src/pybridge.rs:73-92; `__getstate__` / `__setstate__` are the same two calls, so this is what
// `pickle.dump(games, f)` of src/c4a0/training.py:62-63 runs every generation).  serde_cbor 0.11.2 (rust/Cargo.lock) writes the
// derive(Serialize) structs of types.rs:37-48,63-71,103-110 and c4r.rs:13-17 as definite-length maps keyed by field name in
// declaration order, unsigned integers in their shortest form, and an f32 as a half float whenever `f32::from(f16::from_f32(x))
// == x` (infinities and NaN always as halves 7c00 / fc00 / 7e00):
//
//   {"results": [ {"metadata": {"game_id": u, "player0_id": u, "player1_id": u},
//                  "samples": [ {"pos": {"mask": u, "value": u}, "policy": [f x 7], "q_penalty": f, "q_no_penalty": f}, ... ]}, ... ]}
//
// The generator hands over ~0.5 M samples per second of play; this codec runs at memory speed on one host core (tens of millions
// of samples per second), where the per-sample Python loop it replaces managed 25-37 k.
#include <cstdint>
#include <cstring>
#include <string>

#include "../../include/c4a0_hip.h"
#include "c4_host.hpp"

namespace {

// ---------------------------------------------------------------------------------------------------------------- encoder
inline uint8_t* put_head(uint8_t* p, unsigned major, uint64_t n) {
  const uint8_t m = static_cast<uint8_t>(major << 5);
  if (n < 24) { *p++ = m | static_cast<uint8_t>(n); return p; }
  if (n < (1ull << 8)) { *p++ = m | 24; *p++ = static_cast<uint8_t>(n); return p; }
  if (n < (1ull << 16)) { *p++ = m | 25; *p++ = static_cast<uint8_t>(n >> 8); *p++ = static_cast<uint8_t>(n); return p; }
  if (n < (1ull << 32)) {
    *p++ = m | 26;
    for (int s = 24; s >= 0; s -= 8) *p++ = static_cast<uint8_t>(n >> s);
    return p;
  }
  *p++ = m | 27;
  for (int s = 56; s >= 0; s -= 8) *p++ = static_cast<uint8_t>(n >> s);
  return p;
}

// text-string keys with their one-byte headers (all shorter than 24 bytes)
static const char K_RESULTS[] = "\x67" "results";
static const char K_METADATA[] = "\x68" "metadata";
static const char K_SAMPLES[] = "\x67" "samples";
static const char K_GAME_ID[] = "\x67" "game_id";
static const char K_PLAYER0[] = "\x6a" "player0_id";
static const char K_PLAYER1[] = "\x6a" "player1_id";
static const char K_POS[] = "\x63" "pos";
static const char K_POLICY[] = "\x66" "policy";
static const char K_QPEN[] = "\x69" "q_penalty";
static const char K_QNOPEN[] = "\x6c" "q_no_penalty";
static const char K_MASK[] = "\x64" "mask";
static const char K_VALUE[] = "\x65" "value";

template <size_t N>
inline uint8_t* put_key(uint8_t* p, const char (&k)[N]) {
  std::memcpy(p, k, N - 1);
  return p + (N - 1);
}

// Is the f32 with these bits exactly a half float?  If so *h = its bits.  (What `f32::from(f16::from_f32(x)) == x` decides for a
// finite x: the conversion rounds to nearest, so the round trip is the identity exactly for the representable values; -0.0 == 0.0
// holds and f16::from_f32 keeps the sign, so -0.0 is the half 8000.)
inline bool f32_is_half(uint32_t u, uint16_t* h) {
  const uint32_t sign = (u >> 16) & 0x8000u, e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
  if (e == 0) {                       // zero, or an f32 subnormal (< 2^-126: far below the smallest half 2^-24)
    if (m != 0) return false;
    *h = static_cast<uint16_t>(sign);
    return true;
  }
  const int ex = static_cast<int>(e) - 127;
  if (ex >= -14 && ex <= 15) {        // a normal half: 10 mantissa bits
    if (m & 0x1fffu) return false;
    *h = static_cast<uint16_t>(sign | static_cast<uint32_t>(ex + 15) << 10 | m >> 13);
    return true;
  }
  if (ex >= -24 && ex <= -15) {       // a subnormal half: a multiple of 2^-24
    const uint32_t full = m | 0x800000u;          // 1.m as a 24-bit integer, value = full * 2^(ex - 23)
    const int shift = -ex - 1;                    // half mantissa = full >> shift  (ex = -15 -> 14, ex = -24 -> 23)
    if (full & ((1u << shift) - 1u)) return false;
    *h = static_cast<uint16_t>(sign | full >> shift);
    return true;
  }
  return false;
}

inline uint8_t* put_f32(uint8_t* p, float x) {
  uint32_t u;
  std::memcpy(&u, &x, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) {   // serde_cbor: infinities by sign, every NaN as 7e00
    const uint16_t h = (u & 0x7fffffu) ? 0x7e00u : ((u >> 31) ? 0xfc00u : 0x7c00u);
    *p++ = 0xf9; *p++ = static_cast<uint8_t>(h >> 8); *p++ = static_cast<uint8_t>(h);
    return p;
  }
  uint16_t h;
  if (f32_is_half(u, &h)) {
    *p++ = 0xf9; *p++ = static_cast<uint8_t>(h >> 8); *p++ = static_cast<uint8_t>(h);
    return p;
  }
  *p++ = 0xfa;
  *p++ = static_cast<uint8_t>(u >> 24); *p++ = static_cast<uint8_t>(u >> 16); *p++ = static_cast<uint8_t>(u >> 8); *p++ = static_cast<uint8_t>(u);
  return p;
}

inline uint64_t head_len(uint64_t n) { return n < 24 ? 1 : n < (1ull << 8) ? 2 : n < (1ull << 16) ? 3 : n < (1ull << 32) ? 5 : 9; }

inline uint64_t f32_len(float x) {
  uint32_t u;
  std::memcpy(&u, &x, 4);
  uint16_t h;
  return ((u & 0x7f800000u) == 0x7f800000u || f32_is_half(u, &h)) ? 3 : 5;
}

// ---------------------------------------------------------------------------------------------------------------- decoder
struct Reader {
  const uint8_t* d;
  uint64_t n, i = 0;
  std::string err;

  bool fail(const std::string& what) {
    if (err.empty()) err = "invalid CBOR at byte " + std::to_string(i) + ": " + what;
    return false;
  }
  // head of a data item: major type, additional info, argument
  bool head(unsigned* major, unsigned* info, uint64_t* arg) {
    if (i >= n) return fail("truncated input");
    const uint8_t b = d[i++];
    *major = b >> 5;
    *info = b & 31u;
    if (*info < 24) { *arg = *info; return true; }
    if (*info > 27) return fail("unsupported additional info (indefinite lengths are not produced by serde_cbor here)");
    const unsigned len = 1u << (*info - 24);
    if (n - i < len) return fail("truncated input");
    uint64_t v = 0;
    for (unsigned k = 0; k < len; ++k) v = v << 8 | d[i++];
    *arg = v;
    return true;
  }
  bool length(unsigned want_major, uint64_t* out) {
    unsigned major, info;
    if (!head(&major, &info, out)) return false;
    if (major != want_major) return fail("expected CBOR major type " + std::to_string(want_major) + ", got " + std::to_string(major));
    return true;
  }
  bool uint(uint64_t* out) {
    unsigned major, info;
    if (!head(&major, &info, out)) return false;
    if (major != 0) return fail("expected unsigned integer");
    return true;
  }
  template <size_t N>
  bool key(const char (&k)[N]) {       // k = header byte + text
    if (n - i >= N - 1 && std::memcmp(d + i, k, N - 1) == 0) { i += N - 1; return true; }
    return fail(std::string("expected field \"") + (k + 1) + "\"");
  }
  bool map_of(uint64_t fields) {
    uint64_t got;
    if (!length(5, &got)) return false;
    if (got != fields) return fail("expected a map of " + std::to_string(fields) + " fields");
    return true;
  }
  bool f32(float* out) {
    unsigned major, info;
    uint64_t arg;
    if (!head(&major, &info, &arg)) return false;
    if (major == 0) { *out = static_cast<float>(arg); return true; }
    if (major == 7 && info == 25) {
      const uint32_t h = static_cast<uint32_t>(arg), sign = (h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
      uint32_t u;
      if (e == 31) u = sign | 0x7f800000u | m << 13;
      else if (e != 0) u = sign | (e + 112u) << 23 | m << 13;
      else if (m == 0) u = sign;
      else {                                       // subnormal half: m * 2^-24, exact in f32
        const float v = static_cast<float>(m) * 5.9604644775390625e-08f;
        std::memcpy(&u, &v, 4);
        u |= sign;
      }
      std::memcpy(out, &u, 4);
      return true;
    }
    if (major == 7 && info == 26) { const uint32_t u = static_cast<uint32_t>(arg); std::memcpy(out, &u, 4); return true; }
    if (major == 7 && info == 27) { double v; std::memcpy(&v, &arg, 8); *out = static_cast<float>(v); return true; }
    return fail("expected a float");
  }
};

}  // namespace

extern "C" {

// Two-call pattern: dst == NULL -> *n_written = the exact size of the document; otherwise it is written (cap >= that size).
int c4_records_to_cbor(const c4_game_metadata* metas, const uint32_t* counts, uint64_t n_games, const c4_sample_rec* recs,
                       uint64_t n_records, uint8_t* dst, uint64_t cap, uint64_t* n_written) {
  if ((n_games && (metas == nullptr || counts == nullptr)) || (n_records && recs == nullptr) || n_written == nullptr)
    return c4host::fail(C4_ERR_BAD_ARG, "c4_records_to_cbor: null argument");
  uint64_t total = 0;
  for (uint64_t g = 0; g < n_games; ++g) total += counts[g];
  if (total != n_records) return c4host::fail(C4_ERR_BAD_ARG, "c4_records_to_cbor: the counts sum to " + std::to_string(total) + " records, " + std::to_string(n_records) + " given");
  // the size first (also when writing: the destination is checked against it, not against a bound)
  constexpr uint64_t GAME_KEYS = 1 + 9 + 1 + 8 + 11 + 11 + 8;                  // map heads and keys of a game, without the integers
  constexpr uint64_t SAMPLE_KEYS = 1 + 4 + 1 + 5 + 6 + 7 + 1 + 10 + 13;        // ... of a sample, without the integers and floats
  uint64_t size = 1 + 8 + head_len(n_games) + n_games * GAME_KEYS + n_records * SAMPLE_KEYS;
  for (uint64_t g = 0; g < n_games; ++g)
    size += head_len(metas[g].game_id) + head_len(metas[g].player0_id) + head_len(metas[g].player1_id) + head_len(counts[g]);
  for (uint64_t k = 0; k < n_records; ++k) {
    const c4_sample_rec& r = recs[k];
    size += head_len(r.mask) + head_len(r.value) + f32_len(r.q_penalty) + f32_len(r.q_no_penalty);
    for (int c = 0; c < 7; ++c) size += f32_len(r.policy[c]);
  }
  if (dst == nullptr) { *n_written = size; return C4_OK; }
  if (cap < size) return c4host::fail(C4_ERR_BAD_ARG, "c4_records_to_cbor: the document takes " + std::to_string(size) + " bytes, room for " + std::to_string(cap));
  uint8_t* p = dst;
  *p++ = 0xa1;
  p = put_key(p, K_RESULTS);
  p = put_head(p, 4, n_games);
  const c4_sample_rec* r = recs;
  for (uint64_t g = 0; g < n_games; ++g) {
    *p++ = 0xa2;
    p = put_key(p, K_METADATA);
    *p++ = 0xa3;
    p = put_key(p, K_GAME_ID); p = put_head(p, 0, metas[g].game_id);
    p = put_key(p, K_PLAYER0); p = put_head(p, 0, metas[g].player0_id);
    p = put_key(p, K_PLAYER1); p = put_head(p, 0, metas[g].player1_id);
    p = put_key(p, K_SAMPLES);
    p = put_head(p, 4, counts[g]);
    for (uint32_t k = 0; k < counts[g]; ++k, ++r) {
      *p++ = 0xa4;
      p = put_key(p, K_POS);
      *p++ = 0xa2;
      p = put_key(p, K_MASK); p = put_head(p, 0, r->mask);
      p = put_key(p, K_VALUE); p = put_head(p, 0, r->value);
      p = put_key(p, K_POLICY);
      *p++ = 0x87;
      for (int c = 0; c < 7; ++c) p = put_f32(p, r->policy[c]);
      p = put_key(p, K_QPEN); p = put_f32(p, r->q_penalty);
      p = put_key(p, K_QNOPEN); p = put_f32(p, r->q_no_penalty);
    }
  }
  *n_written = static_cast<uint64_t>(p - dst);
  if (*n_written != size) return c4host::fail(C4_ERR_BAD_ARG, "c4_records_to_cbor: internal size mismatch");   // (cannot happen: one formula, two walks)
  return C4_OK;
}

int c4_cbor_to_records(const uint8_t* src, uint64_t len, c4_game_metadata* metas, uint32_t* counts, uint64_t cap_games,
                       c4_sample_rec* recs, uint64_t cap_records, uint64_t* n_games_out, uint64_t* n_records_out) {
  if ((src == nullptr && len) || n_games_out == nullptr || n_records_out == nullptr)
    return c4host::fail(C4_ERR_BAD_ARG, "c4_cbor_to_records: null argument");
  const bool fill = metas != nullptr || counts != nullptr || recs != nullptr;
  if (fill && (metas == nullptr || counts == nullptr || (recs == nullptr && cap_records)))
    return c4host::fail(C4_ERR_BAD_ARG, "c4_cbor_to_records: metas, counts and recs must be given together (all NULL = count only)");
  Reader r{src, len};
  uint64_t n_games = 0, n_recs = 0;
  bool ok = r.map_of(1) && r.key(K_RESULTS) && r.length(4, &n_games);
  if (ok && n_games > len) ok = r.fail("array longer than the input");          // a game takes > 1 byte: bounds the loops below
  if (ok && fill && n_games > cap_games) return c4host::fail(C4_ERR_BAD_ARG, "c4_cbor_to_records: " + std::to_string(n_games) + " games, room for " + std::to_string(cap_games));
  for (uint64_t g = 0; ok && g < n_games; ++g) {
    c4_game_metadata m{};
    uint64_t n_s = 0;
    ok = r.map_of(2) && r.key(K_METADATA) && r.map_of(3) && r.key(K_GAME_ID) && r.uint(&m.game_id) && r.key(K_PLAYER0) && r.uint(&m.player0_id) &&
         r.key(K_PLAYER1) && r.uint(&m.player1_id) && r.key(K_SAMPLES) && r.length(4, &n_s);
    if (ok && n_s > len) ok = r.fail("array longer than the input");
    if (ok && n_s > 0xffffffffull) ok = r.fail("more than 2^32 samples in a game");
    if (!ok) break;
    if (fill) {
      if (n_recs + n_s > cap_records) return c4host::fail(C4_ERR_BAD_ARG, "c4_cbor_to_records: more than the " + std::to_string(cap_records) + " records there is room for");
      metas[g] = m;
      counts[g] = static_cast<uint32_t>(n_s);
    }
    for (uint64_t k = 0; ok && k < n_s; ++k) {
      c4_sample_rec s{};
      uint64_t n_pol = 0;
      ok = r.map_of(4) && r.key(K_POS) && r.map_of(2) && r.key(K_MASK) && r.uint(&s.mask) && r.key(K_VALUE) && r.uint(&s.value) && r.key(K_POLICY) &&
           r.length(4, &n_pol);
      if (ok && n_pol != 7) ok = r.fail("policy must have 7 entries");
      for (int c = 0; ok && c < 7; ++c) ok = r.f32(&s.policy[c]);
      ok = ok && r.key(K_QPEN) && r.f32(&s.q_penalty) && r.key(K_QNOPEN) && r.f32(&s.q_no_penalty);
      if (ok && fill) {
        s.game_id = m.game_id;
        s.meta = static_cast<uint32_t>(k & 0xffffu) | (k + 1 == n_s ? 1u << 16 : 0u);   // index | terminal-sample flag, as the generator writes it
        recs[n_recs + k] = s;
      }
    }
    n_recs += n_s;
  }
  if (ok && r.i != len) ok = r.fail("trailing bytes");
  if (!ok) return c4host::fail(C4_ERR_BAD_ARG, r.err);
  *n_games_out = n_games;
  *n_records_out = n_recs;
  return C4_OK;
}

// ------------------------------------------------------------------------------------------------------ split_train_test's permutation
// `results.shuffle(&mut StdRng::seed_from_u64(seed))` (rust/src/pybridge.rs:110-112; rand 0.10.1): seed_from_u64 expands the seed with
// PCG32 into a ChaCha12 key, the generator's u32 stream is the words of blocks 0, 1, ... in order, and SliceRandom::shuffle is
// Durstenfeld from the bottom index up with the index draws batched: one u32 in [0, (n+1)(n+2)...(n+k)) serves k consecutive
// positions (the longest such product that fits a u32), peeled off by % and /; the u32 itself comes from Canon's widening-multiply
// method with one bias-reducing retry.  (The oracle's c4o_shuffle_games restates the same thing generically and is pinned by the
// crate's published value-stability vectors; tests compare the two.)
namespace {

struct ChaCha12Stream {
  uint32_t key[8];
  uint64_t counter = 0;
  uint32_t buf[16];
  int idx = 16;

  explicit ChaCha12Stream(uint64_t seed) {         // rand_core SeedableRng::seed_from_u64
    uint64_t state = seed;
    for (int i = 0; i < 8; ++i) {
      state = state * 6364136223846793005ull + 11634580027462260723ull;
      const uint32_t xs = static_cast<uint32_t>(((state >> 18) ^ state) >> 27), rot = static_cast<uint32_t>(state >> 59);
      key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
  }
  static inline uint32_t rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
  static inline void quarter(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
    a += b; d ^= a; d = rotl(d, 16);
    c += d; b ^= c; b = rotl(b, 12);
    a += b; d ^= a; d = rotl(d, 8);
    c += d; b ^= c; b = rotl(b, 7);
  }
  void refill() {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; ++i) s[4 + i] = key[i];
    s[12] = static_cast<uint32_t>(counter);
    s[13] = static_cast<uint32_t>(counter >> 32);
    s[14] = s[15] = 0;
    ++counter;
    uint32_t x[16];
    std::memcpy(x, s, sizeof x);
    for (int r = 0; r < 12; r += 2) {
      quarter(x[0], x[4], x[8], x[12]); quarter(x[1], x[5], x[9], x[13]); quarter(x[2], x[6], x[10], x[14]); quarter(x[3], x[7], x[11], x[15]);
      quarter(x[0], x[5], x[10], x[15]); quarter(x[1], x[6], x[11], x[12]); quarter(x[2], x[7], x[8], x[13]); quarter(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; ++i) buf[i] = x[i] + s[i];
    idx = 0;
  }
  uint32_t next_u32() {
    if (idx == 16) refill();
    return buf[idx++];
  }
  uint32_t below(uint32_t bound) {                  // random_range(..bound), bound > 0
    const uint64_t m = static_cast<uint64_t>(next_u32()) * bound;
    uint32_t hi = static_cast<uint32_t>(m >> 32);
    const uint32_t lo = static_cast<uint32_t>(m);
    if (lo > 0u - bound) {
      const uint32_t new_hi = static_cast<uint32_t>((static_cast<uint64_t>(next_u32()) * bound) >> 32);
      if (static_cast<uint64_t>(lo) + new_hi > 0xffffffffull) hi += 1;
    }
    return hi;
  }
};

}  // namespace

int c4_shuffle_games(uint64_t seed, uint64_t n_games, uint32_t* order) {
  if (n_games && order == nullptr) return c4host::fail(C4_ERR_BAD_ARG, "c4_shuffle_games: null output");
  if (n_games >= 0xffffffffull) return c4host::fail(C4_ERR_BAD_ARG, "c4_shuffle_games: at most 2^32 - 2 games");
  for (uint64_t i = 0; i < n_games; ++i) order[i] = static_cast<uint32_t>(i);
  if (n_games <= 1) return C4_OK;                  // shuffle() returns before touching the generator
  ChaCha12Stream rng(seed);
  uint32_t chunk = 0, left = 1;                    // position 0 always swaps with itself: no draw
  for (uint32_t i = 0; i < n_games; ++i) {         // the index for position i is uniform on [0, i]
    const uint32_t bound = i + 1;
    uint32_t index;
    if (left == 0) {                               // a new chunk: the longest bound (bound+1) ... that fits a u32
      uint32_t product = bound, next = bound + 1;
      for (;;) {
        const uint64_t p = static_cast<uint64_t>(product) * next;
        if (p > 0xffffffffull) break;
        product = static_cast<uint32_t>(p);
        ++next;
      }
      chunk = rng.below(product);
      left = next - bound;
    }
    if (--left == 0) {
      index = chunk;
    } else {
      index = chunk % bound;
      chunk /= bound;
    }
    const uint32_t t = order[i];
    order[i] = order[index];
    order[index] = t;
  }
  return C4_OK;
}

}  // extern "C"
