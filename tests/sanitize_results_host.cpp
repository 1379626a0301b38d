/* AddressSanitizer / UBSan harness for the library's host-side result functions (c4a0_amd/csrc/c4_results_host.hip), compiled
 * with g++ on the CPU by tests/test_results_host_sanitized.py (sanitizers run on the CPU build only).  `pickle.loads` hands
 * c4_cbor_to_records bytes from a file: the decoder must refuse anything malformed without reading or writing outside its buffers.
 *   1. random results: size query, encode into a buffer of exactly that size, decode into tables of exactly the counted sizes,
 *      compare every field (game_id / meta of the records as the generator writes them);
 *   2. 300 000 damaged documents (byte flips, truncations, insertions, integers blown up): count-only and filling decodes, into
 *      tables sized the way results.py sizes them (len / 53 + 1 games, len / 59 + 1 records), each allocation exact;
 *   3. c4_shuffle_games: a permutation for every length up to 3 000, identity below two games. */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/c4a0_hip.h"

namespace c4host {
static std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
}  // namespace c4host

static uint64_t g_state = 0x243F6A8885A308D3ull;
static uint64_t rnd() {
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static uint64_t rnd_uint() {   /* every width serde_cbor packs to */
  switch (rnd() % 6) {
    case 0: return rnd() % 24;
    case 1: return rnd() % 256;
    case 2: return rnd() % 65536;
    case 3: return rnd() & 0xFFFFFFFFull;
    case 4: return rnd();
    default: return ~0ull;
  }
}
static float rnd_f32() {
  const float specials[] = {0.0f, -0.0f, 0.5f, 1.0f / 7.0f, 65504.0f, 65520.0f, 5.9604645e-08f, 2.9802322e-08f, 6.1035156e-05f, 1e-45f, 1.0f, -1.0f};
  const uint64_t k = rnd() % 8;
  if (k < 3) return specials[rnd() % (sizeof specials / sizeof specials[0])];
  if (k == 3) { const uint32_t u = 0x7F800000u | (uint32_t)(rnd() & 1) << 31 | (rnd() % 3 == 0 ? (uint32_t)rnd() & 0x7FFFFFu : 0u); float f; memcpy(&f, &u, 4); return f; }
  if (k == 4) { const uint32_t u = ((uint32_t)rnd() & 0xFFFFE000u); float f; memcpy(&f, &u, 4); return f; }   /* 10 mantissa bits: often a half */
  const uint32_t u = (uint32_t)rnd();
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static bool same_f32(float a, float b) {
  uint32_t x, y;
  memcpy(&x, &a, 4);
  memcpy(&y, &b, 4);
  if ((x & 0x7F800000u) == 0x7F800000u && (x & 0x7FFFFFu)) return (y & 0x7F800000u) == 0x7F800000u && (y & 0x7FFFFFu);   /* every NaN is written as 7e00 */
  return x == y;
}
#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, c4host::g_err.c_str()); return 1; } } while (0)

int main() {
  std::vector<uint8_t> keep;   /* one valid document for the mutation pass */
  for (int round = 0; round < 300; ++round) {
    const uint64_t n_games = rnd() % 9;
    c4_game_metadata* metas = (c4_game_metadata*)malloc(n_games * sizeof *metas + 1);
    uint32_t* counts = (uint32_t*)malloc(n_games * 4 + 1);
    uint64_t n_recs = 0;
    for (uint64_t g = 0; g < n_games; ++g) {
      metas[g] = {rnd_uint(), rnd_uint(), rnd_uint()};
      counts[g] = (uint32_t)(rnd() % 6);
      n_recs += counts[g];
    }
    c4_sample_rec* recs = (c4_sample_rec*)malloc(n_recs * sizeof *recs + 1);
    for (uint64_t k = 0; k < n_recs; ++k) {
      recs[k].game_id = 0; recs[k].meta = 0;
      recs[k].mask = rnd_uint(); recs[k].value = rnd_uint();
      for (int c = 0; c < 7; ++c) recs[k].policy[c] = rnd_f32();
      recs[k].q_penalty = rnd_f32(); recs[k].q_no_penalty = rnd_f32();
    }
    uint64_t size = 0, written = 0;
    REQUIRE(c4_records_to_cbor(metas, counts, n_games, recs, n_recs, nullptr, 0, &size) == C4_OK);
    uint8_t* doc = (uint8_t*)malloc(size);
    REQUIRE(c4_records_to_cbor(metas, counts, n_games, recs, n_recs, doc, size, &written) == C4_OK && written == size);
    if (size > 1) REQUIRE(c4_records_to_cbor(metas, counts, n_games, recs, n_recs, doc, size - 1, &written) == C4_ERR_BAD_ARG);
    uint64_t ng = 0, nr = 0;
    REQUIRE(c4_cbor_to_records(doc, size, nullptr, nullptr, 0, nullptr, 0, &ng, &nr) == C4_OK && ng == n_games && nr == n_recs);
    c4_game_metadata* m2 = (c4_game_metadata*)malloc(ng * sizeof *m2 + 1);
    uint32_t* c2 = (uint32_t*)malloc(ng * 4 + 1);
    c4_sample_rec* r2 = (c4_sample_rec*)malloc(nr * sizeof *r2 + 1);
    REQUIRE(c4_cbor_to_records(doc, size, m2, c2, ng, r2, nr, &ng, &nr) == C4_OK);
    if (nr) REQUIRE(c4_cbor_to_records(doc, size, m2, c2, ng, r2, nr - 1, &ng, &nr) == C4_ERR_BAD_ARG);   /* too little room is refused, not overrun */
    uint64_t k = 0;
    for (uint64_t g = 0; g < n_games; ++g) {
      REQUIRE(m2[g].game_id == metas[g].game_id && m2[g].player0_id == metas[g].player0_id && m2[g].player1_id == metas[g].player1_id && c2[g] == counts[g]);
      for (uint32_t i = 0; i < counts[g]; ++i, ++k) {
        REQUIRE(r2[k].mask == recs[k].mask && r2[k].value == recs[k].value && r2[k].game_id == metas[g].game_id);
        REQUIRE(r2[k].meta == (i | (i + 1 == counts[g] ? 1u << 16 : 0u)));
        for (int c = 0; c < 7; ++c) REQUIRE(same_f32(r2[k].policy[c], recs[k].policy[c]));
        REQUIRE(same_f32(r2[k].q_penalty, recs[k].q_penalty) && same_f32(r2[k].q_no_penalty, recs[k].q_no_penalty));
      }
    }
    if (n_recs >= 3 && keep.empty()) keep.assign(doc, doc + size);
    free(metas); free(counts); free(recs); free(doc); free(m2); free(c2); free(r2);
  }
  REQUIRE(!keep.empty());
  uint64_t accepted = 0, refused = 0;
  for (int it = 0; it < 300000; ++it) {
    std::vector<uint8_t> d = keep;
    const int edits = 1 + (int)(rnd() % 3);
    for (int e = 0; e < edits && !d.empty(); ++e) {
      const size_t at = rnd() % d.size();
      switch (rnd() % 6) {
        case 0: d[at] = (uint8_t)rnd(); break;
        case 1: d[at] ^= (uint8_t)(1u << (rnd() % 8)); break;
        case 2: d.resize(at); break;                                             /* truncated */
        case 3: d.insert(d.begin() + at, (uint8_t)rnd()); break;
        case 4: d[at] = (uint8_t)((d[at] & 0xE0) | (24 + rnd() % 8)); break;      /* a length / integer head blown up to 1-8 following bytes or an indefinite form */
        default: d.erase(d.begin() + at); break;
      }
    }
    uint8_t* doc = (uint8_t*)malloc(d.size() + (d.empty() ? 1 : 0));               /* exact: a read past the end is caught */
    if (!d.empty()) memcpy(doc, d.data(), d.size());
    uint64_t ng = 0, nr = 0;
    const int rc = c4_cbor_to_records(doc, d.size(), nullptr, nullptr, 0, nullptr, 0, &ng, &nr);
    const uint64_t cap_g = d.size() / 53 + 1, cap_r = d.size() / 59 + 1;
    c4_game_metadata* m2 = (c4_game_metadata*)malloc(cap_g * sizeof *m2);
    uint32_t* c2 = (uint32_t*)malloc(cap_g * 4);
    c4_sample_rec* r2 = (c4_sample_rec*)malloc(cap_r * sizeof *r2);
    uint64_t ng2 = 0, nr2 = 0;
    const int rc2 = c4_cbor_to_records(doc, d.size(), m2, c2, cap_g, r2, cap_r, &ng2, &nr2);
    REQUIRE(rc == rc2);                                                            /* the bounds results.py relies on hold for every accepted document */
    if (rc == C4_OK) { REQUIRE(ng == ng2 && nr == nr2 && ng <= cap_g && nr <= cap_r); ++accepted; } else { REQUIRE(rc == C4_ERR_BAD_ARG); ++refused; }
    free(doc); free(m2); free(c2); free(r2);
  }
  for (uint64_t n = 0; n <= 3000; n += (n < 40 ? 1 : 37)) {
    uint32_t* order = (uint32_t*)malloc(n * 4 + (n ? 0 : 1));
    REQUIRE(c4_shuffle_games(rnd(), n, order) == C4_OK);
    std::vector<uint8_t> seen(n, 0);
    for (uint64_t i = 0; i < n; ++i) { REQUIRE(order[i] < n && !seen[order[i]]); seen[order[i]] = 1; }
    if (n < 2) for (uint64_t i = 0; i < n; ++i) REQUIRE(order[i] == i);
    free(order);
  }
  REQUIRE(c4_shuffle_games(1, 0xFFFFFFFFull, nullptr) == C4_ERR_BAD_ARG);
  printf("sanitized ok: 300 round trips, %llu damaged documents accepted (still well-formed) and %llu refused\n", (unsigned long long)accepted, (unsigned long long)refused);
  return 0;
}
