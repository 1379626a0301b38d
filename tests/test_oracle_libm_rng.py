"""Oracle: third-party arithmetic the path depends on.

* glibc 2.35 expf/logf (what Rust f32::exp / f32::ln call; mcts.rs:379,430,451-453): the
  restatement is swept against the HOST libm.  The default run strides the sweep; `-m slow`
  runs every bit pattern.
* rand 0.10.1 / chacha20 0.10.1 / rand_core 0.10.1 (Cargo.lock:1585-1593,269-277,1621-1622;
  call site mcts.rs:214-222).  The crates' source is not in /root/reference; every stage of
  `StdRng::seed_from_u64(s)` -> `WeightedIndex<f32>::sample` is pinned by a vector the crates
  publish in their own unit tests (value-stability tests: they exist so that these values do
  not change between releases): the ChaCha block (RFC 7539, eSTREAM), StdRng = ChaCha12 word
  order (`test_stdrng_construction`), the PCG32 seed expansion (`rand_core`'s
  `test_seed_from_u64` value-breakage constant) and UniformFloat / cumulative weights /
  partition_point (`WeightedIndex`'s `value_stability` under the crate's Pcg32 test
  generator).  The vectors are those of rand 0.8/0.9 and rand_core 0.6/0.9; a silent
  value-breaking change in 0.10.1 cannot be excluded without its source.
"""
import ctypes as C
import struct

import numpy as np
import pytest

from oracle import c4oracle as O


def _sweep(fn, lo, hi, stride):
    bad = C.c_uint32(0)
    n = fn(lo, hi, stride, C.byref(bad))
    return n, bad.value


def test_expf_strided_sweep_vs_host_libm():
    L = O.lib()
    # negative inputs down to -inf/NaN range, positive up to +inf/NaN range
    for lo, hi in [(0x80000000, 0xFFFFFFFF), (0x00000000, 0x7FFFFFFF)]:
        n, bad = _sweep(L.c4o_sweep_expf, lo, hi, 977)
        assert n == 0, hex(bad)
    # the two inputs where only the fused r = fma(InvLn2N, x, -kd) matches
    for x in (-float.fromhex("0x1.f8cbb2p+5"), float.fromhex("0x1.04845ep+5")):
        xi = struct.unpack("<I", struct.pack("<f", x))[0]
        assert _sweep(L.c4o_sweep_expf, xi, xi, 1)[0] == 0


def test_expf_dense_window_vs_host_libm():
    # softmax arguments live in [-30, 0]: sweep that window densely
    L = O.lib()
    lo = struct.unpack("<I", struct.pack("<f", -0.0))[0]
    hi = struct.unpack("<I", struct.pack("<f", -30.0))[0]
    n, bad = _sweep(L.c4o_sweep_expf, lo, hi, 13)
    assert n == 0, hex(bad)


def test_logf_strided_sweep_vs_host_libm():
    L = O.lib()
    n, bad = _sweep(L.c4o_sweep_logf, 0x00000000, 0xFFFFFFFF, 977)
    assert n == 0, hex(bad)
    # integer arguments (ln of visit counts, mcts.rs:379) and probabilities k/n
    xs = np.arange(0, 200001, dtype=np.float32)
    ys = np.empty_like(xs)
    L.c4o_host_logf(xs.ctypes.data_as(C.POINTER(C.c_float)), ys.ctypes.data_as(C.POINTER(C.c_float)), xs.size)
    mine = np.array([L.c4o_logf(float(x)) for x in xs[:5000]], dtype=np.float32)
    assert np.array_equal(mine.view(np.uint32), ys[:5000].view(np.uint32))


@pytest.mark.slow
def test_expf_logf_exhaustive_vs_host_libm():
    L = O.lib()
    assert _sweep(L.c4o_sweep_expf, 0, 0xFFFFFFFF, 1)[0] == 0
    assert _sweep(L.c4o_sweep_logf, 0, 0xFFFFFFFF, 1)[0] == 0


def test_special_values():
    L = O.lib()
    assert L.c4o_expf(float("-inf")) == 0.0 and L.c4o_expf(0.0) == 1.0
    assert L.c4o_logf(0.0) == float("-inf") and L.c4o_logf(1.0) == 0.0
    assert np.isnan(L.c4o_logf(-1.0)) and np.isnan(L.c4o_expf(float("nan")))


# ---- RNG -------------------------------------------------------------------------------------
def _words_to_bytes(ws):
    return b"".join(struct.pack("<I", w) for w in ws)


def test_chacha20_rfc7539_zero_key_block():
    # RFC 7539 section 2.3.2-style all-zero key/nonce/counter keystream block
    blk = O.chacha_block(bytes(32), 0, 20)
    assert _words_to_bytes(blk)[:16].hex() == "76b8e0ada0f13d90405d6ae55386bd28"
    assert _words_to_bytes(blk).hex() == (
        "76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
        "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")


def test_chacha12_estream_zero_key_block():
    blk = O.chacha_block(bytes(32), 0, 12)
    assert _words_to_bytes(blk)[:32].hex() == "9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f"


def test_chacha_counter_is_words_12_13():
    a = O.chacha_block(bytes(32), 1, 20)
    # RFC 7539 Appendix A.1 test vector #2: zero key, block counter 1
    assert _words_to_bytes(a)[:16].hex() == "9f07e7be5551387a98ba977c732d080d"


def test_stdrng_construction_vector_of_the_rand_crate():
    """rand's own unit test `test_stdrng_construction` (rand 0.8/0.9 src/rngs/std.rs; the crate is a
    third-party dependency whose source is NOT in /root/reference, so the vector is quoted from the
    published crate): StdRng::from_seed([1,0,0,0, 23,0,0,0, 200,1,0,0, 210,30,0,0, 0 x16]).next_u64()
    == 10719222850664546238, and StdRng::from_rng(that rng).next_u64() == 14064965282130556830.
    It pins: StdRng = ChaCha12, output = words 0,1,... of block 0 little-endian, and that from_rng
    consumes the next 32 bytes of the stream as the new seed."""
    seed = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16)
    w = O.chacha_block(seed, 0, 12)
    assert w[0] | (w[1] << 32) == 10719222850664546238
    seed1 = b"".join(struct.pack("<I", x) for x in w[2:10])
    w1 = O.chacha_block(seed1, 0, 12)
    assert w1[0] | (w1[1] << 32) == 14064965282130556830
    for rounds in (8, 20):   # the vector distinguishes the round count
        assert O.chacha_block(seed, 0, rounds)[0] != w[0]


SEED_TABLE = [  # SURVEY Appendix A.1 (independent scratch restatement): seed -> key, first u32, u01
    (0, "ecf273f981b5cd4587f0467306ad6cadd0d0a3e33317e767f29bea72d78a7dfe", 0xCD2C6F7F, 0.8014591932296753),
    (1, "ead81d725d26104e899c3bf842ce782ebad303da9997d2c2120256ac7366fb1b", 0xD3301861, 0.8249526023864746),
    (42, "a48fa17b58323d0aeab8a1cc690114b82b8cc87518b4f7548d446ea1e4df20f2", 0x222724A2, 0.13340973854064941),
    (43, "229d6fa798b1d804d9b7592c388dcfd97283da0efd1677faf49aa9925a051bb3", 0xEC8A28F6, 0.923983097076416),
    (86, "f2e4cf3b78b70c603ba6adfa48db318b943946f9b794b57548ec3cc81eecb70c", 0xF57683D2, 0.9588395357131958),
]


def test_seed_expansion_and_first_word_regression():
    L = O.lib()
    for seed, key_hex, first, u01 in SEED_TABLE:
        assert O.seed_key(seed).hex() == key_hex
        assert L.c4o_rng_first_u32(seed) == first
        got = np.frombuffer(struct.pack("<I", 0x3F800000 | (first >> 9)), dtype=np.float32)[0] - np.float32(1.0)
        assert float(got) == u01


def test_seed_from_u64_value_breakage_constant_of_rand_core():
    """rand_core's own unit test `test_seed_from_u64` (rand_core 0.5-0.9 src/lib.rs) ends with a
    "value-breakage test": a SeedableRng with an 8-byte seed, read little-endian, gives
    seed_from_u64(0) == 5029875928683246316.  It pins the expansion the oracle (and the device)
    use: PCG32 with MUL 6364136223846793005 / INC 11634580027462260723, the state advanced BEFORE
    each output, XSH-RR output, words copied little-endian in order."""
    key = O.seed_key(0)
    assert int.from_bytes(key[:8], "little") == 5029875928683246316


class _Pcg32:
    """rand_pcg::Pcg32 = Lcg64Xsh32 (the generator behind rand's `crate::test::rng(seed)`,
    `Pcg32::new(seed, 11634580027462260723)`): needed only to replay the crate's vectors."""
    MUL, M64 = 6364136223846793005, (1 << 64) - 1

    def __init__(self, state, stream):
        self.inc = ((stream << 1) | 1) & self.M64
        self.state = (state + self.inc) & self.M64
        self._step()

    def _step(self):
        self.state = (self.state * self.MUL + self.inc) & self.M64

    def next_u32(self):
        st = self.state
        self._step()
        xs = (((st >> 18) ^ st) >> 27) & 0xFFFFFFFF
        rot = st >> 59
        return ((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xFFFFFFFF


def test_weighted_index_value_stability_vector_of_the_rand_crate():
    """rand's own `value_stability` test of WeightedIndex (rand 0.8/0.9
    src/distributions/weighted_index.rs): with `crate::test::rng(701)`, ten samples of
    WeightedIndex::new([0.7f32, 0.1, 0.1, 0.1]) are [0, 0, 0, 1, 0, 0, 2, 3, 0, 0].  The oracle's
    c4o_weighted_index consumes one next_u32 per sample, so this pins UniformFloat<f32>::new (the
    scale loop), ::sample ((u >> 9 | exponent 0) - 1.0) * scale + low, the left-to-right f32
    cumulative sums and partition_point(w <= x).  Trailing zero weights change neither the total
    nor the chosen index."""
    rng = _Pcg32(701, 11634580027462260723)
    w = [0.7, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0]
    got = [O.weighted_index(w, rng.next_u32()) for _ in range(10)]
    assert got == [0, 0, 0, 1, 0, 0, 2, 3, 0, 0]
    # the same generator replays the crate's f64 vector through a plain-Python restatement of the
    # same algorithm (next_u64 = low word first): a check of _Pcg32 itself
    rng = _Pcg32(701, 11634580027462260723)
    cum, total, out = [1.0, 1.0 + 0.999, 1.0 + 0.999 + 0.998], 1.0 + 0.999 + 0.998 + 0.997, []
    scale = total
    while scale * (1.0 - 2.0 ** -52) + 0.0 >= total:
        scale = struct.unpack("<d", struct.pack("<Q", struct.unpack("<Q", struct.pack("<d", scale))[0] - 1))[0]
    for _ in range(10):
        lo = rng.next_u32()
        u = (rng.next_u32() << 32) | lo
        x = (struct.unpack("<d", struct.pack("<Q", 0x3FF0000000000000 | (u >> 12)))[0] - 1.0) * scale + 0.0
        out.append(sum(1 for c in cum if c <= x))
    assert out == [2, 2, 1, 3, 2, 1, 3, 3, 2, 1]


def test_slice_shuffle_value_stability_vectors_of_the_rand_crate():
    """rand's own `value_stability_slice` test (rand 0.9 src/seq/slice.rs; third-party source not in /root/reference, vector
    quoted from the published crate): with `crate::test::rng(414)`, `[0, 1, .., 12].shuffle(&mut r)` gives
    [5, 11, 0, 8, 7, 12, 6, 4, 9, 3, 1, 2, 10], and on the SAME generator `[0..=12].partial_shuffle(&mut r, 6)` then returns
    ([7, 12, 6, 8, 1, 9], [0, 11, 2, 3, 4, 5, 10]).  Thirteen elements: positions 1..11 share the first index chunk (one u32 below
    12!), position 12 opens the second (below 13 * .. * 19), and the partial shuffle starts its chooser at n = 7 -- so the vectors
    pin the direction of the walk, the chunk bounds, the % / order of peeling, and `random_range(..bound)` on u32 (Canon's method:
    both results depend on the high word of next_u32 * bound).  This is what split_train_test's permutation rests on
    (pybridge.rs:110-112)."""
    rng = _Pcg32(414, 11634580027462260723)
    assert O.partial_shuffle_with(list(range(13)), 13, rng.next_u32) == [5, 11, 0, 8, 7, 12, 6, 4, 9, 3, 1, 2, 10]
    out = O.partial_shuffle_with(list(range(13)), 6, rng.next_u32)
    assert out[7:] == [7, 12, 6, 8, 1, 9] and out[:7] == [0, 11, 2, 3, 4, 5, 10]


def test_shuffle_games_is_a_permutation_and_a_function_of_the_seed():
    """c4o_shuffle_games = the same walk on StdRng::seed_from_u64(seed) (ChaCha12 words in order): a permutation, reproducible,
    seed-dependent, identity for fewer than two games (shuffle() returns before drawing), and a prefix property of the chooser:
    the first chunk serves positions 1..11 whatever the length, so lists of 12 and 13 games agree on where games 0..11 went
    relative to each other only up to the last swap -- checked against a plain-Python walk fed by the oracle's own ChaCha words."""
    for n in (0, 1):
        assert O.shuffle_games(99, n).tolist() == list(range(n))
    a, b = O.shuffle_games(1337, 1000), O.shuffle_games(1337, 1000)
    assert a.tolist() == b.tolist() and sorted(a.tolist()) == list(range(1000)) and a.tolist() != O.shuffle_games(1338, 1000).tolist()
    # an independent walk in Python over the same word stream
    for seed, n in ((0, 2), (1337, 13), (1337, 14), ((1 << 64) - 1, 300), (42, 5000)):
        key = O.seed_key(seed)
        words, blk = [], 0

        def next_u32():
            nonlocal blk
            if not words:
                words.extend(O.chacha_block(key, blk, 12))
                blk += 1
            return words.pop(0)

        def below(bound):
            m = next_u32() * bound
            hi, lo = m >> 32, m & 0xFFFFFFFF
            if lo > ((-bound) & 0xFFFFFFFF):
                hi += (lo + ((next_u32() * bound) >> 32)) >> 32
            return hi

        items, chunk, left = list(range(n)), 0, 1
        for i in range(n):
            if left == 0:
                product, nxt = i + 1, i + 2
                while product * nxt <= 0xFFFFFFFF:
                    product, nxt = product * nxt, nxt + 1
                chunk, left = below(product), nxt - (i + 1)
            left -= 1
            if left == 0:
                j = chunk
            else:
                j, chunk = chunk % (i + 1), chunk // (i + 1)
            items[i], items[j] = items[j], items[i]
        assert O.shuffle_games(seed, n).tolist() == items, (seed, n)


def test_weighted_index_semantics():
    # x = u01 * scale; index = #cumulative weights <= x  (partition_point)
    w = [0.25, 0.25, 0.0, 0.25, 0.0, 0.0, 0.25]
    assert O.weighted_index(w, 0) == 0
    assert O.weighted_index(w, 0x40000000) == 1          # u01 = 0.25 -> cum[0] <= x -> 1
    assert O.weighted_index(w, 0x80000000) == 3          # 0.5 -> skips the zero-weight column 2
    assert O.weighted_index(w, 0xFFFFFFFF) == 6          # top of the range lands in the last non-zero weight
    one_hot = [0, 0, 0, 0, 1.0, 0, 0]
    for u in (0, 1 << 31, 0xFFFFFFFF):
        assert O.weighted_index(one_hot, u) == 4
    for bad in ([0.0] * 7, [0.5, -0.1, 0, 0, 0, 0, 0.6], [float("nan")] + [0.1] * 6):
        with pytest.raises(ValueError):
            O.weighted_index(bad, 123)


def test_sample_move_seed_formula():
    # mcts.rs:215 seed = game_id * (42 + n_moves): game 0 always seeds 0; 43*42 == 42*43 collide
    p = [1 / 7] * 7
    assert O.sample_move(0, 0, p, 1.0) == O.sample_move(0, 5, p, 1.0)
    assert O.sample_move(43, 0, p, 1.0) == O.sample_move(42, 1, p, 1.0)
    cols = {O.sample_move(g, 0, p, 4.0) for g in range(200)}
    assert cols == set(range(7))
    # zero-probability moves are never sampled, at any temperature
    q = [0.0, 0.5, 0.0, 0.5, 0.0, 0.0, 0.0]
    for g in range(300):
        assert O.sample_move(g, 3, q, 2.0) in (1, 3)
