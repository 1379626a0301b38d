"""Oracle vs the reference's game-rule known-answer tests (rust/src/c4r.rs:474-653,
rust/proptest-regressions/c4r.txt).  Each test names the reference test it restates."""
import ctypes as C
import random

import numpy as np
import pytest

from oracle import c4oracle as O

NONE, PLAYER_WIN, OPPONENT_WIN, DRAW = 0, 1, 2, 3

DRAW_MOVES = [0, 1, 2, 3, 4, 5] * 3 + [5, 4, 3, 2, 1, 0] * 3 + [6] * 6  # c4r.rs:504-516


def test_playing_moves_works():  # c4r.rs:475-487
    pos = O.Pos(0, 0)
    L = O.lib()
    for col in range(7):
        for row in range(6):
            pos = O.make_move(pos, col)
            assert pos is not None
            assert L.c4o_get(C.byref(pos), row, col) == 0  # CellValue::Opponent after the flip
        assert not (O.legal_mask(pos) >> col) & 1
        assert O.make_move(pos, col) is None


def test_row_win():  # c4r.rs:490-495
    assert O.terminal_state(O.from_moves([0, 0, 1, 1, 2, 2, 3])) == OPPONENT_WIN


def test_col_win():  # c4r.rs:498-501
    assert O.terminal_state(O.from_moves([6, 0, 6, 0, 6, 0, 6])) == OPPONENT_WIN


def test_draw():  # c4r.rs:504-520
    pos = O.from_moves(DRAW_MOVES)
    assert O.terminal_state(pos) == DRAW
    assert O.terminal_value(pos, 0.01) == (DRAW, 0.0, 0.0)


def test_to_str():  # c4r.rs:523-546
    pos = O.from_moves(DRAW_MOVES)
    expected = [
        "🔵🔴🔵🔴🔵🔴🔵",
        "🔵🔴🔵🔴🔵🔴🔴",
        "🔵🔴🔵🔴🔵🔴🔵",
        "🔴🔵🔴🔵🔴🔵🔴",
        "🔴🔵🔴🔵🔴🔵🔵",
        "🔴🔵🔴🔵🔴🔵🔴",
    ]
    assert O.to_rows(pos) == expected
    assert O.from_rows(expected).key() == pos.key()


def _legal_str(pos):
    m = O.legal_mask(pos)
    return "".join("O" if (m >> c) & 1 else "X" for c in range(7))


def test_legal_moves():  # c4r.rs:549-579
    pos = O.Pos(0, 0)
    assert _legal_str(pos) == "OOOOOOO"
    pos = O.from_moves([0, 1, 2, 3, 4, 5] * 3 + [5, 4, 3, 2, 1, 0] * 2)
    assert _legal_str(pos) == "OOOOOOO"
    for col, want in [(5, "OOOOOXO"), (4, "OOOOXXO"), (3, "OOOXXXO"), (2, "OOXXXXO"), (1, "OXXXXXO"), (0, "XXXXXXO")]:
        pos = O.make_move(pos, col)
        assert _legal_str(pos) == want
    for _ in range(6):
        pos = O.make_move(pos, 6)
    assert _legal_str(pos) == "XXXXXXX"


def test_flip_h_symmetrical():  # c4r.rs:603-608
    pos = O.from_moves([3, 3, 3])
    assert O.flip_h(pos).key() == pos.key()
    assert O.flip_h(O.flip_h(pos)).key() == pos.key()


def random_pos(rng: random.Random) -> O.Pos:
    """c4r.rs:610-629 `random_pos` strategy: up to 500 random columns, illegal ones skipped,
    stop at a terminal position."""
    pos = O.Pos(0, 0)
    for _ in range(rng.randrange(0, 500)):
        if O.terminal_state(pos) != NONE:
            break
        mov = rng.randrange(7)
        if (O.legal_mask(pos) >> mov) & 1:
            pos = O.make_move(pos, mov)
    return pos


REGRESSION_POS = [  # rust/proptest-regressions/c4r.txt:7-8 (mask, value)
    (0b0000000000000000000000000000100000010000001000000100000010000101,
     0b0000000000000000000000000000000000010000000000000100000000000001),
    (0b0000000000000000000000000000100000010000001100000110000111000111,
     0b0000000000000000000000000000000000010000000000000100000101000101),
]


def test_flip_h_and_string_roundtrip_properties():  # c4r.rs:631-645 + regressions
    rng = random.Random(1337)
    cases = [O.Pos(m, v) for m, v in REGRESSION_POS] + [random_pos(rng) for _ in range(300)]
    for pos in cases:
        assert O.flip_h(O.flip_h(pos)).key() == pos.key()
        assert O.from_rows(O.to_rows(pos)).key() == pos.key()


def test_regression_positions_render():  # the rendering recorded in c4r.txt:7-8
    assert O.to_rows(O.Pos(*REGRESSION_POS[0])) == [
        "🔵⚫⚫⚫⚫⚫⚫", "🔴⚫⚫⚫⚫⚫⚫", "🔵⚫⚫⚫⚫⚫⚫", "🔴⚫⚫⚫⚫⚫⚫", "🔵⚫⚫⚫⚫⚫⚫", "🔴⚫🔵⚫⚫⚫⚫"]
    assert O.to_rows(O.Pos(*REGRESSION_POS[1])) == [
        "🔵⚫⚫⚫⚫⚫⚫", "🔴⚫⚫⚫⚫⚫⚫", "🔵⚫⚫⚫⚫⚫⚫", "🔴⚫⚫⚫⚫⚫🔵", "🔵🔴⚫⚫⚫⚫🔵", "🔴🔵🔴⚫⚫⚫🔴"]


def test_win_masks():  # c4r.rs:165-224: 24 horizontal + 21 vertical + 12 + 12 diagonal
    L = O.lib()
    masks = [L.c4o_win_mask(i) for i in range(69)]
    assert len(set(masks)) == 69
    assert all(bin(m).count("1") == 4 and m < (1 << 42) for m in masks)
    assert masks[0] == 0b1111 and masks[24] == (1 | 1 << 7 | 1 << 14 | 1 << 21)
    assert masks[45] == (1 | 1 << 8 | 1 << 16 | 1 << 24)          # diagonal up-right from (0,0)
    assert masks[57] == (1 << 21 | 1 << 15 | 1 << 9 | 1 << 3)     # diagonal down-right from (3,0)


def test_terminal_value_with_ply_penalty():  # c4r.rs:253-263
    pos = O.from_moves([0, 0, 1, 1, 2, 2, 3])
    t, qp, qn = O.terminal_value(pos, 0.01)
    assert t == OPPONENT_WIN
    assert qp == np.float32(-1.0) + np.float32(0.01) * np.float32(7.0) and qn == -1.0
    # PlayerWin only arises for hand-made positions (the mover already has four)
    pw = O.Pos(0b1111, 0b1111)
    t, qp, qn = O.terminal_value(pw, 0.01)
    assert t == PLAYER_WIN and qn == 1.0 and qp == np.float32(1.0) - np.float32(0.01) * np.float32(4.0)
    assert O.terminal_value(O.Pos(0, 0), 0.01)[0] == NONE


def test_planes_layout():  # c4r.rs:378-392 / pybridge.rs:202-221
    pos = O.from_moves([3, 3, 0])  # mover = player 1 (ply 3): own piece at (1,3); opponent at (0,3),(0,0)
    pl = O.planes(pos)
    assert pl.shape == (2, 6, 7) and pl.dtype == np.float32
    value_bits = np.array([(pos.value >> i) & 1 for i in range(42)], dtype=np.float32)
    opp_bits = np.array([((pos.mask & ~pos.value) >> i) & 1 for i in range(42)], dtype=np.float32)
    assert np.array_equal(pl[0].reshape(-1), value_bits)
    assert np.array_equal(pl[1].reshape(-1), opp_bits)
    assert pl[0, 1, 3] == 1 and pl[1, 0, 3] == 1 and pl[1, 0, 0] == 1 and pl.sum() == 3


def test_shift_and_terminal_equals_mask_scan():
    """The HIP kernels detect four-in-a-row with shifts instead of the 69-mask scan
    ('any method, result is boolean'); pin the equivalence on random boards."""
    rng = random.Random(7)
    L = O.lib()
    masks = [L.c4o_win_mask(i) for i in range(69)]
    NOT_COL = [0] * 4
    FULL = (1 << 42) - 1
    colmask = lambda c: sum(1 << (r * 7 + c) for r in range(6))
    # horizontal / diagonal shifts must not wrap around the 7-wide rows
    ok3 = FULL & ~(colmask(4) | colmask(5) | colmask(6))  # start columns 0..3

    def has4(x):
        h = x & (x >> 1) & (x >> 2) & (x >> 3) & ok3
        v = x & (x >> 7) & (x >> 14) & (x >> 21)
        d1 = x & (x >> 8) & (x >> 16) & (x >> 24) & ok3
        d2 = (x >> 21) & (x >> 15) & (x >> 9) & (x >> 3) & ok3  # down-right: start (row>=3, col<=3)
        return bool(h | v | d1 | d2)

    for _ in range(20000):
        x = rng.getrandbits(42)
        if rng.random() < 0.5:
            x &= rng.getrandbits(42)
        want = any(bin(x & m).count("1") == 4 for m in masks)
        assert has4(x) == want, hex(x)


def test_batch_helpers_equal_the_single_position_functions():
    """c4o_pos_ops_batch / c4o_random_positions (what the million-position GPU parity test runs against): the batch loop adds
    nothing of its own, and every generated position is reachable by legal play (value within mask, gravity, alternating counts)."""
    import ctypes as C
    n = 20000
    mask, value = O.random_positions_np(n, seed=5)
    col = np.random.default_rng(1).integers(-1, 8, size=n).astype(np.int32)
    wm, wv, wl, wt, wq = O.pos_ops_batch(mask, value, col, 0.01)
    L = O.lib()
    colmask = [sum(1 << (r * 7 + c) for r in range(6)) for c in range(7)]
    for i in range(n):
        m, v = int(mask[i]), int(value[i])
        assert v & ~m == 0 and m < (1 << 42)
        for c in range(7):                                   # gravity: a column's stones are contiguous from the bottom row
            h = bin(m & colmask[c]).count("1")
            assert m & colmask[c] == sum(1 << (r * 7 + c) for r in range(h))
        ply = bin(m).count("1")
        assert bin(v).count("1") == ply // 2                 # the side to move has made floor(ply / 2) moves
        p = O.Pos(m, v)
        assert wl[i] == L.c4o_legal_mask(C.byref(p))
        a, b = C.c_float(), C.c_float()
        assert wt[i] == L.c4o_terminal_value(C.byref(p), 0.01, C.byref(a), C.byref(b))
        if wt[i]:
            assert (wq[i, 0], wq[i, 1]) == (a.value, b.value)
        nx = O.Pos()
        ok = L.c4o_make_move(C.byref(p), int(col[i]), C.byref(nx))
        assert (int(wm[i]), int(wv[i])) == ((nx.mask, nx.value) if ok else (0, 0))
