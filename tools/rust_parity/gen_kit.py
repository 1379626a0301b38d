#!/usr/bin/env python3
"""Generates the Rust parity kit: `parity_kit.rs` (a #[test] for the REFERENCE crate) and `expected.txt`
(what this repository's restatements answer for the same cases).

Why: the reference samples every move with rand 0.10.1 (`StdRng::seed_from_u64`, `WeightedIndex<f32>`;
rust/src/mcts.rs:214-222, rust/Cargo.lock), serialises results with serde_cbor 0.11.2 (rust/src/pybridge.rs:73-92)
and splits them with rand's slice shuffle (pybridge.rs:110-120).  None of these crates' sources is in
/root/reference and the build container has no Rust toolchain, so the restatements (oracle/c4_oracle.c
c4o_seed_from_u64 / c4o_chacha_block / c4o_weighted_index / c4o_sample_move / c4o_shuffle_games, the library's c4_records_to_cbor
and c4_shuffle_games behind c4a0_amd/results.py) are pinned by
the crates' published vectors of EARLIER versions only.  A maintainer with cargo closes the gap in
one command -- see README.md beside this file.

The case tables below are the single source of both files: run `python tools/rust_parity/gen_kit.py` after
changing them (tests/test_rust_parity_kit.py fails if the committed files are stale)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

U64 = (1 << 64) - 1

# ---- StdRng::seed_from_u64(seed): the first two u32 words --------------------------------------------------------------
SEEDS = [0, 1, 42, 43, 1337, 42 * 43, (1 << 32) + 5, 1 << 63, U64, (U64 * 42) & U64]


def _f32(x):
    return np.float32(x)


def _bits(x) -> int:
    return int(np.float32(x).view(np.uint32))


def _visits(v):
    """Node::policy (mcts.rs:396-412): child visit counts as f32, summed left to right, divided."""
    v = [np.float32(c) for c in v]
    s = np.float32(0)
    for c in v:
        s = np.float32(s + c)
    return [np.float32(c / s) for c in v]


def _schedule(n_moves):   # self_play.rs:283-301: temperature by ply
    return 4.0 if n_moves < 4 else 2.0 if n_moves < 8 else 1.0


POLICIES = {
    "uniform": [np.float32(1.0) / np.float32(7.0)] * 7,
    "typical": _visits([10, 12, 30, 20, 15, 8, 4]),
    "peaked": _visits([1, 0, 95, 2, 0, 1, 0]),
    "one_hot": _visits([0, 0, 0, 99, 0, 0, 0]),
    "two_full_columns": _visits([0, 40, 0, 30, 29, 0, 0]),
    "edge": _visits([50, 0, 0, 0, 0, 0, 49]),
    "n1400": _visits([120, 180, 310, 420, 200, 110, 59]),
}


def move_cases():
    cases = []
    for gid in (0, 1, 42, 43, 1337, (1 << 32) + 5, 1 << 63, U64):
        for n_moves in (0, 1, 3, 4, 7, 8, 20, 41):
            for name in ("typical", "peaked") if n_moves not in (0, 41) else ("uniform", "two_full_columns", "n1400"):
                cases.append((gid, n_moves, _schedule(n_moves), POLICIES[name]))
    # temperatures the schedule never produces, and the degenerate shapes
    for t in (0.0, 0.5, 1.0, 3.0):
        for name in ("typical", "one_hot", "edge", "uniform"):
            cases.append((7, 5, t, POLICIES[name]))
    return cases


# ---- one PlayGamesResult of two games for serde_cbor::to_vec ------------------------------------------------------------
# (moves from the empty board, per-sample policy, q_penalty, q_no_penalty); values chosen to hit every float width
# serde_cbor packs to (f16 when lossless, else f32) and every unsigned width (1, 2, 3, 5, 9 bytes)
CBOR_GAMES = [
    dict(meta=(23, 0, 300), moves=[3, 3, 4],
         samples=[(POLICIES["uniform"], 0.0, -0.0), (POLICIES["typical"], 0.5, -1.0), (POLICIES["one_hot"], 0.96875, 1.0),
                  (POLICIES["uniform"], -0.9599999785423279, 65504.0)]),
    dict(meta=(U64, 70000, (1 << 32) + 5), moves=[0, 6, 0, 6, 0, 6, 1],
         samples=[(POLICIES["peaked"], 1.0e-8, 0.333333343267), (POLICIES["edge"], -0.123456789, 2.0 ** -24)]),
]

# 13 and 14 games: the last position of the first index chunk (12!) and the first of the second; 100: several chunks
SPLITS = [(n, frac, seed) for n in (1, 2, 5, 10, 13, 14, 37, 100) for frac, seed in ((0.5, 1337), (0.8, 0))] + [(37, 0.5, U64)]


def _make_move(mask, value, col):   # c4r.rs:58-72 + invert :125-129
    for row in range(6):
        bit = 1 << (7 * row + col)
        if not mask & bit:
            mask |= bit
            value |= bit
            return mask, (~value) & mask
    raise ValueError("column full")


def _game_positions(moves):
    m = v = 0
    out = [(m, v)]
    for c in moves:
        m, v = _make_move(m, v, c)
        out.append((m, v))
    return out


def expected_lines():
    from c4a0_amd.results import GameMetadata, GameResult, PlayGamesResult, Sample
    from oracle import c4oracle as O

    lines = ["c4a0-parity-kit v1"]
    for s in SEEDS:
        w = O.chacha_block(O.seed_key(s), 0, 12)
        lines.append(f"rng {s} {w[0]:08x} {w[1]:08x}")
    for gid, n_moves, t, pol in move_cases():
        tempered = O.apply_temperature(pol, t)
        try:
            col = O.sample_move(gid, n_moves, pol, t)
        except ValueError:
            col = -1
        lines.append("move %d %d %08x %s -> %s %d" % (gid, n_moves, _bits(t), ",".join("%08x" % _bits(p) for p in pol),
                                                      ",".join("%08x" % _bits(p) for p in tempered), col))
    results = []
    for g in CBOR_GAMES:
        pos = _game_positions(g["moves"])
        samples = [Sample(pos[i][0], pos[i][1], [float(np.float32(p)) for p in pol], float(np.float32(qp)), float(np.float32(qn)))
                   for i, (pol, qp, qn) in enumerate(g["samples"])]
        results.append(GameResult(GameMetadata(*g["meta"]), samples))
    pgr = PlayGamesResult(results)
    lines.append("cbor " + pgr.to_cbor().hex())
    for n, frac, seed in SPLITS:
        games = [GameResult(GameMetadata(i, 0, 0), [Sample(i, 0, [0.0] * 7, 0.0, 0.0)]) for i in range(n)]
        train, test = PlayGamesResult(games).split_train_test(frac, seed)
        lines.append("split %d %08x %d -> %d %s" % (n, _bits(frac), seed, len(train), ",".join(str(s.mask) for s in train + test)))
    return lines


RUST_HEADER = '''//! c4a0 parity kit -- a test for the REFERENCE crate (advait/c4a0, rust/), generated by
//! tools/rust_parity/gen_kit.py of the MI355X self-play generator.  It prints what the crate's pinned
//! dependencies (rand 0.10.1 / chacha20 / rand_core, serde_cbor 0.11.2) really answer at the three places
//! where that generator restates them without access to their sources:
//!
//!   rng   `StdRng::seed_from_u64(seed)`: the first two u32 words
//!   move  rust/src/mcts.rs:214-222 line for line: seed = game_id * (42 + n_moves), `apply_temperature`,
//!         `WeightedIndex::new(policy).unwrap().sample(&mut rng)` (policy and temperature as f32 bit patterns;
//!         the tempered policy is printed too: it pins f32::ln / f32::exp)
//!   cbor  `serde_cbor::to_vec(&PlayGamesResult { .. })` of a fixed two-game result (= `to_cbor()`, pybridge.rs:73-76)
//!   split `results.shuffle(&mut StdRng::seed_from_u64(seed))` + the train count (pybridge.rs:110-116), on game indices
//!
//! Install:  cp parity_kit.rs <reference>/rust/src/  and add `#[cfg(test)] mod parity_kit;` to rust/src/lib.rs
//! Run:      cargo test --release parity_kit -- --nocapture | grep -E '^(c4a0-parity-kit|rng|move|cbor|split) ' > out.txt
//! Check:    python tools/rust_parity/check.py out.txt        (in the generator's repository)
//!
//! (`--release`: mcts.rs:215 multiplies u64 without wrapping_mul; the test itself wraps explicitly.)
use crate::c4r::Pos;
use crate::mcts::apply_temperature;
use crate::pybridge::PlayGamesResult;
use crate::types::{GameMetadata, GameResult, Policy, Sample};
use rand::{
    distr::{weighted::WeightedIndex, Distribution, StandardUniform},
    rngs::StdRng,
    seq::SliceRandom,
    SeedableRng,
};

fn pol(bits: &[u32; 7]) -> Policy {
    core::array::from_fn(|i| f32::from_bits(bits[i]))
}

fn hex7(p: &Policy) -> String {
    p.iter().map(|x| format!("{:08x}", x.to_bits())).collect::<Vec<_>>().join(",")
}
'''


def rust_source():
    out = [RUST_HEADER]
    out.append("const SEEDS: &[u64] = &[%s];\n" % ", ".join(str(s) for s in SEEDS))
    out.append("// (game_id, n_moves, temperature bits, policy bits)")
    out.append("const MOVES: &[(u64, usize, u32, [u32; 7])] = &[")
    for gid, n_moves, t, pol in move_cases():
        out.append("    (%d, %d, 0x%08x, [%s])," % (gid, n_moves, _bits(t), ", ".join("0x%08x" % _bits(p) for p in pol)))
    out.append("];\n")
    out.append("// (n_games, train_frac bits, seed)")
    out.append("const SPLITS: &[(usize, u32, u64)] = &[%s];\n" % ", ".join("(%d, 0x%08x, %d)" % (n, _bits(f), s) for n, f, s in SPLITS))
    out.append("fn cbor_result() -> PlayGamesResult {\n    let mut results = Vec::new();")
    for g in CBOR_GAMES:
        out.append("    {\n        let moves: &[usize] = &[%s];" % ", ".join(str(m) for m in g["moves"]))
        out.append("        let mut positions = vec![Pos::default()];")
        out.append("        for &m in moves {\n            let next = positions.last().unwrap().make_move(m).unwrap();\n            positions.push(next);\n        }")
        out.append("        let samples = vec![")
        for i, (pol, qp, qn) in enumerate(g["samples"]):
            out.append("            Sample { pos: positions[%d].clone(), policy: pol(&[%s]), q_penalty: f32::from_bits(0x%08x), q_no_penalty: f32::from_bits(0x%08x) },"
                       % (i, ", ".join("0x%08x" % _bits(p) for p in pol), _bits(qp), _bits(qn)))
        out.append("        ];")
        out.append("        results.push(GameResult { metadata: GameMetadata { game_id: %d, player0_id: %d, player1_id: %d }, samples });\n    }" % g["meta"])
    out.append("    PlayGamesResult { results }\n}\n")
    out.append('''#[test]
fn print_parity_lines() {
    println!("c4a0-parity-kit v1");
    for &s in SEEDS {
        let mut rng = StdRng::seed_from_u64(s);
        let a: u32 = StandardUniform.sample(&mut rng);
        let b: u32 = StandardUniform.sample(&mut rng);
        println!("rng {} {:08x} {:08x}", s, a, b);
    }
    for (game_id, n_moves, t_bits, p_bits) in MOVES.iter() {
        // mcts.rs:214-222
        let seed = game_id.wrapping_mul(((Pos::N_ROWS * Pos::N_COLS) + n_moves) as u64);
        let mut rng = StdRng::seed_from_u64(seed);
        let policy = pol(p_bits);
        let tempered = apply_temperature(&policy, f32::from_bits(*t_bits));
        let col: i64 = match WeightedIndex::new(tempered) {
            Ok(dist) => dist.sample(&mut rng) as i64,
            Err(_) => -1, // the reference unwrap()s: a panic
        };
        println!("move {} {} {:08x} {} -> {} {}", game_id, n_moves, t_bits, hex7(&policy), hex7(&tempered), col);
    }
    let cbor = serde_cbor::to_vec(&cbor_result()).unwrap();
    println!("cbor {}", cbor.iter().map(|b| format!("{:02x}", b)).collect::<String>());
    for &(n, frac_bits, seed) in SPLITS {
        // pybridge.rs:110-116 on game indices
        let mut rng = StdRng::seed_from_u64(seed);
        let mut results: Vec<usize> = (0..n).collect();
        results.shuffle(&mut rng);
        let n_train = (results.len() as f32 * f32::from_bits(frac_bits)).round() as usize;
        let order = results.iter().map(|i| i.to_string()).collect::<Vec<_>>().join(",");
        println!("split {} {:08x} {} -> {} {}", n, frac_bits, seed, n_train, order);
    }
}
''')
    return "\n".join(out)


def main():
    with open(os.path.join(HERE, "parity_kit.rs"), "w") as f:
        f.write(rust_source())
    with open(os.path.join(HERE, "expected.txt"), "w") as f:
        f.write("\n".join(expected_lines()) + "\n")
    print("wrote parity_kit.rs and expected.txt:", len(move_cases()), "move cases,", len(SEEDS), "seeds,", len(SPLITS), "splits")


if __name__ == "__main__":
    main()
