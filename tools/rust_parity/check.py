#!/usr/bin/env python3
"""Compares the output of the reference-side test (parity_kit.rs, run with cargo in the reference's rust/ directory)
with what this repository's restatements answer.

    python tools/rust_parity/check.py out.txt [--live]

Default: against the committed `expected.txt`.  --live: the expected lines are recomputed now from the oracle
(oracle/libc4oracle.so) and c4a0_amd/results.py instead of read from the file.

Exit code 0 when every `rng`, `move`, `cbor` and `split` line agrees (the split's train count AND game order:
`split_train_test` restates rand's slice shuffle since round 6, include/c4a0_hip.h c4_shuffle_games)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def parse(lines):
    out = {}
    for l in lines:
        l = l.strip()
        if not l or l.split(" ", 1)[0] not in ("c4a0-parity-kit", "rng", "move", "cbor", "split"):
            continue   # cargo's own chatter
        if l.startswith("c4a0-parity-kit"):
            out["version"] = l
            continue
        key, _, val = l.partition(" -> ") if " -> " in l else (l.split(" ")[0], "", l.split(" ", 1)[1])
        out[key] = val
    return out


def main(argv):
    args = [a for a in argv if not a.startswith("--")]
    if len(args) != 1:
        print(__doc__)
        return 2
    got = parse(open(args[0]).read().splitlines())
    if "--live" in argv:
        sys.path.insert(0, HERE)
        import gen_kit
        want = parse(gen_kit.expected_lines())
    else:
        want = parse(open(os.path.join(HERE, "expected.txt")).read().splitlines())
    if got.get("version") != want.get("version"):
        print(f"kit version mismatch: output says {got.get('version')!r}, expected {want.get('version')!r}")
        return 2
    bad = ok = 0
    for key, w in want.items():
        if key == "version":
            continue
        g = got.get(key)
        kind = key.split(" ")[0]
        if g is None:
            print(f"MISSING  {key}")
            bad += 1
        elif g == w:
            ok += 1
        else:
            what = {"rng": "StdRng::seed_from_u64 output words", "move": "tempered policy / sampled column", "cbor": "serde_cbor bytes",
                    "split": "train count or order"}[kind]
            print(f"DIFFERS ({what})  {key}\n    reference: {g}\n    here:      {w}")
            bad += 1
    extra = [k for k in got if k not in want]
    for k in extra:
        print(f"UNEXPECTED line in the output: {k}")
    print(f"{ok} lines agree, {bad + len(extra)} failures")
    return 0 if bad == 0 and not extra else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
