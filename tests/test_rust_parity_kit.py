"""tools/rust_parity: the kit a maintainer with cargo uses to pin move sampling, the CBOR wire format and the split
against the real rand 0.10.1 / serde_cbor 0.11.2.  Here: the committed, generated files are what the
oracle and results.py answer today, the Rust text carries the same cases, and the checker tells agreement from
disagreement."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIT = os.path.join(ROOT, "tools", "rust_parity")
sys.path.insert(0, KIT)


def test_generated_files_are_current():
    import gen_kit

    assert open(os.path.join(KIT, "expected.txt")).read() == "\n".join(gen_kit.expected_lines()) + "\n"
    assert open(os.path.join(KIT, "parity_kit.rs")).read() == gen_kit.rust_source()


def test_rust_text_and_expected_file_hold_the_same_cases():
    import gen_kit

    rs = open(os.path.join(KIT, "parity_kit.rs")).read()
    exp = [l for l in open(os.path.join(KIT, "expected.txt")).read().splitlines()]
    moves = [l for l in exp if l.startswith("move ")]
    assert len(moves) == len(gen_kit.move_cases()) == rs.count("\n    (") and len(moves) >= 100
    for gid in ("0", "42", "43", "18446744073709551615"):           # game ids 0, 42, 43 (colliding seeds) and 2^64 - 1 (wrapping seed)
        assert f"    ({gid}, " in rs
    for l in moves[:5] + moves[-5:]:
        gid, n_moves, t_bits, pol = l.split(" -> ")[0].split(" ")[1:5]
        assert f"({gid}, {n_moves}, 0x{t_bits}, [{', '.join('0x' + p for p in pol.split(','))}])" in rs
    assert "serde_cbor::to_vec" in rs and "results.shuffle(&mut rng)" in rs and "game_id.wrapping_mul" in rs
    assert sum(l.startswith("rng ") for l in exp) == len(gen_kit.SEEDS) and sum(l.startswith("split ") for l in exp) == len(gen_kit.SPLITS)


def _check(path, *flags):
    return subprocess.run([sys.executable, os.path.join(KIT, "check.py"), path, *flags], capture_output=True, text=True)


def test_checker_accepts_agreement_and_names_disagreement(tmp_path):
    exp = open(os.path.join(KIT, "expected.txt")).read()
    ok = tmp_path / "ok.txt"
    ok.write_text("   Compiling c4a0_rust v0.1.0\nrunning 1 test\n" + exp + "test parity_kit::print_parity_lines ... ok\n")
    r = _check(str(ok))
    assert r.returncode == 0 and " 0 failures" in r.stdout, r.stdout
    assert _check(str(ok), "--live").returncode == 0
    lines = exp.splitlines()
    i = next(k for k, l in enumerate(lines) if l.startswith("move 42 "))
    wrong = lines[:]
    wrong[i] = wrong[i][:-1] + str((int(wrong[i][-1]) + 1) % 7)     # another column
    bad = tmp_path / "bad.txt"
    bad.write_text("\n".join(wrong) + "\n")
    r = _check(str(bad))
    assert r.returncode == 1 and "DIFFERS (tempered policy / sampled column)" in r.stdout
    # the split's game order is rand's slice shuffle restated (round 6): another order with the same train count is a failure too
    j = next(k for k, l in enumerate(lines) if l.startswith("split 5 "))
    head, order = lines[j].rsplit(" ", 1)
    dev = lines[:]
    dev[j] = head + " " + ",".join(reversed(order.split(",")))
    devf = tmp_path / "dev.txt"
    devf.write_text("\n".join(dev) + "\n")
    r = _check(str(devf))
    assert r.returncode == 1 and "DIFFERS (train count or order)" in r.stdout
    missing = tmp_path / "missing.txt"
    missing.write_text("\n".join(l for l in lines if not l.startswith("cbor ")) + "\n")
    assert _check(str(missing)).returncode == 1
