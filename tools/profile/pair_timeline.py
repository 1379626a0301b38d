#!/usr/bin/env python3
"""Reduce a rocprofv3 kernel trace of the bench to the paired graph's steady-state schedule: per stream (queue) the
sequence tower / GEMMs / output + step with start offsets inside a round, the gaps between consecutive kernels of a
session, and how much of every kernel ran beside which kernel of the other session."""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    if "conv_tower" in n: return "T"
    if "head_gemm" in n: return "G"
    if "out_step" in n: return "S"
    return None
ev = []
for r in rows:
    k = short(r["Kernel_Name"])
    if k is None: continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id", "?"), r.get("Stream_Id", r.get("Queue_Id", "?"))))
ev.sort()
# keep the last third (steady state of the timed region)
ev = ev[len(ev) * 2 // 3:]
byq = collections.defaultdict(list)
for e in ev: byq[e[3]].append(e)
print("kernels per queue:", {q: len(v) for q, v in byq.items()})
for q, v in byq.items():
    # split into rounds at every T
    rounds, cur = [], []
    for e in v:
        if e[2] == "T" and cur: rounds.append(cur); cur = []
        cur.append(e)
    rounds = [r for r in rounds if [x[2] for x in r] == ["T", "G", "G", "G", "S"]]
    if not rounds: continue
    n = len(rounds)
    dur = [sum(r[i][1] - r[i][0] for r in rounds) / n / 1e3 for i in range(5)]
    gap = [sum(r[i + 1][0] - r[i][1] for r in rounds) / n / 1e3 for i in range(4)]
    period = (rounds[-1][0][0] - rounds[0][0][0]) / (n - 1) / 1e3 if n > 1 else 0
    print(f"queue {q}: {n} rounds, period {period:.1f} us; durations T {dur[0]:.1f} W {dur[1]:.1f} N {dur[2]:.1f} N {dur[3]:.1f} S {dur[4]:.1f} (sum {sum(dur):.1f}); "
          f"gaps T-W {gap[0]:.1f} W-N {gap[1]:.1f} N-N {gap[2]:.1f} N-S {gap[3]:.1f}; S-next T {period - sum(dur) - sum(gap):.1f}")
qs = list(byq)
if len(qs) >= 2:
    a, b = byq[qs[0]], byq[qs[1]]
    names = {}
    for q in qs[:2]:
        i = 0
        for e in byq[q]:
            if e[2] == "T": i = 0
            names[e] = e[2] + (str(i) if e[2] == "G" else ""); i += e[2] == "G"
    ov = collections.Counter(); tot = collections.Counter()
    j0 = 0
    for e in a:
        tot[names[e]] += e[1] - e[0]
        for f in b:
            if f[1] <= e[0]: continue
            if f[0] >= e[1]: break
            ov[(names[e], names[f])] += min(e[1], f[1]) - max(e[0], f[0])
    print("share of a kernel's time (first queue) that ran beside a kernel of the other queue:")
    for k in ["T", "G0", "G1", "G2", "S"]:
        print("  ", k, "  ".join(f"{o}: {100 * ov[(k, o)] / max(1, tot[k]):.0f}%" for o in ["T", "G0", "G1", "G2", "S"]),
              f"  alone: {100 * (1 - sum(ov[(k, o)] for o in ['T', 'G0', 'G1', 'G2', 'S']) / max(1, tot[k])):.0f}%")
    # one round of the first queue with everything the other queue ran meanwhile
    rs = [e for e in a if e[2] == "T"]
    if len(rs) > 4:
        t0, t1 = rs[-3][0], rs[-2][0]
        print("one round (us from the first queue's tower start):")
        for e in sorted([x for x in a + b if t0 <= x[0] < t1 + 1]):
            print(f"   q{qs.index(e[3])} {names[e]:3s} {1e-3 * (e[0] - t0):7.1f} .. {1e-3 * (e[1] - t0):7.1f}")
