#!/usr/bin/env python3
"""The schedule of two sessions' kernels AS THE DEVICE RAN IT (BASELINE config 2, the paired graph of the bench).

rocprofv3's kernel trace serialises the two queues (every kernel runs alone under it: tools/profile/pair_timeline.sh shows
periods of 2 x the chain), so the stamps come from the kernels themselves: a library built with -DC4_TIMELINE
(c4_timeline.hpp) lets every workgroup of the tower, the GEMMs and the output + step launch record {kind, tag, block, CU,
start, end}.  This tool plays the bench's workload, records a few replays in steady state and prints, per session and
launch: start / end relative to the round, the span between the first workgroup's start and the last one's end, the
workgroups' own mean duration, how long the launch's workgroups had to wait for a CU (start spread), what ran beside it.

    python tools/build_variant.py tl WORK -DC4_TIMELINE
    C4A0_HIP_LIB=libc4a0_hip_tl.so python tools/pair_timeline.py [--sessions 2] [--offset-stage 1] [--rounds 8] [--npz out.npz]"""
import argparse, collections, ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=4096)
ap.add_argument("--n-mcts", type=int, default=100)
ap.add_argument("--blocks", type=int, default=4)
ap.add_argument("--channels", type=int, default=32)
ap.add_argument("--sessions", type=int, default=2)
ap.add_argument("--offset-stage", type=int, default=1)
ap.add_argument("--steps-per-graph", type=int, default=64)
ap.add_argument("--preroll", type=int, default=3000)
ap.add_argument("--rounds", type=int, default=6, help="rounds of the recorded replay to print")
ap.add_argument("--gemm-config", default=None)
ap.add_argument("--npz", default=None, help="save the raw records")
ap.add_argument("--from-npz", default=None, help="analyse saved records instead of running (no GPU needed)")
args = ap.parse_args()

if args.from_npz:
    rec = np.load(args.from_npz)["rec"]
    print(f"{len(rec)} workgroup records from {args.from_npz}")
else:
    from c4a0_amd._lib import lib
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
    from c4a0_amd.session import DeviceSession, capture_pair
    dev = torch.device("cuda:0")
    L = lib()
    torch.manual_seed(1337)
    net = InferenceNet(ConnectFourNet(ModelConfig(args.blocks, args.channels, 4, 2)), dev, dtype=torch.bfloat16, gemm_config=args.gemm_config)
    P, U = args.sessions, args.steps_per_graph
    sessions, streams = [], []
    n_games = args.games * 40
    for p in range(P):
        sp = DeviceSession((args.games + P - 1 - p) // P, args.n_mcts, 6.6, 0.01, device=dev, planes_dtype=torch.bfloat16)
        sp.set_games([(i, 0, 0) for i in range(p, n_games, P)])
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            sp.bind(st)
            sp.start()
        st.synchronize()
        sessions.append(sp)
        streams.append(st)
    if P == 2:
        g = capture_pair(sessions, streams, net, U, offset_stage=args.offset_stage)
        def replay():
            with torch.cuda.stream(streams[0]):
                g.replay()
    else:
        net.latency_mode = False
        g = sessions[0].capture_steps(net, U, stream=streams[0])
        def replay():
            with torch.cuda.stream(streams[0]):
                g.replay()
    for _ in range(args.preroll // U):
        replay()
    torch.cuda.synchronize()

    cap = 1 << 20
    buf = torch.zeros(16 + 32 * cap, dtype=torch.uint8, device=dev)
    hdr = torch.tensor([0, cap], dtype=torch.int64, device=dev)
    buf[:16].copy_(hdr.view(torch.uint8))
    torch.cuda.synchronize()
    for name in ("c4_debug_timeline_session", "c4_debug_timeline_tower", "c4_debug_timeline_gemm"):
        f = getattr(L, name)
        f.restype, f.argtypes = C.c_int, [C.c_void_p]
        assert f(C.c_void_p(buf.data_ptr())) == 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(streams[0])
    for _ in range(3):
        replay()
    ev1.record(streams[0])
    torch.cuda.synchronize()
    for name in ("c4_debug_timeline_session", "c4_debug_timeline_tower", "c4_debug_timeline_gemm"):
        getattr(L, name)(None)
    raw = buf.cpu().numpy()
    n = int(raw[:8].view(np.uint64)[0])
    print(f"{n} workgroup records; 3 replays of {U} rounds in {ev0.elapsed_time(ev1):.3f} ms = {1e3 * ev0.elapsed_time(ev1) / (3 * U):.1f} us per round")
    rec = raw[16:16 + 32 * min(n, cap)].view(np.dtype([("kind", "<u4"), ("tag", "<u4"), ("block", "<u4"), ("hw", "<u4"), ("t0", "<u8"), ("t1", "<u8")]))
    if args.npz:
        os.makedirs(os.path.dirname(os.path.abspath(args.npz)), exist_ok=True)
        np.savez_compressed(args.npz, rec=rec)

# ---- launches: records of one (kind, tag) whose start times cluster; a new launch begins where block 0 starts again
tick = 0.01   # us per s_memrealtime tick (100 MHz)
t_base = int(rec["t0"].min())
launches = []
for key in sorted(set(zip(rec["kind"].tolist(), rec["tag"].tolist()))):
    r = rec[(rec["kind"] == key[0]) & (rec["tag"] == key[1])]
    r = r[np.argsort(r["t0"], kind="stable")]
    n_blocks = int(r["block"].max()) + 1
    # every launch has each block once: cut the sorted stream whenever a block id repeats
    seen, cur = set(), []
    for x in r:
        b = int(x["block"])
        if b in seen:
            launches.append((key, np.array(cur, dtype=r.dtype)))
            seen, cur = set(), []
        seen.add(b)
        cur.append(x)
    if cur:
        launches.append((key, np.array(cur, dtype=r.dtype)))
kind_name = {1: "T", 2: "G", 3: "S"}
# session of a tag: tower out / gemm y / slots pointers differ per session; group tags by which session's chain they follow
# (chain order inside a session: T, G, G, G, S): assign by nearest preceding tower in time with non-overlapping chains
L2 = []
for key, r in launches:
    cu = (r["hw"] >> 28).astype(np.int64) * 1000 + ((r["hw"] >> 13) & 7).astype(np.int64) * 100 + ((r["hw"] >> 12) & 1).astype(np.int64) * 50 + ((r["hw"] >> 8) & 15)
    L2.append(dict(kind=kind_name[key[0]], tag=key[1], n=len(r), t0=(int(r["t0"].min()) - t_base) * tick, t1=(int(r["t1"].max()) - t_base) * tick,
                   last_start=(int(r["t0"].max()) - t_base) * tick, wg_us=float((r["t1"] - r["t0"]).mean()) * tick, cus=len(set(cu.tolist())), cu=cu))
L2.sort(key=lambda d: d["t0"])
# launches are labelled kind + the index of their tag (output / slots pointer) in order of first appearance
tag_idx = {}
for d in L2:
    tag_idx.setdefault(d["kind"], {})
    tag_idx[d["kind"]].setdefault(d["tag"], len(tag_idx[d["kind"]]))
mid = L2[len(L2) // 2]["t0"]                       # a window in the middle of the recording: 2 x rounds tower launches (both sessions)
towers = [d for d in L2 if d["kind"] == "T" and d["t0"] >= mid]
t_start = towers[0]["t0"] if towers else mid
t_end = towers[min(len(towers) - 1, 2 * args.rounds)]["t0"] if towers else mid + 1000
print(f"window of {args.rounds} rounds from t = {t_start:.1f} us; columns: kind/tag#, workgroups, CUs used, first start .. last end (us), "
      f"last workgroup start - first (wait for CUs), mean workgroup duration, CUs shared with the launches running at its start")
for d in L2:
    if d["t0"] < t_start - 200 or d["t0"] > t_end:
        continue
    beside = [f"{o['kind']}{tag_idx[o['kind']][o['tag']]}({len(set(d['cu'].tolist()) & set(o['cu'].tolist()))} CUs)" for o in L2
              if o is not d and o["t0"] < d["t1"] and o["t1"] > d["t0"]]
    if d["t0"] >= t_start:
        print(f"  {d['kind']}{tag_idx[d['kind']][d['tag']]:<2d} wgs {d['n']:4d} cus {d['cus']:3d}  {d['t0'] - t_start:8.1f} .. {d['t1'] - t_start:8.1f}  "
              f"span {d['t1'] - d['t0']:6.1f}  start spread {d['last_start'] - d['t0']:6.1f}  wg {d['wg_us']:6.1f}   beside: {' '.join(beside)}")
# period per S tag
tags_S = sorted({d["tag"] for d in L2 if d["kind"] == "S"})
for tg in tags_S:
    ts = [d["t0"] for d in L2 if d["kind"] == "S" and d["tag"] == tg]
    if len(ts) > 2:
        print(f"session with S tag#{tag_idx['S'][tg]}: {len(ts)} rounds recorded, period {np.diff(ts).mean():.1f} us (min {np.diff(ts).min():.1f}, max {np.diff(ts).max():.1f})")
