#!/usr/bin/env python3
"""Refresh profiles/step_kernel_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the
bench command, summarised by tools/profile/summarize_pmc.py, and the bench line of one of those runs.
    python tools/profile/update_traffic.py <traffic_FETCH_SIZE.json> <traffic_WRITE_SIZE.json> <pmc bench json> <label> [<fused FETCH json> <fused WRITE json>]
The optional pair holds the same counters for c4_out_step_kernel (the launch of the timed region: output layers + step)."""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def step_hash():
    sys.path.insert(0, root)
    import bench
    return bench.step_kernel_source_hash()


path = os.path.join(root, "profiles", "step_kernel_traffic.json")
old = json.load(open(path))
f, w = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
bench = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
label = sys.argv[4]
hist = old.get("history", {})
hist[f"before: {old.get('note', '')[:80]}"] = {k: old[k] for k in ("FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch", "hbm_bytes_per_launch_raw", "hbm_bytes_per_launch") if k in old}
fk, wk = f["FETCH_SIZE"], w["WRITE_SIZE"]
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
raw, corr = (fk + wk) * 1024, (2 * fk + wk) * 1024
new = dict(old)
new.update({"launches_averaged": int(min(f["dispatches_averaged"], w["dispatches_averaged"])), "FETCH_SIZE_KB_per_launch": round(fk, 1),
            "WRITE_SIZE_KB_per_launch": round(wk, 1), "hbm_bytes_per_launch_raw": int(raw), "hbm_bytes_per_launch": int(corr),
            "algorithmic_bytes_per_launch": int(alg), "ratio_raw": round(raw / alg, 2), "ratio_fetch_doubled": round(corr / alg, 2),
            "history": hist, "note": label, "step_kernel_source_hash": step_hash(),
            "command": old["command"].replace("tools/profile/run_r02.sh", "tools/profile/run_r04.sh").replace("tools/profile/run_r03.sh", "tools/profile/run_r04.sh")})
new["command"] = new["command"].replace("tools/profile/run_r04.sh", "tools/profile/run_r06.sh").replace("tools/profile/run_r05.sh", "tools/profile/run_r06.sh")
if len(sys.argv) > 6:
    ff, fw = json.load(open(sys.argv[5])), json.load(open(sys.argv[6]))
    rl = bench["roofline"]
    alg_f = rl["algorithmic_bytes_per_launch"]
    head = rl.get("head_out_operand_bytes_per_launch") or 0
    raw_f, corr_f = (ff["FETCH_SIZE"] + fw["WRITE_SIZE"]) * 1024, (2 * ff["FETCH_SIZE"] + fw["WRITE_SIZE"]) * 1024
    new["c4_out_step_kernel"] = {"launches_averaged": int(min(ff["dispatches_averaged"], fw["dispatches_averaged"])),
                                 "FETCH_SIZE_KB_per_launch": round(ff["FETCH_SIZE"], 1), "WRITE_SIZE_KB_per_launch": round(fw["WRITE_SIZE"], 1),
                                 "hbm_bytes_per_launch_raw": int(raw_f), "hbm_bytes_per_launch": int(corr_f),
                                 "algorithmic_tree_bytes_per_launch": int(alg_f), "head_out_operand_bytes_per_launch": int(head),
                                 "ratio_raw_to_tree_plus_head_bytes": round(raw_f / (alg_f + head), 2),
                                 "ratio_fetch_doubled_to_tree_plus_head_bytes": round(corr_f / (alg_f + head), 2),
                                 "note": "the fused launch also reads the heads' last hidden activations (rows x 2F bf16) and writes logprobs / q: counted in head_out_operand_bytes"}
json.dump(new, open(path, "w"), indent=1)
print(json.dumps({k: new[k] for k in ("FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "ratio_raw", "ratio_fetch_doubled")}))
