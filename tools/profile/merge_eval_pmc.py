#!/usr/bin/env python3
"""Merge the per-pass summaries of tools/profile/run_eval_pmc.sh into one JSON per backend with the
derived figures the bench line quotes (MFMA busy fraction of the evaluator's kernels).
    python tools/profile/merge_eval_pmc.py gpurun_out/<tag>_evalpmc_M2048 > profiles/r03_evaluator_pmc.json"""
import csv
import glob
import json
import re
import sys

d = sys.argv[1]
N_SIMD = 1024   # 256 CUs x 4
out = {"source": "rocprofv3 --pmc passes over tools/evaluator_probe.py (one pass per counter group, kernels serialised by the profiler), M = %s rows" % re.search(r"M(\d+)", d).group(1),
       "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES counts cycles (16 per v_mfma_f32_16x16x32_bf16); per launch",
       "backends": {}}


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].strip()[:100]


for backend in ("hip0", "hipblaslt0"):
    merged = {}
    for f in sorted(glob.glob(f"{d}/{backend}_p*.json")):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        for k, v in j.items():
            if any(x in k for x in ("c4_", "Cijk")):
                merged.setdefault(k, {}).update(v)
    dur = {}
    try:
        for row in csv.DictReader(open(f"{d}/{backend}_kernel_stats.csv")):
            dur[short(row["Name"])] = float(row["AverageNs"])
    except Exception:
        pass
    res = {}
    for k, v in merged.items():
        kk = re.sub(r"^void ", "", k)
        ns = next((x for n, x in dur.items() if n[:60] == kk[:60]), None)
        waves, wc = v.get("SQ_WAVES", 0), v.get("SQ_WAVE_CYCLES", 0)
        derived = {}
        if ns and waves:
            cyc_per_wave = 4.0 * wc / waves
            clock = cyc_per_wave / ns            # GHz, if a wavefront lives about as long as the launch
            derived = {"duration_us": ns / 1e3, "est_clock_GHz_if_waves_span_the_launch": clock,
                       "mfma_busy_frac_of_chip": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (N_SIMD * ns * min(clock, 2.4)),
                       "mfma_busy_frac_of_wave_time": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, 4.0 * wc) * (1.0),
                       "wait_any_frac": v.get("SQ_WAIT_ANY", 0) / max(1.0, wc), "wait_inst_frac": v.get("SQ_WAIT_INST_ANY", 0) / max(1.0, wc),
                       "active_inst_frac": v.get("SQ_ACTIVE_INST_ANY", 0) / max(1.0, wc),
                       "l2_hit_rate": v.get("TCC_HIT_sum", 0) / max(1.0, v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)),
                       "lds_bank_conflict_frac_of_lds_cycles": v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, v.get("SQ_LDS_IDX_ACTIVE", 0)),
                       "tcp_to_tcc_read_GBps": v.get("TCP_TCC_READ_REQ_sum", 0) * 64.0 / ns if ns else None}
        res[k] = {"derived": derived, "counters": {c: x for c, x in v.items() if c != "dispatches_averaged"}}
    # weighted by time AND by launches per forward pass (a kernel name that serves all three hidden layers counts three times)
    for k, r in res.items():
        r["derived"]["launches_averaged"] = merged[k].get("dispatches_averaged", 1)
    tot = sum(r["derived"].get("duration_us", 0) * r["derived"]["launches_averaged"] for r in res.values())
    if tot:
        out["backends"][backend] = {"kernels": res, "evaluator_mfma_busy_frac_of_chip_time_weighted":
                                    sum(r["derived"].get("duration_us", 0) * r["derived"]["launches_averaged"] * r["derived"].get("mfma_busy_frac_of_chip", 0) for r in res.values()) / tot}
    else:
        out["backends"][backend] = {"kernels": res}
print(json.dumps(out, indent=1))
