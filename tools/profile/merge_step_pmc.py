#!/usr/bin/env python3
"""The step kernel's counter summary (profiles/r0N_step_kernel_pmc.json) out of the per-pass summaries run_r06.sh leaves:
    python tools/profile/merge_step_pmc.py gpurun_out/r06p > profiles/r06_step_kernel_pmc.json
Reads sq_<G>.json + sq_active_<G>.json (two --pmc passes of 8 SQ counters each, tools/tree_roofline.py --games G), the
FETCH_SIZE / WRITE_SIZE passes at 65 536 games and tree_sweep.json (algorithmic bytes and device-clock time per launch)."""
import json
import os
import sys

d = sys.argv[1]
out = {"command": "tools/profile/run_r06.sh: rocprofv3 --kernel-trace --pmc <8 SQ counters per pass> -- python3 tools/tree_roofline.py --games G --steps 40 --preroll 1500 "
                  "(averages over the 40 measured launches); a timed launch carries 8 (65 536 games) / 1 (2 048) timing-helper wavefronts: divide by the 8 192 / 256 game wavefronts",
       "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x4 = shader cycles), summed over wavefronts"}
for g in (65536, 2048):
    raw = {}
    for f in (f"sq_{g}.json", f"sq_active_{g}.json"):
        raw.update(json.load(open(os.path.join(d, f))))
    waves = g // 8
    out[f"games_{g}"] = {"raw_per_launch": raw,
                         "per_wave": {k + "_per_wave": round(v / waves, 1) for k, v in raw.items() if isinstance(v, float) and k not in ("SQ_WAVES", "GRBM_GUI_ACTIVE")}}
try:
    f, w = json.load(open(os.path.join(d, "traffic65536_FETCH_SIZE.json"))), json.load(open(os.path.join(d, "traffic65536_WRITE_SIZE.json")))
    sweep = [json.loads(l) for l in open(os.path.join(d, "tree_sweep.json")) if l.startswith("{")]
    row = next((r for r in (sweep[0].get("sweep", sweep) if isinstance(sweep[0], dict) and "sweep" in sweep[0] else sweep) if r.get("games_per_launch", r.get("games")) == 65536), None)
    t = {"FETCH_SIZE_KB": f["FETCH_SIZE"], "WRITE_SIZE_KB": w["WRITE_SIZE"]}
    if row:
        alg = row.get("algorithmic_bytes_per_launch")
        t.update({"algorithmic_bytes_per_launch": alg, "ratio_raw": round((f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024 / alg, 2),
                  "ratio_fetch_doubled": round((2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024 / alg, 2), "device_clock_us": row.get("device_clock_us", row.get("avg_kernel_us"))})
    out["traffic_65536"] = t
except Exception as e:  # the traffic passes are optional
    out["traffic_65536"] = {"error": repr(e)}
pw = out["games_65536"]["per_wave"]
waves_per_simd = 4
out["derived_65536"] = {
    "wavefront_lifetime_cycles": round(4 * pw["SQ_WAVE_CYCLES_per_wave"]),
    "issue_cycles_per_wavefront": round(4 * pw["SQ_ACTIVE_INST_ANY_per_wave"]),
    "wavefronts_per_simd": waves_per_simd,
    "issue_slots_taken": round(pw["SQ_ACTIVE_INST_ANY_per_wave"] * waves_per_simd / pw["SQ_WAVE_CYCLES_per_wave"], 3),
    "valu_share_of_issue_cycles": round(pw["SQ_ACTIVE_INST_VALU_per_wave"] / pw["SQ_ACTIVE_INST_ANY_per_wave"], 3),
    "valu_pipe_busy": round(pw["SQ_ACTIVE_INST_VALU_per_wave"] * waves_per_simd / pw["SQ_WAVE_CYCLES_per_wave"], 3),
    "scalar_pipe_busy": round(pw["SQ_ACTIVE_INST_SCA_per_wave"] * waves_per_simd / pw["SQ_WAVE_CYCLES_per_wave"], 3),
    "reading": "a wavefront issues for issue_cycles of its lifetime; four wavefronts share a SIMD (127-128 VGPRs), so issue_slots_taken of the SIMD's cycles have "
               "some wavefront of it issuing -- an upper bound on how full the issue stage is, because instructions of DIFFERENT types from different wavefronts can "
               "issue in the same cycle; the vector ALU alone is busy valu_pipe_busy of the time (a wave64 VALU instruction occupies the SIMD for one quad-cycle).  Either "
               "way the kernel is bound by instruction issue and the dependent-load latency four wavefronts cannot hide, not by bytes"}
print(json.dumps(out, indent=1))
