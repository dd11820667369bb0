#!/usr/bin/env python
"""rocprofv3 databases of tools/gpu/counters.sh -> the JSON files bench.py quotes (profiles/cfar_bits_pmc.json,
extract_pmc.json, icp_sq.json).  Nothing is transcribed by hand: every number comes out of the .db of this round's
passes; the text summaries (tools/rocpd_summary.py) of the same databases are committed next to them.
FETCH_SIZE / WRITE_SIZE: KiB as rocprofv3 reports them; corrections as calibrated in round 1 on a known byte count
(profiles/r01_run3_pmc_calibration.txt, tools/pmc_calib.hip): fetch bytes = 2 x FETCH_SIZE, write bytes = WRITE_SIZE."""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
CFAR_FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 4096   # frames per CFAR launch of the passes (counters.sh)
G = os.path.join(ROOT, "gpurun_out")


ROWS = {}     # kernel -> counter rows of the last database read (dispatches x shader engines)


def counters(name):
    """-> {kernel: {counter: avg}}, {kernel: avg duration us} of gpurun_out/<tag>_<name>.db"""
    db = sqlite3.connect(os.path.join(G, "%s_%s.db" % (tag, name)))
    c, d = {}, {}
    for kname, cn, avg, dur, rows in db.execute("select name, counter_name, avg(counter_value), avg(duration), count(*) from "
                                                "pmc_events group by name, counter_name"):
        k = kname.split("(")[0].replace("void ", "")
        c.setdefault(k, {})[cn] = avg
        d[k] = dur / 1e3
        ROWS[k] = rows
    return c, d


def kernel_us(name):
    db = sqlite3.connect(os.path.join(G, "%s_%s.db" % (tag, name)))
    return {n.split("(")[0].replace("void ", ""): (cnt, avg / 1e3, mn / 1e3, mx / 1e3) for n, cnt, avg, mn, mx in
            db.execute("select name, count(*), avg(duration), min(duration), max(duration) from kernels group by name")}


def kernel_series_us(name, kernel):
    """durations (us) of every dispatch of `kernel`, in start order"""
    db = sqlite3.connect(os.path.join(G, "%s_%s.db" % (tag, name)))
    return [d / 1e3 for n, d in db.execute("select name, duration from kernels order by start")
            if n.split("(")[0].replace("void ", "") == kernel]


def have(name):
    return os.path.exists(os.path.join(G, "%s_%s.db" % (tag, name)))


def filters():
    """the resident cloud filters (cf_cast_bbox / cf_downsample_radix / cf_radius_filter): traffic and SQ counters of
    `python tools/extract_times.py 512` -> profiles/filters_pmc.json"""
    files = "profiles/%s_%%s.txt" % tag
    f, du = counters("filters_fetch")
    w, _ = counters("filters_write")
    c, _ = counters("filters_sq")
    ks = sorted(x for x in f if x.startswith("cf_") and du[x] > 10.0)    # (the 64-bit fallback kernel returns at once)
    nf = 512
    per = {}
    for k in ks:
        v = c.get(k, {})
        per[k] = {"avg_us_per_launch": du[k], "fetch_bytes": 2 * 1024 * f[k]["FETCH_SIZE"],
                  "write_bytes": 1024 * w.get(k, {}).get("WRITE_SIZE", 0.0)}
        if v:
            per[k].update({"valu_active_frac": v["SQ_ACTIVE_INST_VALU"] * 4.0 / (32.0 * v["SQ_BUSY_CYCLES"]),
                           "wave_wait_frac": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
                           "valu_insts_per_frame": v["SQ_INSTS_VALU"] * 32.0 / nf, "salu_insts_per_frame": v["SQ_INSTS_SALU"] * 32.0 / nf,
                           "lds_insts_per_frame": v["SQ_INSTS_LDS"] * 32.0 / nf})
    fb, wb = sum(p["fetch_bytes"] for p in per.values()), sum(p["write_bytes"] for p in per.values())
    out = {"kernels": " + ".join(ks),
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (one pass each) on `python tools/extract_times.py 512` "
                     "(512 bench frames 1024 x 512 per launch, ~11 400 raw points per frame)",
           "source_files": [files % "filters_kernels", files % "filters_fetch", files % "filters_write", files % "filters_sq"],
           "frames_per_launch": nf, "correction": "fetch bytes = 2 * FETCH_SIZE, write bytes = WRITE_SIZE",
           "per_kernel": per, "traffic_bytes_per_launch": fb + wb, "traffic_bytes_per_frame": (fb + wb) / nf,
           "us_per_launch": sum(p["avg_us_per_launch"] for p in per.values())}
    json.dump(out, open(os.path.join(ROOT, "profiles", "filters_pmc.json"), "w"), indent=1)
    print("wrote profiles/filters_pmc.json from the %s passes" % tag)


def main():
    if have("filters_fetch"):
        filters()
    if not have("cfar_bits_fetch"):
        return
    files = "profiles/%s_%%s.txt" % tag
    # ---- CFAR bit-stream kernel ----
    f, _ = counters("cfar_bits_fetch")
    w, _ = counters("cfar_bits_write")
    us = kernel_us("cfar_bits_kernels")
    k = [x for x in f if x.startswith("cfar_u8_ring")][0]
    fetch, write = f[k]["FETCH_SIZE"], w[k]["WRITE_SIZE"]
    rows, cols, frames = 1024, 512, CFAR_FRAMES
    out = {"kernel": k, "source": "tools/gpu/counters.sh %s: rocprofv3 --kernel-trace [--pmc FETCH_SIZE | --pmc WRITE_SIZE], "
                                  "one pass each, on `python tools/cfar_sweep.py --only --bits --frames %d`" % (tag, frames),
           "source_files": [files % "cfar_bits_kernels", files % "cfar_bits_fetch", files % "cfar_bits_write"],
           "frames_per_launch": frames, "rows": rows, "cols": cols, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
           "correction": "fetch bytes = 2 * FETCH_SIZE, write bytes = WRITE_SIZE (profiles/r01_run3_pmc_calibration.txt)",
           "fetch_bytes": 2 * 1024 * fetch, "write_bytes": 1024 * write,
           "traffic_bytes_per_launch": 2 * 1024 * fetch + 1024 * write,
           "algorithmic_bytes_per_launch": 1.125 * rows * cols * frames, "survey_bytes_per_launch": 2.0 * rows * cols * frames,
           "launches": us[k][0], "avg_us_per_launch": us[k][1], "min_us_per_launch": us[k][2], "max_us_per_launch": us[k][3]}
    out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
    out["frac_of_hbm_peak_rocprof"] = out["algorithmic_bytes_per_launch"] / (out["avg_us_per_launch"] * 1e-6) / 8e12
    # the device reaches its sustained clocks after ~100 launches (profiles/r05_cfar_series.txt): the launches of the second half
    # of the pass are what the bench line's `roofline.ms_per_launch` (timed after the launch time has settled) compares with
    ser = kernel_series_us("cfar_bits_kernels", k)
    if len(ser) >= 40:
        half = ser[len(ser) // 2:]
        out["avg_us_per_launch_second_half_of_the_pass"] = sum(half) / len(half)
        out["avg_us_per_launch_first_20"] = sum(ser[:20]) / 20.0
        out["frac_of_hbm_peak_rocprof_sustained"] = out["algorithmic_bytes_per_launch"] / (out["avg_us_per_launch_second_half_of_the_pass"] * 1e-6) / 8e12
    json.dump(out, open(os.path.join(ROOT, "profiles", "cfar_bits_pmc.json"), "w"), indent=1)
    # ---- extraction ----
    f, du = counters("extract_fetch")
    w, _ = counters("extract_write")
    ks = sorted(x for x in f if x.startswith("extract_"))
    fk = {x: f[x]["FETCH_SIZE"] for x in ks}
    wk = {x: w.get(x, {}).get("WRITE_SIZE", 0.0) for x in ks}
    nf = 256
    fb, wb = 2 * 1024 * sum(fk.values()), 1024 * sum(wk.values())
    out = {"kernels": " + ".join(ks),
           "source": "tools/gpu/counters.sh %s: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, one pass each, on "
                     "`python tools/extract_times.py 256`" % tag,
           "source_files": [files % "extract_fetch", files % "extract_write"],
           "frames_per_launch": nf, "rows": 1024, "cols": 512, "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
           "correction": "fetch bytes = 2 * FETCH_SIZE, write bytes = WRITE_SIZE", "fetch_bytes": fb, "write_bytes": wb,
           "traffic_bytes_per_launch": fb + wb, "traffic_bytes_per_frame": (fb + wb) / nf,
           "avg_us_per_launch": {x: du[x] for x in ks}}
    json.dump(out, open(os.path.join(ROOT, "profiles", "extract_pmc.json"), "w"), indent=1)
    # ---- ICP loop + prep kernels ----
    c, du = counters("icp_sq")

    def sq(k):
        v = c[k]
        return {"kernel": k, "avg_us_per_launch": du[k], "per_shader_engine": v,
                "valu_active_frac": v["SQ_ACTIVE_INST_VALU"] * 4.0 / (32.0 * v["SQ_BUSY_CYCLES"]),
                "wave_wait_frac": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]}
    # (the launch set holds one profiled launch of the PROF build next to the plain launches: the kernel with the most
    # dispatches is the one the step runs)
    loops = sorted((x for x in c if x.startswith("icp_sweep_kernel")), key=lambda x: -ROWS[x])
    preps = sorted((x for x in c if x.startswith("icp_sweep_prep_kernel")), key=lambda x: -du[x])
    loop = sq(loops[0])
    out = {"source": files % "icp_sq" + " (tools/gpu/counters.sh %s: rocprofv3 --kernel-trace --pmc SQ_* -- python "
                                       "tools/stage_times.py --batch 4096 --icp-variants 0 --p2plane-only)" % tag,
           "kernel": loop["kernel"] + " (4096 p2plane30 jobs per launch)", "avg_us_per_launch": loop["avg_us_per_launch"],
           "per_shader_engine": loop["per_shader_engine"], "valu_active_frac": loop["valu_active_frac"],
           "valu_active_note": "SQ_ACTIVE_INST_VALU quad-cycles x 4 / (32 SIMDs per shader engine x SQ_BUSY_CYCLES)",
           "wave_wait_frac": loop["wave_wait_frac"],
           "wave_wait_note": "SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of their resident time the waves sit in s_waitcnt / s_barrier"}
    if preps:
        out["prep"] = sq(preps[0])
    json.dump(out, open(os.path.join(ROOT, "profiles", "icp_sq.json"), "w"), indent=1)
    print("wrote profiles/cfar_bits_pmc.json, extract_pmc.json, icp_sq.json from the %s passes" % tag)


if __name__ == "__main__":
    main()
