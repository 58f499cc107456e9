#!/usr/bin/env python3
"""Turns gpurun_out/final/ (made by tools/collect_profiles.sh on the GPU box) into the committed evidence under profiles/:
per config the rocprofv3 kernel-stats summary, a PMC summary (HBM traffic per launch with the guide's gfx950 correction, VALU /
SALU / LDS wave-instructions per launch and per sample, the effective clock) and profiles/traffic_cfgN.json, the small file
bench.py reads `roofline.traffic` and `roofline.issue_ceiling` from (valid only for the kernel sources it was measured with:
bench.kernel_source_sha()).  Prints the BASELINE.md results table.   usage: python tools/summarize_profiles.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r3_final"
src = "gpurun_out/final"
os.makedirs("profiles", exist_ok=True)
sys.path.insert(0, os.getcwd())
import bench as _bench  # noqa: E402


def newest(pattern):
    g = glob.glob(pattern)
    return max(g, key=os.path.getmtime) if g else None


def pmc(d):
    """{kernel: {counter: avg per launch}}, {kernel: avg ns}, launches-per-kernel; warm-up launches included (same work)"""
    f = newest(f"{src}/{d}/*/*_counter_collection.csv")
    if not f:
        return None
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "alac" not in r["Kernel_Name"]:
            continue
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[r["Kernel_Name"]][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return ({k: {c: sum(v) / len(v) for c, v in d2.items()} for k, d2 in agg.items()},
            {k: sum(v.values()) / len(v) for k, v in dur.items()})


def stats_rows(c):
    f = newest(f"{src}/stats_{c}/*/*_kernel_stats.csv")
    if not f:
        return None, []
    shutil.copy(f, f"profiles/{tag}_kernel_stats_{c}.csv" if c != "cfg2" else f"profiles/{tag}_kernel_stats.csv")
    return f, [r for r in csv.DictReader(open(f)) if "alac" in r["Name"]]


table = []
for c in ("cfg2", "cfg3", "cfg4", "cfg5"):
    _, rows = stats_rows(c)
    if not rows:
        continue
    fe, wr, sq = pmc(f"pmc_fetch_{c}"), pmc(f"pmc_write_{c}"), pmc(f"pmc_sq_{c}")
    bj = f"{src}/{c}.json" if c != "cfg2" else f"{src}/bench_default.json"
    bench = json.load(open(bj)) if os.path.exists(bj) else None
    kernels = sorted({r["Name"] for r in rows})
    per = {}
    for k in kernels:
        per[k] = {"avg_ns_stats": float([r for r in rows if r["Name"] == k][0]["AverageNs"]),
                  "calls_stats": int([r for r in rows if r["Name"] == k][0]["Calls"])}
        for nm, p in (("fetch", fe), ("write", wr), ("sq", sq)):
            if p and k in p[0]:
                per[k][nm] = p[0][k]
                per[k][nm + "_avg_ns"] = p[1][k]
    tot = lambda nm, ctr: sum(per[k].get(nm, {}).get(ctr, 0.0) for k in kernels)   # noqa: E731  (both launches of a pair)
    fetch_kb, write_kb = tot("fetch", "FETCH_SIZE"), tot("write", "WRITE_SIZE")
    main = max(kernels, key=lambda k: per[k]["avg_ns_stats"])
    gui = per[main].get("write", {}).get("GRBM_GUI_ACTIVE")
    clock = gui / 8 / per[main]["write_avg_ns"] if gui else None
    out = {
        "command": "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py --config N --steps 12 --warmup 3 "
                   "--no-cpu-baseline --no-host-path --no-big-batch --no-in-flight   (tools/collect_profiles.sh: one --pmc pass per "
                   "counter group, no other tracing domain)",
        "workload": c, "dominant_kernel": main, "kernels": per,
        "traffic": {
            "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb,
            "gfx950_correction": "FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads on gfx950 (MI355X_MICROARCH.md,"
                                 " HBM section): read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact",
            "read_bytes_per_launch": 2 * fetch_kb * 1024, "write_bytes_per_launch": write_kb * 1024,
            "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
            "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"] if bench else None,
            "note": "two-pass kernels: pass 0 parks channel A's reconstructed samples in the upper half of the packet's output slot "
                    "(4 bytes per sample frame) and pass 1 reads them back; that round trip replaces a Rice-only pre-scan of channel A",
        },
        "issue": {"valu_wave_instr_per_launch": tot("sq", "SQ_INSTS_VALU"), "salu_wave_instr_per_launch": tot("sq", "SQ_INSTS_SALU"),
                  "lds_wave_instr_per_launch": tot("sq", "SQ_INSTS_LDS"),
                  "samples_per_launch": bench["config"]["samples_per_step_per_gpu"] if bench else None},
        "effective_clock_GHz": clock,
    }
    if bench:
        s = bench["config"]["samples_per_step_per_gpu"]
        out["issue"]["valu_wave_instr_per_sample"] = out["issue"]["valu_wave_instr_per_launch"] / s
    json.dump(out, open(f"profiles/{tag}_pmc_summary_{c}.json", "w"), indent=1)
    json.dump({"workload": c, "kernel": main, "hbm_bytes_per_launch": out["traffic"]["hbm_bytes_per_launch"],
               "valu_wave_instr_per_launch": out["issue"]["valu_wave_instr_per_launch"], "effective_clock_GHz": clock,
               "kernel_source_sha": _bench.kernel_source_sha(),   # bench.py reports these figures only for these kernel sources
               "source": f"profiles/{tag}_pmc_summary_{c}.json"}, open(f"profiles/traffic_{c}.json", "w"))
    if bench:
        shutil.copy(bj, f"profiles/{tag}_bench_{c}.json")
        table.append((c, bench, out, per, main))
    print(c, main, "avg ns", round(per[main]["avg_ns_stats"]), "traffic MB", round(out["traffic"]["hbm_bytes_per_launch"] / 1e6, 1),
          "VALU/sample", round(out["issue"].get("valu_wave_instr_per_sample", 0), 3), "clock", clock)
# the issue-bound regime
_, rows = stats_rows("cfg2_32768")
sq = pmc("pmc_sq_cfg2_32768")
if rows and sq:
    k = max(sq[0], key=lambda kk: sq[1][kk])
    samples = 32768 * 8192
    json.dump({"workload": "cfg2 at 32768 packets", "kernel": k, "avg_ns": sq[1][k], "counters_avg_per_launch": sq[0][k],
               "valu_wave_instr_per_sample": sq[0][k]["SQ_INSTS_VALU"] / samples,
               "effective_clock_GHz": sq[0][k].get("GRBM_GUI_ACTIVE", 0) / 8 / sq[1][k]},
              open(f"profiles/{tag}_pmc_summary_cfg2_32768.json", "w"), indent=1)
    print("cfg2@32768", k, "VALU/sample", sq[0][k]["SQ_INSTS_VALU"] / samples)
for extra in ("cfg2_8192", "cfg2_16384", "cfg2_32768", "two_ranks_one_gpu_gloo", "cfg1_m4a"):
    if os.path.exists(f"{src}/{extra}.json"):
        shutil.copy(f"{src}/{extra}.json", f"profiles/{tag}_bench_{extra}.json" if extra != "cfg1_m4a" else f"profiles/{tag}_cfg1_m4a.json")
print()
print("| config | packets / GPU | dominant kernel | ms / batch | GPU Msamples/s | algorithmic GB/s | fraction of 8 TB/s | HBM traffic / algorithmic | VALU wave-instr / sample | fraction of the issue ceiling | CPU 1 thread Msamples/s | GPU / CPU-1T | CPU all cores Msamples/s |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for c, j, out, per, main in table:
    cb = j.get("cpu_baseline") or {}
    ips = out["issue"].get("valu_wave_instr_per_sample")
    clock = out["effective_clock_GHz"] or 2.4
    ceil = 1024 * clock * 1e3 / 4 / ips if ips else None
    kms = j["roofline"]["kernel_ms"]
    frac_issue = j["config"]["samples_per_step_per_gpu"] / (kms * 1e-3) / 1e6 / ceil if ceil else None
    print(f"| {c} | {j['config']['packets_per_gpu']} | `{main.replace('alac_decode_', '').replace('_kernel', '')}` | {j['ms_per_step']:.3f} | "
          f"{j['value']:.0f} | {j['roofline']['achieved']:.1f} | {j['roofline']['frac'] * 100:.2f} % | "
          f"{out['traffic']['hbm_bytes_per_launch'] / j['roofline']['algorithmic_bytes_per_launch']:.2f} | {ips:.2f} | "
          f"{(frac_issue or 0) * 100:.0f} % | {cb.get('value', 0):.1f} | {j['value'] / cb['value'] if cb.get('value') else 0:.0f}x | "
          f"{cb.get('all_cores_value', 0):.0f} | parity={j.get('parity_vs_oracle')}")
if os.path.exists(f"{src}/cfg1_m4a.json"):
    print(open(f"{src}/cfg1_m4a.json").read())
