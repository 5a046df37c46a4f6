#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_<tag>*/) into the tracked summaries under profiles/.

    python profiles/summarize.py <tag> <config>      e.g.  r01 c2

Inputs (written on the GPU box by the commands in profiles/README.md):
    gpurun_out/prof_<tag>/<config>_kernel_stats.csv              rocprofv3 --kernel-trace --stats
    gpurun_out/prof_<tag>_fetch/<config>_counter_collection.csv  rocprofv3 --kernel-trace --pmc FETCH_SIZE
    gpurun_out/prof_<tag>_write/<config>_counter_collection.csv  rocprofv3 --kernel-trace --pmc WRITE_SIZE
Outputs:
    profiles/<tag>_<config>_kernel_stats.csv   copy of the --stats summary
    profiles/<tag>_<config>_pmc_summary.csv    per-kernel mean FETCH_SIZE / WRITE_SIZE (KB, raw)
    profiles/traffic.json                      per-launch HBM bytes used by bench.py's roofline.traffic

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes read, so bytes = (2*FETCH + WRITE) * 1024.
(The 2x is confirmed here by k_dyn_colsum, whose compulsory read is the whole dynamic tensor.)
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short_name(k):
    """bench.py's name of a kernel (the last template argument of k_transition* is the form of the precedence
    update: 0 fp32 copy, 1 bit shadow, 2 first step building the shadow)."""
    import re
    m = re.match(r"(?:void )?k_(?:big_|macs2d_wave_|macs3d_wave_)?transition(?:_macs3?)?<(.*)>", k)
    if m:
        targs = [x.strip() for x in m.group(1).split(",")]
        # the form is the last template argument, except k_transition_macs[3]<G, NC, MODE, WC | WL> (compile-time width / sides last)
        mode = targs[2] if ("transition_macs" in k and len(targs) == 4) else targs[-1]
        # & 3: the form; 4 = TAP_MODE_MERGED (the run-of-rows expansion, round 5)
        mode = str(int(mode) & 3) if mode.isdigit() else mode
        return {"0": "transition_copy", "1": "transition", "2": "transition_first"}.get(mode, "transition")
    for pat, name in (("k_mask_step", "mask_step"), ("k_env_step", "env_step"), ("k_rolling_window", "rolling_window"),
                      ("k_rolling_step", "rolling_step"), ("k_macs2d_step", "macs_step"), ("k_macs3d_step", "macs_step"),
                      ("k_big_wave_episode", "episode"), ("k_big_wave_step", "env_step"), ("k_macs2d_wave_step", "macs_step"),
                      ("k_macs3d_wave_step", "macs_step"),
                      ("k_episode", "episode"), ("k_dyn_bits", "dyn_bits")):
        if pat in k:
            return name
    return None


def main():
    global tag, cfg
    tag, cfg = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out")
    # the --stats summary, our kernels row by row, everything else (torch / rocprim / runtime copy and fill kernels of the
    # set-up code) summed in one line
    rows = list(csv.DictReader(open(os.path.join(src, "prof_%s" % tag, "%s_kernel_stats.csv" % cfg))))
    import re
    ours = [r for r in rows if re.match(r"^(void )?k_", r["Name"])]
    rest = [r for r in rows if r not in ours]
    with open(os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, cfg)), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
        w.writeheader()
        for r in ours:
            w.writerow(r)
        if rest:
            tot = sum(float(r["TotalDurationNs"]) for r in rest)
            calls = sum(int(r["Calls"]) for r in rest)
            w.writerow({"Name": "(other: %d torch / rocprim / runtime kernels of the set-up code)" % len(rest), "Calls": calls,
                        "TotalDurationNs": int(tot), "AverageNs": tot / max(calls, 1),
                        "Percentage": round(sum(float(r["Percentage"]) for r in rest), 4), "MinNs": "", "MaxNs": "", "StdDev": ""})
    means = collections.defaultdict(dict)
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        path = os.path.join(src, "prof_%s_%s" % (tag, kind), "%s_counter_collection.csv" % cfg)
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            means[k][counter] = sum(v) / len(v)
            means[k]["calls_" + counter] = len(v)
    with open(os.path.join(ROOT, "profiles", "%s_%s_pmc_summary.csv" % (tag, cfg)), "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "calls", "FETCH_SIZE_KiB_mean_raw", "WRITE_SIZE_KiB_mean", "HBM_bytes_per_launch=(2*FETCH+WRITE)*1024"])
        for k, m in sorted(means.items()):
            fs, ws = m.get("FETCH_SIZE", 0.0), m.get("WRITE_SIZE", 0.0)
            w.writerow([k, m.get("calls_FETCH_SIZE", 0), "%.3f" % fs, "%.3f" % ws, int((2 * fs + ws) * 1024)])
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    # per-kernel average duration from the --stats pass of the same command
    avg_ns = {}
    for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, cfg)))):
        avg_ns[r["Name"]] = float(r["AverageNs"])


    for k, m in means.items():
        name = short_name(k)
        if name is None:
            continue
        traffic["%s:%s" % (cfg, name)] = dict(
            bytes=int((2 * m.get("FETCH_SIZE", 0.0) + m.get("WRITE_SIZE", 0.0)) * 1024), profile=tag,
            kernel_us=round(avg_ns.get(k, 0.0) / 1e3, 3) or None, kernel=k[:80])
    traffic["_note"] = ("per-launch HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes and the "
                        "kernel's average duration from the --kernel-trace --stats pass of the same bench.py command; "
                        "`profile` names the profile set (profiles/<profile>_<config>_*.csv) the entry comes from")
    traffic["_source"] = "profiles/summarize.py"
    json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
    # bench lines under profiles/ come from clean runs (gpurun_out/bench_<tag>/<cfg>.json), never from the
    # profiled run, whose timings the tracer perturbs
    bj = os.path.join(src, "bench_%s" % tag, "%s.json" % cfg)
    if os.path.exists(bj) and os.path.getsize(bj):
        shutil.copy(bj, os.path.join(ROOT, "profiles", "%s_%s_bench.json" % (tag, cfg)))
    print(json.dumps({k: v for k, v in traffic.items() if k.startswith(cfg + ":")}, indent=1))


if __name__ == "__main__":
    main()
