#!/usr/bin/env python3
"""Collate what tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/:

    profiles/<tag>_bench.json                  the default bench line (16 streams)
    profiles/<tag>_bench_serial.json           bench.py --streams 1 (what roofline.ms_per_launch is measured on)
    profiles/<tag>_kernel_stats_serial.csv     rocprofv3 --kernel-trace --stats of bench.py --streams 1
    profiles/<tag>_kernel_stats_pipelined.csv  the same for the default (overlapped launches run longer each)
    profiles/<tag>_pmc.json                    per-kernel counters (separate --pmc passes)
    profiles/pmc_latest.json                   HBM bytes per launch of the two big kernels (bench.py's roofline.traffic)

    python tools/summarize_profiles.py r01
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void ", "")
    for cut in ("<", "("):
        k = name.find(cut)
        if k > 0 and not name.startswith("misti::correct_kernel") and not name.startswith("misti::spectrum_kernel"):
            name = name[:k]
    if name.startswith("misti::correct_kernel") or name.startswith("misti::spectrum_kernel"):
        name = name[:name.find("(")]
    return name[:80]


def stats_csv(src_dir, dst):
    files = glob.glob(os.path.join(src_dir, "*", "*_kernel_stats.csv"))
    if not files:
        print("no kernel stats under", src_dir)
        return
    files.sort(key=os.path.getmtime)          # gpurun_out/ accumulates: the newest run of this tag
    rows = list(csv.reader(open(files[-1])))
    with open(dst, "w", newline="") as f:
        wr = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        wr.writerow(rows[0])
        for r in rows[1:]:
            r[0] = short(r[0])
            wr.writerow(r)
    print("wrote", dst)


def pmc_last(src_dir):
    """{kernel: {counter: value of the kernel's last launch}} - the last launches are the timed bench batches."""
    out, seen = {}, {}
    found = sorted(glob.glob(os.path.join(src_dir, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    for fn in found[-1:]:                     # the newest run of this tag only
        for r in csv.DictReader(open(fn)):
            k = short(r["Kernel_Name"])
            key = (k, r["Counter_Name"])
            if int(r["Dispatch_Id"]) >= seen.get(key, -1):
                seen[key] = int(r["Dispatch_Id"])
                out.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
                out[k]["grid_size"] = int(r["Grid_Size"])
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    for a, b in (("bench.json", "_bench.json"), ("bench_serial.json", "_bench_serial.json"), ("bench_config3.json", "_bench_config3.json"),
                 ("bench_config4.json", "_bench_config4.json"), ("bench_config5.json", "_bench_config5.json"),
                 ("bench_config3_search.json", "_bench_config3_search.json"), ("bench_dist_weak.json", "_bench_dist_weak_1rank.json"),
                 ("bench_dist_strong_config4.json", "_bench_dist_strong_config4_1rank.json"),
                 ("bench_dist_strong_config5.json", "_bench_dist_strong_config5_1rank.json")):
        p = os.path.join(src, a)
        if os.path.exists(p) and os.path.getsize(p):
            line = [l for l in open(p) if l.startswith("{")][-1]
            json.dump(json.loads(line), open(os.path.join(dst, tag + b), "w"), indent=1)
            print("wrote", tag + b)
    stats_csv(os.path.join(src, "trace_serial"), os.path.join(dst, tag + "_kernel_stats_serial.csv"))
    stats_csv(os.path.join(src, "trace_pipelined"), os.path.join(dst, tag + "_kernel_stats_pipelined.csv"))
    counters = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        for k, d in pmc_last(os.path.join(src, sub)).items():
            counters.setdefault(k, {}).update(d)
    if counters:
        note = ("rocprofv3 --pmc passes (separate runs: FETCH_SIZE / WRITE_SIZE / SQ_*) of `bench.py --streams 1 --steps 4` "
                "(config 2, 4096 candidates, numT=128); values of each kernel's last launch (a timed bench batch). "
                "FETCH_SIZE / WRITE_SIZE are KB as rocprofv3 reports them; on gfx950 FETCH_SIZE counts wide coalesced reads at half "
                "their bytes (MI355X_MICROARCH.md, HBM section): doubled before comparing with a byte count.")
        json.dump({"note": note, "counters": counters}, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)
        hbm = {}
        # kernel 1 of the timed batches: the chain launch with the trunk following (first batch of a context: plain)
        k1 = "misti::correct_follow_kernel" if any(k.startswith("misti::correct_follow_kernel") for k in counters) else "misti::correct_kernel<true, 64"
        for key, pat in (("correct", k1), ("post", "misti::post_kernel"), ("spectrum", "misti::spectrum_kernel<true>")):
            for k, d in counters.items():
                if k.startswith(pat) and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                    hbm[key + "_hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        json.dump({"workload": "config2",
                   "source": "profiles/%s_pmc.json (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KB -> bytes)" % tag, **hbm},
                  open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
        print("wrote", tag + "_pmc.json, pmc_latest.json:", hbm)


if __name__ == "__main__":
    main()
