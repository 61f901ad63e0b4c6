#!/usr/bin/env python3
"""Collate what tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/:

    profiles/<tag>_bench.json                  the default bench line (16 streams)
    profiles/<tag>_bench_serial.json           bench.py --streams 1 (what roofline.ms_per_launch is measured on)
    profiles/<tag>_kernel_stats_serial_<workload>.csv   rocprofv3 --kernel-trace --stats of bench.py --workload W --streams 1
    profiles/<tag>_kernel_stats_pipelined.csv  the same for the default (overlapped launches run longer each)
    profiles/<tag>_pmc_<workload>.json         per-kernel counters (separate --pmc passes), config2 / config2x16 / config3 / config5
    profiles/pmc_latest.json                   per workload: HBM bytes and SQ counters per launch of the two big kernels
                                               (bench.py's roofline.traffic and roofline.valu)

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


WORKLOADS = ("config2", "config2x16", "config3", "config5")
# kernel 1 of a workload; a packed launch is TWO kernels (the packed phase and the chains that yielded, resumed one per wave):
# their counters are added
KERNEL1 = {"config2": ("misti::correct_follow_kernel",), "config2x16": ("misti::correct_follow_kernel",),
           "config3": ("misti::correct_kernel<true, 6>", "misti::correct_resume_kernel"),
           "config5": ("misti::correct_kernel<true, 6>", "misti::correct_resume_kernel")}
SQ = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY",
      "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
      "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64",
      "SQ_INSTS_VALU_CVT")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    for a, b in (("bench.json", "_bench.json"), ("bench_serial.json", "_bench_serial.json"), ("bench_config3.json", "_bench_config3.json"),
                 ("bench_config4.json", "_bench_config4.json"), ("bench_config5.json", "_bench_config5.json"), ("bench_config2x16.json", "_bench_config2x16.json"),
                 ("bench_config3_search.json", "_bench_config3_search.json"), ("bench_config3_basinhopping.json", "_bench_config3_basinhopping.json"),
                 ("bench_driver_steps20.json", "_bench_driver_steps20.json"), ("bench_default_fit.json", "_bench_default_fit.json"),
                 ("bench_dist_weak.json", "_bench_dist_weak_1rank.json"),
                 ("bench_dist_strong_config4.json", "_bench_dist_strong_config4_1rank.json"),
                 ("bench_dist_strong_config5.json", "_bench_dist_strong_config5_1rank.json")):
        p = os.path.join(src, a)
        if os.path.exists(p) and os.path.getsize(p):
            lines = [l for l in open(p) if l.startswith("{")]
            if lines:
                json.dump(json.loads(lines[-1]), open(os.path.join(dst, tag + b), "w"), indent=1)
                print("wrote", tag + b)
    for wl in WORKLOADS:
        stats_csv(os.path.join(src, "trace_serial_" + wl), os.path.join(dst, "%s_kernel_stats_serial_%s.csv" % (tag, wl)))
    stats_csv(os.path.join(src, "trace_pipelined"), os.path.join(dst, tag + "_kernel_stats_pipelined.csv"))
    bid = os.path.join(src, "build_id.txt")
    build_id = open(bid).read().strip() if os.path.exists(bid) else None
    latest = {"build_id": build_id,
              "note": "per workload: HBM bytes per launch of the two big kernels (FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE, KB -> bytes) and the SQ "
                      "counters of the same launches; bench.py reads this for roofline.traffic and roofline.valu (stored, not measured in the bench run)",
              "workloads": {}}
    for wl in WORKLOADS:
        counters = {}
        for sub in ("pmc_fetch_", "pmc_write_", "pmc_sq_", "pmc_sq2_", "pmc_f64_"):
            for k, d in pmc_last(os.path.join(src, sub + wl)).items():
                counters.setdefault(k, {}).update(d)
        if not counters:
            continue
        note = ("rocprofv3 --pmc passes (separate runs: FETCH_SIZE / WRITE_SIZE / SQ_* / SQ_THREAD_CYCLES_VALU ...) of `bench.py --workload %s --streams 1 --steps 4 "
                "--no-extra-legs`; values of each kernel's last launch (a timed bench batch). "
                "FETCH_SIZE / WRITE_SIZE are KB as rocprofv3 reports them; on gfx950 FETCH_SIZE counts wide coalesced reads at half "
                "their bytes (MI355X_MICROARCH.md, HBM section): doubled before comparing with a byte count. SQ_WAVE_CYCLES / SQ_WAIT_* / "
                "SQ_ACTIVE_INST_* count quad-cycles, SQ_INSTS_* instructions per wave." % wl)
        json.dump({"note": note, "workload": wl, "counters": counters}, open(os.path.join(dst, "%s_pmc_%s.json" % (tag, wl)), "w"), indent=1)
        entry = {"source": "profiles/%s_pmc_%s.json" % (tag, wl), "build_id": build_id}
        for key, pats in (("correct", KERNEL1[wl]), ("post", ("misti::post_kernel",)), ("spectrum", ("misti::spectrum_kernel<true",))):
            hbm, v, names = 0.0, {}, []
            for pat in pats:
                hits = [k for k in counters if k.startswith(pat)]
                if key == "spectrum" and len(hits) > 1:      # two instantiations (one / four waves per workgroup): the one the timed batches ran
                    hits = [max(hits, key=lambda k: counters[k].get("SQ_INSTS_VALU", 0.0))]
                for k in hits:
                    d = counters[k]
                    names.append(k)
                    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                        hbm += (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
                    for c in SQ:
                        if c in d:
                            v[c] = v.get(c, 0.0) + d[c]
            if hbm:
                entry[key + "_hbm_bytes_per_launch"] = hbm
            if "SQ_INSTS_VALU" in v:
                if v.get("SQ_THREAD_CYCLES_VALU") and v.get("SQ_ACTIVE_INST_VALU"):
                    # mean share of the 64 lanes a VALU instruction had live.  Both counters tick once per 4-cycle issue (the spectrum
                    # kernel, 44 live lanes of 64 in its main loop and fewer around it, reads 0.51)
                    v["lane_occupancy"] = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64.0)
                v["kernel"] = " + ".join(names)
                entry[key + "_valu"] = v
        latest["workloads"][wl] = entry
        print("wrote %s_pmc_%s.json:" % (tag, wl), {k: v for k, v in entry.items() if k.endswith("per_launch")})
    # the replicate epilogue at 65 536 spectra x 1 000 replicates (bench.py: llk_roofline_leg; its last launch under the two TCC passes)
    llk = {}
    for sub in ("pmc_write_llk", "pmc_fetch_llk"):
        for k, d in pmc_last(os.path.join(src, sub)).items():
            if k.startswith("misti::llk_kernel"):
                llk.update(d)
    if "WRITE_SIZE" in llk:
        w, f = llk["WRITE_SIZE"] * 1024.0, 2.0 * llk.get("FETCH_SIZE", 0.0) * 1024.0
        rec = {"kernel": "misti::llk_kernel", "spectra": 65536, "replicates": 1000, "WRITE_SIZE_KB": llk["WRITE_SIZE"], "FETCH_SIZE_KB": llk.get("FETCH_SIZE"),
               "write_bytes": w, "fetch_bytes_corrected": f, "hbm_bytes_per_launch": w + f, "algorithmic_bytes": 8.0 * 65536 * 1000 + 56.0 * 65536 + 72.0 * 1000,
               "grid_size": llk.get("grid_size"), "build_id": build_id,
               "note": "rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE (separate passes) of `bench.py --workload config4 --streams 1 --steps 4`: the LAST launch of "
                       "misti::llk_kernel = misti_llk_dev on 65 536 spectra x 1 000 replicates; KB as reported, FETCH_SIZE doubled (gfx950: MI355X_MICROARCH.md); "
                       "16-byte-per-lane streaming stores, for which WRITE_SIZE is exact"}
        json.dump(rec, open(os.path.join(dst, "%s_pmc_llk.json" % tag), "w"), indent=1)
        latest["llk"] = rec
        print("wrote %s_pmc_llk.json:" % tag, w + f, "bytes against", rec["algorithmic_bytes"], "algorithmic")
    # the default fit on the headline grid
    dcnt = {}
    for sub in ("pmc_sq_config2_default", "pmc_f64_config2_default"):
        for k, d in pmc_last(os.path.join(src, sub)).items():
            dcnt.setdefault(k, {}).update(d)
    if dcnt:
        json.dump({"note": "rocprofv3 --pmc passes of `bench.py --workload config2 --fit default --streams 1 --steps 4`: per kernel, its last launch", "build_id": build_id,
                   "counters": dcnt}, open(os.path.join(dst, "%s_pmc_config2_default.json" % tag), "w"), indent=1)
        print("wrote %s_pmc_config2_default.json" % tag)
    stats_csv(os.path.join(src, "trace_serial_config2_default"), os.path.join(dst, "%s_kernel_stats_serial_config2_default.csv" % tag))
    if latest["workloads"]:
        json.dump(latest, open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
        print("wrote pmc_latest.json")


if __name__ == "__main__":
    main()
