#!/usr/bin/env python
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite) runs of bench.py into small JSON/CSV files for
profiles/.  Usage: summarize_rocpd2.py <run dir> <profiles dir> <tag> [kernel-substring ...]

<run dir>/stats/  : --kernel-trace --stats run           -> <tag>_kernel_stats.csv
<run dir>/fetch/, <run dir>/write/ (optional) : --pmc FETCH_SIZE / WRITE_SIZE passes
For every kernel substring given (default: k_gfstack) a <tag>_<name>_summary.json with the
per-dispatch average / min / max duration, launch geometry, registers, LDS and -- when the PMC
passes exist -- HBM bytes per launch (FETCH_SIZE in KB doubled for 16 B/lane streams, as
MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE)."""
import csv
import glob
import json
import os
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    if "distribution_elementwise_grid_stride_kernel" in name:
        return "at::native::distribution_elementwise_grid_stride_kernel<double,...normal_kernel> (torch: library fill)"
    return name if len(name) <= 120 else name[:117] + "..."


def find_db(d):
    c = glob.glob(os.path.join(d, "**", "*results.db"), recursive=True)
    return c[0] if c else None


def main(src, dst, tag, *kernels):
    kernels = kernels or ("k_gfstack",)
    os.makedirs(dst, exist_ok=True)
    db = sqlite3.connect(find_db(os.path.join(src, "stats")))
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for r in rows:
            w.writerow([short(r[0]), r[1], "%.3f" % r[2], "%.3f" % r[3], "%.3f" % r[4]])
    for kn in kernels:
        disp = list(db.execute("select name, duration, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, "
                               "sgpr_count, lds_size from kernels where name like ? order by start", ("%" + kn + "%",)))
        if not disp:
            print("no dispatch of", kn)
            continue
        # several kernels may match (the stacking path times every chain-group size once per problem shape
        # before settling; "k_quadform" also names k_quadform_small): the summary is of the instance that took
        # the most TIME in total -- give the template arguments ("k_quadform<128>") to pin one
        by_name = {}
        for d in disp:
            by_name.setdefault(d[0], []).append(d)
        exact = max(by_name, key=lambda k: sum(x[1] for x in by_name[k]))
        others = {k.replace("void ", ""): len(v) for k, v in by_name.items() if k != exact}
        disp = by_name[exact]
        durs = [d[1] for d in disp]
        summary = dict(kernel=disp[0][0].replace("void ", ""), dispatches=len(disp),
                       avg_us=sum(durs) / len(durs) / 1e3, median_us=sorted(durs)[len(durs) // 2] / 1e3,
                       min_us=min(durs) / 1e3, max_us=max(durs) / 1e3,
                       # the timed launches of bench.py are the LAST `steps` ones of a run; the first few (initial
                       # evaluation, warm-up) run on cold tables and clocks
                       avg_us_after_first_4=(sum(durs[4:]) / len(durs[4:]) / 1e3) if len(durs) > 8 else None,
                       grid=disp[0][2], workgroup=disp[0][3], vgpr=disp[0][4], agpr=disp[0][5],
                       sgpr=disp[0][6], lds=disp[0][7])
        for sub, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            p = find_db(os.path.join(src, sub)) if os.path.isdir(os.path.join(src, sub)) else None
            if not p:
                continue
            d2 = sqlite3.connect(p)
            vals = [r[0] for r in d2.execute("select value from counters_collection where kernel_name = ? "
                                             "and counter_name like ?", (exact, key))]
            if not vals:   # (kernel names of the counter table may carry a different decoration)
                vals = [r[0] for r in d2.execute("select value from counters_collection where kernel_name like ? "
                                                 "and counter_name like ?",
                                                 ("%" + exact.replace("void ", "").split("(")[0] + "%", key))]
            if vals:
                summary[key + "_KB_per_launch_raw"] = sum(vals) / len(vals)
        if "FETCH_SIZE_KB_per_launch_raw" in summary:
            summary["hbm_read_bytes_per_launch_corrected"] = summary["FETCH_SIZE_KB_per_launch_raw"] * 1024 * 2
        if "WRITE_SIZE_KB_per_launch_raw" in summary:
            summary["hbm_write_bytes_per_launch"] = summary["WRITE_SIZE_KB_per_launch_raw"] * 1024
        if others:
            summary["other_matching_kernels_not_summarised"] = others
        name = (kn[2:] if kn.startswith("k_") else kn).replace("<", "").replace(">", "").replace(",", "_")
        with open(os.path.join(dst, "%s_%s_summary.json" % (tag, name)), "w") as fh:
            json.dump(summary, fh, indent=1)
        print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
