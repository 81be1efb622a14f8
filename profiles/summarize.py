#!/usr/bin/env python3
"""Turns rocprofv3 CSV output (kernel-trace --stats, and separate --pmc FETCH_SIZE / WRITE_SIZE passes)
into the per-round summary kept under profiles/.

  python profiles/summarize.py <tag> <stats_dir> [<fetch_dir> <write_dir>] [--workload uk64m --gpus 1]

HBM traffic per launch follows MI355X_MICROARCH.md (HBM / rocprofv3 PMC slots): FETCH_SIZE and WRITE_SIZE
are in KiB, collected in separate passes; on gfx950 FETCH_SIZE under-reports wide coalesced reads by
exactly 2x, so reads are doubled ("corrected"); other access widths are uncalibrated, both raw and
corrected figures are kept.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read_csv(pattern):
    rows = []
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict(zip(sys.argv[1:], sys.argv[2:]))
    tag, stats_dir = args[0], args[1]
    here = os.path.dirname(os.path.abspath(__file__))
    out = {"tag": tag, "workload": opts.get("--workload", "uk64m"), "n_gpus": int(opts.get("--gpus", "1"))}
    stats = read_csv(os.path.join(stats_dir, "**", "*_kernel_stats.csv"))
    out["kernel_stats"] = [{"name": r["Name"], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                            "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])} for r in stats]
    if len(args) >= 4:
        per = defaultdict(lambda: defaultdict(list))
        for which, d in (("FETCH_SIZE", args[2]), ("WRITE_SIZE", args[3])):
            for r in read_csv(os.path.join(d, "**", "*_counter_collection.csv")):
                if r["Counter_Name"] == which:
                    per[r["Kernel_Name"]][which].append(float(r["Counter_Value"]))
        traffic = {}
        for k, v in per.items():
            f = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])) * 1024
            w = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])) * 1024
            traffic[k] = {"launches": len(v["FETCH_SIZE"]), "fetch_bytes_raw": f, "write_bytes": w,
                          "hbm_bytes_raw": f + w, "hbm_bytes_corrected": 2 * f + w}
        out["traffic_per_launch"] = traffic
        tj = {"workload": out["workload"], "n_gpus": out["n_gpus"], "source": "profiles/%s_summary.json" % tag}
        for k, v in traffic.items():
            short = k.split("(")[0]
            tj[short + "_bytes_per_launch"] = v["hbm_bytes_corrected"]
        # HBM bytes of one time step = bytes of all of the library's kernels over the profiled run / its time steps
        # (--steps; a chunk pass covers up to 96 steps, so a per-kernel figure alone would not be per step)
        ours = {k: v for k, v in traffic.items() if k.split("(")[0].startswith("k_")}
        run_bytes = sum(v["hbm_bytes_corrected"] * v["launches"] for v in ours.values())
        steps = float(opts.get("--steps", "5000"))
        tj["run_bytes"] = run_bytes
        tj["steps"] = steps
        tj["step_bytes_per_launch"] = run_bytes / steps
        chunks = max([v["launches"] for k, v in ours.items() if k.split("(")[0] == "k_chunk_marks"] + [0])
        if chunks:
            tj["chunk_passes"] = chunks
            tj["chunk_pass_bytes"] = run_bytes / chunks
        with open(os.path.join(here, "traffic.json"), "w") as fh:
            json.dump(tj, fh, indent=1)
    with open(os.path.join(here, "%s_summary.json" % tag), "w") as fh:
        json.dump(out, fh, indent=1)
    lines = ["# rocprofv3 summary %s (%s, %d GPU)" % (tag, out["workload"], out["n_gpus"]), "",
             "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    for r in sorted(out["kernel_stats"], key=lambda r: -r["total_ms"]):
        lines.append("| %s | %d | %.2f | %.2f | %.2f |" % (r["name"], r["calls"], r["avg_us"], r["total_ms"], r["pct"]))
    if "traffic_per_launch" in out:
        lines += ["", "| kernel | launches | FETCH raw MB | WRITE MB | HBM corrected MB |", "|---|---|---|---|---|"]
        for k, v in out["traffic_per_launch"].items():
            lines.append("| %s | %d | %.2f | %.2f | %.2f |" % (k, v["launches"], v["fetch_bytes_raw"] / 1e6, v["write_bytes"] / 1e6,
                                                             v["hbm_bytes_corrected"] / 1e6))
    with open(os.path.join(here, "%s_summary.md" % tag), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
