#!/usr/bin/env python3
"""Per-round profile summary from rocprofv3's rocpd databases (ROCm 7.2 writes SQLite by default): kernel statistics of a
`--kernel-trace --stats` run and HBM traffic per launch from two separate `--pmc` passes (FETCH_SIZE, WRITE_SIZE).

  python profiles/summarize_db.py <tag> <stats.db> <fetch.db> <write.db> [--workload uk64m --steps 5000]

HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3 PMC slots): the counters are in KiB, collected in separate passes; on
gfx950 FETCH_SIZE tallies 128-byte requests at 64 B, so reads are doubled ("corrected"); WRITE_SIZE is taken as it is.  Access
patterns other than wide streaming reads are uncalibrated there, so both raw and corrected figures are kept.  Achieved
bandwidth of a kernel = corrected bytes of all its launches / their total duration in the --stats run; frac = that / 8000 GB/s."""
import json
import os
import sqlite3
import sys
from collections import defaultdict

PEAK_GBS = 8000.0


def kernels(db):
    cur = sqlite3.connect(db).cursor()
    names = {r[0]: r[1] for r in cur.execute("select id, display_name from rocpd_info_kernel_symbol")}
    rows = cur.execute("select kernel_id, start, end from rocpd_kernel_dispatch").fetchall()
    return names, rows


def counter(db, which):
    cur = sqlite3.connect(db).cursor()
    per = defaultdict(list)
    q = ("select s.display_name, e.counter_value from pmc_events e join rocpd_kernel_dispatch k on k.dispatch_id = e.dispatch_id "
         "join rocpd_info_kernel_symbol s on s.id = k.kernel_id where e.counter_name = ?")
    for name, v in cur.execute(q, (which,)):
        per[name.split("(")[0]].append(float(v))
    return per


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict(zip(sys.argv[1:], sys.argv[2:]))
    tag, stats_db, fetch_db, write_db = args[:4]
    steps = float(opts.get("--steps", "5000"))
    here = os.path.dirname(os.path.abspath(__file__))
    names, rows = kernels(stats_db)
    dur = defaultdict(list)
    for kid, s, e in rows:
        dur[names[kid].split("(")[0]].append((e - s) / 1e3)
    fetch, write = counter(fetch_db, "FETCH_SIZE"), counter(write_db, "WRITE_SIZE")
    total_us = sum(sum(v) for k, v in dur.items() if k.startswith("k_"))
    out = {"tag": tag, "workload": opts.get("--workload", "uk64m"), "steps": steps, "n_gpus": 1, "kernels": {}}
    run_bytes = 0.0
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if not k.startswith("k_"):
            continue
        f = sum(fetch.get(k, [])) * 1024.0
        w = sum(write.get(k, [])) * 1024.0
        corrected = 2.0 * f + w
        run_bytes += corrected
        out["kernels"][k] = {"calls": len(v), "avg_us": sum(v) / len(v), "max_us": max(v), "total_ms": sum(v) / 1e3, "pct": 100.0 * sum(v) / total_us,
                             "fetch_bytes_raw": f, "write_bytes": w, "hbm_bytes_corrected": corrected,
                             "hbm_bytes_per_launch": corrected / max(1, len(v)),
                             "achieved_GBs": corrected / (sum(v) * 1e-6) / 1e9 if sum(v) else 0.0,
                             "frac_of_peak": corrected / (sum(v) * 1e-6) / 1e9 / PEAK_GBS if sum(v) else 0.0}
    out["device_us_per_step"] = total_us / steps
    out["hbm_bytes_per_step"] = run_bytes / steps
    out["run_achieved_GBs"] = run_bytes / (total_us * 1e-6) / 1e9
    with open(os.path.join(here, "%s_summary.json" % tag), "w") as fh:
        json.dump(out, fh, indent=1)
    lines = ["# rocprofv3 summary %s (%s, 1 GPU, %d steps)" % (tag, out["workload"], steps), "",
             "device time of the library's kernels: %.1f ms = %.3f us per time step; HBM bytes (corrected) %.1f MB = %.1f KB per time step; "
             "%.1f GB/s over the run" % (total_us / 1e3, total_us / steps, run_bytes / 1e6, run_bytes / steps / 1e3, out["run_achieved_GBs"]), "",
             "| kernel | calls | avg us | max us | total ms | % | FETCH raw MB | WRITE MB | HBM corrected MB | GB/s | frac of 8 TB/s |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    for k, r in out["kernels"].items():
        lines.append("| %s | %d | %.2f | %.1f | %.2f | %.1f | %.1f | %.1f | %.1f | %.0f | %.4f |" % (
            k, r["calls"], r["avg_us"], r["max_us"], r["total_ms"], r["pct"], r["fetch_bytes_raw"] / 1e6, r["write_bytes"] / 1e6,
            r["hbm_bytes_corrected"] / 1e6, r["achieved_GBs"], r["frac_of_peak"]))
    with open(os.path.join(here, "%s_summary.md" % tag), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
