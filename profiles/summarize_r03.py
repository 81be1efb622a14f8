#!/usr/bin/env python3
"""Round-3 profile summary from rocprofv3's rocpd databases (tools/profile_passes_r03.sh): per kernel of the library
  * durations (calls, average, maximum, total) from the --kernel-trace --stats pass;
  * HBM-side bytes that are REQUEST-SIZE EXACT: reads = 32 B x TCC_EA0_RDREQ_32B + 64 B x TCC_EA0_RDREQ_64B + 128 B x
    TCC_EA0_RDREQ_128B (checked: the three add up to TCC_EA0_RDREQ), writes = 64 B x TCC_EA0_WRREQ_64B + 32 B x the rest --
    instead of FETCH_SIZE x 2, which MI355X_MICROARCH.md prescribes for wide coalesced streams only and which over-states
    gather kernels (the derived FETCH_SIZE / WRITE_SIZE are kept beside it); tools/micro/tcc_calib.hip calibrates the formula on
    known byte counts (`--calib`);
  * the SQ view: VALU busy = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x clock x duration), resident wavefronts per SIMD =
    SQ_WAVE_CYCLES x 4 / (1024 x clock x duration), the share of wave time spent waiting; the clock is GRBM_GUI_ACTIVE / 8 /
    duration where the dispatch is long enough, else the 2.1-2.4 GHz band is given;
  * with the counting build's tallies (profiles/<tag>_work_<preset>.json): draws and Philox blocks per second, useful bytes over
    fetched bytes.
  python profiles/summarize_r03.py <tag> <dir with the passes> [--workload uk64m --steps 5000] | --calib"""
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict

HBM_PEAK_GBS = 8000.0
N_SIMD = 1024.0            # 256 CUs x 4
CLOCK_MAX_GHZ = 2.4
# One Philox4x32-10 block + the draw's own arithmetic, in VALU instructions per (member, four-step slot) pair, is measured, not
# assumed: SQ_INSTS_VALU / pairs.  The VALU roof: every SIMD issues one wave-instruction per 4 cycles... per wave; a SIMD retires one
# VALU wave-instruction every 4 cycles at most (64 lanes over 16... SIMD-32: 2 cycles issue; SQ_ACTIVE_INST_VALU counts quad-cycles of
# VALU activity per wave), so "busy" = active quad-cycles x 4 / elapsed cycles per SIMD.


def kname(display):
    """'void k_chunk_draw<false>(Dev, unsigned int)' -> 'k_chunk_draw' (template instances of one kernel are one kernel here)."""
    n = display.split("(")[0].strip()
    n = re.sub(r"<.*>$", "", n)
    return n.split()[-1] if n else n


def db_path(d, name):
    p = os.path.join(d, name, "%s_results.db" % name)
    return p if os.path.exists(p) else None


def durations(db):
    cur = sqlite3.connect(db).cursor()
    names = {r[0]: r[1] for r in cur.execute("select id, display_name from rocpd_info_kernel_symbol")}
    per = defaultdict(list)
    for kid, s, e in cur.execute("select kernel_id, start, end from rocpd_kernel_dispatch"):
        per[kname(names[kid])].append((e - s) / 1e3)
    return per


def counters(db):
    """{counter: {kernel: [sum over instances per dispatch, ...]}} and the dispatch durations of that pass."""
    if not db:
        return {}, {}
    cur = sqlite3.connect(db).cursor()
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    dur = defaultdict(dict)
    for name, disp, cname, val, d in cur.execute("select name, dispatch_id, counter_name, counter_value, duration from pmc_events"):
        k = kname(name)
        acc[cname][k][disp] += float(val)
        dur[k][disp] = d / 1e3
    return {c: {k: list(v.values()) for k, v in per.items()} for c, per in acc.items()}, {k: list(v.values()) for k, v in dur.items()}


def total(cs, cname, k):
    return sum(cs.get(cname, {}).get(k, []))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict(zip(sys.argv[1:], sys.argv[2:]))
    calib = "--calib" in sys.argv
    tag, d = args[0], args[1]
    here = os.path.dirname(os.path.abspath(__file__))
    cs = {}
    pass_dur = {}
    for name in ("rd_a", "rd_b", "wr", "at", "fetch", "write", "sq_a", "sq_b"):
        c, du = counters(db_path(d, name))
        cs.update(c)
        pass_dur[name] = du
    if calib:
        known = {"cal_stream_read16": ("read", 1 << 30), "cal_stream_read4": ("read", 1 << 30), "cal_gather4": ("read", 4 << 26), "cal_gather32": ("read", 32 << 24),
                 "cal_stream_write16": ("write", 1 << 30), "cal_scatter4": ("write", 4 << 26), "cal_atomic4": ("atomic", 4 << 26)}
        lines = ["# counter calibration (tools/micro/tcc_calib.hip): memory-side requests on known byte counts", "",
                 "| kernel | known useful MB | RDREQ | _32B | _64B | _128B | exact read MB | FETCH_SIZE MB (x2) | WRREQ | _64B | exact write MB | WRITE_SIZE MB | ATOMIC | L2 hit |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
        out = {}
        for k, (kind, nbytes) in known.items():
            rd, r32, r64, r128 = (total(cs, c, k) for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
            wr, w64 = total(cs, "TCC_EA0_WRREQ_sum", k), total(cs, "TCC_EA0_WRREQ_64B_sum", k)
            at, hit, miss = total(cs, "TCC_EA0_ATOMIC_sum", k), total(cs, "TCC_HIT_sum", k), total(cs, "TCC_MISS_sum", k)
            fs, ws = total(cs, "FETCH_SIZE", k) * 1024.0, total(cs, "WRITE_SIZE", k) * 1024.0
            exact_r = 32 * r32 + 64 * r64 + 128 * r128
            exact_w = 64 * w64 + 32 * (wr - w64)
            out[k] = {"kind": kind, "known_bytes": nbytes, "RDREQ": rd, "RDREQ_32B": r32, "RDREQ_64B": r64, "RDREQ_128B": r128, "read_bytes_exact": exact_r,
                      "FETCH_SIZE_bytes": fs, "WRREQ": wr, "WRREQ_64B": w64, "write_bytes_exact": exact_w, "WRITE_SIZE_bytes": ws, "ATOMIC": at,
                      "l2_hit_rate": hit / (hit + miss) if hit + miss else None}
            lines.append("| %s | %.1f | %.3g | %.3g | %.3g | %.3g | %.1f | %.1f (%.1f) | %.3g | %.3g | %.1f | %.1f | %.3g | %s |" % (
                k, nbytes / 1e6, rd, r32, r64, r128, exact_r / 1e6, fs / 1e6, 2 * fs / 1e6, wr, w64, exact_w / 1e6, ws / 1e6, at,
                "%.2f" % (hit / (hit + miss)) if hit + miss else "-"))
        json.dump(out, open(os.path.join(here, "%s_tcc_calibration.json" % tag), "w"), indent=1)
        open(os.path.join(here, "%s_tcc_calibration.md" % tag), "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))
        return
    steps = float(opts.get("--steps", "5000"))
    workload = opts.get("--workload", "uk64m")
    dur = durations(db_path(d, "kt"))
    lib = {k: v for k, v in dur.items() if k.startswith("k_")}
    total_us = sum(sum(v) for v in lib.values())
    work = None
    wpath = os.path.join(here, "%s_work_%s.json" % (tag, workload))
    if os.path.exists(wpath):
        work = json.load(open(wpath))
    out = {"tag": tag, "workload": workload, "steps": steps, "n_gpus": 1, "device_us_per_step": total_us / steps, "kernels": {}}
    run_r = run_w = run_fs = run_ws = 0.0
    for k, v in sorted(lib.items(), key=lambda kv: -sum(kv[1])):
        rd, r32, r64, r128 = (total(cs, c, k) for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
        wr, w64 = total(cs, "TCC_EA0_WRREQ_sum", k), total(cs, "TCC_EA0_WRREQ_64B_sum", k)
        exact_r, exact_w = 32 * r32 + 64 * r64 + 128 * r128, 64 * w64 + 32 * (wr - w64)
        fs, ws = total(cs, "FETCH_SIZE", k) * 1024.0, total(cs, "WRITE_SIZE", k) * 1024.0
        hit, miss, at = total(cs, "TCC_HIT_sum", k), total(cs, "TCC_MISS_sum", k), total(cs, "TCC_EA0_ATOMIC_sum", k)
        t_s = sum(v) * 1e-6
        # the SQ passes have their own (profiled) durations: utilisation figures are taken against those
        def sq_frac(cname, pname):
            du = sum(pass_dur.get(pname, {}).get(k, []))
            return (total(cs, cname, k), du * 1e-6)
        valu, t_a = sq_frac("SQ_ACTIVE_INST_VALU", "sq_a")
        wavec, _ = sq_frac("SQ_WAVE_CYCLES", "sq_a")
        insts, _ = sq_frac("SQ_INSTS_VALU", "sq_a")
        waves, _ = sq_frac("SQ_WAVES", "sq_a")
        wait_any, t_b = sq_frac("SQ_WAIT_ANY", "sq_b")
        act_any, _ = sq_frac("SQ_ACTIVE_INST_ANY", "sq_b")
        gui, _ = sq_frac("GRBM_GUI_ACTIVE", "sq_b")
        clock = gui / 8.0 / t_b / 1e9 if t_b and gui else None                    # GHz; reads high on short dispatches (MICROARCH, DVFS)
        clk = min(CLOCK_MAX_GHZ, clock) if clock and clock > 1.0 else None
        def per_simd(x, t, ghz):
            return x * 4.0 / (N_SIMD * ghz * 1e9 * t) if t else None
        rec = {"calls": len(v), "avg_us": sum(v) / len(v), "max_us": max(v), "total_ms": sum(v) / 1e3, "pct": 100.0 * sum(v) / total_us,
               "read_requests": rd, "read_requests_32B_64B_128B": [r32, r64, r128], "read_bytes_exact": exact_r, "write_requests": wr, "write_requests_64B": w64,
               "write_bytes_exact": exact_w, "hbm_bytes_exact": exact_r + exact_w, "hbm_bytes_per_launch": (exact_r + exact_w) / len(v),
               "achieved_GBs": (exact_r + exact_w) / t_s / 1e9 if t_s else 0.0, "frac_of_hbm_peak": (exact_r + exact_w) / t_s / 1e9 / HBM_PEAK_GBS if t_s else 0.0,
               "FETCH_SIZE_bytes": fs, "WRITE_SIZE_bytes": ws, "fetch_x2_plus_write_bytes": 2 * fs + ws, "memory_side_atomics": at,
               "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
               "SQ_ACTIVE_INST_VALU": valu, "SQ_INSTS_VALU": insts, "SQ_WAVE_CYCLES": wavec, "SQ_WAVES": waves, "SQ_WAIT_ANY": wait_any, "SQ_ACTIVE_INST_ANY": act_any,
               "clock_GHz_from_GRBM": clock,
               "valu_busy_at_2.4GHz": per_simd(valu, t_a, 2.4), "valu_busy_at_2.1GHz": per_simd(valu, t_a, 2.1), "valu_busy_at_measured_clock": per_simd(valu, t_a, clk) if clk else None,
               "waves_per_simd_at_2.4GHz": per_simd(wavec, t_a, 2.4), "waves_per_simd_at_2.1GHz": per_simd(wavec, t_a, 2.1),
               "wait_share_of_wave_cycles": wait_any / wavec if wavec else None, "valu_share_of_wave_cycles": valu / wavec if wavec else None}
        out["kernels"][k] = rec
        run_r += exact_r; run_w += exact_w; run_fs += fs; run_ws += ws
    out["hbm_bytes_exact_per_step"] = (run_r + run_w) / steps
    out["hbm_bytes_fetchx2_per_step"] = (2 * run_fs + run_ws) / steps
    out["run_achieved_GBs"] = (run_r + run_w) / (total_us * 1e-6) / 1e9
    if work:
        w = work["counts"]
        draw_ms = sum(out["kernels"][k]["total_ms"] for k in ("k_chunk_draw", "k_chunk_units") if k in out["kernels"])
        insts = sum(out["kernels"][k]["SQ_INSTS_VALU"] for k in ("k_chunk_draw", "k_chunk_units") if k in out["kernels"])
        out["work"] = dict(work, draws_per_s=w["draws"] / (total_us * 1e-6), philox_blocks_per_s=w["philox_blocks"] / (total_us * 1e-6),
                           draws_per_s_in_draw_kernels=w["draws"] / (draw_ms * 1e-3) if draw_ms else None,
                           pairs_per_s_in_draw_kernels=w["pairs"] / (draw_ms * 1e-3) if draw_ms else None,
                           valu_wave_instructions_per_pair=insts / w["pairs"] if w["pairs"] else None,
                           useful_over_fetched=work["useful_bytes"] / (run_r + run_w) if run_r + run_w else None)
    with open(os.path.join(here, "%s_summary.json" % tag), "w") as fh:
        json.dump(out, fh, indent=1)
    L = ["# rocprofv3 summary %s (%s, 1 GPU, %d steps)" % (tag, workload, steps), "",
         "device time of the library's kernels: %.1f ms = %.3f us per time step; HBM-side bytes (request-size exact) %.1f MB = %.1f KB per time step = %.1f GB/s over the run "
         "(FETCH_SIZE x 2 + WRITE_SIZE would say %.1f MB)" % (total_us / 1e3, total_us / steps, (run_r + run_w) / 1e6, (run_r + run_w) / steps / 1e3, out["run_achieved_GBs"], (2 * run_fs + run_ws) / 1e6), "",
         "| kernel | calls | avg us | max us | max/avg | total ms | % | rd req (32/64/128 B) M | read MB | write MB | HBM GB/s | frac of 8 TB/s | FETCHx2+WRITE MB | L2 hit |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for k, r in out["kernels"].items():
        a = r["read_requests_32B_64B_128B"]
        L.append("| %s | %d | %.2f | %.1f | %.1f | %.2f | %.1f | %.1f / %.1f / %.1f | %.1f | %.1f | %.0f | %.4f | %.1f | %s |" % (
            k, r["calls"], r["avg_us"], r["max_us"], r["max_us"] / r["avg_us"], r["total_ms"], r["pct"], a[0] / 1e6, a[1] / 1e6, a[2] / 1e6, r["read_bytes_exact"] / 1e6, r["write_bytes_exact"] / 1e6,
            r["achieved_GBs"], r["frac_of_hbm_peak"], r["fetch_x2_plus_write_bytes"] / 1e6, "%.2f" % r["l2_hit_rate"] if r["l2_hit_rate"] is not None else "-"))
    L += ["", "SQ counters (sums over the run; busy / resident figures per SIMD over the kernel's own profiled duration, at 2.4 and 2.1 GHz -- the chip lowers its clock under load):", "",
          "| kernel | VALU wave-insts | VALU active quad-cyc | wave quad-cyc | VALU busy per SIMD (2.4 / 2.1 GHz) | resident waves per SIMD (2.4 / 2.1) | wait share of wave time | clock from GRBM GHz |", "|---|---|---|---|---|---|---|---|"]
    for k, r in out["kernels"].items():
        if not r["SQ_WAVE_CYCLES"]:
            continue
        L.append("| %s | %.3g | %.3g | %.3g | %.2f / %.2f | %.2f / %.2f | %.2f | %s |" % (
            k, r["SQ_INSTS_VALU"], r["SQ_ACTIVE_INST_VALU"], r["SQ_WAVE_CYCLES"], r["valu_busy_at_2.4GHz"] or 0, r["valu_busy_at_2.1GHz"] or 0, r["waves_per_simd_at_2.4GHz"] or 0,
            r["waves_per_simd_at_2.1GHz"] or 0, r["wait_share_of_wave_cycles"] or 0, "%.2f" % r["clock_GHz_from_GRBM"] if r["clock_GHz_from_GRBM"] else "-"))
    if work:
        w, x = work["counts"], out["work"]
        L += ["", "What the pass worked on (counting build, same workload; records equal the golden's: %s):" % work["golden_match"], "",
              "Infected citizen-steps %d; log entries marked %d, keys %d, claims %d, records %d; items drawn %d, members staged %d (%d through an index list); (member, four-step slot) pairs %d "
              "(%d with a draw), Philox blocks %d, Bernoulli draws %d, hits %d; units %d; route pairs %d, riders ranked %d, bus draws %d." % (
                  work["infected_citizen_steps"], w["entries"], w["keys"], w["claims"], w["records"], w["items"], w["members"], w["members_through_index"], w["pairs"], w["pairs_active"],
                  w["philox_blocks"], w["draws"], w["hits"], w["units"], w["route_pairs"], w["riders_ranked"], w["bus_draws"]), "",
              "draws / s over the run %.3g (%.3g inside the draw kernels); Philox blocks / s %.3g; VALU wave-instructions per pair in the draw kernels %.1f; useful bytes %.1f MB "
              "(%s) = %.3f of the bytes fetched + written." % (x["draws_per_s"], x["draws_per_s_in_draw_kernels"] or 0, x["philox_blocks_per_s"], x["valu_wave_instructions_per_pair"] or 0,
                                                              work["useful_bytes"] / 1e6, work["useful_bytes_model"], x["useful_over_fetched"] or 0)]
    with open(os.path.join(here, "%s_summary.md" % tag), "w") as fh:
        fh.write("\n".join(L) + "\n")
    print("\n".join(L))


if __name__ == "__main__":
    main()
