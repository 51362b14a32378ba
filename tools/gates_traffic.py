#!/usr/bin/env python3
"""usage: tools/gates_traffic.py <pmc_summary.txt> <pipelined_kernel_stats.csv> <tag> [bench line of the traced run] [trace_by_grid output of the same trace] > profiles/<tag>_gates_traffic.json
HBM-side traffic of the fp32 gates kernels at the bench default (256 sessions, aprilv0 dims) from a committed rocprofv3 PMC pass
(tools/pmc_pass.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes, kernel-trace only), in the schema bench.py reads
(roofline.traffic), and the same kernels' durations from the kernel trace of the default (pipelined) invocation."""
import csv
import json
import re
import sys

D, H, SESS = 512, 1024, 256
KERNELS = [("gemm_f32_kernel<4, 4, 1, 0, 0, 0, 1", "one problem per launch"), ("gemm_f32_zkernel<4, 4, 1,", "two problems per launch"),
           ("gemm_f32_zkernel_walk<4, 4, 1,", "three problems per launch (768 tiles on 512 walking workgroups)")]


def main():
    pmc, stats, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    ctr = {}
    for line in open(pmc):
        m = re.match(r"(void .*?)\s+n=\s*(\d+)\s+avg_us=\s*([\d.]+)\s+(.*)", line)
        if not m:
            continue
        for key, _ in KERNELS:
            if key in m.group(1) and (key.startswith("gemm_f32_zkernel_walk") or "walk" not in m.group(1) or "walk" in key):
                for kv in m.group(4).split():
                    k, v = kv.split("=")
                    if k in ("FETCH_SIZE", "WRITE_SIZE"):
                        ctr.setdefault(key, {})[k] = (float(v), int(m.group(2)))
    per, tot_n, tot_tr, tot_alg, tot_rows = {}, 0, 0.0, 0.0, 0.0
    for key, what in KERNELS:
        c = ctr.get(key)
        if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        fetch_kb, n = c["FETCH_SIZE"]
        write_kb = c["WRITE_SIZE"][0]
        rows = write_kb / 8.0                       # u and c: 8 KB written per row
        layers = rows / SESS
        alg = 2 * D * 4 * H * 4 * layers + rows * (D * 4 * 2 + H * 4 * 3)
        tr = (2 * fetch_kb + write_kb) * 1024.0     # gfx950: FETCH_SIZE counts 128-byte requests at 64 bytes -> doubled
        per[what] = {"kernel": key + " ...>", "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "launches_counted": n, "rows": round(rows, 1),
                     "layers": round(layers, 3), "traffic_bytes": int(tr), "algorithmic_bytes": int(alg), "ratio": round(tr / alg, 3)}
        tot_n += n; tot_tr += n * tr; tot_alg += n * alg; tot_rows += n * rows
    times = {}
    for r in csv.DictReader(open(stats)):
        for key, what in KERNELS:
            if key in r["Name"] and (("walk" in key) == ("walk" in r["Name"])):
                times[what] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    rk = None
    if times:
        calls = sum(c for c, _ in times.values())
        us = sum(c * t for c, t in times.values()) / calls
        rows_by = {"one problem per launch": 256.0, "two problems per launch": 512.0, "three problems per launch (768 tiles on 512 walking workgroups)": 768.0}
        rows = sum(c * rows_by[w] for w, (c, _) in times.items()) / calls
        rows_src = "256 / 512 / 768 rows for one / two / three problems"
        if len(sys.argv) > 4:      # (the z-batched kernel also takes the four-problem launches of merged flights: rows per launch from the same run's launch plans)
            bl = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
            rows = float(bl["roofline"]["gates_clock"]["rows_per_launch"])
            rows_src = "rows per launch from the launch plans of the same run (%s: roofline.gates_clock)" % sys.argv[4]
        per_class = None
        if len(sys.argv) > 5:      # tools/trace_by_grid.py output of the same trace: the z-batched kernel holds one- AND two-problem launches; the grid's z tells them apart
            per_class = {}
            for line in open(sys.argv[5]):
                m = re.match(r"(.*?)\s+grid (\S+)\s+n\s+(\d+)\s+mean\s+([\d.]+) us", line)
                if not m or "4, 4, 1, 0, 0, 0, 1" not in m.group(1):
                    continue
                probs = 3 if "walk" in m.group(1) else int(m.group(2).split("x")[2])
                per_class[str(probs)] = {"launches": int(m.group(3)), "avg_us": float(m.group(4)), "rows": 256 * probs}
            calls = sum(v["launches"] for v in per_class.values())
            us = sum(v["launches"] * v["avg_us"] for v in per_class.values()) / calls
            rows = sum(v["launches"] * v["rows"] for v in per_class.values()) / calls
            rows_src = "256 rows per problem, problems per launch from the grid sizes of the same trace (%s)" % sys.argv[5]
        tf = 2.0 * rows * (2 * D) * (4 * H) / (us * 1e-6) / 1e12
        rk = {"source": stats + " (rocprofv3 --kernel-trace --stats of the default, pipelined invocation)", "weighted_avg_us_per_launch": round(us, 2),
              "rows_per_launch": round(rows, 1), "rows_source": rows_src, "weighted_tflops": round(tf, 2), "frac_of_157.3": round(tf / 157.3, 4),
              "per_kernel_us": {w: {"calls": c, "avg_us": round(t, 2)} for w, (c, t) in times.items()},
              "by_problems_per_launch": per_class}
    out = {"source": "%s (rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tools/pmc_pass.sh %s_b256 --ingest lockstep --steps 4 --warmup 2 --no-sweep --no-cpu-baseline --no-config5 --steady-steps 0 --profile-steps 1)" % (pmc, tag),
           "kernel": "gemm_f32_kernel / gemm_f32_zkernel / gemm_f32_zkernel_walk <4, 4, 1, ...>: LSTM gates GEMM + BasicNorm row scale + cell, hand-scheduled K-split loop",
           "sessions_per_gpu": SESS, "rows_per_launch": round(tot_rows / tot_n, 1), "layers_per_launch": round(tot_rows / tot_n / SESS, 3),
           "per_kernel": per,
           "weighting": "the average over the launches the PMC pass counted (rows of a launch from its WRITE_SIZE: 8 KB per row)",
           "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes for wide coalesced reads -> doubled; WRITE_SIZE taken as reported",
           "traffic_bytes_per_launch": int(tot_tr / tot_n), "algorithmic_bytes_per_launch": int(tot_alg / tot_n),
           "note": "traffic / algorithmic = %.2f: the activation rows are fetched once per XCD (8 private L2s); the counters see L2 misses, Infinity-Cache hits included" % (tot_tr / tot_alg),
           "rocprof_kernel_time": rk}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
