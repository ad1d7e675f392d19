#!/usr/bin/env python3
"""Markdown report of one tools/profile_bench.sh output directory.

  tools/profile_report.py gpurun_out/prof_TAG "title" KERNEL [KERNEL ...] > profiles/rN_x.md

Kernel-stats table of the --kernel-trace --stats pass, the bench line that pass printed, the mean duration of the
launches [TIMED_SLICE] (environment, python slice, default "-100:") of the first KERNEL - the ones a workload's in-step
HIP-event timing brackets -, and for every
KERNEL the per-launch means of the counter passes (FETCH_SIZE, WRITE_SIZE, SQ counters; separate runs).
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import summarize_rocprof  # noqa: E402


def one(pattern):
    hits = glob.glob(pattern, recursive=True)
    return hits[0] if hits else None


def counter_means(path, kernel):
    sums, counts = {}, {}
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            c = r["Counter_Name"]
            sums[c] = sums.get(c, 0.0) + float(r["Counter_Value"])
            counts[c] = counts.get(c, 0) + 1
    return {c: (sums[c] / counts[c], counts[c]) for c in sums}


def main(prof, title, kernels):
    stats = one(os.path.join(prof, "stats", "**", "*kernel_stats.csv"))
    summarize_rocprof.main(stats, title)
    line = [ln for ln in open(os.path.join(prof, "stats.log")) if ln.startswith("{")]
    bench = json.loads(line[-1]) if line else None
    trace = one(os.path.join(prof, "stats", "**", "*kernel_trace.csv"))
    if trace and bench and bench.get("roofline", {}).get("kernel_ms") is not None:
        rows = [r for r in csv.DictReader(open(trace)) if kernels[0] in r["Kernel_Name"]]
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        lo, hi = (int(v) if v else None for v in os.environ.get("TIMED_SLICE", "-100:").split(":"))
        tail = d[lo:hi]
        print(f"\nAgreement check on this profiled run: HIP events inside the bench give `roofline.kernel_ms` = "
              f"{bench['roofline']['kernel_ms']}; launches [{lo if lo is not None else ''}:{hi if hi is not None else ''}] of "
              f"`{kernels[0]}` in the rocprofv3 kernel trace (the {len(tail)} the events bracket) average "
              f"{sum(tail) / len(tail):.1f} us (all {len(d)} launches: {sum(d) / len(d):.1f} us).")
    print("\nCounter passes of the same command (separate runs), mean per launch:\n")
    print("| kernel | counter | mean per launch | launches |")
    print("|---|---|---:|---:|")
    for k in kernels:
        for sub in ("fetch", "write", "sq"):
            path = one(os.path.join(prof, sub, "**", "*counter_collection.csv"))
            if not path:
                continue
            for c, (mean, n) in sorted(counter_means(path, k).items()):
                print(f"| `{k}` | {c} | {mean:,.1f} | {n} |")
    print("\nFETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 128-byte requests as 64, so bytes read = "
          "2 x FETCH_SIZE x 1024.  SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs against GRBM_GUI_ACTIVE / 8 XCDs is the "
          "matrix-pipe utilisation.")
    if bench:
        print("\nBench line of the profiled (kernel-trace) run:\n")
        print("```json\n" + json.dumps(bench) + "\n```")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
