#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV into a short table (kernel names trimmed)."""
import csv
import re
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)              # drop the argument list
    name = re.sub(r"^void\s+", "", name)
    return name[:90]


def main(path: str, title: str) -> None:
    rows = list(csv.DictReader(open(path)))
    print(f"# {title}\n")
    print(f"source: rocprofv3 --kernel-trace --stats ({path.split('/')[-1]})\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for r in rows:
        if float(r["Percentage"]) < 0.05:
            continue
        print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
              f"{float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | "
              f"{float(r['Percentage']):.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "rocprofv3 kernel stats")
