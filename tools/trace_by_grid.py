#!/usr/bin/env python3
"""Mean duration per (kernel, grid size) from a rocprofv3 kernel trace CSV: tells apart the layers one kernel serves."""
import csv
import re
import sys
from collections import defaultdict


def main(path, pattern):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if pattern and not re.search(pattern, name):
            continue
        grid = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        acc[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (name, grid), d in sorted(acc.items()):
        tail = d[len(d) // 2:]
        print(f"{name:45s} grid {str(grid):18s} calls {len(d):4d}  mean {sum(d)/len(d):8.1f} us  second half {sum(tail)/len(tail):8.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
