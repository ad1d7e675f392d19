# Kernel trace of the training bench: mean duration of every kernel over the timed steps, per step.
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/train_trace; rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
python3 - $O <<'PY'
import csv, glob, os, sys
p = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(p)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time of the run {tot / 1e6:.2f} ms")
for r in rows[:28]:
    print(f"{r['Name'].split('(')[0][:58]:58s} x{int(r['Calls']):4d} avg {float(r['AverageNs']) / 1e3:7.1f} us  {float(r['Percentage']):5.2f} %")
PY
tail -1 $O/log.txt | cut -c1-200
