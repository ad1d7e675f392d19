# L1 (TCP) counter pass over the training step: requests to the L1, requests it passes on to the L2, per kernel.
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/train_tcp; rm -rf $O; mkdir -p $O
timeout -s KILL 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O -- python3 $R/bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
python3 - $O <<'PY'
import csv, glob, sys, os
p = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
s, n = {}, {}
for r in csv.DictReader(open(p)):
    k = (r["Kernel_Name"].split("(")[0][:50], r["Counter_Name"])
    s[k] = s.get(k, 0.0) + float(r["Counter_Value"]); n[k] = n.get(k, 0) + 1
names = sorted({k[0] for k in s})
for nm in names:
    m = {c: s[(nm, c)] / n[(nm, c)] for (k, c) in s if k == nm}
    if m.get("TCP_TCC_READ_REQ_sum", 0) < 2e5: continue
    print(f"{nm:50s} " + "  ".join(f"{c[4:-4]}={m[c]:.3g}" for c in sorted(m)))
PY
