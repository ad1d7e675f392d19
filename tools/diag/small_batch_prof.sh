# Kernel trace of 200 forwards at a small batch: per-kernel mean durations and the gaps between consecutive kernels.
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  prec=${cfg%%:*}; B=${cfg##*:}
  O=$R/gpurun_out/sb_${prec}_$B; rm -rf $O; mkdir -p $O
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/diag/small_batch_trace.py $prec $B > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
  python3 - $O $prec $B <<'PY'
import csv, glob, os, sys
p = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-8 * 100:]            # the last 100 forwards
names = {}
gaps = []
for i, r in enumerate(rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = r["Kernel_Name"].split("(")[0][:60]
    names.setdefault(n, []).append(d)
    if i:
        gaps.append((int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3)
print(f"== {sys.argv[2]} B={sys.argv[3]}: span per forward {(int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3 / 100:.1f} us, "
      f"kernel time {sum(sum(v) for v in names.values()) / 100:.1f} us, mean gap {sum(gaps) / len(gaps):.2f} us")
for n, v in names.items():
    print(f"   {n:60s} x{len(v) / 100:.0f}  {sum(v) / len(v):6.2f} us")
PY
done
