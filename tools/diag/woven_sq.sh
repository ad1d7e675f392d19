# SQ counter pass over the bench for both forms of the bf16 conv stack (env switch), one box.
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in phase woven; do
  export PNGPD_CONV_STACK_BF16=$v
  O=$R/gpurun_out/woven_sq_$v; mkdir -p $O
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads > $O/log.txt 2>&1 || exit 1
  python3 - $O $v <<'PY'
import csv, glob, sys, os
p = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
s, n = {}, {}
for r in csv.DictReader(open(p)):
    if "conv_stack_bf16" in r["Kernel_Name"]:
        c = r["Counter_Name"]; s[c] = s.get(c, 0.0) + float(r["Counter_Value"]); n[c] = n.get(c, 0) + 1
print(sys.argv[2], {c: round(s[c] / n[c]) for c in sorted(s)}, n.get("SQ_BUSY_CYCLES"))
m = {c: s[c] / n[c] for c in s}
print(sys.argv[2], "pipe busy %.3f" % ((m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (m["SQ_BUSY_CYCLES"] / 8)))
PY
done
