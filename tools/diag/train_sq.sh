# SQ counter pass and kernel time of the training step's conv3 product/max kernel for the modes given (env switch).
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export PNGPD_TRAIN_CONV3=$v
  O=$R/gpurun_out/train_sq_$v; rm -rf $O; mkdir -p $O
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O -- python3 $R/bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline > $O/log.txt 2>&1 || exit 1
  python3 - $O $v <<'PY'
import csv, glob, sys, os, json
p = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
s, n = {}, {}
for r in csv.DictReader(open(p)):
    if "conv3_pool_train" in r["Kernel_Name"]:
        c = r["Counter_Name"]; s[c] = s.get(c, 0.0) + float(r["Counter_Value"]); n[c] = n.get(c, 0) + 1
m = {c: s[c] / n[c] for c in s}
print(sys.argv[2], {c: round(m[c]) for c in sorted(m)})
print(sys.argv[2], "cycles per launch %.0f, pipe busy %.3f" % (m["GRBM_GUI_ACTIVE"] / 8, (m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (m["GRBM_GUI_ACTIVE"] / 8)))
line = [l for l in open(os.path.join(sys.argv[1], "log.txt")) if l.startswith("{")]
if line:
    j = json.loads(line[-1]); print(sys.argv[2], "ms/step", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms"])
PY
done
