# Full GPU suite, smoke, the default bench line, the voxel line and the small-batch latency table, each under its own limit.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_r3; mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c1-600 $O/bench.json
timeout -k 10 200 python bench.py --workload voxel > $O/bench_voxel.json 2> $O/bench_voxel.err || { tail -20 $O/bench_voxel.err; exit 1; }
cut -c1-600 $O/bench_voxel.json
timeout -k 10 300 python tools/latency_probe.py > $O/latency.log 2>&1 || { tail -20 $O/latency.log; exit 1; }
grep -v "^$" $O/latency.log | tail -24
