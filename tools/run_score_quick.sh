# forward tests, the small-batch latency probe, then the headline bench at P=750 and P=500 (no CPU baseline)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_forward_gpu.py tests/test_c_host_gpu.py tests/test_crop_gpu.py tests/test_timed_gpu.py -x -q --timeout 200 --timeout-method thread > $O/z_pytest.log 2>&1 || { tail -30 $O/z_pytest.log; exit 1; }
tail -1 $O/z_pytest.log
PYTHONPATH=$R timeout -k 10 200 python $R/tools/latency_probe.py > $O/z_latency.log 2>&1 || { tail -20 $O/z_latency.log; exit 1; }
grep "us per call" $O/z_latency.log
timeout -k 10 200 python $R/bench.py --no-cpu-baseline | python3 -c "import sys,json; l=json.loads(sys.stdin.read()); print('P750', l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['roofline']['frac'])" || exit 1
timeout -k 10 200 python $R/bench.py --no-cpu-baseline --points 500 --batch 8192 | python3 -c "import sys,json; l=json.loads(sys.stdin.read()); print('P500', l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['roofline']['frac'])" || exit 1
