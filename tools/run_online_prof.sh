# crop tests + the online workload under rocprofv3 (kernel stats)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_crop_gpu.py tests/test_dataset_crop_gpu.py -x -q --timeout 200 --timeout-method thread > $O/w_pytest.log 2>&1 || { tail -20 $O/w_pytest.log; exit 1; }
tail -1 $O/w_pytest.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_w_online
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_w_online -- python3 $R/bench.py --workload online --steps 20 --warmup 5 --no-cpu-baseline > $O/w_online_prof.log 2>&1 || { tail -20 $O/w_online_prof.log; exit 1; }
grep '"metric"' $O/w_online_prof.log | cut -c1-400
python3 $R/tools/summarize_rocprof.py $O/prof_w_online/*/*_kernel_stats.csv online | grep -E "cull|gather|scatter|cell"
