# voxel tests + the voxel workload under rocprofv3 (kernel stats)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_voxel_gpu.py -x -q --timeout 200 --timeout-method thread > $O/vx_pytest.log 2>&1 || { tail -20 $O/vx_pytest.log; exit 1; }
tail -1 $O/vx_pytest.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_voxel
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_voxel -- python3 $R/bench.py --workload voxel --steps 50 --warmup 10 --no-cpu-baseline > $O/vx_prof.log 2>&1 || { tail -20 $O/vx_prof.log; exit 1; }
grep '"metric"' $O/vx_prof.log | cut -c1-300
python3 $R/tools/summarize_rocprof.py $O/prof_voxel/*/*_kernel_stats.csv voxel | head -20
