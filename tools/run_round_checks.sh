set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $O/t_pytest.log 2>&1 || { tail -20 $O/t_pytest.log; exit 1; }
tail -1 $O/t_pytest.log
timeout -k 10 200 python bench.py --workload train > $O/t_train.json 2> $O/t_train.err || { tail -20 $O/t_train.err; exit 1; }
cat $O/t_train.json
timeout -k 10 200 python bench.py --workload online > $O/t_online.json 2> $O/t_online.err || { tail -20 $O/t_online.err; exit 1; }
cat $O/t_online.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_t_train -- python3 $R/bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline > $O/t_train_prof.log 2>&1 || { tail -20 $O/t_train_prof.log; exit 1; }
tail -1 $O/t_train_prof.log
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_t_online -- python3 $R/bench.py --workload online --steps 20 --warmup 5 --no-cpu-baseline > $O/t_online_prof.log 2>&1 || { tail -20 $O/t_online_prof.log; exit 1; }
tail -1 $O/t_online_prof.log
