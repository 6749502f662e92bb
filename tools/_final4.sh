mkdir -p gpurun_out/r05p
O=$(pwd)/gpurun_out/r05p
(timeout -k 10 800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log); tail -4 $O/pytest.log
MCG_DP_REHEARSE_NCCL=1 MASTER_PORT=37751 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary 0 > $O/dp1_f32x3.json 2> $O/dp1.err; cut -c1-400 $O/dp1_f32x3.json
MCG_SINGLE_DEVICE=1 MCG_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --secondary 0 > $O/dp2_gloo_f32x3.json 2> $O/dp2.err; cut -c1-400 $O/dp2_gloo_f32x3.json; tail -2 $O/dp2.err
MCG_BENCH_DETAIL=$O/bench_detail.json timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench.err; cat $O/bench_default.json
