set -x
mkdir -p gpurun_out/r05d
O=$(pwd)/gpurun_out/r05d
export MCG_LIB_PATH=$(pwd)/mocogan-chainer_amd/lib/variants/lib_stamps.so
timeout -k 10 300 python tools/stamp_phases_v2.py --tile 7 > $O/stamps_tile7.txt 2> $O/stamps.err
timeout -k 10 300 python tools/stamp_phases_v2.py --tile 8 > $O/stamps_tile8.txt 2>> $O/stamps.err
unset MCG_LIB_PATH
cat $O/stamps_tile7.txt $O/stamps_tile8.txt
for cfg in "f32 32" "bf16 256"; do set -- $cfg; timeout -k 10 200 python tools/bench_train.py --mfma $1 --batchsize $2 --data cached --out $O/bench_train.json 2>> $O/train.err | tail -1; done
timeout -k 10 200 python tools/bench_train.py --mfma f32 --batchsize 32 --data jpeg --out $O/bench_train.json 2>> $O/train.err | tail -1
timeout -k 10 200 python tools/bench_train.py --mfma f32 --batchsize 32 --data synthetic --out $O/bench_train.json 2>> $O/train.err | tail -1
# data-parallel rehearsal over nccl (world of one) under the kernel tracer
R=$(pwd)
export MCG_DP_REHEARSE_NCCL=1 MASTER_PORT=37741 TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/dp_trace -o dp -- python3 $R/bench.py --dtype f32 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline --secondary 0 > $O/dp_bench.json 2> $O/dp_trace.err
cd $R
unset MCG_DP_REHEARSE_NCCL
T=$(find $O/dp_trace -name '*kernel_trace.csv' | head -1)
python tools/trace_dp_overlap.py $T 2 > $O/dp1_nccl_trace_summary.txt 2>&1
cat $O/dp1_nccl_trace_summary.txt | head -40
rm -rf $O/dp_trace
