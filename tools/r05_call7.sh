mkdir -p gpurun_out/r05g
O=$(pwd)/gpurun_out/r05g
R=$(pwd)
export MCG_LIB_PATH=$R/mocogan-chainer_amd/lib/variants/lib_early2.so
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_guardband.py -x -q -m gpu -k "lds_dma or guard or bf16_gemm_outputs" > $O/pytest_early2.log 2>&1; echo "pytest rc $?" >> $O/pytest_early2.log; tail -3 $O/pytest_early2.log
unset MCG_LIB_PATH
grep -q "pytest rc 0" $O/pytest_early2.log || exit 1
bash tools/ab_variant.sh r05g 512 7 bf16s early2
cat $O/ab.log
export MCG_DP_REHEARSE_NCCL=1 MASTER_PORT=37741 TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/dp_trace -o dp -- python3 $R/bench.py --dtype f32 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline --secondary 0 > $O/dp_bench.json 2> $O/dp_trace.err
cd $R
unset MCG_DP_REHEARSE_NCCL
T=$(find $O/dp_trace -name '*kernel_trace.csv' | head -1)
python tools/trace_dp_overlap.py $T 1 > $O/dp1_nccl_trace_summary.txt 2>&1
head -30 $O/dp1_nccl_trace_summary.txt
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("$T")))
c=collections.Counter(r['Kernel_Name'][:90] for r in rows)
for n,k in c.most_common(400):
    if any(x in n.lower() for x in ('ccl','reduce','copy','fill')): print(k, n)
PY
rm -rf $O/dp_trace
