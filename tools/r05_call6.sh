mkdir -p gpurun_out/r05f
O=$(pwd)/gpurun_out/r05f
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_guardband.py -x -q -m gpu -k "lds_dma or guard or split_fp32 or bf16_gemm_outputs" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "pytest rc 0" $O/pytest.log || exit 1
bash tools/ab_variant.sh r05f 512 7 bf16s noskew
bash tools/ab_variant.sh r05f 512 8 bf16s noskew
bash tools/ab_variant.sh r05f 64 7 f32x3 noskew
bash tools/ab_variant.sh r05f 64 10 f32x3 noskew
cat $O/ab.log
