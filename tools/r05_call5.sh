mkdir -p gpurun_out/r05e
O=$(pwd)/gpurun_out/r05e
tools/bin/lds_fill_probe > $O/fill_probe.txt 2>&1
head -8 $O/fill_probe.txt
for cfg in "f32 32" "bf16 256"; do set -- $cfg
  timeout -k 10 200 python bench.py --dtype $1 --batch $2 --steps 50 --no-cpu-baseline --secondary 0 2>> $O/bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench.py', d['dtype'], d['config']['per_gpu_batch'], d['value'], d['ms_per_step'], d['ms_per_step_median'], d['config'].get('input_ready_early'))" | tee -a $O/bench_same_box.txt
  timeout -k 10 200 python tools/bench_train.py --mfma $1 --batchsize $2 --data cached --out $O/bench_train.json 2>> $O/train.err | tail -1
done
timeout -k 10 200 python tools/bench_train.py --mfma f32 --batchsize 32 --data jpeg --out $O/bench_train.json 2>> $O/train.err | tail -1
MCG_LOADER_SHM=0 timeout -k 10 200 python tools/bench_train.py --mfma bf16 --batchsize 256 --data cached --out $O/bench_train.json 2>> $O/train.err | tail -1
timeout -k 10 200 python tools/bench_train.py --mfma bf16 --batchsize 256 --data cached --loader_workers 16 --out $O/bench_train.json 2>> $O/train.err | tail -1
tail -3 $O/train.err
