# Product path against bench.py on one box, with the loader's uint8 batches handed to TrainStep (MCG_LOADER_U8=1, round 6) and with the
# float batch of rounds 1-5 (=0): alternating runs.   usage: bash tools/ab_train_u8.sh <outdir under gpurun_out>
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2; do
  for cfg in "bf16 256" "f32x3 32"; do set -- $cfg
    python3 bench.py --dtype $1 --batch $2 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 > $O/b.json 2>/dev/null
    python3 -c "import json;d=json.load(open('$O/b.json'));print('bench.py $1 b$2', round(d['value'],1), d['ms_per_step'])" >> $O/ab.log
    for u8 in 0 1; do
      MCG_LOADER_U8=$u8 python3 tools/bench_train.py --mfma $1 --batchsize $2 --data cached --loader_workers 8 --iters 50 --out $O/train.json 2>>$O/err.log | tail -1 > $O/t.json
      python3 -c "import json;d=json.load(open('$O/t.json'));print('bench_train $1 b$2 MCG_LOADER_U8=$u8', {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('clips_per_s','ms_per_iteration')})" >> $O/ab.log
    done
  done
done
cat $O/ab.log
