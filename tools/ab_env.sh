# A/B of an environment switch on one box: alternating bench.py runs (no CPU baseline, no secondary workloads)
# usage: bash tools/ab_env.sh <outdir under gpurun_out> <VAR> <value A> <value B> "<dtype> <batch>" ["<dtype> <batch>" ...]
O=gpurun_out/$1; V=$2; A=$3; B=$4; shift 4; mkdir -p $O
for cfg in "$@"; do set -- $cfg
  for rep in 1 2; do for val in $A $B; do
    env $V=$val python3 bench.py --dtype $1 --batch $2 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 > $O/b.json 2>/dev/null
    python3 -c "import json;d=json.load(open('$O/b.json'));print('$1 b$2 $V=$val', round(d['value'],1), d['ms_per_step'])" >> $O/ab.log
  done; done
done
cat $O/ab.log
