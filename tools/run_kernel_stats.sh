set -e
TAG=${1:-r05_j}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
# tile tables first (so that the profiled pass has no tuning launches)
python3 bench.py --dtype f32 --batch 32 --steps 3 --warmup 2 --no-cpu-baseline --secondary 0 --overlap 0 --save-tiles $O/tiles_f32.json > $O/pre_f32.json 2> $O/pre.err
python3 bench.py --dtype bf16 --batch 256 --steps 3 --warmup 2 --no-cpu-baseline --secondary 0 --overlap 0 --save-tiles $O/tiles_bf16.json > $O/pre_bf16.json 2>> $O/pre.err
python3 bench.py --dtype f32x3 --batch 32 --steps 3 --warmup 2 --no-cpu-baseline --secondary 0 --overlap 0 --save-tiles $O/tiles_f32x3.json > $O/pre_f32x3.json 2>> $O/pre.err
cd /tmp
for cfg in "f32 32" "bf16 256" "f32x3 32"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -o k -- python3 $R/bench.py --dtype $1 --batch $2 --steps 10 --warmup 3 --no-cpu-baseline --secondary 0 --overlap 0 --tiles $O/tiles_$1.json > $O/bench_$1.json 2> $O/prof_$1.err
  cp $(find $O/prof_$1 -name '*kernel_stats.csv' | head -1) $O/${TAG}_$1_b$2_kernel_stats.csv
  rm -rf $O/prof_$1
done
ls -la $O
