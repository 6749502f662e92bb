# A/B timing of the default library against lib/variants/lib_<X>.so on one box (bench_layers.py, D_V dc2..dc4)
# usage: bash tools/ab_variant.sh <outdir under gpurun_out> <batch> <tile> <precision> <variant> [...]
#   build a variant with: MCG_LIB_PATH=$(pwd)/mocogan-chainer_amd/lib/variants/lib_<X>.so MCG_HIPCC_FLAGS=-D... python mocogan-chainer_amd/build.py
[ $# -ge 5 ] || { echo "usage: bash tools/ab_variant.sh <outdir> <batch> <tile> <precision> <variant> [...]" >&2; exit 2; }
O=gpurun_out/$1; B=$2; T=$3; P=$4; shift 4; mkdir -p $O
for lib in "$@"; do
  [ -f "$(pwd)/mocogan-chainer_amd/lib/variants/lib_$lib.so" ] || { echo "missing mocogan-chainer_amd/lib/variants/lib_$lib.so" >&2; exit 1; }
done
for rep in 1 2; do
for lib in default "$@"; do
  if [ $lib != default ]; then export MCG_LIB_PATH=$(pwd)/mocogan-chainer_amd/lib/variants/lib_$lib.so; else unset MCG_LIB_PATH; fi
  echo "== $lib $P b$B tile $T" >> $O/ab.log
  python3 tools/bench_layers.py --batch $B --precision $P --net D_V --tile $T 2>> $O/ab.err | grep -E "dc[234]" >> $O/ab.log
done; done
