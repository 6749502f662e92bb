# Alternating bench_layers.py timings of a LIST OF TILE CODES on the default library and on lib/variants/lib_<X>.so (one box)
# usage: bash tools/ab_tiles.sh <outdir under gpurun_out> <precision> <pass> "<batches>" "<tile codes>" "<layer filter>" <variant|-> [...]
#   e.g. bash tools/ab_tiles.sh r06_idle f32x3 dgrad "32 64" "10 1010 2010 7 1007 2007" "dc[34]" base_r06
[ $# -ge 7 ] || { echo "usage: bash tools/ab_tiles.sh <outdir> <precision> <pass> \"<batches>\" \"<tiles>\" \"<layer regex>\" <variant|-> [...]" >&2; exit 2; }
O=gpurun_out/$1; P=$2; PASS=$3; BS=$4; TS=$5; LAY=$6; shift 6; mkdir -p $O
LIBS="default"
for lib in "$@"; do
  [ "$lib" = "-" ] && continue
  [ -f "$(pwd)/mocogan-chainer_amd/lib/variants/lib_$lib.so" ] || { echo "missing mocogan-chainer_amd/lib/variants/lib_$lib.so" >&2; exit 1; }
  LIBS="$LIBS $lib"
done
for rep in 1 2; do
for lib in $LIBS; do
  if [ $lib != default ]; then export MCG_LIB_PATH=$(pwd)/mocogan-chainer_amd/lib/variants/lib_$lib.so; else unset MCG_LIB_PATH; fi
  for B in $BS; do for T in $TS; do
    echo "== $lib $P b$B tile $T rep $rep" >> $O/ab.log
    python3 tools/bench_layers.py --batch $B --precision $P --net D_V --only $PASS --tile $T 2>> $O/ab.err | grep -E "$LAY" >> $O/ab.log
  done; done
done; done
