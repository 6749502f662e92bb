# Timing ablations of the patch-stationary input gradient on one box: the default library against lib/variants/lib_<X>.so builds
# with -DMCG_PATCH_BURST (round 3's load placement) / -DMCG_PP_NOLOADS / _NOBAR / _NOEPI (results garbage).
# usage: bash tools/ab_patch.sh <outdir under gpurun_out> <variant> [<variant> ...]
#   build a variant with: MCG_LIB_PATH=$(pwd)/mocogan-chainer_amd/lib/variants/lib_<X>.so MCG_HIPCC_FLAGS=-D... python mocogan-chainer_amd/build.py
[ $# -ge 2 ] || { echo "usage: bash tools/ab_patch.sh <outdir> <variant> [...]" >&2; exit 2; }
O=gpurun_out/$1; shift; mkdir -p $O
for lib in "$@"; do
  [ -f "$(pwd)/mocogan-chainer_amd/lib/variants/lib_$lib.so" ] || { echo "missing mocogan-chainer_amd/lib/variants/lib_$lib.so" >&2; exit 1; }
done
for rep in 1 2; do
for lib in default "$@"; do
  if [ $lib != default ]; then export MCG_LIB_PATH=$(pwd)/mocogan-chainer_amd/lib/variants/lib_$lib.so; else unset MCG_LIB_PATH; fi
  echo "== $lib bf16s b512" >> $O/ab.log; python3 tools/bench_layers.py --batch 512 --precision bf16s --layer dc2 --only dgrad --tile 9 --net D_V 2>> $O/ab.err | grep dgrad >> $O/ab.log
  echo "== $lib f32x3 b64" >> $O/ab.log; python3 tools/bench_layers.py --batch 64 --precision f32x3 --layer dc2 --only dgrad --tile 9 --net D_V 2>> $O/ab.err | grep dgrad >> $O/ab.log
done; done
cat $O/ab.log
