# Timing ablations of the patch-stationary input gradient on one box: the default library against lib/variant_<X>.so builds of
# the dgrad translation unit with -DMCG_PATCH_BURST (round 3's load placement) / -DMCG_PP_NOLOADS / _NOBAR / _NOEPI (results garbage)
O=gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2; do
for lib in default "$@"; do
  if [ $lib != default ]; then export MCG_LIB_PATH=$(pwd)/mocogan-chainer_amd/lib/variant_$lib.so; else unset MCG_LIB_PATH; fi
  echo "== $lib bf16s b512" >> $O/ab.log; python3 tools/bench_layers.py --batch 512 --precision bf16s --layer dc2 --only dgrad --tile 9 --net D_V 2>/dev/null | grep dgrad >> $O/ab.log
  echo "== $lib f32x3 b64" >> $O/ab.log; python3 tools/bench_layers.py --batch 64 --precision f32x3 --layer dc2 --only dgrad --tile 9 --net D_V 2>/dev/null | grep dgrad >> $O/ab.log
done; done
cat $O/ab.log
