mkdir -p gpurun_out/r05h
O=$(pwd)/gpurun_out/r05h
tools/bin/lds_fill_probe 2>&1 | head -8 > $O/simd_map.txt; cat $O/simd_map.txt
bash tools/ab_variant.sh r05h 512 7 bf16s noloads nobar nofragwait nomfma
bash tools/ab_variant.sh r05h 512 8 bf16s noloads nobar nofragwait nomfma
cat $O/ab.log
