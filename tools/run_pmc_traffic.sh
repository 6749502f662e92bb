#!/bin/bash
# Fabric traffic of every D_V conv launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes) -> JSON for bench.py.
# usage (on the GPU box, from the repository root): bash tools/run_pmc_traffic.sh <outdir> [per-GPU batch n = 32] [precision f32 | bf16s]
# (bf16s: the bf16-stored layers dc2..dc4 -- the 4-channel first layer keeps fp32 tensors and is not part of that table)
set -e
OUT=$(realpath -m "$1"); N=${2:-32}; PREC=${3:-f32}; ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
for B in $((2 * N)) $N; do
    python3 tools/bench_layers.py --net D_V --batch $B --precision $PREC --autotune --save-tiles "$OUT/tiles_b$B.json" > /dev/null 2>&1
    for C in FETCH_SIZE WRITE_SIZE; do
        c=$(echo $C | tr A-Z a-z)
        (cd /tmp && rocprofv3 --pmc $C --output-format csv -d "$OUT/${c}_b$B" -o pmc -- python3 "$ROOT/tools/bench_layers.py" --net D_V --batch $B --precision $PREC --tiles "$OUT/tiles_b$B.json" > "$OUT/bench_layers_b$B.log" 2> "$OUT/${c}_b$B.err")
        cp "$(find "$OUT/${c}_b$B" -name '*counter_collection.csv' | head -1)" "$OUT/${c}_b$B.csv"
        rm -rf "$OUT/${c}_b$B"
    done
done
python3 tools/pmc_traffic.py "$OUT/fetch_size_b$((2 * N)).csv" "$OUT/write_size_b$((2 * N)).csv" "$OUT/bench_layers_b$((2 * N)).log" \
    "$OUT/fetch_size_b$N.csv" "$OUT/write_size_b$N.csv" "$OUT/bench_layers_b$N.log" "$OUT/dv_conv_traffic.json" $N
