#!/bin/bash
# SQ / LDS / clock counters of the D_V conv GEMM launches (two --pmc passes) -> text table.
# usage (GPU box, repository root): bash tools/run_pmc_util.sh <outdir> <precision f32|bf16s> <batch> [tiles.json]
set -e
OUT=$(realpath -m "$1"); PREC=$2; B=$3; TILES=${4:-}; ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--net D_V --batch $B --precision $PREC"
if [ -n "$TILES" ]; then ARGS="$ARGS --tiles $TILES"; else python3 tools/bench_layers.py $ARGS --autotune --save-tiles "$OUT/tiles.json" > /dev/null 2>&1; ARGS="$ARGS --tiles $OUT/tiles.json"; fi
MOPS=SQ_INSTS_VALU_MFMA_MOPS_F32; [ "$PREC" != f32 ] && MOPS=SQ_INSTS_VALU_MFMA_MOPS_BF16
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY $MOPS SQ_ACTIVE_INST_VALU \
    --output-format csv -d "$OUT/p1" -o pmc -- python3 "$ROOT/tools/bench_layers.py" $ARGS > "$OUT/layers_p1.log" 2> "$OUT/p1.err")
(cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/p2" -o pmc -- python3 "$ROOT/tools/bench_layers.py" $ARGS > "$OUT/layers_p2.log" 2> "$OUT/p2.err")
cp "$(find "$OUT/p1" -name '*counter_collection.csv' | head -1)" "$OUT/p1.csv"
cp "$(find "$OUT/p2" -name '*counter_collection.csv' | head -1)" "$OUT/p2.csv"
rm -rf "$OUT/p1" "$OUT/p2"
python3 tools/pmc_kernel_util.py "$OUT/p1.csv" "$OUT/p2.csv" | tee "$OUT/util.txt"
