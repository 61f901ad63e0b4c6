#!/bin/bash
# Runs on the GPU box (from the repo root): the bench line plus the rocprofv3 passes the
# numbers in DESIGN.md section 6 / profiles/ come from.  Output goes to gpurun_out/<tag>/;
# tools/summarize_profiles.py turns it into profiles/<tag>_*.{csv,json}.
#
#   tools/profile_round.sh r01
#
# Counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass) and use
# one stream so that a launch's counters belong to that launch alone.
set -e -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
SHORT="--steps 4 --warmup 2 --streams 1 --no-cpu-baseline"

python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python bench.py --streams 1 --steps 16 --warmup 4 --no-cpu-baseline > "$OUT/bench_serial.json" 2>> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_serial" -- python bench.py --streams 1 --steps 16 --warmup 4 --no-cpu-baseline > "$OUT/bench_serial_prof.json" 2>> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_pipelined" -- python bench.py --no-cpu-baseline > "$OUT/bench_pipelined_prof.json" 2>> "$OUT/bench.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python bench.py $SHORT > /dev/null 2>> "$OUT/bench.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python bench.py $SHORT > /dev/null 2>> "$OUT/bench.err"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
    --output-format csv -d "$OUT/pmc_sq" -- python bench.py $SHORT > /dev/null 2>> "$OUT/bench.err"
echo "profile_round $TAG done"
