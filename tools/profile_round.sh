#!/bin/bash
# Runs on the GPU box (from the repo root): the bench lines plus the rocprofv3 passes the
# numbers in DESIGN.md section 6 / profiles/ come from.  Output goes to gpurun_out/<tag>/;
# tools/summarize_profiles.py turns it into profiles/<tag>_*.{csv,json,txt}.
#
#   bash tools/profile_round.sh r03 && python tools/summarize_profiles.py r03
#
# Counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with a trace domain) and use
# one stream so that a launch's counters belong to that launch alone.  Under rocprofv3 the
# program itself follows `--` (python3 bench.py ...): no env / bash -c hop.
set -e -o pipefail
TAG=${1:-r05}
PART=${2:-all}          # all | bench | trace | pmc
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -c "from misti_amd import _lib; print(_lib.build_id())" > "$OUT/build_id.txt"      # the counters below belong to THIS build (profiles/pmc_latest.json: build_id)
SHORT="--steps 4 --warmup 2 --streams 1 --no-cpu-baseline --no-extra-legs --min-seconds 0"
WORKLOADS="config2 config2x16 config3 config5"

if [ "$PART" = all ] || [ "$PART" = bench ]; then
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench default done"
python3 bench.py --streams 1 --steps 16 --warmup 4 --no-cpu-baseline --no-extra-legs > "$OUT/bench_serial.json" 2>> "$OUT/bench.err"
for wl in config3 config4 config5; do
  python3 bench.py --workload $wl --steps 64 --warmup 8 --cpu-seconds 8 > "$OUT/bench_$wl.json" 2>> "$OUT/bench.err"
  echo "bench $wl done"
done
python3 bench.py --workload config2x16 --steps 32 --warmup 4 --no-cpu-baseline > "$OUT/bench_config2x16.json" 2>> "$OUT/bench.err"
python3 bench.py --steps 20 --warmup 20 > "$OUT/bench_driver_steps20.json" 2>> "$OUT/bench.err"      # the driver's own invocation shape (one 20-step region per repeat)
python3 bench.py --fit default --no-cpu-baseline --no-extra-legs > "$OUT/bench_default_fit.json" 2>> "$OUT/bench.err"
python3 bench.py --workload config3-search --steps 2 --warmup 1 > "$OUT/bench_config3_search.json" 2>> "$OUT/bench.err"
if python3 bench.py --workload config3-basinhopping --steps 1 --warmup 1 > "$OUT/bench_config3_basinhopping.json" 2>> "$OUT/bench.err"; then echo "basinhopping done"; else echo "basinhopping leg failed (see bench.err)"; fi
# the RCCL path with one rank (rehearsal of --gpus N): weak (headline) and strong (ONE grid sharded; configs 4 and 5)
python3 bench.py --force-dist --no-cpu-baseline --no-extra-legs > "$OUT/bench_dist_weak.json" 2>> "$OUT/bench.err"
python3 bench.py --force-dist --scaling strong --workload config4 --steps 64 --warmup 8 --no-cpu-baseline --no-extra-legs > "$OUT/bench_dist_strong_config4.json" 2>> "$OUT/bench.err"
python3 bench.py --force-dist --scaling strong --workload config5 --steps 64 --warmup 8 --no-cpu-baseline --no-extra-legs > "$OUT/bench_dist_strong_config5.json" 2>> "$OUT/bench.err"
echo "bench dist done"
fi

if [ "$PART" = all ] || [ "$PART" = trace ]; then
for wl in $WORKLOADS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_serial_$wl" -- python3 bench.py --workload $wl --streams 1 --steps 16 --warmup 4 --no-cpu-baseline --no-extra-legs --min-seconds 0 > "$OUT/bench_serial_prof_$wl.json" 2>> "$OUT/bench.err"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_pipelined" -- python3 bench.py --no-cpu-baseline --no-extra-legs > "$OUT/bench_pipelined_prof.json" 2>> "$OUT/bench.err"
echo "kernel traces done"
fi

if [ "$PART" = all ] || [ "$PART" = pmc ]; then
# the replicate epilogue (the HBM-write-bound kernel): bench.py --workload config4 runs misti_llk_dev at 256 x 1 000 and, last, at 65 536 x 1 000
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_llk" -- python3 bench.py --workload config4 --steps 4 --warmup 2 --streams 1 --no-cpu-baseline --min-seconds 0 > /dev/null 2>> "$OUT/bench.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_llk" -- python3 bench.py --workload config4 --steps 4 --warmup 2 --streams 1 --no-cpu-baseline --min-seconds 0 > /dev/null 2>> "$OUT/bench.err"
echo "pmc llk done"
# the default fit on the headline grid (no PMC pass existed for any default-fit workload)
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
    --output-format csv -d "$OUT/pmc_sq_config2_default" -- python3 bench.py --workload config2 --fit default $SHORT > /dev/null 2>> "$OUT/bench.err"
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT \
    --output-format csv -d "$OUT/pmc_f64_config2_default" -- python3 bench.py --workload config2 --fit default $SHORT > /dev/null 2>> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_serial_config2_default" -- python3 bench.py --workload config2 --fit default --streams 1 --steps 16 --warmup 4 --no-cpu-baseline --no-extra-legs --min-seconds 0 > "$OUT/bench_serial_prof_config2_default.json" 2>> "$OUT/bench.err"
echo "pmc default fit done"
for wl in $WORKLOADS; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$wl" -- python3 bench.py --workload $wl $SHORT > /dev/null 2>> "$OUT/bench.err"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$wl" -- python3 bench.py --workload $wl $SHORT > /dev/null 2>> "$OUT/bench.err"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
      --output-format csv -d "$OUT/pmc_sq_$wl" -- python3 bench.py --workload $wl $SHORT > /dev/null 2>> "$OUT/bench.err"
  rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY \
      --output-format csv -d "$OUT/pmc_sq2_$wl" -- python3 bench.py --workload $wl $SHORT > /dev/null 2>> "$OUT/bench.err" || echo "second SQ pass failed for $wl"
  # fp64 arithmetic priced as fp64: wave-instructions by class - an FMA is two flops per lane, the rest one
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT \
      --output-format csv -d "$OUT/pmc_f64_$wl" -- python3 bench.py --workload $wl $SHORT > /dev/null 2>> "$OUT/bench.err" || echo "fp64 class pass failed for $wl"
  echo "pmc $wl done"
done
fi
echo "profile_round $TAG $PART done"
