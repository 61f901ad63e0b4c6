#!/bin/bash
# Runs on the GPU box: the parity reports and the chain stamps behind DESIGN.md section 2 / 7 (output under gpurun_out/<tag>rep/; copy into profiles/).
#   bash tools/reports_round.sh r05        (needs scratch/libmisti_stamp.so and scratch/libmisti_work.so: python -m misti_amd.build --out ... -DMISTI_STAMP / -DMISTI_WORK_COUNTERS=1)
TAG=${1:-r05}
OUT=gpurun_out/${TAG}rep
mkdir -p "$OUT"
python3 tools/parity_report.py > "$OUT/parity_goldens.txt" 2> "$OUT/err.txt"
echo "parity report done"
for s in 1 2 3 4 5 6 7 8; do
  python3 tools/random_campaign.py --ref tests/golden/campaign_seed$s.json.gz --models 600 --seed $s > "$OUT/random_campaign_seed$s.txt" 2>> "$OUT/err.txt"
done
echo "campaign reports done"
for wl in config2 config3 config5; do
  if [ -f scratch/libmisti_stamp.so ]; then MISTI_LIB_AB=1 MISTI_LIB=scratch/libmisti_stamp.so python3 tools/stamp_run.py $wl > "$OUT/stamp_$wl.txt" 2>> "$OUT/err.txt"; fi
  if [ -f scratch/libmisti_work.so ]; then MISTI_LIB_AB=1 MISTI_LIB=scratch/libmisti_work.so python3 tools/stamp_run.py $wl > "$OUT/work_$wl.txt" 2>> "$OUT/err.txt"; fi
done
hipcc --offload-arch=gfx950 -O3 -o "$OUT/exec_skip" tools/ub/exec_skip.hip 2>> "$OUT/err.txt" && "$OUT/exec_skip" > "$OUT/exec_skip.txt" 2>> "$OUT/err.txt"
echo "reports_round $TAG done"
