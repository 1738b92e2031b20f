#!/bin/bash
# Copies the outputs of tools/prof_round.sh <tag> from gpurun_out/ (scratch) into profiles/ (tracked).
# usage (in the build container, after the gpurun call): bash tools/copy_profiles.sh r04
T=$1
cd "$(dirname "$0")/.."
for f in gpurun_out/prof_${T}_*_kernel_stats.csv; do
  [ -f "$f" ] && cp "$f" profiles/$(basename "$f" | sed "s/^prof_//")
done
for d in gpurun_out/pmc_${T}_*; do
  [ -d "$d" ] || continue
  tag=$(basename "$d" | sed "s/^pmc_//")
  for c in FETCH_SIZE WRITE_SIZE; do
    [ -f "$d/$c/run_counter_collection.csv" ] && cp "$d/$c/run_counter_collection.csv" profiles/${tag}_pmc_${c}.csv
  done
done
[ -f gpurun_out/${T}_prof_round_summary.txt ] && grep -v "^[EWI]2026" gpurun_out/${T}_prof_round_summary.txt > profiles/${T}_prof_round_summary.txt
ls profiles | grep "^${T}_" | wc -l
