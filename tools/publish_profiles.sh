#!/bin/bash
# Turn one tools/collect_profiles.sh collection (gpurun_out/<tag>/) into the tracked summaries profiles/<prefix>_*.
#   usage: tools/publish_profiles.sh <tag> <prefix>      e.g.  tools/publish_profiles.sh r3s r03
# (a tag collected twice holds two runs per directory: the newest files are taken and the pass directories are reduced to them)
set -e
T=gpurun_out/$1
P=profiles/$2
newest() { ls -t $1 | head -1; }
cp "$(newest "$T/stats/run/*/*_kernel_stats.csv")" ${P}_bench_steps5_kernel_stats.csv
cp "$(newest "$T/stats_seq/run/*/*_kernel_stats.csv")" ${P}_bench_steps5_sequential_kernel_stats.csv
for pass in fetch write sq; do      # the summarisers glob a pass directory: leave only the newest run in it
  keep=$(newest "$T/$pass/run/*/*_counter_collection.csv")
  for f in $T/$pass/run/*/*_counter_collection.csv; do [ "$f" = "$keep" ] || rm -f "$f" "${f%_counter_collection.csv}"_*.csv; done
done
python tools/pmc_to_summary_csv.py $T/fetch/run FETCH_SIZE ${P}_pmc_fetch_size_summary.csv
python tools/pmc_to_summary_csv.py $T/write/run WRITE_SIZE ${P}_pmc_write_size_summary.csv
python tools/traffic_from_pmc.py $T/fetch/run $T/write/run ${P}_traffic.json
python tools/sq_summary.py $T/sq/run ${P}_sq_counters_summary.csv
tail -1 $T/bench_default.log > ${P}_bench_line.json
ls -la ${P}_*
