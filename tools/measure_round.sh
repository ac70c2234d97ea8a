#!/bin/bash
# One round's profile evidence, run ON the GPU box:  gpurun -- 'bash tools/measure_round.sh r02m'
#   per workload: rocprofv3 --kernel-trace --stats of tools/prof_driver.py (3 steps) and two --pmc passes (FETCH_SIZE, WRITE_SIZE;
#   one counter per pass, no trace domains with --pmc), all under gpurun_out/<tag>/.  tools/pmc_summary.py condenses the pmc
#   passes into profiles/<round>_pmc_traffic.json afterwards (here, where git is).
set -u
TAG=${1:-r02m}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT/pmc
cd /tmp; export TMPDIR=/tmp
# config  level  n  size
while read CFG L N S; do
  KEY=${CFG}_L${L}_${N}x${S}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$KEY -o s -- python $R/tools/prof_driver.py $N $S $L 3 > $OUT/${KEY}_driver.json 2> $OUT/${KEY}_stats.err
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc/${KEY}_$C -o p -- python $R/tools/prof_driver.py $N $S $L 1 > /dev/null 2> $OUT/pmc/${KEY}_$C.err
  done
  f=$(find $OUT/stats_$KEY -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/${KEY}_kernel_stats.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    f=$(find $OUT/pmc/${KEY}_$C -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $OUT/pmc/${KEY}_$C.csv
  done
  rm -rf $OUT/stats_$KEY $OUT/pmc/${KEY}_FETCH_SIZE $OUT/pmc/${KEY}_WRITE_SIZE
  cat $OUT/${KEY}_driver.json
done <<LIST
metric 3 65536 65536
3 1 65536 65536
5shape 3 65536 131072
LIST
ls -la $OUT $OUT/pmc
