#!/bin/bash
# HBM-side traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass, no trace domains) of the library AS BUILT, plus the
# rocprofv3 --kernel-trace --stats summary of the same driver command:   gpurun -- 'bash tools/pmc_traffic.sh [tag]'   (the last GPU action of a round)
# Output under gpurun_out/<tag>/; `python tools/pmc_summary.py gpurun_out/<tag> r03` (where git is) condenses it into profiles/r03_pmc_traffic.json,
# every record stamped with the zjni_build_stamp() the driver printed — bench.py quotes roofline.traffic only when that equals its own library's.
# A list line is "<config> <level> <n> <size> [ENV=.. for the --pmc passes only]".  Counter collection SERIALISES kernels, and which of the flag kernel and the match kernel (two streams) it
# lets go first varies from pass to pass — so the metric line runs with ZJNI_NEED_INLINE=1 (flag kernel ahead of the match kernel on one stream: every frame has its
# flags from its first round; in production they arrive during the first ~30 ms) and a second key, metricnoflags, with ZJNI_NEED=0 gives the other bound.
TAG=${1:-r03pmc}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT/pmc
cd /tmp; export TMPDIR=/tmp
while read CFG L N S ENVS; do
  [ -z "$CFG" ] && continue
  KEY=${CFG}_L${L}_${N}x${S}
  # the driver: tools/prof_driver.py <n> <size> <level> <steps>; config 4 (dictionary, 4 KiB JSON-like records, bench.py's own dictionary): tools/prof_cdict.py <n> <level> <steps> 1 bench.  PROF_* settings (the data the
  # driver generates: PROF_DATA=xml = config 1's slices of the xml fixture) belong to the stats pass as well as to the counter passes.
  DATA_ENVS=$(echo $ENVS | tr ' ' '\n' | grep '^PROF_' | tr '\n' ' ')
  if [ "$CFG" = 4 ]; then DRV3="python $R/tools/prof_cdict.py $N $L 3 1 bench"; DRV1="python $R/tools/prof_cdict.py $N $L 1 1 bench"; else DRV3="python $R/tools/prof_driver.py $N $S $L 3"; DRV1="python $R/tools/prof_driver.py $N $S $L 1"; fi
  timeout 300 env $DATA_ENVS rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$KEY -o s -- $DRV3 > $OUT/${KEY}_driver.json 2> $OUT/${KEY}_stats.err
  f=$(find $OUT/stats_$KEY -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/${KEY}_kernel_stats.csv
  rm -rf $OUT/stats_$KEY
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 env $ENVS rocprofv3 --pmc $C --output-format csv -d $OUT/pmc/${KEY}_$C -o p -- $DRV1 > $OUT/pmc/${KEY}_${C}_driver.json 2> $OUT/pmc/${KEY}_$C.err
    f=$(find $OUT/pmc/${KEY}_$C -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $OUT/pmc/${KEY}_$C.csv
    rm -rf $OUT/pmc/${KEY}_$C
  done
  tail -1 $OUT/${KEY}_driver.json | cut -c1-400
done <<LIST
${PMC_LIST:-metric 3 65536 65536 ZJNI_NEED_INLINE=1}
LIST
ls $OUT $OUT/pmc
