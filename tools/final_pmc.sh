#!/bin/bash
# HBM traffic of the level-3 metric workload (rocprofv3 --pmc, one counter per pass, no trace domains): gpurun -- 'bash tools/final_pmc.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r02zz/pmc; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 60 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o p -- python $R/tools/prof_driver.py 65536 65536 3 1 > /dev/null 2> $OUT/$C.err
  f=$(find $OUT/$C -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $OUT/metric_L3_65536x65536_$C.csv
  rm -rf $OUT/$C
done
ls -la $OUT
