# HBM-side traffic of one library configuration on the metric workload: bash tools/pmc_one.sh <tag> ENV=.. ENV=..   (rocprofv3 --pmc, one counter per pass, no trace domains)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  env "$@" timeout 120 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o p -- python $R/tools/prof_driver.py 65536 65536 3 1 > /dev/null 2> $OUT/$C.err
  f=$(find $OUT/$C -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $OUT/$C.csv
  rm -rf $OUT/$C
done
python3 - <<PY
import csv, collections
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    try:
        for r in csv.DictReader(open("$OUT/%s.csv" % C)):
            k = r.get("Kernel_Name", "?").split("(")[0]; tot[k] += float(r.get("Counter_Value", 0)); cnt[k] += 1
    except Exception as e: print(C, "failed", e); continue
    for k, v in sorted(tot.items(), key=lambda x: -x[1])[:6]: print("$TAG", C, k[:60], "launches", cnt[k], "sum %.3e per launch %.3e" % (v, v / cnt[k]))
PY
