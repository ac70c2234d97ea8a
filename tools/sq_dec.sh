# SQ instruction / cycle counters of the decode kernels on config 2 (one rocprofv3 --pmc pass, no trace domains; counter collection serialises the kernels):
#   bash tools/sq_dec.sh <tag> [ENV=.. ...]   -> one line per (kernel launch of the LAST decompress call, counter)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  rm -rf $OUT/sqd_$TAG; env "$@" timeout 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/sqd_$TAG -o p -- python $R/bench.py --config 2 --steps 1 --warmup 1 --skip-cpu > /dev/null 2> $OUT/sqd_$TAG.err
  f=$(find $OUT/sqd_$TAG -name '*counter_collection.csv' | head -1)
  python - <<PY
import csv, collections
rows = collections.OrderedDict()
try:
    for r in csv.DictReader(open("$f")):
        k = r.get("Kernel_Name", "?").split("(")[0]
        if "zj_dec" not in k: continue
        rows.setdefault((int(r["Dispatch_Id"]), k), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    keys = sorted(rows)
    last = max(i for i, kk in enumerate(keys) if "prep" in kk[1])
    for kk in keys[last:]:
        c = rows[kk]; print("$TAG", kk[1][:34].ljust(34), " ".join("%s %.3e" % (n.replace("SQ_", ""), v) for n, v in sorted(c.items())))
except Exception as e: print("$TAG sq failed", e); print(open("$OUT/sqd_$TAG.err").read()[-600:])
PY
done
rm -rf $OUT/sqd_$TAG
