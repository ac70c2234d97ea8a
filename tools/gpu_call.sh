#!/bin/bash
# One parametrised GPU-box script for every measurement call of a round (replaces the per-call scripts of rounds 3-4):
#   gpurun --timeout T -- 'bash tools/gpu_call.sh <name> "<step>" "<step>" ...'        -> gpurun_out/<name>.txt (and printed)
# Steps are shell lines evaluated with these helpers in scope (V = a variant built here by tools/build_variant.sh <V>; "prod" = the product library):
#   ab  "<V>[:ENV=..,ENV=..] ..." [n size level]      A/B of variants on one box through tools/ab.sh (STEPS=3): compress call, match kernel, decode, size fingerprint
#   sq  <V> [ENV=..]                                  SQ instruction / cycle counters of the level-3 match kernel (tools/sq_counters.sh)
#   zlprof <V> [n size level]                         one run of a -DZL_PROFILE build: per-phase cycles, per-wave cycles, frame finish rounds
#   gputest <pytest args>                             python -m pytest -m gpu ... (tail)
#   bench <bench.py args>                             one bench.py line
#   kstats <name> <cmd...>                            rocprofv3 --kernel-trace --stats of a command -> gpurun_out/<name>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
LIBD=$R/zstd-jni_amd/lib
NAME=$1; shift
libof() { [ "$1" = prod ] && echo $LIBD/libzjni_amd.so || echo $LIBD/libzjni_amd_$1.so; }
ab() {
  local f=$OUT/ab_lines_$$.txt; : > $f
  for spec in $1; do local v=${spec%%:*}; local envs=""; [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' '); echo "$(echo $spec | tr ':=,/' '____') ZJNI_LIB=$(libof $v) $envs" >> $f; done
  echo "== ab ${2:-65536} x ${3:-65536} L${4:-3}"; STEPS=${STEPS:-3} bash tools/ab.sh $f ${2:-65536} ${3:-65536} ${4:-3}; rm -f $f
}
sq() { local v=$1; shift; echo "== SQ counters $v $*"; bash tools/sq_counters.sh $v ZJNI_LIB=$(libof $v) "$@" | grep "match_run\|match_wide\|failed"; }
zlprof() { local v=$1; echo "== ZL_PROFILE $v"; AB_TAG=$v ZJNI_LIB=$(libof $v) ZJNI_NEED_INLINE=${INLINE:-0} timeout 120 python tools/prof_driver.py ${2:-65536} ${3:-65536} ${4:-3} 1 2>&1 | grep -v "^$" | tail -${TAIL:-14}; }
gputest() { echo "== pytest -m gpu $*"; timeout ${PYTEST_TIMEOUT:-900} python -m pytest -m gpu -x -q "$@" 2>&1 | tail -6; }
bench() { echo "== bench.py $*"; timeout 600 python bench.py "$@" 2>&1 | tail -2; }
kstats() { local n=$1; shift; echo "== rocprofv3 kernel stats $n: $*"; (cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/ks_$n; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_$n -o p -- "$@" > $OUT/ks_$n.log 2>&1); f=$(find $OUT/ks_$n -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/${n}_kernel_stats.csv && head -12 $f | cut -c1-160; rm -rf $OUT/ks_$n; }
# ktrace <V> [n size level]: kernel timeline (start / end in ms after the first kernel of the LAST compress call) from rocprofv3 --kernel-trace of tools/prof_driver.py
ktrace() { local v=$1; echo "== kernel timeline $v"; (cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/kt_$v; AB_TAG=$v ZJNI_LIB=$(libof $v) timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$v -o p -- python $R/tools/prof_driver.py ${2:-65536} ${3:-65536} ${4:-3} 1 > $OUT/kt_$v.log 2>&1); f=$(find $OUT/kt_$v -name '*kernel_trace.csv' | head -1); python3 - "$f" <<'P'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
last = max(i for i, r in enumerate(rows) if "classify" in r[2])          # the last compress call starts with its classify kernel
t0 = rows[last][0]
for s, e, k in rows[last:last + 16]: print("  %-44s start %8.3f ms  end %8.3f ms  (%.3f)" % (k, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
P
rm -rf $OUT/kt_$v; }
# ctrace <name> <cmd...>: kernels >= 1 ms and memory copies >= 1 ms of a command on one time axis (rocprofv3 --kernel-trace --memory-copy-trace)
ctrace() { local n=$1; shift; echo "== copy + kernel timeline $n: $*"; (cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/ct_$n; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/ct_$n -o p -- "$@" > $OUT/ct_$n.log 2>&1); python3 - $OUT/ct_$n <<'P'
import csv, sys, glob
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:40]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "?"))[:40]))
import os
mn = float(os.environ.get("CT_MIN_MS", "1")) * 1e6; pat = os.environ.get("CT_GREP", "")
rows = [r for r in rows if r[1] - r[0] >= mn and (not pat or any(p in r[2] for p in pat.split(",")))]; rows.sort()
if rows:
    t0 = rows[0][0]
    win = os.environ.get("CT_AROUND", "")           # "name,first,last": the rows between the first-th and the last-th kernel of that name (counted from 0)
    if win:
        nm, a, b = win.split(","); hits = [r for r in rows if nm in r[2]]
        if len(hits) > int(b): lo, hi = hits[int(a)][0] - 150000000, hits[int(b)][1] + 100000000; rows = [r for r in rows if lo <= r[0] <= hi]
    else: rows = rows[-90:]
    for s, e, k in rows: print("  %9.1f -> %9.1f ms (%7.1f)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, k))
P
rm -rf $OUT/ct_$n; }
{ for step in "$@"; do eval "$step"; done; } > $OUT/$NAME.txt 2>&1
cat $OUT/$NAME.txt
