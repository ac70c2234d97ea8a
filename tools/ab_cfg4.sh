# Same-box A/B of bench configs 4 and 3 over library variants: tools/ab_cfg4.sh old vA vB ... (zstd-jni_amd/lib/libzjni_amd_<v>.so;
# "new" = the library as built).  Box-to-box variance is 5-10 %, so only lines from ONE call compare.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
L=zstd-jni_amd/lib
cp $L/libzjni_amd.so /tmp/new.so
run() { python bench.py --config $1 --skip-cpu --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$2', d['config']['name'], 'compress %.2f GiB/s  call %.1f ms' % (d['compress_GiBps_per_gpu'], d['kernel_ms']['compress_call']), {k: round(v,1) for k,v in d['kernel_ms'].items() if 'match' in k or 'rest' in k})"; }
for rep in 1 2; do
for v in "$@"; do
  if [ $v = new ]; then cp /tmp/new.so $L/libzjni_amd.so; else cp $L/libzjni_amd_$v.so $L/libzjni_amd.so; fi
  touch $L/libzjni_amd.so; run 4 $v; run 3 $v
done; done
cp /tmp/new.so $L/libzjni_amd.so
