R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for t in old new old new; do
  if [ $t = old ]; then export ZJNI_LIB=$R/zstd-jni_amd/lib/libzjni_amd_old.so; else unset ZJNI_LIB; fi
  AB_TAG=$t ZJNI_PROFILE=1 python tools/prof_cdict.py 262144 3 2 1 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['tag'], 'compress %.1f ms %.2f GiB/s' % (d['compress_ms'], d['compress_GiBps']), d['entropy_kcycles_per_frame'])"
done
