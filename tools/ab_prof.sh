# per-wave placement and cycle profile of the match kernel (library built with -DZL_PROFILE): one step per variant, raw printf output kept
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
export ZJNI_LIB=$R/zstd-jni_amd/lib/libzjni_amd_prof.so
while read tag envs; do
  [ -z "$tag" ] && continue
  env AB_TAG=$tag $envs timeout 120 python $R/tools/prof_driver.py 65536 65536 3 2 > $OUT/prof_$tag.txt 2>&1
  grep '"tag"' $OUT/prof_$tag.txt | cut -c1-200
done < $1
