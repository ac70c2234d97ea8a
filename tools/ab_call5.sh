# block-level and window-level cycle profile of the multi-block kernel on bench config 1's data (xml slices)   -> gpurun_out/call5.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{ PROF_DATA=xml ZJNI_LIB=$R/zstd-jni_amd/lib/libzjni_amd_zxprof.so AB_TAG=zxprof timeout 120 python tools/prof_driver.py 1024 1048576 3 1 2>&1 | grep "^zx \|compress_ms" | sort | awk 'NR % 4 == 1' | cut -c1-330 | head -50; } > $OUT/call5.txt 2>&1
cat $OUT/call5.txt
