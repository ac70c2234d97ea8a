# an experimental build of the library beside the product's: bash tools/build_variant.sh <suffix> <extra hipcc flags...>  -> zstd-jni_amd/lib/libzjni_amd_<suffix>.so
# (run here: hipcc cross-compiles; the .so travels to the GPU box; tools/prof_driver.py takes it through ZJNI_LIB)
R=$(cd $(dirname $0)/.. && pwd); S=$1; shift
ulimit -s unlimited 2>/dev/null || ulimit -s $(ulimit -H -s)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DZJNI_BUILD_STAMP=\"variant_$S\" "$@" -o $R/zstd-jni_amd/lib/libzjni_amd_$S.so $R/zstd-jni_amd/csrc/zj_kernels.hip 2>&1 | grep -v "occupancy target\|^ *[0-9]* |\|\^\|warnings generated" 
ls -la $R/zstd-jni_amd/lib/libzjni_amd_$S.so
