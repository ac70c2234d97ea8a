#!/bin/bash
# A/B of library variants over bench configs on one box: bash tools/ab_cfg.sh "<configs>" "<label>=<ENV=..,ENV=..> ..."   -> compress / decompress GiB/s and the kernel times per (config, label)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
CFGS=$1; shift
for cfg in $CFGS; do for spec in "$@"; do
  label=${spec%%=*}; envs=$(echo "${spec#*=}" | tr ',' ' ')
  echo -n "config $cfg $label: "; env $envs timeout 400 python bench.py --config $cfg --steps ${AB_STEPS:-4} --warmup 1 --skip-cpu ${AB_ARGS:-} 2>&1 | tail -1 | python -c "
import sys,json
try:
    j=json.loads(sys.stdin.read()); print('value %.2f compress %s decompress %s' % (j['value'], j.get('compress_GiBps_per_gpu') and round(j['compress_GiBps_per_gpu'],2), j.get('decompress_GiBps_per_gpu') and round(j['decompress_GiBps_per_gpu'],1)), {k:round(v,2) for k,v in j['kernel_ms'].items() if v > 0.05})
except Exception as e: print('FAILED', e)"
done; done
