#!/bin/bash
# A/B of decode variants on one box: bash tools/ab_dec.sh "<label>=<ENV=..,ENV=..> ..."   (ZJNI_LIB=<path> selects a variant library); config 2, 8 steps, two rounds per label
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for round in 1 2; do for spec in "$@"; do
  label=${spec%%=*}; envs=$(echo "${spec#*=}" | tr ',' ' ')
  echo -n "$label: "; env $envs timeout 300 python bench.py --config ${AB_CONFIG:-2} --steps 8 --warmup 2 --skip-cpu 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1), {k:round(v,2) for k,v in j['kernel_ms'].items() if v > 0.05})"
done; done
