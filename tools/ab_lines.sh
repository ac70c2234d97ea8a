# like tools/ab.sh, but prints the driver's whole JSON line per variant (phase cycles etc.): bash tools/ab_lines.sh <variants file> n size level steps
R=${GRAFT_REPO_ROOT:-/root/repo}
while read tag envs; do
  [ -z "$tag" ] && continue
  env AB_TAG=$tag $envs timeout 90 python $R/tools/prof_driver.py ${2:-65536} ${3:-65536} ${4:-3} ${5:-3} 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['tag'], 'compress %.1f ms match %.1f' % (d['compress_ms'], d['stages_ms']['match']), d.get('entropy_kcycles_per_frame'))"
done < $1
