# ISA census of one kernel of a (reduced) translation unit: bash tools/isa/census.sh file.hip <mangled-name-prefix> [hipcc flags]
F=$1; K=$2; shift; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S "$@" -o /tmp/isa/census.s $F 2>&1 | grep -E "error" | head
python3 - "$K" <<'P'
import re, sys, collections
s = open('/tmp/isa/census.s').read(); K = sys.argv[1]
m = re.search(r'^(%s\w*):' % K, s, re.M); st = m.start(); en = s.index('.Lfunc_end', st); body = s[st:en]
open('/tmp/isa/census_body.s', 'w').write(body)
c = collections.Counter(l.split()[0] for l in body.splitlines() if l.startswith('\t') and not l.strip().startswith(('.', ';')))
print('instructions', sum(c.values()))
for k in ('flat_load', 'global_load', 'ds_read', 'ds_write', 'global_store', 'flat_store', 'scratch', 's_waitcnt'):
    print(' ', k, sum(v for n, v in c.items() if n.startswith(k)))
for l in s[en:en + 4000].splitlines():
    if any(k in l for k in ('NumVgprs', 'ScratchSize', 'Occupancy', 'LDSByteSize', 'NumSgprs')): print(' ', l.strip())
P
