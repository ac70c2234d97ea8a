#!/usr/bin/env python3
"""Static instructions of zj_enc_match_run_kernel per round_t variant (the `case` line of ZLaneR::round the code was inlined from) and source line.
usage: by_variant.py listing_g.s kernel_substring header.h caseline=label ...   e.g.  118=COUNT 119=POST 120=START 121=SEARCH"""
import collections, re, sys
path, name, hdr = sys.argv[1], sys.argv[2], sys.argv[3]
cases = dict((int(a.split("=")[0]), a.split("=")[1]) for a in sys.argv[4:])
on = False; cur = ("-", 0, "?")
tot = collections.Counter(); perv = collections.Counter(); book = collections.Counter()
for ln in open(path, errors="replace"):
    if not on:
        if re.match(r"^[_A-Za-z0-9]*%s[_A-Za-z0-9]*:" % re.escape(name), ln): on = True
        continue
    if ln.startswith(".Lfunc_end"): break
    s = ln.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+).*?;\s*(\S+?):(\d+):\d+(.*)$", s)
    if m:
        f = m.group(3).split("/")[-1]; l = int(m.group(4)); v = "-"
        for mm in re.finditer(re.escape(hdr) + r":(\d+):\d+", m.group(5)):
            if int(mm.group(1)) in cases: v = cases[int(mm.group(1))]
        cur = (v, l, f); continue
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"): continue
    op = s.split()[0]
    if not re.match(r"^[a-z_0-9]+$", op): continue
    tot[cur] += 1; perv[cur[0]] += 1
    if re.match(r"^s_(and|or|andn2|xor|orn2|not|mov|cselect)_b64$|saveexec|^s_cbranch|^s_branch", op): book[cur[0]] += 1
print("per variant:", dict(perv), "bookkeeping:", dict(book))
vs = sorted(set(k[0] for k in tot))
lines = sorted(set((k[2], k[1]) for k in tot))
print("%-26s" % "file:line" + "".join("%8s" % v for v in vs))
for f, l in lines:
    row = [tot.get((v, l, f), 0) for v in vs]
    if max(row) >= 6: print("%-26s" % (f + ":" + str(l)) + "".join("%8d" % x for x in row))
