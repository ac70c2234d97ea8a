#!/usr/bin/env python3
"""Static instructions of one kernel attributed to source lines (listing from hipcc -S -gline-tables-only).
usage: by_line.py listing.s kernel_substring [file_substring]
For every source line: instructions, of which exec-mask salu + branches (bookkeeping) and v_mov.  Inlined code is attributed to
the innermost .loc (the line of the callee)."""
import collections, re, sys
path, name = sys.argv[1], sys.argv[2]
only = sys.argv[3] if len(sys.argv) > 3 else None
files = {}
for ln in open(path, errors="replace"):
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', ln) or re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"', ln)
    if m: files[int(m.group(1))] = m.group(2).split("/")[-1]
on = False; cur = (0, 0)
tot = collections.Counter(); book = collections.Counter(); mov = collections.Counter(); mul = collections.Counter()
for ln in open(path, errors="replace"):
    if not on:
        if re.match(r"^[_A-Za-z0-9]*%s[_A-Za-z0-9]*:" % re.escape(name), ln): on = True
        continue
    if ln.startswith(".Lfunc_end"): break
    s = ln.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"): continue
    op = s.split()[0]
    if not re.match(r"^[a-z_0-9]+$", op): continue
    tot[cur] += 1
    if re.match(r"^s_(and|or|andn2|xor|orn2|not|mov|cselect)_b64$|saveexec|^s_cbranch|^s_branch", op): book[cur] += 1
    if op.startswith("v_mov") or op.startswith("v_accvgpr"): mov[cur] += 1
    if op.startswith("v_mul_lo") or op.startswith("v_mul_hi") or op.startswith("v_mad_u64"): mul[cur] += 1
rows = sorted(tot.items(), key=lambda kv: (kv[0][0], kv[0][1]))
print(f"{'file:line':28s} {'instr':>6s} {'bookkeep':>8s} {'v_mov':>6s} {'mul':>4s}")
ft = collections.Counter()
for (f, l), n in rows:
    fn = files.get(f, str(f)); ft[fn] += n
    if only and only not in fn: continue
    print(f"{fn + ':' + str(l):28s} {n:6d} {book[(f, l)]:8d} {mov[(f, l)]:6d} {mul[(f,l)]:4d}")
print("per file:", dict(ft), "total", sum(tot.values()))
