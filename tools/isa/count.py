#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc -S listing (device asm).
usage: count.py listing.s kernel_substring [--top N]
Prints: total instructions, per-class counts (salu exec-mask bookkeeping, branches, v_mov, 64-bit shifts/adds, multiplies,
memory), and the N most frequent mnemonics.  Used for the profiles/r04_* instruction-diet records."""
import collections, re, sys

def kernel_lines(path, name):
    out, on = [], False
    for ln in open(path, errors="replace"):
        if not on:
            if re.match(r"^[_A-Za-z0-9]*%s[_A-Za-z0-9]*:" % re.escape(name), ln): on = True
            continue
        if ln.startswith(".Lfunc_end"): break
        out.append(ln)
    return out

def census(lines):
    c = collections.Counter()
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"): continue
        m = s.split()[0]
        if not re.match(r"^[a-z_0-9]+$", m): continue
        c[m] += 1
    return c

CLASSES = [
    ("exec-mask salu", r"^s_(and|or|andn2|xor|orn2|not|mov|cselect|nand|nor)_(b64|saveexec_b64)$|^s_(and|or|andn2|xor)_saveexec_b64$"),
    ("branch", r"^s_cbranch|^s_branch"),
    ("s_waitcnt/nop", r"^s_waitcnt|^s_nop"),
    ("other salu", r"^s_"),
    ("v_mov", r"^v_mov_b32|^v_mov_b64|^v_accvgpr"),
    ("v_cndmask", r"^v_cndmask"),
    ("v_cmp", r"^v_cmp"),
    ("64-bit shift", r"^v_(lshl|lshr|ashr)(rev)?_b64"),
    ("64-bit addr add", r"^v_lshl_add_u64|^v_add_co|^v_addc"),
    ("multiply", r"^v_mul|^v_mad_u64|^v_mad_u32|^v_mad_i"),
    ("byte perm/align", r"^v_perm|^v_alignb|^v_bfe|^v_bfi|^v_alignbit"),
    ("global load", r"^global_load|^buffer_load|^flat_load"),
    ("global store/atomic", r"^global_store|^buffer_store|^flat_store|^global_atomic|^flat_atomic"),
    ("lds", r"^ds_"),
    ("scratch", r"^scratch_"),
    ("other valu", r"^v_"),
]

def main():
    path, name = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    lines = kernel_lines(path, name)
    c = census(lines)
    total = sum(c.values())
    print(f"kernel *{name}*: {total} static instructions")
    left = dict(c)
    for label, pat in CLASSES:
        n = sum(v for k, v in left.items() if re.match(pat, k))
        for k in [k for k in left if re.match(pat, k)]: del left[k]
        print(f"  {label:22s} {n:6d}  {100.0 * n / max(total, 1):5.1f} %")
    if left: print("  unclassified", left)
    print("  top mnemonics:", ", ".join(f"{k} {v}" for k, v in c.most_common(top)))
    meta = [ln.strip() for ln in open(path, errors="replace") if name in ln and (".num_vgpr" in ln or ".numbered_sgpr" in ln or "scratch" in ln.lower() and ".set" in ln)]
    for m in meta[:6]: print("  ", m.split(".set ")[-1][-60:])

if __name__ == "__main__":
    main()
