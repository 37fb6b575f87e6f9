#!/usr/bin/env python3
"""Where the result stores sit in the instruction stream of a kernel (device assembly from `hipcc --cuda-device-only -S`): per kernel the
instruction count, the store count, a 40-bin histogram of the store positions, the distribution of the gaps between consecutive stores and the
opcode mix.  Usage: tools/isa_store_histogram.py file.s [name-substring]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r"\n(_Z[^\n:]*):[^\n]*\n", txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split("s_endpgm")[0]
    if want not in name or "s_endpgm" not in parts[i + 1]:
        continue
    ins = [l.strip() for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    st = [k for k, l in enumerate(ins) if l.startswith(("buffer_store_dwordx4", "buffer_store_dwordx2", "global_store"))]
    print(name[:100], len(ins), "instructions", len(st), "stores")
    n = len(ins)
    bins = [0] * 40
    for k in st:
        bins[k * 40 // n] += 1
    print(" store positions (40 bins):", bins)
    gaps = [b - a for a, b in zip(st, st[1:])]
    c = collections.Counter(min(g, 100) // 10 * 10 for g in gaps)
    print(" gaps between stores (instructions, binned by 10):", sorted(c.items()))
    ops = collections.Counter(l.split()[0] for l in ins if l.split())
    print(" opcode mix:", ops.most_common(28))
