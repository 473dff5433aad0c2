#!/usr/bin/env python3
"""Registers, code size and s_waitcnt summary of the kernels in a device assembly file whose names contain a pattern:
   hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S file.hip -o out.s; tools/isa_stats.py out.s PATTERN"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)^\.Lfunc_end\d+:', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    stats = {k: re.search(r'\.set ' + re.escape(name) + r'\.' + k + r', (\d+)', s) for k in ("num_vgpr", "num_agpr", "numbered_sgpr", "private_seg_size")}
    ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    print(name[:90])
    print("  ", {k: v and int(v.group(1)) for k, v in stats.items()}, "instructions", len(ins))
    c = collections.Counter(ins)
    print("   vmem loads", sum(v for k, v in c.items() if k.startswith(("global_load", "buffer_load", "flat_load"))), "stores",
          sum(v for k, v in c.items() if k.startswith(("global_store", "buffer_store", "flat_store"))), "scratch", sum(v for k, v in c.items() if k.startswith("scratch_")),
          "s_load", sum(v for k, v in c.items() if k.startswith("s_load")), "ds", sum(v for k, v in c.items() if k.startswith("ds_")))
    print("   waits", collections.Counter(w.strip() for w in re.findall(r's_waitcnt ([^\n]*)', body)).most_common(10))
