#!/usr/bin/env python
"""Classify the address of a 'Memory access fault by GPU node-N ... on address 0x...' line against a faultdump
table (tools/faultdump/faultdump.c): device allocation / host buffer handed to a copy / mapped file / nothing."""
import re
import sys

dump, line = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
m = re.search(r"address (0x[0-9a-fA-F]+)", line)
if not m:
    print("no fault address in:", line)
    sys.exit(0)
addr = int(m.group(1), 16)
print(f"fault address {addr:#x} (page {addr >> 12:#x})")
try:
    text = open(dump).read()
except OSError as e:
    print("no dump:", e)
    sys.exit(0)
hits = []
for l in text.splitlines():
    mm = re.match(r"\s*(\w[\w*]*)\s+a=(0x[0-9a-f]+|\(nil\))\s+b=(0x[0-9a-f]+|\(nil\))\s+n=(\d+)", l)
    if mm:
        kind, a, b, n = mm.group(1), mm.group(2), mm.group(3), int(mm.group(4))
        for name, p in (("a", a), ("b", b)):
            if p != "(nil)":
                base = int(p, 16)
                if base <= addr < base + max(n, 1) + 4096:
                    hits.append(f"{kind} {name}={p} n={n} (offset {addr - base:+d})")
        continue
    mm = re.match(r"([0-9a-f]+)-([0-9a-f]+) (\S+) \S+ \S+ \S+\s*(.*)", l)
    if mm and int(mm.group(1), 16) <= addr < int(mm.group(2), 16):
        hits.append(f"maps: {l.strip()}")
print("\n".join(hits) if hits else "address is in no recorded allocation, copy operand or mapping")
