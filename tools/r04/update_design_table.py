#!/usr/bin/env python
"""Puts tools/bench_table.py's table of profiles/r04_bench.json between the bench-table markers of DESIGN.md."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
table = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_table.py"), os.path.join(ROOT, "profiles", "r04_bench.json"),
                        os.path.join(ROOT, "profiles", "r03_bench.json")], capture_output=True, text=True, check=True).stdout
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a, b = s.index("<!-- bench-table:begin -->"), s.index("<!-- bench-table:end -->")
s = s[:a] + "<!-- bench-table:begin -->\n" + table.rstrip() + "\n" + s[b:]
open(p, "w").write(s)
print(table)
