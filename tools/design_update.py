"""Replace the numbers table of DESIGN.md section 7 (between the numbers:begin / numbers:end markers) with
tools/design_numbers.py's output on the committed bench lines."""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
table = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "design_numbers.py"),
                                 os.path.join(ROOT, "profiles", "r06_bench_line_default.json"),
                                 os.path.join(ROOT, "profiles", "r06_bench_line_k20.json")], text=True)
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
a = s.index("<!-- numbers:begin")
a = s.index("\n", a) + 1
b = s.index("<!-- numbers:end -->")
open(path, "w").write(s[:a] + table + s[b:])
print("DESIGN.md section 7 table: %d lines" % table.count("\n"))
