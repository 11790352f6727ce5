#!/usr/bin/env python3
"""Instruction histogram of one kernel from a hipcc -save-temps .s file: tools/isa_hist.py file.s mangled_substring"""
import collections, re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r"^(_Z\w*%s\w*):" % re.escape(key), s, re.M)
name = m.group(1)
i = s.index(name + ":"); j = s.index(".Lfunc_end", i)
ops = collections.Counter()
for line in s[i:j].splitlines():
    t = line.strip()
    if not t or t.startswith((".", ";")) or t.endswith(":"): continue
    ops[t.split()[0]] += 1
print(name, "total", sum(ops.values()))
for k, v in ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40): print(f"  {k:28s}{v}")
