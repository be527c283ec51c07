#!/usr/bin/env python3
"""Compressed op trace of one kernel from a hipcc -save-temps .s file (M=mfma, G=global_load_lds, D=ds_read, ...)."""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
s = open(path).read()
m = re.search(r"^(" + pat + r"[^\n:]*):[^\n]*\n(.*?)\n\s+s_endpgm", s, re.S | re.M)
if not m:
    sys.exit("kernel not found")
ops = []
for line in m.group(2).split("\n"):
    l = line.strip()
    if not l or l.startswith(";"):
        continue
    if l.startswith(".LBB") and l.endswith(":"):
        ops.append("\n" + l)
        continue
    if l.startswith("."):
        continue
    op = l.split()[0]
    if op.startswith("v_mfma"):
        ops.append("M")
    elif op.startswith("global_load_lds") or (op.startswith("buffer_load") and l.endswith("lds")):
        ops.append("G")
    elif op.startswith("global_load") or op.startswith("buffer_load"):
        ops.append("L")
    elif op.startswith("global_store") or op.startswith("global_atomic"):
        ops.append("S")
    elif op.startswith("ds_read"):
        ops.append("D")
    elif op.startswith("ds_write"):
        ops.append("W")
    elif op.startswith("s_barrier"):
        ops.append("|BAR|")
    elif op.startswith("s_waitcnt"):
        ops.append("w(" + l.split(None, 1)[1].replace(" ", "") + ")")
    elif op.startswith("s_setprio"):
        ops.append("p" + l.split()[1])
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        ops.append("<" + op.replace("s_cbranch_", "") + " " + l.split()[-1] + ">")
    elif op.startswith("v_"):
        ops.append("v")
    elif op.startswith("s_"):
        ops.append("s")
print(" ".join(ops))
