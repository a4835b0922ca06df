"""Static look at a device assembly file (hipcc -S --cuda-device-only): per kernel, the number of vector-memory loads and the number of
s_waitcnt vmcnt(...) points that have at least one load outstanding (= dependent round trips along the straight-line code; loops count
once).  A kernel whose waits are as many as its loads requests one thing at a time.   python tools/asm_roundtrips.py build/asm/x.s"""
import re, sys
name, loads, pend, trips, out = None, 0, 0, 0, []
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name, loads, pend, trips = m.group(1), 0, 0, 0
        continue
    if name is None:
        continue
    t = line.strip()
    if t.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        loads += 1; pend += 1
    elif t.startswith("s_waitcnt") and "vmcnt" in t:
        n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
        if pend > n:
            trips += 1; pend = n
    elif t.startswith("s_endpgm"):
        out.append((name, loads, trips)); name = None
for n, l, t in sorted(out, key=lambda r: -r[2]):
    if l:
        print(f"{t:4d} waits / {l:4d} loads  {n[:110]}")
