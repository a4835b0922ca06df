"""Static look at a device assembly file (hipcc -S --cuda-device-only): per kernel, the number of vector-memory loads and the number of
dependent round trips along the straight-line code (loops count once): an s_waitcnt vmcnt(n) starts a new round trip when it waits for a load
that was issued after the previous such wait — draining one batch of requests step by step (vmcnt(7), vmcnt(6), ...) is ONE trip.  A kernel
with as many trips as loads requests one thing at a time.   python tools/asm_roundtrips.py build/asm/x.s"""
import re, sys
out, name = [], None
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name, issued, done, at_trip, trips = m.group(1), 0, 0, 0, 0
        continue
    if name is None:
        continue
    t = line.strip()
    if t.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        issued += 1
    elif t.startswith("s_waitcnt") and "vmcnt" in t:
        target = issued - int(re.search(r"vmcnt\((\d+)\)", t).group(1))
        if target > done:
            if target > at_trip:
                trips += 1
                at_trip = issued
            done = target
    elif t.startswith(".Lfunc_end"):            # (not s_endpgm: a kernel whose DMA waves return early has several)
        out.append((name, issued, trips)); name = None
for n, l, t in sorted(out, key=lambda r: -r[2]):
    if l:
        print(f"{t:4d} trips / {l:4d} loads  {n[:110]}")
