"""Per kernel of a `hipcc -S` listing: every s_waitcnt that names vmcnt inside a LOOP block, with the loads / stores of that loop.
A wave's vmcnt counts loads AND stores in issue order, so a loop that stores and then waits for any vector-memory load with a count
the compiler could not establish (vmcnt(0)) waits for its own stores' HBM round trip every iteration.
usage: python tools/asm_loop_waits.py FILE.s [kernel-name-substring ...]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
keys = sys.argv[2:]
kern = None
blocks = []          # (kernel, label, in_loop, [instr lines])
cur = None
for ln in lines:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        kern = m.group(1); cur = None; continue
    if kern is None: continue
    m = re.match(r"^(\.LBB\d+_\d+|; %bb\.\d+):(.*)", ln)
    if m:
        cur = [kern, m.group(1), "Loop" in m.group(2), []]; blocks.append(cur); continue
    if "s_endpgm" in ln: kern = None; cur = None; continue
    if cur is not None:
        if "in Loop" in ln or "Loop Header" in ln or "Inner Loop" in ln: cur[2] = True
        t = ln.strip()
        if t and not t.startswith(";"): cur[3].append(t)
def short(k):
    m = re.search(r"\d+([a-z_0-9]+_kernel)I(.*?)E", k)
    return (m.group(1) + "<" + m.group(2) + ">") if m else k[:60]
seen = {}
for k, lab, loop, ins in blocks:
    if keys and not any(s in k for s in keys): continue
    if not loop: continue
    w = [i for i in ins if i.startswith("s_waitcnt") and "vmcnt" in i]
    if not w: continue
    ld = sum(1 for i in ins if re.match(r"(global|buffer|flat)_load", i)); st = sum(1 for i in ins if re.match(r"(global|buffer|flat)_store", i))
    mf = sum(1 for i in ins if i.startswith("v_mfma"))
    print(f"{short(k):60s} {lab:12s} loads {ld:3d} stores {st:3d} mfma {mf:3d}  waits: {', '.join(x.split(None, 1)[1] for x in w)}")
