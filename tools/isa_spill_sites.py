"""Where a kernel's scratch traffic sits, by source line and loop depth (no GPU needed).
   hipcc ... -gline-tables-only --save-temps --cuda-device-only -c qa_capi.hip
   python tools/isa_spill_sites.py FILE.s SUBSTRING_OF_MANGLED_NAME [rows]
Every scratch_load / scratch_store is attributed to the last .loc before it and to the number of loops (backward branches)
that enclose it; v_readlane / v_writelane (spilled SGPRs) are counted per depth."""
import collections, re, sys

path, sub = sys.argv[1], sys.argv[2]
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 40
text = open(path).read()
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', text):
    files[m.group(1)] = m.group(3)
for f in re.split(r"\n(?=_Z\w+:)", text):
    name = f.split(":", 1)[0]
    if sub not in name:
        continue
    lines = f.split("\n")
    labels, loops = {}, []
    for i, l in enumerate(lines):
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = i
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i: loops.append((labels[t], i))
    # merge loops with the same header
    hdr = {}
    for a, b in loops: hdr[a] = max(hdr.get(a, 0), b)
    loops = sorted(hdr.items())
    def depth(i): return sum(1 for a, b in loops if a <= i <= b)
    cur = ("?", 0)
    ld, st = collections.Counter(), collections.Counter()
    lane = collections.Counter()
    for i, l in enumerate(lines):
        m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
        if m: cur = (files.get(m.group(1), m.group(1)), int(m.group(2)))
        elif re.match(r"\s+scratch_load", l): ld[cur + (depth(i),)] += 1
        elif re.match(r"\s+scratch_store", l): st[cur + (depth(i),)] += 1
        elif re.search(r"v_readlane|v_writelane", l): lane[depth(i)] += 1
    print(name, "scratch loads", sum(ld.values()), "stores", sum(st.values()), "| lane ops by loop depth", dict(sorted(lane.items())))
    bydepth = collections.Counter()
    for k, v in ld.items(): bydepth[("ld", k[2])] += v
    for k, v in st.items(): bydepth[("st", k[2])] += v
    print("   by loop depth:", " ".join(f"{a}@{d}:{v}" for (a, d), v in sorted(bydepth.items(), key=lambda x: (x[0][1], x[0][0]))))
    keys = sorted(set(ld) | set(st), key=lambda k: (-k[2], -(ld[k] + st[k])))
    for k in keys[:rows]:
        print(f"   depth {k[2]}  {k[0]}:{k[1]:5d}  loads {ld[k]:3d}  stores {st[k]:3d}")
