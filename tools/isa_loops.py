"""Loop-level statistics of a kernel's gfx950 assembly (no GPU needed): where the spills, scalar loads and waits sit.
   hipcc ... --save-temps -c qa_capi.hip     (writes qa_capi-hip-amdgcn-amd-amdhsa-gfx950.s)
   python tools/isa_loops.py FILE.s SUBSTRING_OF_MANGLED_NAME [min_len]
e.g. python tools/isa_loops.py /tmp/isa/qa_capi-hip-amdgcn-amd-amdhsa-gfx950.s integrate_csILb1ELb0E"""
import re
import sys

PATS = [("valu", r"^\s+v_"), ("salu", r"^\s+s_(?!waitcnt|nop|cbranch|branch|load|buffer_load)"), ("smem", r"^\s+s_(buffer_)?load"), ("ds", r"^\s+ds_"),
        ("gload", r"^\s+global_load"), ("gstore", r"^\s+global_store"), ("scr_ld", r"^\s+scratch_load"), ("scr_st", r"^\s+scratch_store"),
        ("waitcnt", r"^\s+s_waitcnt"), ("lane", r"v_readlane|v_writelane"), ("call", r"s_swappc"), ("rcp", r"v_rcp_f32|v_rsq|v_sqrt"),
        ("div", r"v_div_scale"), ("cvtub", r"v_cvt_f32_ubyte"), ("dsmin", r"ds_min")]


def main():
    path, sub = sys.argv[1], sys.argv[2]
    min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    text = open(path).read()
    funcs = re.split(r"\n(?=_Z\w+:)", text)
    for f in funcs:
        name = f.split(":", 1)[0]
        if sub not in name:
            continue
        lines = f.split("\n")
        print(name, len(lines), "lines")
        labels = {}
        for i, l in enumerate(lines):
            m = re.match(r"(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = i
        hdr = {}
        for i, l in enumerate(lines):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
            if m:
                t = m.group(1) or m.group(2)
                if t in labels and labels[t] < i:
                    hdr[labels[t]] = max(hdr.get(labels[t], 0), i)
        total = {k: sum(1 for l in lines if re.search(p, l)) for k, p in PATS}
        print("whole kernel:", " ".join(f"{k} {v}" for k, v in total.items()))
        for a, b in sorted(hdr.items()):
            if b - a < min_len:
                continue
            body = lines[a:b + 1]
            depth = sum(1 for x, y in hdr.items() if x < a and y > b)
            st = {k: sum(1 for l in body if re.search(p, l)) for k, p in PATS}
            print(f"{'  ' * depth}loop {a}-{b} ({b - a} lines): " + " ".join(f"{k} {v}" for k, v in st.items() if v))


if __name__ == "__main__":
    main()
