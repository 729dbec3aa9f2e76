"""Register / spill / scratch / LDS use of every gfx950 kernel in built objects (no GPU needed).
   python tools/kernel_resources.py [qaray_amd/lib/obj/qa_capi.o ...] [--grep qa_integrate_cs]
Unbundles the device code object from each host object and reads the AMDGPU metadata note."""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True)
    return r.stdout.splitlines()


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as td:
        co = os.path.join(td, "dev.co")
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], stderr=subprocess.PIPE)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], stderr=subprocess.PIPE, text=True)
        if r.returncode != 0 or not os.path.exists(co):
            co = obj   # already a device code object
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], stdout=subprocess.PIPE, text=True).stdout
    out = []
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        f = {}
        for key in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                    "group_segment_fixed_size", "max_flat_workgroup_size"):
            m = re.search(r"\." + key + r":\s+(\d+)", blk)
            f[key] = int(m.group(1)) if m else -1
        m = re.search(r"\.name:\s+(\S+)", blk)
        f["name"] = m.group(1) if m else "?"
        out.append(f)
    return out


def main():
    args = sys.argv[1:]
    pat = None
    if "--grep" in args:
        i = args.index("--grep")
        pat = args[i + 1]
        del args[i:i + 2]
    objs = args or [os.path.join(ROOT, "qaray_amd", "lib", "obj", f) for f in sorted(os.listdir(os.path.join(ROOT, "qaray_amd", "lib", "obj")))
                    if f.endswith(".o") and "-hip-" not in f and "-host-" not in f]   # (the objects, not the compiler temporaries next to them)
    for obj in objs:
        ks = kernels_of(obj)
        names = demangle([k["name"] for k in ks])
        print(f"== {os.path.relpath(obj, ROOT)}")
        print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'sspill':>6} {'scratchB':>8} {'ldsB':>6}  kernel")
        for k, n in sorted(zip(ks, names), key=lambda kn: kn[1]):
            n = re.sub(r"^void ", "", n)
            n = re.sub(r"\(.*$", "", n)
            if pat and pat not in n:
                continue
            print(f"{k['vgpr_count']:5d} {k['agpr_count']:5d} {k['sgpr_count']:5d} {k['vgpr_spill_count']:6d} {k['sgpr_spill_count']:6d} "
                  f"{k['private_segment_fixed_size']:8d} {k['group_segment_fixed_size']:6d}  {n}")


if __name__ == "__main__":
    main()
