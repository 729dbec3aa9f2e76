"""Finds a code-generation hazard of hipcc (ROCm 7.2, clang 22) in gfx950 assembly: register-allocator spill code placed at the head
of a control-flow join block BEFORE the instruction that re-enables the lanes which skipped the preceding divergent region.

    .LBB8_459:                                        ; %Flow3060
        scratch_store_dwordx2 off, v[120:121], off offset:320   ; 8-byte Folded Spill     <- only the lanes still enabled store
        s_or_b64 exec, exec, s[8:9]                                                        <- the others come back here
        ...
    .LBB8_798:
        scratch_load_dwordx2 v[120:121], off, off offset:320    ; 8-byte Folded Reload    <- every lane loads: the lanes that did not
                                                                                              store get what the scratch slot held before
That is the cause of the photon-walk nondeterminism recorded in round 1 / 2 (profiles/round03/photon_walk_root_cause.txt): the
lanes' good register contents (a loop-invariant 1.0f) were replaced by stale scratch memory.  A reload before the mask is widened is
harmless (masked lanes keep their register lanes); a STORE there is the hazard when the value is live in the masked lanes.

   hipcc ... --save-temps --cuda-device-only -c file.hip ;  python tools/isa_exec_spill_check.py FILE.s [--all]
Prints, per kernel / device function, the spill stores that precede an `s_or_b64 exec, exec, ...` in their basic block with nothing
but spill code, waits and nops before them; exit status 1 when any kernel listed in --fail-on (substring of the mangled name) has one."""
import re, sys

path = sys.argv[1]
show_all = "--all" in sys.argv
text = open(path).read()
total = 0
for f in re.split(r"\n(?=_Z\w+:)", text):
    name = f.split(":", 1)[0]
    if not name.startswith("_Z"):
        continue
    lines = f.split("\n")
    sites = []
    i = 0
    while i < len(lines):
        if re.match(r"\.LBB\d+_\d+:", lines[i]):
            j = i + 1
            stores = []
            while j < len(lines):
                l = lines[j].strip()
                if not l or l.startswith(";") or l.startswith(".loc") or l.startswith(".Ltmp") or l.startswith(".cfi"):
                    j += 1; continue
                if re.match(r"scratch_store_\w+ .*Folded Spill", l):
                    stores.append((j, l)); j += 1; continue
                if re.match(r"(scratch_load_\w+ .*Folded Reload|s_waitcnt|s_nop|v_writelane_b32|v_readlane_b32)", l):
                    j += 1; continue
                if re.match(r"s_or_b64 exec, exec,", l) and stores:
                    sites.append((lines[i].split(":")[0], stores))
                break
            i = j
        else:
            i += 1
    if sites or show_all:
        n = sum(len(s) for _, s in sites)
        total += n
        print(f"{name[:110]}: {n} spill store(s) ahead of the mask restore in {len(sites)} block(s)")
        for lab, st in sites[:6]:
            for ln, l in st:
                print(f"      {lab}: {l[:100]}")
print("total hazardous spill stores:", total)
