"""Static scan of a gfx950 assembly listing for software-visible MFMA hazards (the wait states the compiler has to provide
with independent instructions or s_nop: CDNA3/4 ISA guide, "Manually inserted wait states"; LLVM GCNHazardRecognizer,
gfx940 tables).  Written to pin down why filter_block64_kernel differs from run to run when the SLP vectorizer is on
(DESIGN.md 3.2b'):

  hipcc -O3 ... -S --cuda-device-only filter_mid.hip -o x.s ;  python tools/mfma_hazard_scan.py x.s [kernel-substring]

For every v_mfma it follows the straight-line code behind it (stops at labels / branches / barriers) and records, per
hazard class, the smallest number of wait states (instructions issued in between, `s_nop n` = n + 1) found anywhere:
  RAW-valu   MFMA result -> read by a VALU / LDS / VMEM instruction        needs passes + 3   (16x16x32 bf16: 8 passes -> 11)
  WAW-valu   MFMA result -> overwritten by a VALU instruction              needs passes + 3
  WAR-srcC   MFMA SrcC   -> overwritten by a VALU instruction              needs passes + 1 (16x16: 9?)  [reported, not judged]
  RAW-load   VGPR written by an in-flight LDS/VMEM load -> MFMA reads it with no s_waitcnt in between   [always a bug]
Prints every pair below the required distance."""
import re
import sys

PASSES = {"16x16x32": 8, "32x32x16": 16, "16x16x16": 8, "32x32x8": 16, "16x16x4": 8, "32x32x2": 16, "4x4x4": 2, "32x32x64": 16,
          "16x16x128": 8}


def regs(tok):
    """'v[12:15]' / 'v7' / 'a[0:3]' -> set of ('v'|'a', n)"""
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b", tok):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def parse(path, want):
    ins, on = [], False
    for ln in open(path):
        s = ln.split(";")[0].rstrip()
        if re.match(r"^[A-Za-z_.$][\w.$]*:", s):
            name = s.split(":")[0]
            if not name.startswith(".L") and not name.startswith("BB"):
                on = want in name
            if on:
                ins.append(("label", name, [], ln))
            continue
        if not on or not s.strip() or s.strip().startswith("."):
            continue
        t = s.strip().split(None, 1)
        ops = [o.strip() for o in t[1].split(",")] if len(t) > 1 else []
        ins.append((t[0], ops, ln))
    return ins


def main():
    path, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "filter_block64")
    ins = parse(path, want)
    found = {}
    n_mfma = 0
    for i, it in enumerate(ins):
        if it[0] == "label" or not it[0].startswith("v_mfma"):
            continue
        n_mfma += 1
        op, ops = it[0], it[1]
        shape = re.search(r"(\d+x\d+x\d+)", op).group(1)
        need = PASSES.get(shape, 8) + 3
        dst, srcc = regs(ops[0]), regs(ops[3]) if len(ops) > 3 else set()
        ws = 0
        for j in range(i + 1, min(i + 40, len(ins))):
            o = ins[j]
            if o[0] == "label" or o[0].startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_setpc")):
                break
            if o[0] == "s_nop":
                ws += int(o[1][0], 0) + 1
                continue
            if ws >= need + 8:
                break
            if o[0].startswith("v_mfma"):                 # MFMA -> MFMA dependencies are interlocked in hardware (same-shape SrcC) -- skip
                ws += 1
                continue
            if o[0].startswith(("s_", "v_accvgpr")) and not o[0].startswith("v_accvgpr"):
                ws += 1
                continue
            is_store = o[0].startswith(("ds_write", "ds_store", "global_store", "buffer_store", "flat_store"))
            is_load = o[0].startswith(("ds_read", "ds_load", "global_load", "buffer_load", "flat_load"))
            w = set() if is_store else (regs(o[1][0]) if o[1] else set())
            r = set().union(*[regs(x) for x in (o[1] if is_store else o[1][1:])]) if o[1] else set()
            kind = None
            if r & dst:
                kind = "RAW-valu" if not (is_store or is_load) else "RAW-mem"
            elif w & dst and not is_load:
                kind = "WAW-valu"
            elif w & srcc and not is_load and not (srcc & dst):
                kind = "WAR-srcC"
            if kind:
                key = (kind, shape)
                if key not in found or ws < found[key][0]:
                    found[key] = (ws, it[2].strip(), o[2].strip())
                if kind in ("RAW-valu", "RAW-mem", "WAW-valu") and ws < need:
                    print(f"!! {kind}: {ws} wait states < {need}\n     {it[2].strip()}\n     {o[2].strip()}")
            ws += 1
    print(f"{n_mfma} MFMAs scanned in kernels matching '{want}'")
    for (kind, shape), (ws, a, b) in sorted(found.items()):
        print(f"min distance {kind:9s} {shape}: {ws} wait states\n     {a}\n     {b}")


if __name__ == "__main__":
    main()
