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


# ---------------------------------------------------------------------------------------------------------------------------
# The accumulation-chain rule (DESIGN.md 3.2b', measured on gfx950 with hipcc 7.2; tools/repro_filter_block64.sh):
#   when the first MFMA of a NEW accumulation chain B is issued after the last MFMA of a finished chain A whose accumulator
#   registers are read (v_accvgpr_read / VALU / store) only LATER, beside B's MFMAs, A's last-written register can come back
#   wrong unless the wave idles >= 8 wait states (s_nop) between the two chains -- or reads A completely before B starts.
# chain_gap_scan() finds every such (A_last, B_first, late read) triple in a listing and reports the explicit idle wait states
# (sum of s_nop n + 1) between A_last and B_first; tests/test_host_logic.py fails on any triple below CHAIN_GAP_MIN.
CHAIN_GAP_MIN = 8


def _kernels(path):
    """[(name, [(opcode, [operands], line_no)])] per global function of the listing, instructions in program order"""
    out, cur, name = [], None, None
    for no, ln in enumerate(open(path), 1):
        s = ln.split(";")[0].rstrip()
        m = re.match(r"^([A-Za-z_$][\w.$]*):", s)
        if m and not m.group(1).startswith((".L", "BB", "$")):
            name, cur = m.group(1), []
            out.append((name, cur))
            continue
        if cur is None or not s.strip() or s.strip().startswith(".") or re.match(r"^\S+:", s.strip()):
            continue
        t = s.strip().split(None, 1)
        cur.append((t[0], [o.strip() for o in t[1].split(",")] if len(t) > 1 else [], no))
    return [(n, c) for n, c in out if any(i[0].startswith("v_mfma") for i in c)]


_STORES = ("ds_write", "ds_store", "global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "ds_add", "ds_max", "ds_min")


def chain_gap_scan(path, want=""):
    """-> {"kernels": n, "mfma": n, "switches": [(kernel, line A_last, line B_first, line first late read, idle wait states)]}"""
    res = {"kernels": 0, "mfma": 0, "switches": []}
    for name, ins in _kernels(path):
        if want not in name:
            continue
        res["kernels"] += 1
        n = len(ins)
        # per instruction: registers written / read (MFMA: dst, srcC separately)
        wr, rd, mf = [set()] * n, [set()] * n, [None] * n
        for i, (op, ops, _) in enumerate(ins):
            if op.startswith("v_mfma"):
                res["mfma"] += 1
                dst = frozenset(regs(ops[0]))
                srcc = frozenset(regs(ops[3])) if len(ops) > 3 else frozenset()
                mf[i] = (dst, srcc)
                wr[i], rd[i] = dst, set().union(*[regs(x) for x in ops[1:]])
            elif op.startswith(_STORES):
                rd[i] = set().union(*[regs(x) for x in ops]) if ops else set()
            elif op.startswith(("s_", "buffer_wbl2", "buffer_inv")):
                pass
            elif ops:
                wr[i] = regs(ops[0])
                rd[i] = set().union(*[regs(x) for x in ops[1:]]) if len(ops) > 1 else set()
        last = {}            # dst tuple -> index of the last MFMA that wrote it (the open end of its chain)
        fresh = set()        # tuples (of `last`) written by a non-MFMA instruction since: their next MFMA starts a new chain
        for p in range(n):
            if mf[p] is None:
                if wr[p]:
                    for t in list(last):
                        if t & wr[p]:
                            fresh.add(t)
                continue
            dst, srcc = mf[p]
            start = dst not in last or dst in fresh or srcc != dst
            if start:
                for a, e in list(last.items()):
                    if a == dst or a & dst or a in fresh:
                        continue
                    # is A read after p before any of its registers is written again?
                    late, pending = None, set(a)
                    for r in range(p + 1, min(n, p + 1500)):
                        if mf[r] is not None and mf[r][0] == a:
                            break                               # A's chain goes on (interleaved chains): not finished at p
                        hit = rd[r] & pending if mf[r] is None else (rd[r] & pending) - set()
                        if hit and not (mf[r] is not None and mf[r][1] == a and mf[r][0] == a):
                            late = r
                            break
                        pending -= wr[r]
                        if not pending:
                            break
                    if late is None:
                        continue
                    idle = sum(int(ins[j][1][0], 0) + 1 for j in range(e + 1, p) if ins[j][0] == "s_nop")
                    res["switches"].append((name, ins[e][2], ins[p][2], ins[late][2], idle))
            last[dst] = p
            fresh.discard(dst)
            for t in list(last):                                # a tuple overwritten (partly) by this MFMA under another shape
                if t != dst and t & dst:
                    del last[t]
                    fresh.discard(t)
    return res


def chain_gap_report(path, want=""):
    r = chain_gap_scan(path, want)
    bad = [s for s in r["switches"] if s[4] < CHAIN_GAP_MIN]
    print(f"{path}: {r['kernels']} kernels, {r['mfma']} MFMAs, {len(r['switches'])} chain switches with late reads of the finished "
          f"accumulator, {len(bad)} of them with fewer than {CHAIN_GAP_MIN} idle wait states")
    for k, a, b, c, idle in bad[:12]:
        print(f"   {k[:60]}: last MFMA of the finished chain at line {a}, first MFMA of the new chain at line {b}, late read at line {c}, idle {idle}")
    return r, bad



def asm_lds_reads_are_waited_for(path, want=""):
    """knn.hip (fp6 kernel): the 8-byte half of a fragment is read by an asm ds_read_b64 hipcc knows nothing about.  -> (reads found,
    reads whose destination registers are touched before any s_waitcnt lgkmcnt behind them)"""
    n = bad = 0
    for name, ins in _kernels(path):
        if want not in name:
            continue
        for i, (op, ops, no) in enumerate(ins):
            if op != "ds_read_b64" or not ops:
                continue
            dst = regs(ops[0])
            n += 1
            waited = False
            for j in range(i + 1, min(len(ins), i + 600)):
                o, oo, _ = ins[j]
                if o == "s_waitcnt" and any("lgkmcnt" in x for x in oo):
                    waited = True
                rd = set().union(*[regs(x) for x in (oo if o.startswith(_STORES) else oo[1:])]) if oo else set()
                wr = regs(oo[0]) if oo and not o.startswith(("s_",) + _STORES) else set()
                if rd & dst:
                    bad += 0 if waited else 1
                    break
                if wr & dst:
                    break
    return n, bad
