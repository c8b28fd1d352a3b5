"""Audit of the asm-owned registers of scan_mfma_w4_kernel (run after every edit).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S -o /tmp/k.s vrod_amd/csrc/kernels_mfma_w4.hip --cuda-device-only
    python scripts/audit_w4.py /tmp/k.s

The kernel names a[0:255] literally in its inline-asm MFMAs.  That is only sound when the
compiler itself keeps out of the AGPRs wherever the accumulators are live: no v_accvgpr_* and
no a[..] operand outside ;;#ASMSTART/;;#ASMEND from the first asm MFMA on (the accumulators are
dead in the prologue of a query block: its first MFMAs are the C = 0 form, so compiler
temporaries parked in AGPRs there are harmless -- hipcc does that for the L2 variant), no
scratch anywhere (spills go to AGPRs first), accum_offset <= 256; the compiler never names m0 (the
asm LDS-DMA pieces own it) and emits no vmcnt wait of its own inside a block of the scan loop.

scan_mfma_w4a_kernel also loads its A fragments with inline-asm global_load_dwordx4 (uncounted by
hipcc): between such a load and the counted wait that names its registers the data is in flight, so
the compiler must never copy, move or overwrite a register that is the destination of an asm load
-- checked here: no instruction outside ;;#ASMSTART/;;#ASMEND names one of those registers.
"""
import re
import sys


def _vregs(tok):
    """registers named by one operand token: v12 -> {12}, v[8:11] -> {8..11}"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def audit_asm_loaded_registers(name, body):
    """Between the first asm load and the last asm MFMA (the main loop and its tile epilogues) compiler code
    must not WRITE a VGPR that is the destination of an asm load, and must not READ one that feeds an MFMA
    (the A ring: its data is in flight between the load and the counted wait; only asm touches it).  Row
    norms loaded the same way (L2) are read by compiler code behind their wait: reads of those are fine."""
    lines = body.split("\n")
    loaded, ring = set(), set()
    first = last = -1
    inasm = False
    for ln, line in enumerate(lines):
        if "ASMSTART" in line:
            inasm = True
            continue
        if "ASMEND" in line:
            inasm = False
            continue
        code = line.split(";")[0]
        if not inasm:
            continue
        if re.search(r"\bglobal_load_dwordx4\b", code):
            loaded |= _vregs(code.split()[1].rstrip(","))
            if first < 0:
                first = ln
        if "v_mfma" in code:
            last = ln
    if not loaded:
        return 0
    for line in lines:   # MFMA operands that come from asm loads
        code = line.split(";")[0]
        if "v_mfma" in code:
            ops = [t.strip() for t in code.split(None, 1)[1].split(",")]
            ring |= (_vregs(ops[1]) | _vregs(ops[2])) & loaded
    n_bad = 0
    inasm = False
    for ln, line in enumerate(lines):
        if "ASMSTART" in line:
            inasm = True
            continue
        if "ASMEND" in line:
            inasm = False
            continue
        if inasm or ln < first or ln > last:
            continue
        code = line.split(";")[0].strip()
        if not code or code.endswith(":") or code.startswith("."):
            continue
        toks = re.findall(r"\bv\[\d+:\d+\]|\bv\d+\b", code)
        if not toks:
            continue
        is_store = re.match(r"(global_store|ds_write|buffer_store|global_atomic|ds_add|ds_max|ds_min)", code) is not None
        written = set() if is_store else _vregs(toks[0])
        read = set()
        for t in (toks if is_store else toks[1:]):
            read |= _vregs(t)
        if written & loaded or read & ring:
            n_bad += 1
            print(f"{name}: compiler code touches asm-loaded registers: {code}")
    print(f"{name}: {len(loaded)} VGPRs are destinations of asm loads ({len(ring)} feed MFMAs), {n_bad} compiler instructions touch them in the loop")
    return n_bad


def audit(path):
    text = open(path).read()
    bad = 0
    for m in re.finditer(r"^(_ZN4vrod\d+scan_mfma_w4_kernel\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        bad += audit_asm_loaded_registers(name, body)
        inasm = False
        n_out = n_mfma = n_pro = 0
        last_touch = first_mfma = -1
        labels, branches = {}, []
        blk_mfma, blk_waits = 0, []   # the basic block being read: its asm MFMAs, its compiler-emitted vmcnt waits
        for ln, line in enumerate(body.split("\n") + [".LBB_end:"]):
            lab = re.match(r"^(\.LBB\w+):", line)
            if lab:
                # The scan loop keeps 16-32 LDS-DMA pieces in flight that hipcc's counter bookkeeping cannot see (asm); its own
                # waits are asm too.  A vmcnt wait the COMPILER puts into a block of the loop (it does when compiler-visible
                # loads / stores are left pending at the loop's back edge) drains that queue every K-tile: +16 %, same results.
                # (Not asked of the every-score sample form, <METRIC, 1, SPLIT>: one tile per work-group, 64 K compiler-visible
                #  score stores behind it -- there is no queue to keep.)
                if blk_mfma >= 32 and not re.search(r"kernelILi\dELi1ELb", name):
                    for w in blk_waits:
                        n_out += 1
                        print(f"{name}: compiler-emitted vmcnt wait inside a block of {blk_mfma} MFMAs: {w}")
                blk_mfma, blk_waits = 0, []
                labels[lab.group(1)] = ln
            br = re.search(r"\bs_c?branch\w*\s+(\.LBB\w+)", line.split(";")[0])
            if br:
                branches.append((ln, br.group(1)))
            if "ASMSTART" in line:
                inasm = True
                continue
            if "ASMEND" in line:
                inasm = False
                continue
            code = line.split(";")[0]
            if inasm:
                if "v_mfma" in code:
                    n_mfma += 1
                    blk_mfma += 1
                    if first_mfma < 0:
                        first_mfma = ln
                continue
            if re.search(r"\bs_waitcnt\b.*\bvmcnt\(", code):
                blk_waits.append(line.strip())
            if re.search(r"\bm0\b", code):   # M0 is set by the asm LDS-DMA pieces only: the compiler must have no use of its own for it
                n_out += 1
                print(f"{name}: compiler names m0: {line.strip()}")
            if "scratch_" in code or ((("v_accvgpr" in code) or re.search(r"\ba\[?\d+", code)) and n_mfma > 0):
                n_out += 1
                print(f"{name}: compiler touches an AGPR / scratch: {line.strip()}")
            elif "v_accvgpr" in code or re.search(r"\ba\[?\d+", code):
                n_pro += 1
                last_touch = ln
        # the parked values must not be re-read after the accumulators came alive: no branch from
        # the MFMA region back to (or before) the last compiler AGPR use
        for ln, target in branches:
            if last_touch >= 0 and ln > first_mfma >= 0 and labels.get(target, 1 << 30) <= last_touch:
                n_out += 1
                print(f"{name}: back-edge from line {ln} to {target} (line {labels[target]}) re-enters the region where the compiler uses AGPRs")
        print(f"{name}: {n_mfma} asm MFMAs, {n_pro} compiler AGPR uses in the prologue (accumulators dead), {n_out} violations")
        bad += n_out + (n_mfma == 0)
    for m in re.finditer(r"\.amdhsa_kernel (_ZN4vrod\d+scan_mfma_w4_kernel\w+)\n(.*?)\.end_amdhsa_kernel", text, re.S):
        name, desc = m.group(1), m.group(2)
        acc = int(re.search(r"\.amdhsa_accum_offset (\d+)", desc).group(1))
        nv = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1))
        priv = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc).group(1))
        ok = acc <= 256 and nv >= acc + 256 and nv <= 512 and priv == 0
        print(f"{name}: accum_offset {acc}, next_free_vgpr {nv}, private {priv} -> {'ok' if ok else 'BAD'}")
        bad += not ok
    return bad


if __name__ == "__main__":
    sys.exit(1 if audit(sys.argv[1]) else 0)
