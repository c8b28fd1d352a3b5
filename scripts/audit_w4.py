"""Audit of the asm-owned accumulator file of scan_mfma_w4_kernel (run after every edit).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S -o /tmp/k.s vrod_amd/csrc/kernels_mfma.hip --cuda-device-only
    python scripts/audit_w4.py /tmp/k.s

The kernel names a[0:255] literally in its inline-asm MFMAs.  That is only sound when the
compiler itself keeps out of the AGPRs wherever the accumulators are live: no v_accvgpr_* and
no a[..] operand outside ;;#ASMSTART/;;#ASMEND from the first asm MFMA on (the accumulators are
dead in the prologue of a query block: its first MFMAs are the C = 0 form, so compiler
temporaries parked in AGPRs there are harmless -- hipcc does that for the L2 variant), no
scratch anywhere (spills go to AGPRs first), accum_offset <= 256.
"""
import re
import sys


def audit(path):
    text = open(path).read()
    bad = 0
    for m in re.finditer(r"^(_ZN4vrod19scan_mfma_w4_kernel\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        inasm = False
        n_out = n_mfma = n_pro = 0
        last_touch = first_mfma = -1
        labels, branches = {}, []
        for ln, line in enumerate(body.split("\n")):
            lab = re.match(r"^(\.LBB\w+):", line)
            if lab:
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
                    if first_mfma < 0:
                        first_mfma = ln
                continue
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
    for m in re.finditer(r"\.amdhsa_kernel (_ZN4vrod19scan_mfma_w4_kernel\w+)\n(.*?)\.end_amdhsa_kernel", text, re.S):
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
