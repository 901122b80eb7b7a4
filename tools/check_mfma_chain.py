#!/usr/bin/env python
"""Audit of the compiler-scheduled MFMAs in the gfx950 assembly of csrc/*.hip for the pattern that produced wrong results in the first build of
csrc/groupnorm.hip::convout_bwd_kernel: an accumulate chain through TWO DIFFERENT MFMA opcodes issued back to back --
    v_mfma_f32_16x16x32_bf16 D, A, B, 0        (8 passes)
    v_mfma_f32_16x16x16_bf16 D, A2, B2, D      (reads D as its C operand)
hipcc (ROCm 7.2) pads no wait states between the two and the hardware does not interlock the pair: the second reads C before the first has written it
(measured: the first product is lost; with the two opcodes equal the back-to-back chain is the supported one).  Flagged: a v_mfma whose C operand overlaps
the destination of one of the previous 8 MFMAs of a DIFFERENT opcode with fewer than 11 wait states between them (an 8-pass producer + 3; s_nop N = N + 1,
an intervening MFMA = 4, the shortest one's passes, any other instruction = 1).  The -S output carries inline assembly expanded, so the hand-ordered block of
convout_bwd_kernel (four 16x16x32 products, then their four 16x16x16 accumulations: 3 MFMAs = 12 states between every pair) is audited like compiler output:
reordering or shrinking that block below the distance fails here before it fails on the GPU.  (A destination on top of the A / B operand, which the same build also had, is legal and common: vit.hip, linear_rows.hip.)
usage: python tools/check_mfma_chain.py [files...]   (exit 1 on a finding)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "dmvae_amd/csrc", "*.hip")))


def rng(tok):
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return (m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1)))
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return (m.group(1), {int(m.group(2))})
    return None


bad = 0
for f in files:
    out = tempfile.mktemp(suffix=".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(f),
                    "-Wno-unused-value", "-Wno-c++20-extensions", "-S", "--cuda-device-only", f, "-o", out], check=True, stderr=subprocess.DEVNULL)
    ins = [l.strip() for l in open(out) if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
    os.unlink(out)
    n = 0
    recent = []          # (index, opcode, destination) of the last MFMAs
    for k, l in enumerate(ins):
        if not l.startswith("v_mfma"):
            continue
        n += 1
        op = l.split()[0]
        ops = [t.strip() for t in l.split(None, 1)[1].split(",")]
        d, c = rng(ops[0]), (rng(ops[3]) if len(ops) > 3 else None)
        if c:
            for (j, opj, dj) in recent[-8:]:
                if opj != op and dj[0] == c[0] and (dj[1] & c[1]):
                    states = sum(int(q.split()[1]) + 1 if q.startswith("s_nop") else 4 if q.startswith("v_mfma") else 1 for q in ins[j + 1:k])
                    if states < 11:
                        bad += 1
                        print(f"{os.path.basename(f)}: mixed accumulate chain, {states} wait states: {ins[j][:64]}  ->  {l[:80]}")
        if d:
            recent.append((k, op, d))
    print(f"{os.path.basename(f)}: {n} MFMAs audited")
sys.exit(1 if bad else 0)
