#!/usr/bin/env python
"""Audit of the inline-asm MFMAs (hipcc pads no hazards inside / around asm statements): for every v_mfma in the gfx950
assembly of the given .hip files, flag a VALU instruction within the 3 preceding instructions that writes one of its VGPR
operands unless an s_nop sits between them.  usage: python tools/check_asm_hazards.py [files...]  (exit 1 on a finding)"""
import re, subprocess, sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or [os.path.join(ROOT, "dmvae_amd/csrc", f) for f in ("conv_pp.hip", "conv_wgrad_pp.hip")]
bad = 0
for f in files:
    out = tempfile.mktemp(suffix=".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.dirname(f), "-Wno-unused-value", "-S", "--cuda-device-only", f, "-o", out], check=True,
                   stderr=subprocess.DEVNULL)
    ins = [l.strip() for l in open(out) if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
    n = 0
    for k, l in enumerate(ins):
        if not l.startswith("v_mfma"):
            continue
        n += 1
        regs = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", l):
            regs |= set(range(int(a), int(b) + 1))
        window = ins[max(0, k - 3):k]
        for j, p in enumerate(window):
            if p.startswith("v_") and not p.startswith("v_mfma"):
                m = re.match(r"v_\w+\s+v(\d+)|v_\w+\s+v\[(\d+):(\d+)\]", p)
                if m:
                    w = {int(m.group(1))} if m.group(1) else set(range(int(m.group(2)), int(m.group(3)) + 1))
                    if (w & regs) and not any(q.startswith("s_nop") for q in window[j + 1:]):
                        bad += 1
                        print(f"{os.path.basename(f)}: {p}  ->  {l[:80]}")
    print(f"{os.path.basename(f)}: {n} MFMAs audited")
sys.exit(1 if bad else 0)
