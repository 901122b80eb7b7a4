#!/usr/bin/env python
"""Wait / barrier / DMA skeleton of every MFMA loop in a hipcc -S listing.  For each kernel, each innermost loop (the compiler's "Inner Loop Header" /
"in Loop: Header=" block annotations) that holds v_mfma instructions is printed block by block as its sequence of s_waitcnt / s_barrier /
buffer_load..lds / ds_read / v_mfma counts.  An `s_waitcnt vmcnt(0)` inside a K loop that is meant to keep a counted LDS-DMA prefetch queue in flight is a
compiler-inserted drain (its wait-count pass orders every ds_read behind every earlier LDS-DMA it cannot prove disjoint).
usage: python tools/loop_waits.py file.s [kernel-name filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\S+):.*\n', txt, re.M):
    name = m.group(1)
    if flt not in name: continue
    end = txt.find('.Lfunc_end', m.end())
    if end < 0: continue
    lines = txt[m.end():end].split('\n')
    # split into blocks
    blocks, cur = [], ['entry', '', []]
    for l in lines:
        h = re.match(r'^(\.LBB\d+_\d+):(.*)', l) or re.match(r'^; %bb\.(\d+):(.*)', l)
        if h:
            blocks.append(cur); cur = [h.group(1), h.group(2), []]
        elif l.strip().startswith(';') and not cur[2]:
            cur[1] += ' ' + l.strip()      # a nested loop's header annotation continues on comment lines
        else:
            cur[2].append(l.strip())
    blocks.append(cur)
    headers = [b[0] for b in blocks if 'Inner Loop Header' in b[1]]
    for hd in headers:
        key = 'Header=' + hd.lstrip('.L')
        loop = [b for b in blocks if b[0] == hd or key in b[1]]
        nm = sum('v_mfma' in x for b in loop for x in b[2])
        if nm < 4: continue
        print(name[:110], f'loop {hd}: {nm} mfma, {sum(len(b[2]) for b in loop)} lines in {len(loop)} blocks')
        for b in loop:
            seq = []
            for x in b[2]:
                k = None
                if x.startswith('s_waitcnt'): k = x.replace('s_waitcnt ', 'wait ')
                elif x.startswith('s_barrier'): k = 'BARRIER'
                elif x.startswith('buffer_load') and ' lds' in x: k = 'dma'
                elif x.startswith('ds_read'): k = 'ds_read'
                elif x.startswith('v_mfma'): k = 'mfma'
                elif x.startswith('v_readfirstlane'): k = 'readfirstlane'
                elif x.startswith('s_cbranch'): k = x.replace('s_cbranch_', 'br_')
                if k is None: continue
                if seq and seq[-1][0] == k and k in ('dma', 'ds_read', 'mfma'): seq[-1][1] += 1
                else: seq.append([k, 1])
            if seq: print(f'    {b[0]:>10}:', ' | '.join(k if n == 1 else f'{k} x{n}' for k, n in seq))
