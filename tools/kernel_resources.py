"""Developer tool: VGPRs / scratch / occupancy of every k_sweep_q instance from hipcc's resource remarks.
usage: hipcc <Makefile flags> -fno-slp-vectorize -c csrc/xee_sweepq.hip -o /tmp/q.o -Rpass-analysis=kernel-resource-usage 2> res.txt
       python tools/kernel_resources.py res.txt"""
import re
import sys

recs, name = {}, None
for line in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1); recs[name] = {}
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, line)
        if m and name:
            recs[name][key] = int(m.group(1))
for k, v in sorted(recs.items()):
    m = re.search(r"k_sweep_q(_redo)?ILi(\d+)ELi(\d+)ELb([01])(?:ELb([01])ELb([01]))?", k)
    if m:
        redo, nw, r, b1, b2, b3 = m.groups()
        tag = f"redo NW={nw} R={r} baro={b1}" if redo else f"NW={nw} R={r} write_r={b1} baro={b2} alpha1={b3}"
        print(f"{tag:44s} {v}")
