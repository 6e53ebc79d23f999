#!/usr/bin/env python3
"""tools/check_asm.py [kernels.s] -- invariants of the compiled ISA that the kernels' design relies on (run by `make -C
sdrreceiver_amd/csrc asm`): no kernel uses scratch memory (a spill reload is a vector-memory operation behind an
s_waitcnt vmcnt -- in the chunk loops exactly the wait the code is arranged to avoid), and k_dc_chain -- whose scalar
prefetch hands `s_load` destinations from one asm statement to the next -- spills no SGPR (a v_writelane / v_readlane
between request and wait would read stale products and silently change the bit-exact DC estimate)."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "kernels.s"
name, bad = None, []
scratch = {}
lanes = {}
for line in open(path):
    m = re.match(r"^(_ZN4sdrx\w+):", line)
    if m:
        name = m.group(1)
    m = re.search(r";\s*ScratchSize:\s*(\d+)", line)
    if m and name:
        scratch[name] = int(m.group(1))
    if name and ("v_writelane_b32" in line or "v_readlane_b32" in line):
        lanes[name] = lanes.get(name, 0) + 1
for k, v in scratch.items():
    if v:
        bad.append(f"{k}: {v} bytes of scratch")
for k, v in lanes.items():
    if "k_dc_chainE" in k:
        bad.append(f"{k}: {v} v_writelane / v_readlane (an SGPR spill next to the scalar prefetch)")
if bad:
    sys.exit("ISA check failed:\n  " + "\n  ".join(bad))
print(f"ISA check ok: {len(scratch)} kernels, 0 scratch; k_dc_chain keeps its scalar operands in SGPRs")
