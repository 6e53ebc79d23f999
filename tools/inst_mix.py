#!/usr/bin/env python3
"""tools/inst_mix.py [kernels.s] -- static instruction mix of the mix/decimate item in the compiled ISA
(`make -C sdrreceiver_amd/csrc asm` writes kernels.s), next to what the source demands:

  per 1024-sample chunk of a d = 5 VFO, one wave (a packed v_pk_*_f32 on a (re, im) pair = ONE instruction)
    NCO replay  16 x (cmul: 2 pk_mul + 1 pk_fma; n*n: 1 pk_mul; x+y, 1.95-s: 2 scalar; n*norm: 1 pk_mul) = 112
    mix         16 x (cmul: 2 pk_mul + 1 pk_fma)                                                          =  48
    stage 0      8 x (3 pair sums + 4 products + 3 sums + (0 + s))  [4 pk_mul + 7 pk_add]                 =  88
    stage 1      4 x the same                                                                             =  44
    halos       16 complex values shifted one lane (wave_shr:1), 2 v_mov_b32_dpp each                     =  32
    LDS stages  the same 11-instruction dot product, run 2 + 1 + 1 times per lane: ONE rolled instance in the generic routine
                (hb_stage_lds: any depth, frame end, tiled output) + the FOUR unrolled instances of hb_stage_fixed (full chunk
                of a d = 5 leaf: the path 374 of 375 chunks take)

The register-resident part is straight-line code, so the STATIC counts of a mix/decimate body must be exactly
  any-VFO body:  v_pk_mul_f32 = 64 + 32 + 32 + 16 + 4 (generic stage) = 148,  v_pk_add_f32 = 56 + 28 + 7 = 91,  v_pk_fma_f32 = 16 + 16 = 32,  *_dpp = 32
  d = 5 body:    the same + the fixed stages' 11 dot products (44 pk_mul + 77 pk_add: stage 2 instantiated three times -- chunk
                 finished alone, first and second chunk of a pair -- with 2 dot products each, the single-chunk tail 1 + 1, the
                 two-chunk tail 2 + 1) = 192 + 4 / 168 + 7 (its frame-end chunks still take the generic stage) / 32 / 32
  d = 2 body:    64 + 32 + 32 + 16 = 144 / 84 / 32 / 32 (no LDS stage)
Since round 5 mix_item is compiled in those three bodies (kernels.hip, run_item), and the compiler unswitches the any-VFO body's
chunk loop on its loop-invariant output form (tile layout / natural order): it is in the code twice.
The same kernels also hold the two fused late decimations (late_item<5>, <6>: NCO + mix as above, and 3 x Nd products and
sums of the decimating low-pass in inline asm):
  /5:  v_pk_mul_f32 = 64 + 32 + 147 = 243, v_pk_add_f32 = 147, v_pk_fma_f32 = 32
  /6:  v_pk_mul_f32 = 64 + 32 + 219 = 315, v_pk_add_f32 = 219, v_pk_fma_f32 = 32
so the kernel totals are 2 x 148 + 196 + 144 + 243 + 315 = 1194 / 2 x 91 + 175 + 84 + 147 + 219 = 807 / 6 x 32 = 192 / 4 x 32 = 128,
which this script checks (exact arithmetic).  Per chunk of a d = 5 leaf a wave EXECUTES 148 / 91 / 32 / 32 of them (a chunk of a
pair: 146 / 87.5 on average -- the last stage runs once per two chunks)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "sdrreceiver_amd", "csrc", "kernels.s")
lines = open(path).read().splitlines()
want = {"v_pk_mul_f32": 1194, "v_pk_add_f32": 807, "v_pk_fma_f32": 192, "dpp": 128}
ok = True
for sym, label in (("_ZN4sdrx14k_mix_decimateILb1ELi1EEE", "k_mix_decimate<exact, level>=1>"), ("_ZN4sdrx12k_mix_levelsILb1EEE", "k_mix_levels<exact>")):
    start = next((i for i, l in enumerate(lines) if l.startswith(sym) and l.rstrip().endswith(":") or (l.startswith(sym) and ": " in l)), None)
    if start is None:
        print(f"{label}: not found in {path}")
        ok = False
        continue
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = [l.strip() for l in lines[start:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    mn = [l.split()[0] for l in body if l]
    count = {k: sum(1 for m in mn if m == k) for k in ("v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32")}
    count["dpp"] = sum(1 for l in body if re.search(r"\b(row_|wave_)\w+:\d", l))
    other = {"valu_total": sum(1 for m in mn if m.startswith("v_")), "ds": sum(1 for m in mn if m.startswith("ds_")),
             "global": sum(1 for m in mn if m.startswith("global_")), "salu+smem": sum(1 for m in mn if m.startswith("s_")),
             "scratch": sum(1 for m in mn if m.startswith("scratch_"))}
    good = all(count[k] == want[k] for k in want)
    ok &= good
    print(f"{label}: {count}  {'== source count' if good else '!= source count ' + str(want)}  (static totals: {other})")
sys.exit(0 if ok else 1)
