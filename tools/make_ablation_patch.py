#!/usr/bin/env python3
"""tools/make_ablation_patch.py -- (re)generate tools/ablation.patch from the CURRENT sdrreceiver_amd/csrc/kernels.hip.

The phase ablations of mix_item (-DSDRX_ABL_LOAD / CP / NCO / MIX / CARRY / ST0 / ST1 / LDS / STORE / CONFLICT: each removes one
phase; WRONG results by design -- the phase-cost study of profiles/README.md) are not part of the product source: they live in
the patch this script writes, which tools/ab_build.sh applies to a scratch copy.  Run it again after editing mix_item or its helpers (load_run_tile and nco_mix are late_item's too: LOAD / NCO / MIX remove
those phases from both); it
fails loudly when an anchor no longer matches.  (The LDS-DMA prefetch of rounds 3-4, -DSDRX_GLDS, was measured slower twice
and is only in the history: `git show 6224116:sdrreceiver_amd/csrc/kernels.hip`.)"""
import difflib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sdrreceiver_amd", "csrc", "kernels.hip")
EDITS = [
    # (anchor, replacement) -- every anchor must occur exactly `n` times
    ("""    const int first_out = W.s_first_out;
    const int lane16 = (W.s_begin >> 4) + lane;""",
     """#ifndef SDRX_ABL_STORE
    const int first_out = W.s_first_out;
#else
    const int first_out = D.n_in == 12345 ? W.s_first_out : 0x7fffffff; // ablation: nothing is ever emitted
#endif
    const int lane16 = (W.s_begin >> 4) + lane;""", 1),
    ("""        const v4f v = gldv4(src + 64 * i); // tile_unit(c, i, l) = tile_unit(c, 0, l) + 64 i
""",
     """#ifndef SDRX_ABL_LOAD
        const v4f v = gldv4(src + 64 * i); // tile_unit(c, i, l) = tile_unit(c, 0, l) + 64 i
#else
        v4f v = {1.f, 2.f, 3.f, 4.f * i}; // ablation: no global loads
        asm volatile("" : "+v"(v));
#endif
""", 1),
    ("""            o_next = gldv2(cp_of(base + kChunk < W.s_end ? base + kChunk : base)); // (always issued -- and therefore counted, like the stores)
""",
     """#ifndef SDRX_ABL_CP
            o_next = gldv2(cp_of(base + kChunk < W.s_end ? base + kChunk : base)); // (always issued -- and therefore counted, like the stores)
#else
            o_next = v2f{0.9f + 1e-6f * base, 0.1f}; // ablation: no checkpoint load
            asm volatile("" : "+v"(o_next));
#endif
""", 1),
    ("""            o = nco_step_pk(o, rot);
            v2f m = o;
            if (i == 0 && first_ever)
                m = gldv2(last);
            x[i] = cmul(m, x[i]);
        }
    } else {
        nco_mix_fast16(o, rk, x);
    }
}""",
     """#ifndef SDRX_ABL_NCO
            o = nco_step_pk(o, rot);
#else
            asm volatile("" : "+v"(o)); // ablation: keep the value opaque, skip the recurrence
#endif
            v2f m = o;
            if (i == 0 && first_ever)
                m = gldv2(last);
#ifndef SDRX_ABL_MIX
            x[i] = cmul(m, x[i]);
#else
            x[i] = x[i] + m;
#endif
        }
    } else {
#if !defined(SDRX_ABL_NCO) && !defined(SDRX_ABL_MIX)
        nco_mix_fast16(o, rk, x);
#else
#pragma unroll
        for (int i = 0; i < kRun; ++i)
            x[i] = x[i] + o; // ablation: neither the rotations nor the mixer
#endif
    }
}""", 1),
    ("""        halo_stage0(car0, ext0, lane);
""",
     """#ifndef SDRX_ABL_CARRY
        halo_stage0(car0, ext0, lane);
#else
        for (int q = 0; q < 10; ++q) ext0[q] = x[q]; // ablation: no LDS carry, no DPP
#endif
""", 1),
    ("""        halo_stage1(car1, ext1, lane);
""",
     """#ifndef SDRX_ABL_CARRY
        halo_stage1(car1, ext1, lane);
#else
        for (int q = 0; q < 10; ++q) ext1[q] = y[q & 7];
#endif
""", 1),
    ("""        hb_regs<EXACT, 8>(ext0, y);
""",
     """#ifndef SDRX_ABL_ST0
        hb_regs<EXACT, 8>(ext0, y);
#else
#pragma unroll
        for (int j = 0; j < 8; ++j) // ablation: 1 add instead of the dot product
            y[j] = ext0[2 * j] + ext0[2 * j + 10];
#endif
""", 1),
    ("""        hb_regs<EXACT, 4>(ext1, z);
""",
     """#ifndef SDRX_ABL_ST1
        hb_regs<EXACT, 4>(ext1, z);
#else
#pragma unroll
        for (int j = 0; j < 4; ++j)
            z[j] = ext1[2 * j] + ext1[2 * j + 10];
#endif
""", 1),
    ("""            *reinterpret_cast<v4f *>(A2 + 2) = cat2(z[2], z[3]);
        }
""",
     """            *reinterpret_cast<v4f *>(A2 + 2) = cat2(z[2], z[3]);
        }
#ifdef SDRX_ABL_LDS
        if (emit && lane < 32) // ablation: skip the LDS stages, write something that depends on z
            gstv2(out + (base >> D.d) + lane, z[0] + z[1] + z[2] + z[3]);
        continue;
#endif
""", 1),
    ("""        const v2f *w = A + kCarry + 2 * j - 10; // w[0..10], newest = input sample 2j of this chunk
""",
     """#ifndef SDRX_ABL_CONFLICT
        const v2f *w = A + kCarry + 2 * j - 10; // w[0..10], newest = input sample 2j of this chunk
#else
        const v2f *w = A + kCarry + j - 10 + (j >> 6); // ablation: lane stride 1 (conflict-free reads, wrong results)
#endif
""", 1),
]


def main():
    old = open(SRC).read()
    new = old
    for anchor, repl, n in EDITS:
        if new.count(anchor) != n:
            sys.exit(f"anchor occurs {new.count(anchor)} times, expected {n}:\n{anchor[:200]}")
        new = new.replace(anchor, repl)
    rel = "sdrreceiver_amd/csrc/kernels.hip"
    diff = difflib.unified_diff(old.splitlines(keepends=True), new.splitlines(keepends=True), "a/" + rel, "b/" + rel)
    open(os.path.join(ROOT, "tools", "ablation.patch"), "w").writelines(diff)
    print("tools/ablation.patch written:", sum(1 for _ in EDITS), "hooks")


if __name__ == "__main__":
    main()
