#!/bin/bash
# tools/ab_build.sh <name> "<EXTRA flags>" -- an experiment build of libsdrx.so into sdrreceiver_amd/csrc/ab/<name>.so.
# The phase ablations of mix_item
# (-DSDRX_ABL_LOAD/NCO/MIX/CARRY/ST0/ST1/LDS/STORE/CONFLICT: WRONG results by design, the phase-cost study) are no longer
# part of the product's kernels.hip: tools/ablation.patch puts them back into a scratch copy of the sources, which this
# script builds.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2
SCR=$ROOT/sdrreceiver_amd/csrc/ab/src_$NAME
rm -rf "$SCR"; mkdir -p "$SCR/sdrreceiver_amd/csrc" "$SCR/include"
cp "$ROOT"/include/sdrx.h "$SCR/include/"
cp "$ROOT"/sdrreceiver_amd/csrc/{Makefile,*.hip,*.h} "$SCR/sdrreceiver_amd/csrc/"
# (the patch is regenerated from the CURRENT kernels.hip first -- tools/make_ablation_patch.py fails loudly when one of its
# anchors no longer matches the source, instead of `patch` failing on a stale file)
case "$EXTRA" in *SDRX_ABL_*) python3 "$ROOT/tools/make_ablation_patch.py" && (cd "$SCR" && patch -p1 < "$ROOT/tools/ablation.patch") ;; esac
make -C "$SCR/sdrreceiver_amd/csrc" OUT="$ROOT/sdrreceiver_amd/csrc/ab/$NAME.so" EXTRA="$EXTRA" CHECK_ASM="$ROOT/tools/check_asm.py"
echo "built sdrreceiver_amd/csrc/ab/$NAME.so  (SDRX_LIB=<that path> loads it)"
