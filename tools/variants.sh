#!/bin/bash
# build library variants with different -D flags into tools/variants/<name>/libmhaq_fq.so
# usage: tools/variants.sh name "-DX=1 -DY=2" [name2 "flags2" ...]
# The base flags come from the product Makefile (incl. -fno-gpu-approx-transcendentals), so a variant differs
# from the shipped library only by its -D knobs.
set -e
cd "$(dirname "$0")/.."
BASE=$(make -s -C mhaq_amd/csrc print-flags)
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  mkdir -p tools/variants/$name
  /opt/rocm/bin/hipcc $BASE $flags -shared -o tools/variants/$name/libmhaq_fq.so mhaq_amd/csrc/fq_pt.hip mhaq_amd/csrc/fq_pc.hip &
done
wait
