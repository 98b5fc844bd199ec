#!/bin/bash
# build library variants with different -D flags into tools/variants/<name>/libmhaq_fq.so
# usage: tools/variants.sh name "-DX=1 -DY=2" [name2 "flags2" ...]
set -e
cd "$(dirname "$0")/.."
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  mkdir -p tools/variants/$name
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt $flags -shared -o tools/variants/$name/libmhaq_fq.so mhaq_amd/csrc/fq_pt.hip mhaq_amd/csrc/fq_pc.hip &
done
wait
