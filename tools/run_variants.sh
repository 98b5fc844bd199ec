#!/bin/bash
# run kbench (library kernels only) against every built variant
cd "$(dirname "$0")/.."
for d in tools/variants/*/; do
  echo "=== $(basename $d)"
  LD_LIBRARY_PATH=$d timeout -k 5 60 ./tools/kbench ${1:-50176000} ${2:-20} lib | tail -4
done
