#!/bin/bash
# time every tools/variants/<name>/libmhaq_fq.so (and the product library) with tools/ab_kernels.py, twice each
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  python3 tools/ab_kernels.py product
  for d in tools/variants/*/; do
    MHAQ_FQ_LIB=$PWD/${d}libmhaq_fq.so python3 tools/ab_kernels.py $(basename $d)
  done
done
