#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s9
for v in base dev base dev; do
  echo "== $v"
  CENTERCLIP_HIP_LIB=$PWD/ab/lib_$v.so python tools/sel_prof.py 2>&1 | grep "^P="
  CENTERCLIP_HIP_LIB=$PWD/ab/lib_$v.so python tools/time_cluster.py 2>&1 | grep "gauss   p=2.0"
done > gpurun_out/s9/ab.txt 2>&1
cat gpurun_out/s9/ab.txt
