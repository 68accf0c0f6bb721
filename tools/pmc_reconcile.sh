#!/bin/bash
# usage: tools/pmc_reconcile.sh <name> <cmd...>  -> gpurun_out/pmc_rec_<name>/<group>/ : the TCC request counters behind
# FETCH_SIZE / WRITE_SIZE, one counter group per run (counters + kernel trace only), summarised per kernel by
# tools/pmc_reconcile.py.  Groups that name a counter this rocprofv3 does not know fail on their own.
name=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_rec_$name
mkdir -p $out
cd $GRAFT_REPO_ROOT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d $out/g$i -o g$i --output-format csv -- "$@" > $out/g$i.log 2>&1 || echo "group $i ($grp) failed: $(tail -2 $out/g$i.log | tr '\n' ' ')"
done
python tools/pmc_reconcile.py $out
