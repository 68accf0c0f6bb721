#!/bin/bash
# usage: tools/pmc.sh <name> <cmd...>  -> gpurun_out/pmc_<name>/{fetch,write}/ counter CSVs (separate passes,
# counters only with --kernel-trace, as MI355X_MICROARCH.md prescribes)
name=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o fetch --output-format csv -- "$@" > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o write --output-format csv -- "$@" > $out/write.log 2>&1
python tools/pmc_summarize.py $out
