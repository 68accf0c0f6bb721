#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>"  -> abx/lib_<name>.so (an A/B build of the CURRENT sources with
# extra -D flags; the shipped library is rebuilt afterwards).  abx/ is git-ignored but travels (delete it after the experiment: it rides every gpurun push) with the gpurun snapshot.
set -e
cd "$(dirname "$0")/.."
mkdir -p abx
CC_EXTRA_FLAGS="$2" python -m centerclip_amd.build --force > /dev/null
cp centerclip_amd/lib/libcenterclip_hip.so abx/lib_$1.so
python -m centerclip_amd.build --force > /dev/null
echo "abx/lib_$1.so built with: $2"
