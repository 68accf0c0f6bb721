#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>"  -> ab/lib_<name>.so (an A/B build of the CURRENT sources with
# extra -D flags; the shipped library is rebuilt afterwards).  ab/ is git-ignored but travels with the gpurun snapshot.
set -e
cd "$(dirname "$0")/.."
mkdir -p ab
CC_EXTRA_FLAGS="$2" python -m centerclip_amd.build --force > /dev/null
cp centerclip_amd/lib/libcenterclip_hip.so ab/lib_$1.so
python -m centerclip_amd.build --force > /dev/null
echo "ab/lib_$1.so built with: $2"
