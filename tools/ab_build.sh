#!/bin/bash
# A second build of the library with extra -D flags on conv.hip, for same-box kernel comparisons:
#   tools/ab_build.sh NAME -DFLAG...   ->  lite-mkd_amd/build/liblmkd_NAME.so      (run with LMKD_LIB=that path)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
python lite-mkd_amd/build.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c lite-mkd_amd/csrc/conv.hip -o lite-mkd_amd/build/conv_$name.o
objs=$(ls lite-mkd_amd/build/*.o | grep -v "/conv" )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lite-mkd_amd/build/liblmkd_$name.so lite-mkd_amd/build/conv_$name.o $objs
echo lite-mkd_amd/build/liblmkd_$name.so
