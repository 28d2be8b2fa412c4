#!/bin/bash
# Variant ("knock-out") builds of the library: one translation unit recompiled with extra flags, the rest relinked as built.
#     bash tools/exp/ko_build.sh <source in csrc/> <name> [flags ...]     ->  tools/exp/_ko/libbiogpu_<name>.so  (git-ignored)
# e.g. ko_build.sh fastq_ingest.hip nolb -DFQ_KO_LOOKBACK.  A knock-out leaves work out: WRONG results, timing only
# (BG_SO=<that .so> python tools/exp/time_*.py ...).  Run here (hipcc cross-compiles); the .so travels with the snapshot.
set -e
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/rust-bio_amd/csrc
K=$R/tools/exp/_ko
SRC=$1; NAME=$2; shift 2
mkdir -p $K
make -C $C -j8 -s
base=$(basename ${SRC%.*})
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$C -Wno-unused-value -Wno-unused-variable "$@" -c $C/$SRC -o $K/${base}_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $K/libbiogpu_$NAME.so $(ls $C/build/*.o | grep -v "/$base.o") $K/${base}_$NAME.o -ldl -lrt
rm -f $K/${base}_$NAME.o
echo "built $K/libbiogpu_$NAME.so"
