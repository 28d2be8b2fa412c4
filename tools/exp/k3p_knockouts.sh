#!/bin/bash
# Knock-out builds of K3p (banded_fill2p.hip): what each part of a step costs.  The variants compute WRONG results (they
# leave work out); only their fill time is of interest.  Builds tools/exp/_ko/libbiogpu_<name>.so (git-ignored) here,
# on the CPU box; tools/exp/time_banded.py <lib> times them on the GPU.
set -e
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/rust-bio_amd/csrc
K=$R/tools/exp/_ko
mkdir -p $K
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$C -Wno-unused-value"
OBJS=$(ls $C/build/*.o | grep -v banded_fill2p.o)
variant() {  # name, sed script
    sed -E "$2" $C/banded_fill2p.hip > $K/banded_fill2p_$1.hip
    if cmp -s $K/banded_fill2p_$1.hip $C/banded_fill2p.hip; then echo "variant $1: the pattern did not match"; exit 1; fi
    hipcc $FLAGS -c $K/banded_fill2p_$1.hip -o $K/banded_fill2p_$1.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o $K/libbiogpu_$1.so $OBJS $K/banded_fill2p_$1.o -ldl -lrt
    rm -f $K/banded_fill2p_$1.o
    echo "built $1"
}
variant noring  's|\*ring_at\(r, slot\) = \(uint8_t\)cell;|asm volatile("" :: "v"(cell));|; s|\*ring_at\(R \+ r, slot\) = \(uint8_t\)\(cell >> 16\);||'
variant noflush 's|if \(\(t_end \& \(FLUSH - 1\)\) == 0\) flush_tb\(.*$||; s|flush_tb\(0, t_last < 0.*$||'
variant nomerge 's|^( +)merge_rows\(t0 \+ 15\);|\1asm volatile("" :: "v"(SnB[0]), "v"(SnB[1]));|'
variant nohand  's|^( +)hand_over\(t0, t_end\);||; s|^( +)hand_over\(t0 \+ 8, t_end\);||'
variant nochunk 's|const Raw raw = issue_chunk\(t0 \+ 16\);|Raw raw; raw.q[0] = raw.q[1] = 1; raw.b[0] = raw.b[1] = make_int2(NEGS, NEGS);|'
variant w3      's|amdgpu_waves_per_eu\(2, 2\)|amdgpu_waves_per_eu(3, 3)|'
