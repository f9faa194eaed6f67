#!/bin/bash
# Ablation builds of the MLP kernels: dfnet_amd/libabl_<flag>.so = the library with nerfh_mlp.hip compiled under -DDFN_ABL_<flag>
# (timing only: results are garbage).  usage: tools/build_ablations.sh NOSYNC NOBAR NODMA NOEPI NOPE NOLDS
# ADDITIVE forms (results stay correct, the data the kernel runs on does not change — the only kind that is comparable in a power-limited
# kernel): ADD_DMA (every weight piece streamed twice), ADD_VALU (+4 vector instructions per split-f16 conversion piece), ADD_LDS (+1 fragment read per chunk)
set -e
cd "$(dirname "$0")/../dfnet_amd/csrc"
make -j8 > /dev/null
OTHERS=$(ls build/*.o | grep -v nerfh_mlp.o)
mkdir -p build_abl
for f in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -DDFN_ABL_$f -c nerfh_mlp.hip -o build_abl/$f.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libabl_$f.so $OTHERS build_abl/$f.o && echo built $f ) &
done
wait
