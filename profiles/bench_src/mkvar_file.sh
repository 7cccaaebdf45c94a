#!/bin/bash
# usage: bash profiles/bench_src/mkvar_file.sh FILE NAME -DFLAG...  -> exp/libamdarrow_NAME.so (csrc/FILE.hip rebuilt with the flags,
# everything else from csrc/build)
file=$1; name=$2; shift 2
cd /root/repo/arrow-1_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include -w "$@" -c $file.hip -o /tmp/${file}_$name.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/exp/libamdarrow_$name.so /tmp/${file}_$name.o $(ls build/*.o | grep -v "/$file.o")
