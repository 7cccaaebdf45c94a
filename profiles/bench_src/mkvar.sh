#!/bin/bash
# usage: bash exp/mkvar.sh NAME -DFLAG...   -> exp/libamdarrow_NAME.so (groupby.hip rebuilt with the flags)
name=$1; shift
cd /root/repo/arrow-1_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include -w "$@" -c groupby.hip -o /tmp/groupby_$name.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/exp/libamdarrow_$name.so /tmp/groupby_$name.o $(ls build/*.o | grep -v groupby.o)
