#!/bin/bash
# device ISA of one source with the library's own flags -> /tmp/<name>.s
src=$1; shift
name=$(basename $src .hip)
cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 -fno-fast-math "$@" --offload-arch=gfx950 -S --cuda-device-only -I/root/repo/include -o /tmp/$name.s /root/repo/ex4dgs_amd/csrc/$name.hip 2>&1 | grep -E "error" -A3
grep -n "vgpr_count" /tmp/$name.s
