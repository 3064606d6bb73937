#!/bin/bash
# usage: build_variant_dir.sh NAME SRC_DIR "-DFLAG ..."  -> tools/lib_NAME.so from the .hip files of SRC_DIR
set -e
cd "$2"
d=/tmp/var_$1; mkdir -p $d
for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -I/root/repo/gaussian-splatting_amd/csrc $3 -c $f -o $d/${f%.hip}.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /root/repo/tools/lib_$1.so $d/*.o
