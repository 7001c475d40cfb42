#!/bin/bash
# Build libgca_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")/video-graph-ssl_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function"
objs=""
for f in conv3d conv3d_halo conv3d_stem conv3d_pw conv3d_wgrad conv3d_wgrad_ts conv3d_wgrad_stem bn pool misc infonce graph input; do
  $HIPCC $FLAGS -c $f.hip -o $f.o &
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libgca_hip.so conv3d.o conv3d_halo.o conv3d_stem.o conv3d_pw.o conv3d_wgrad.o conv3d_wgrad_ts.o conv3d_wgrad_stem.o bn.o pool.o misc.o infonce.o graph.o input.o
ls -la ../libgca_hip.so
