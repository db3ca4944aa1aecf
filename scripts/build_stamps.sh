#!/bin/bash
# diagnostic build of libaoenv with in-kernel stage stamps (scripts/diag_stamps.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/stamps
for f in rlao_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ ! -f build/stamps/$b.o ] || [ $f -nt build/stamps/$b.o ] || [ rlao_amd/csrc/sh_device.hpp -nt build/stamps/$b.o ] || [ rlao_amd/csrc/common.hpp -nt build/stamps/$b.o ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Irlao_amd/csrc -DAO_STEP_STAMPS -c $f -o build/stamps/$b.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/stamps/libaoenv_stamps.so build/stamps/*.o
