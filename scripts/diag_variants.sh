#!/bin/bash
# A/B builds of the fused step kernel with extra -D flags (timing experiments):  build/variants/libaoenv_<name>.so
#   scripts/diag_variants.sh build name1="-DX=1 -DY=2" name2="-DZ"      (here)
#   scripts/diag_variants.sh run name1 name2 ...                        (GPU box: us per step, photon noise)
set -e
cd "$(dirname "$0")/.."
mode=$1; shift
mkdir -p build/variants
if [ "$mode" = build ]; then
  for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Irlao_amd/csrc $flags \
        -c rlao_amd/csrc/step_kernel.hip -o build/variants/step_kernel_$name.o 2>/dev/null &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libaoenv_$name.so build/variants/step_kernel_$name.o \
        $(ls rlao_amd/csrc/*.o | grep -v step_kernel.o) ) &
  done
  wait
else
  for name in "$@"; do
    AOENV_LIB=build/variants/libaoenv_$name.so python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --min-seconds 0.3 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name: step us', round(1e3*d['ms_per_step'],2), ' kernel us', round(d['kernels']['env_step']['avg_us'],2))"
  done
fi
