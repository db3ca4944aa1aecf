#!/bin/bash
# Timing-only ablation builds of the camera block of the fused step kernel: build/ablate/libaoenv_<mask>.so with
# -DAO_CAM_ABLATE=<mask> (bit 0 no Philox, 1 no inversion, 2 no squeeze / queue, 3 nothing queued, 4 queue passes skipped).
#   scripts/diag_cam_ablate.sh build 0 1 2 4 ...     (here)
#   scripts/diag_cam_ablate.sh run 0 1 2 4 ...       (GPU box: us per step of each)
set -e
cd "$(dirname "$0")/.."
mode=$1; shift
mkdir -p build/ablate
if [ "$mode" = build ]; then
  for m in "$@"; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Irlao_amd/csrc -DAO_CAM_ABLATE=$m \
        -c rlao_amd/csrc/step_kernel.hip -o build/ablate/step_kernel_$m.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ablate/libaoenv_$m.so build/ablate/step_kernel_$m.o \
        $(ls rlao_amd/csrc/*.o | grep -v step_kernel.o) ) &
  done
  wait
else
  for m in "$@"; do
    echo "== ablate mask $m"
    AOENV_LIB=build/ablate/libaoenv_$m.so python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --min-seconds 0.2 |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   step us', round(1e3*d['ms_per_step'],2), ' kernel us', round(d['kernels']['env_step']['avg_us'],2))"
  done
fi
