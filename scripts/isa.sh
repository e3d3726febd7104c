#!/bin/bash
# Device ISA of one kernel source with the library's flags: scripts/isa.sh k_conv.hip [extra flags] -> /tmp/isa/k_conv.s
set -e
src=$1; shift
mkdir -p /tmp/isa
cd /root/repo/irmv_detection_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I ../../include "$@" --cuda-device-only -S $src -o /tmp/isa/${src%.hip}.s
echo /tmp/isa/${src%.hip}.s
