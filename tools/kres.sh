#!/bin/bash
# kernel resources of one csrc file: tools/kres.sh gemm_adaln [extra hipcc flags]   (compiles into /tmp/kres, prints vgpr / scratch / lds / occupancy per kernel)
f=$1; shift
mkdir -p /tmp/kres && cd /tmp/kres
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -mllvm -amdgpu-mfma-vgpr-form=1 -ffast-math -fno-finite-math-only -fgpu-flush-denormals-to-zero -Wno-unused-function -Wno-unused-result "$@" -save-temps=obj -c /root/repo/sea_amd/csrc/$f.hip -o /tmp/kres/$f.o || exit 1
awk '/^\s+\.amdhsa_kernel /{k=$2} /amdhsa_next_free_vgpr|amdhsa_private_segment_fixed_size|amdhsa_group_segment_fixed_size/{v[k]=v[k]" "$1"="$2} /; Occupancy:/{o[k]=$3} END{for(k in v) print k, v[k]}' /tmp/kres/$f-hip-amdgcn-amd-amdhsa-gfx950.s | sed 's/\.amdhsa_//g'
grep -E "^; (Occupancy|ScratchSize|NumVgprs|NumAgprs)" /tmp/kres/$f-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - | head -40
