#!/bin/bash
# tools/isa.sh <kernel-name-substring> : compile csrc/pyz_api.hip for gfx950 with -save-temps into /tmp/pyz_isa and cut one
# kernel's ISA into /tmp/pyz_isa/kernel.s (development aid: wait counts, spills, instruction mix)
set -e
OUT=/tmp/pyz_isa
mkdir -p $OUT
cd "$(dirname "$0")/../bayesian_inference_for_nn_amd/csrc"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wall -Wno-unused-function -Wno-pass-failed ${PYZ_DEFS} pyz_api.hip -o $OUT/libpyz_test.so -save-temps=obj > $OUT/compile.log 2>&1 || { grep -B2 -A6 "error" $OUT/compile.log | head -60; exit 1; }
S=$OUT/pyz_api-hip-amdgcn-amd-amdhsa-gfx950.s
sym=$(grep -o "^_Z[A-Za-z0-9_]*$1[A-Za-z0-9_]*:" $S | head -1 | tr -d ':')
echo "symbol: $sym"
[ -z "$sym" ] && { echo "no such kernel"; exit 1; }
awk -v s="^$sym:" '$0 ~ s {on=1} on {print} on && /s_endpgm/ {exit}' $S > $OUT/kernel.s
wc -l $OUT/kernel.s
grep -A40 "^	.amdhsa_kernel $sym" $S | grep "next_free_vgpr\|next_free_sgpr\|group_segment\|private_segment_fixed\|accum_offset" 
echo "mfma: $(grep -c v_mfma $OUT/kernel.s)  scratch: $(grep -c scratch_ $OUT/kernel.s)"
