#!/bin/bash
# Compile engine.hip with -save-temps and print instruction statistics of one kernel (default: the v6 kernel).
# Usage: bash tools/isa_v6.sh [mangled-kernel-name-substring] [extra hipcc flags...]
cd "$(dirname "$0")/.."
K=${1:-prop_bf16_v6_kernel}; shift
mkdir -p /tmp/isa && cd /tmp/isa
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -L/opt/rocm/lib -lhipblaslt -save-temps=obj "$@" -o /tmp/isa/lib.so /root/repo/semi-supervised-vos_amd/csrc/engine.hip 2>&1 | grep -v "^$" | head -20
awk -v k="$K" 'index($0, k) && /^_ZN7vosprop/ && /: *;/ {p=1} p {print} /\.end_amdhsa_kernel/ {if (p) exit}' engine-hip-amdgcn-amd-amdhsa-gfx950.s > k.s
echo "lines: $(wc -l < k.s)"; grep -E "private_segment_fixed_size|next_free_vgpr|accum_offset" k.s
for k in v_mfma v_exp_f32 v_pk_fma_f32 v_pk_mul_f32 v_pk_add_f32 v_fma_f32 v_add_f32 v_cvt_pk v_accvgpr "s_waitcnt vmcnt" scratch_ v_cndmask ds_read_b128 global_load_lds s_barrier; do echo "$k: $(grep -c "$k" k.s)"; done
