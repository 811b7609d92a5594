#!/bin/bash
# Register / LDS / spill figures of the device kernels of one HIP translation unit (cross-compiles, no GPU needed):
#   tools/kstats.sh hot_amd/csrc/mg_solve.hip [name filter] [-DHOT_AB_KERNELS]
src=$1; filt=${2:-.}; shift 2
d=$(dirname $src)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast -Wno-unused-result -Wno-unused-value -Wno-pass-failed "$@" \
  --cuda-device-only -S -o /tmp/kstats.s $src -I$d 2>/dev/null
awk '/^[ \t]*\.amdhsa_kernel /{name=$2} /\.amdhsa_next_free_vgpr/{v=$2} /\.amdhsa_next_free_sgpr/{s=$2} /\.amdhsa_group_segment_fixed_size/{l=$2} /\.amdhsa_private_segment_fixed_size/{p=$2} /\.amdhsa_accum_offset/{a=$2} /^[ \t]*\.end_amdhsa_kernel/{print name, "vgpr", v, "accum_off", a, "sgpr", s, "lds", l, "scratch", p}' /tmp/kstats.s | c++filt | grep -E "$filt"
