#!/bin/bash
# where the compiler put its s_waitcnt vmcnt / barriers in the tile kernels' loops (AM instantiations); run in the build container
cd "$(dirname "$0")/../torch-rgcn_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -munsafe-fp-atomics -I../../include -S --cuda-device-only -o /tmp/fbt.s rgcn_fbasis_tile.hip 2>&1 | grep -v warning | head -3
for k in "14fbt_fwd_kernelILi16ELi12ELi2ELb1E" "17fbt_dcomps_kernelILi12ELi2ELb1E" "17fbt_dbases_kernelILi12ELi2ELb1E"; do
  awk "/^_ZN12_GLOBAL__N_1$k.*:/,/s_endpgm/" /tmp/fbt.s > /tmp/k_$k.s
  echo "== $k $(wc -l < /tmp/k_$k.s) lines"
  grep -n "s_waitcnt vmcnt\|s_barrier\|Loop Header" /tmp/k_$k.s | head -${1:-80}
done
