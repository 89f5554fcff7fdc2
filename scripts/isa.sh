#!/bin/bash
# usage: scripts/isa.sh <mangled-substring>   -> /tmp/kernel.s plus a few stats
R=/root/repo/combinatorial-rl-tasks_amd
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-pass-failed -S --cuda-device-only -o /tmp/k.s $R/csrc/kernels.hip || exit 1
name=$(grep -oE "^_ZN5zenvk[A-Za-z0-9_]*$1[A-Za-z0-9_]*:" /tmp/k.s | head -1 | tr -d ':')
echo "kernel: $name"
awk "/^$name:/,/s_endpgm/" /tmp/k.s > /tmp/kernel.s
wc -l /tmp/kernel.s
echo "vmcnt waits:"; grep "s_waitcnt vmcnt" /tmp/kernel.s | awk '{printf "%s ", $2}' ; echo
grep -A12 "^\s*.amdhsa_kernel $name" /tmp/k.s | grep -E "next_free_vgpr|next_free_sgpr"
grep -E "^\s+; (ScratchSize|Occupancy|NumVgprs|NumSgprs)" /tmp/k.s | head -0
