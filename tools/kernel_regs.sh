#!/bin/bash
# Per-kernel VGPR / spill / LDS figures of one source of csrc/ (developer tool, no GPU needed).
# usage: tools/kernel_regs.sh raster.hip [extra hipcc flags]
src=$1; shift
cd "$(dirname "$0")/../tinysplat_amd/csrc" || exit 1
extra=""
case $src in raster.hip) extra="-fno-slp-vectorize";; project.hip|binning.hip|densify.hip|shard.hip) extra="-ffp-contract=off";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra "$@" -Rpass-analysis=kernel-resource-usage -c $src -o /tmp/kregs_$$.o 2>&1 |
  awk '/Function Name:/ {name=$NF} /remark:.* Name: / {for(i=1;i<=NF;i++) if($i=="Name:") name=$(i+1)}
       / VGPRs: /{v=$0; sub(/.* VGPRs: /,"",v); sub(/ .*/,"",v)}
       /VGPRs Spill: /{sp=$0; sub(/.*VGPRs Spill: /,"",sp); sub(/ .*/,"",sp)}
       /LDS Size/{l=$0; sub(/.*: /,"",l); sub(/ .*/,"",l); cmd="echo " name " | c++filt"; cmd | getline d; close(cmd); gsub(/\(anonymous namespace\)::/,"",d); sub(/\(.*/,"",d); sub(/^void /,"",d); printf "%4s vgpr %3s spill %6s lds  %s\n", v, sp, l, d}
       /error|warning/ {print}'
rm -f /tmp/kregs_$$.o
