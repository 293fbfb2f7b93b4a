#!/bin/bash
# Compile the kernels for gfx950 and print VGPR / scratch per kernel; dump ISA of one kernel.
# usage: tools/kres.sh [pattern-for-ISA-dump]
set -e
R=/root/repo
hipcc -O3 --offload-arch=gfx950 -std=c++17 $EXTRA -c $R/radio-observer_amd/csrc/ro_kernels.hip -o /tmp/ro_kernels.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|error|warning: v" \
  | sed 's/.*remark: //;s/\[-Rpass.*//' | paste - - - | sed 's/Function Name: //' | cut -c1-200
if [ -n "$1" ]; then
  hipcc -O3 --offload-arch=gfx950 -std=c++17 $EXTRA -S --cuda-device-only $R/radio-observer_amd/csrc/ro_kernels.hip -o /tmp/ro_kernels.s 2>/dev/null
  awk -v pat="$1" 'index($0, pat) == 1 && /:/ {p=1} p {print} /s_endpgm/ {if (p) exit}' /tmp/ro_kernels.s > /tmp/kdump.s
  wc -l /tmp/kdump.s
  grep -n "scratch_store\|scratch_load\|s_barrier\|buffer_load\|buffer_store\|ds_write\|ds_read" /tmp/kdump.s | awk '{print $1, $2}' | awk -F: '{print $1": "$2}' | awk '{if (last!=$2) {printf "\n%s %s", $1, $2; last=$2} else {printf "."}} END{print ""}'
fi
