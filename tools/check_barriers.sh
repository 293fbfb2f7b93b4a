#!/bin/bash
# Static check of the compiled kernels: two s_barrier with no LDS instruction between them means the optimiser has
# moved an exchange's LDS traffic out from between its barriers (seen once: see wg_sync() in ro_kernels.hip).
R=/root/repo
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize $EXTRA -S --cuda-device-only $R/radio-observer_amd/csrc/ro_kernels.hip -o /tmp/ro_kernels_chk.s 2>/dev/null
awk '
/^_ZN2ro.*:$/ { name=$1 }
/ds_write_addtid_b32/ { addtid=1 }
/s_waitcnt.*lgkmcnt\(0\)/ { addtid=0 }
/s_barrier/ { if (addtid) { printf "BARRIER AFTER UN-WAITED ADD-TID WRITES in %s line %d\n", name, NR; bad=1 }
              if (seen && !lds) { printf "ADJACENT BARRIERS in %s line %d\n", name, NR; bad=1 } seen=1; lds=0 }
/ds_read|ds_write|ds_add|buffer_load.*lds/ { lds=1 }
/s_endpgm/ { seen=0; lds=0; addtid=0 }
END { if (!bad) print "barriers ok"; exit bad }' /tmp/ro_kernels_chk.s
