#!/bin/bash
# GPU box: which sysfs files give the shader clock and the package power of the visible GPU to an ordinary user?
for d in /sys/class/drm/card*/device; do
  echo "== $d -> $(readlink -f $d)"
  for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent current_link_speed; do [ -r $d/$f ] && { echo "-- $f"; head -12 $d/$f; }; done
  for h in $d/hwmon/hwmon*; do
    echo "-- $h"
    for f in $h/power1_average $h/power1_input $h/power1_cap $h/freq1_input $h/freq1_label $h/freq2_input $h/temp1_input; do
      [ -r $f ] && echo "$(basename $f) $(cat $f 2>&1)"
    done
  done
done 2>&1 | head -150
ls /sys/class/kfd/kfd/topology/nodes/ 2>&1 | head
python3 - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print("torch device:", p.name, getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None), getattr(p, "pci_domain_id", None))
PY
