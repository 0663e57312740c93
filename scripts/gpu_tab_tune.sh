#!/bin/bash
for b in 512 1024 2048 4096; do
  echo "blocks $b"
  M355_TAB_BLOCKS=$b timeout -k 10 200 python scripts/gpu_tab_probe.py 100000000 2>/dev/null | grep -E '"params": \[(2.5, 1.3|2.0, 5.0)\]' | grep -E '"fn": "(logcdf|sf)"' | cut -c1-150
done
