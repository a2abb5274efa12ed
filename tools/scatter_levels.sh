#!/bin/bash
# per-level cost of the binned scatter: run the experiment with one level (or a group) active at a time
for m in 0x1 0x2 0x4 0x8 0x10 0x20 0x100 0x8000 0x1f 0xffe0 0xffff; do
  echo -n "levels $m: "; ARCN_SCATTER_LEVELS=$m timeout 120 python tools/exp_scatter.py 2>&1 | grep samples
done
