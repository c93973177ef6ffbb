#!/bin/bash
# GPU box: each seed given through the snapped test build and the snapped oracle (tools/debug_fuzz_sweep.py <seed> <seed+1> snap:
# exact equality demanded). usage: tools/snap_check_seeds.sh <seed> ...
for s in "$@"; do python3 tools/debug_fuzz_sweep.py $s $((s + 1)) snap 2>&1 | tail -1 | cut -c1-300 | sed "s/^/snap $s: /"; done
