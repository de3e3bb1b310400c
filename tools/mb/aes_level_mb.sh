#!/bin/bash
# runs tools/mb/aes_level_mb.bin over the variants DESIGN quotes (build it first: see the header of aes_level_mb.hip)
B=tools/mb/aes_level_mb.bin
for st in 0 1; do for sr in 0 1 2; do $B $st $sr 0; done; done
for st in 0 1; do for g in 2 3 4 6; do $B $st 0 $g; $B $st 2 $g; done; done
echo "--- ONE launch, 153 trips per wavefront (no re-staging, no launch gaps)"
$B 0 0 0 153 1; $B 0 1 0 153 1; $B 0 2 2 153 1; $B 0 2 4 153 1
echo "--- 264 workgroups (a level of 528 blocks: some SIMDs run a third trip)"
$B 0 0 0 153 0 264; $B 0 2 4 153 0 264
echo "--- 128 workgroups (half a level per launch)"
$B 0 0 0 153 0 128; $B 0 2 4 153 0 128
