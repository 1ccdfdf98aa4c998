#!/bin/bash
for i in 1 2 3 4; do
  tools/_old/ablate_fused_r2 256 256 q | sed 's/^/r2      : /' | cut -c1-150
  WINO_TOOL_KP=0 tools/ablate_fused 256 256 q | sed 's/^/new kp=0: /' | cut -c1-150
  tools/ablate_fused 256 256 q | sed 's/^/new kp=1: /' | cut -c1-150
done
