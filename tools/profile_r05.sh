#!/bin/bash
# Round-5 evidence: tools/profile_r04.sh with its outputs under gpurun_out/prof_r05/ (copy them to profiles/ as r05_* and hbm_traffic.json)
PROFILE_TAG=r05 exec bash "$(dirname "$0")/profile_r04.sh" "$@"
