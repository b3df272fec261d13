#!/bin/bash
# Round-6 evidence: tools/profile_r04.sh with its outputs under gpurun_out/prof_r06/ (copy them to profiles/ as r06_* and hbm_traffic.json)
PROFILE_TAG=r06 exec bash "$(dirname "$0")/profile_r04.sh" "$@"
