#!/bin/bash
# Box pass of quantile_fast on a rank's row tile: workgroups per CU (GPP_QB_FILL) against the run-in rows every extra segment costs
# (csrc/qf_box.hip launch_hw).  Prints the C4 lines of tools/slice_overhead_other.py per setting.
for f in 1 2 3 4 6 8; do echo "GPP_QB_FILL=$f"; GPP_QB_FILL=$f python tools/slice_overhead_other.py nbh 2>&1 | grep "C4"; done
