#!/bin/bash
# every build the round's GPU scripts use, then the carried-over traffic figure for the shipped one
cd "$(dirname "$0")/.."
(make 2>&1 | tail -1) & (make r4x 2>&1 | tail -1) &
(make variant NAME=r4x_noslab EXTRA="-DPCLSEG_R4X_TAIL -DPCLSEG_R4X_CAM -DPCLSEG_R4X_WIDE -DPCLSEG_R4X_KPIPE -DPCLSEG_R4X_GEOM2" 2>&1 | tail -1) & wait
make stamps EXTRA=-DPCLSEG_R4X 2>&1 | tail -1
python3 scripts/carry_traffic.py profiles/r03_traffic.json ad2e081 profiles/r04_traffic.json | tail -2
python3 scripts/kernel_isa_diff.py ad2e081 > profiles/r04_kernel_isa_vs_r03.txt 2>&1
python3 scripts/kernel_isa_diff.py ad2e081 -DPCLSEG_R4X >> profiles/r04_kernel_isa_vs_r03.txt 2>&1
