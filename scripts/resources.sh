#!/bin/bash
# register / scratch / occupancy of every kernel in the library (compiles to a temp file)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage -o /tmp/lib_res.so pclsegmentation_amd/csrc/pclseg_api.hip 2> /tmp/res.txt
python3 scripts/kernel_resources.py /tmp/res.txt | sort | uniq
