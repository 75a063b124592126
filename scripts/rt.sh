#!/bin/bash
# fast register-pressure check of a few conv_kernel instantiations (compile only)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -c -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage -o /tmp/rt/inst.o /tmp/rt/inst.hip 2> /tmp/rt/res.txt
python3 /root/repo/scripts/kernel_resources.py /tmp/rt/res.txt
