#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel, parsed from
hipcc -Rpass-analysis=kernel-resource-usage output.  usage: kernel_resources.py <stderr file>"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
for b in blocks:
  name = b.split("\n")[0].strip().split()[0]

  def g(k):
    m = re.search(k + r": (\d+)", b)
    return int(m.group(1)) if m else -1
  try:
    nm = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
  except FileNotFoundError:
    nm = name
  nm = nm.replace("pclseg::", "").replace("(pclseg::ConvArgs)", "").replace("(ConvArgs)", "")[:110]
  print("%-112s vgpr=%3d agpr=%3d scratch=%4d occ=%d sgpr=%3d" % (
    nm, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g("SGPRs")))
