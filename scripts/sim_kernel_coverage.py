#!/usr/bin/env python3
"""Which device kernels does the test suite execute?  Runs the GPU suite on the functional simulator with
HIPSIM_TRACE set, and compares the kernel instantiations that were launched with the kernels hipcc emits for gfx950
(hipcc -save-temps device assembly).  A kernel that ships but is never launched by a test is listed.

usage: sim_kernel_coverage.py <device.s> [pytest args ...]       (default pytest args: tests -m gpu -q -k 'not full_size')"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def norm(name):
  name = re.sub(r"^void ", "", name.strip())
  name = name.replace("pclseg::", "").replace("(anonymous namespace)::", "")
  return re.sub(r"\(.*\)$", "", name)


def main():
  asm, args = sys.argv[1], sys.argv[2:] or ["tests", "-m", "gpu", "-q", "-k", "not full_size"]
  syms = re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", open(asm).read(), re.M)
  shipped = {norm(n) for n in subprocess.run(["c++filt"] + syms, capture_output=True, text=True).stdout.splitlines()}
  with tempfile.TemporaryDirectory() as tmp:
    trace = os.path.join(tmp, "trace.txt")
    env = dict(os.environ, PCLSEG_SIM="1", HIPSIM_TRACE=trace)
    r = subprocess.run([sys.executable, "-m", "pytest"] + args, cwd=ROOT, env=env, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1])
    launched = {}
    for line in open(trace):
      k = norm(line.split("\t")[0])
      launched[k] = launched.get(k, 0) + 1
  hit = shipped & set(launched)
  print("device kernels in the gfx950 build: %d; launched by the suite on the simulator: %d (%.0f %%)" % (len(shipped), len(hit), 100.0 * len(hit) / len(shipped)))
  for k in sorted(shipped - hit):
    print("  never launched: %s" % k)
  for k in sorted(set(launched) - shipped):
    print("  launched but not in the device build (name mismatch?): %s" % k)


if __name__ == "__main__":
  main()
