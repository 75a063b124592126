#!/usr/bin/env python3
"""Are the device kernels of the working tree's default build the SAME INSTRUCTIONS as those of a git revision?
Compiles both with hipcc -save-temps (gfx950, no GPU needed) and compares every kernel's instruction stream
(comments, directives and basic-block numbering stripped).  Used in round 4 to show that the shipped library
carries exactly the kernels round 3 verified on an MI355X while the unverified variants live behind -DPCLSEG_CAND
(round 4: -DPCLSEG_R4X).

usage: kernel_isa_diff.py <git revision> [extra hipcc flags for the working-tree build, e.g. -DPCLSEG_CAND]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["pclsegmentation_amd/csrc/pclseg_kernels.h", "pclsegmentation_amd/csrc/pclseg_api.hip",
         "pclsegmentation_amd/csrc/pclseg_graph.h", "include/pclseg.h"]
# the device side only, as assembly (the Makefile's flags; the host compile adds nothing to compare)
HIPCC = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-function", "--cuda-device-only", "-S"]


def build(src_root, out_dir, extra):
  os.makedirs(out_dir, exist_ok=True)
  subprocess.check_call(HIPCC + extra + ["-o", "device.s", os.path.join(src_root, FILES[1])], cwd=out_dir,
                        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
  return os.path.join(out_dir, "device.s")


def kernels(path):
  out, cur, buf = {}, None, []
  for line in open(path):
    m = re.match(r"^(_ZN6pclseg\w+):", line)
    if m:
      cur, buf = m.group(1), []
      continue
    if cur is None:
      continue
    t = line.strip()
    if t.startswith(".Lfunc_end"):
      out[cur], cur = "\n".join(buf), None
      continue
    if not t or t[0] in ";.":
      continue
    buf.append(re.sub(r"\.LBB\d+_", ".LBB_", re.sub(r";.*", "", t).strip()))
  return out


def stream_hashes(kern):
  import hashlib
  return {k: hashlib.sha256(v.encode()).hexdigest()[:24] for k, v in sorted(kern.items())}


def manifest_of_revision(rev):
  """{mangled kernel name: hash of its instruction stream} of the default build of a git revision."""
  with tempfile.TemporaryDirectory() as tmp:
    old = os.path.join(tmp, "old_src")
    for f in FILES:
      os.makedirs(os.path.dirname(os.path.join(old, f)), exist_ok=True)
      with open(os.path.join(old, f), "wb") as fh:
        fh.write(subprocess.check_output(["git", "show", "%s:%s" % (rev, f)], cwd=ROOT))
    return stream_hashes(kernels(build(old, os.path.join(tmp, "a"), [])))


def manifest_of_tree(extra=()):
  with tempfile.TemporaryDirectory() as tmp:
    return stream_hashes(kernels(build(ROOT, os.path.join(tmp, "b"), list(extra))))


def main():
  if sys.argv[1] == "--write-manifest":      # kernel_isa_diff.py --write-manifest <revision> <out.json>
    import json
    json.dump({"revision": sys.argv[2], "hipcc": subprocess.check_output([HIPCC[0], "--version"], text=True).splitlines()[0],
               "kernels": manifest_of_revision(sys.argv[2])}, open(sys.argv[3], "w"), indent=0)
    return 0
  rev, extra = sys.argv[1], sys.argv[2:]
  with tempfile.TemporaryDirectory() as tmp:
    old = os.path.join(tmp, "old_src")
    for f in FILES:
      os.makedirs(os.path.dirname(os.path.join(old, f)), exist_ok=True)
      with open(os.path.join(old, f), "wb") as fh:
        fh.write(subprocess.check_output(["git", "show", "%s:%s" % (rev, f)], cwd=ROOT))
    a = kernels(build(old, os.path.join(tmp, "a"), []))
    b = kernels(build(ROOT, os.path.join(tmp, "b"), extra))
  differ = sorted(k for k in a if k in b and a[k] != b[k])
  print("%s: %d kernels; working tree%s: %d kernels" % (rev, len(a), (" " + " ".join(extra)) if extra else "", len(b)))
  print("identical: %d   differing: %d   only in %s: %d   only in the working tree: %d"
        % (sum(1 for k in a if k in b and a[k] == b[k]), len(differ), rev, sum(1 for k in a if k not in b), sum(1 for k in b if k not in a)))
  for k in differ + sorted(k for k in b if k not in a):
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
    print("  %s %s" % ("differs:" if k in a else "new:    ", name[:150]))
  if os.environ.get("KERNEL_ISA_DIFF_LIST_DROPPED"):      # (carry_traffic.py: were any of them launched by the measured run?)
    for k in sorted(k for k in a if k not in b):
      print("  dropped: %s" % (subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k))
  return 1 if differ and not extra else 0


if __name__ == "__main__":
  sys.exit(main())
