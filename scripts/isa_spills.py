#!/usr/bin/env python3
"""Where do a kernel's register spills sit?  For every kernel of a `hipcc -save-temps` device assembly (gfx950) that
uses scratch memory: the scratch loads / stores outside any loop, and per loop (a backward branch and its target)
the scratch instructions next to the MFMAs of that loop.  A spill in a prologue or epilogue costs one memory round
trip per block; one inside the K loop is paid per K-step.  No GPU needed.

usage: isa_spills.py <file.s> [substring of the demangled kernel name ...]"""
import re
import subprocess
import sys


def kernels(path):
  cur, buf = None, []
  for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m and cur is None:
      cur, buf = m.group(1), []
      continue
    if cur is None:
      continue
    t = line.strip()
    if t.startswith(".Lfunc_end"):
      yield cur, buf
      cur = None
      continue
    buf.append(t)


def analyse(lines):
  labels, ins = {}, []
  for t in lines:
    m = re.match(r"^(\.LBB\w+):", t)
    if m:
      labels[m.group(1)] = len(ins)
      continue
    if not t or t[0] in ";." or t.endswith(":"):
      continue
    ins.append(re.sub(r"\s*;.*", "", t))
  loops = []
  for i, t in enumerate(ins):
    m = re.match(r"s_c?branch\w*\s+(\.LBB\w+)", t)
    if m and labels.get(m.group(1), 1 << 30) <= i:
      loops.append((labels[m.group(1)], i))
  def innermost(i):
    best = None
    for lo, hi in loops:
      if lo <= i <= hi and (best is None or hi - lo < best[1] - best[0]):
        best = (lo, hi)
    return best
  is_scr = lambda t: t.startswith("scratch_")
  is_mfma = lambda t: t.startswith("v_mfma")
  out = {"total": sum(map(is_scr, ins)), "mfma": sum(map(is_mfma, ins)), "outside": 0, "loops": {}}
  for i, t in enumerate(ins):
    if is_scr(t):
      l = innermost(i)
      if l is None:
        out["outside"] += 1
      else:
        out["loops"].setdefault(l, [0, 0])[0 if "load" in t else 1] += 1
  res = []
  for (lo, hi), (ld, st) in sorted(out["loops"].items()):
    res.append((hi - lo + 1, sum(map(is_mfma, ins[lo:hi + 1])), ld, st))
  return out, res


def main():
  path, pats = sys.argv[1], sys.argv[2:]
  for sym, lines in kernels(path):
    out, loops = analyse(lines)
    if not out["total"]:
      continue
    name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip().replace("pclseg::", "")
    name = re.sub(r"\((pclseg::)?\w+Args\)$", "", name)
    if pats and not any(p in name for p in pats):
      continue
    print("%s\n  %d scratch instructions, %d MFMAs in the kernel; %d scratch outside every loop" % (name, out["total"], out["mfma"], out["outside"]))
    for n, mf, ld, st in loops:
      print("  loop of %5d instructions, %4d MFMAs: %3d scratch loads, %3d scratch stores%s" % (n, mf, ld, st, "   <-- in a matrix loop" if mf else ""))


if __name__ == "__main__":
  main()
