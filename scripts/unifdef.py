#!/usr/bin/env python3
"""Resolve `#ifdef SYM` / `#ifndef SYM` ... `#else` ... `#endif` blocks for the symbols given as defined, in place;
every other conditional is left alone.  Used to promote a verified kernel variant: the `#else` branch is deleted.

usage: unifdef.py -DSYM [-DSYM2 ...] file [file ...]"""
import re
import sys


def resolve(text, defined):
  out, stack = [], []     # stack entries: None (foreign conditional) or [keep_now, in_else, sym_defined_branch_first]
  for line in text.split("\n"):
    m = re.match(r"\s*#\s*(ifdef|ifndef|if|else|elif|endif)\b\s*(\w*)", line)
    emit = all(s is None or s[0] for s in stack)
    if not m:
      if emit:
        out.append(line)
      continue
    kind, sym = m.group(1), m.group(2)
    if kind in ("ifdef", "ifndef") and sym in defined:
      stack.append([kind == "ifdef", False])
      continue
    if kind in ("ifdef", "ifndef", "if"):
      stack.append(None)
      if emit:
        out.append(line)
      continue
    if not stack:
      raise SystemExit("unbalanced conditional: %r" % line)
    if kind in ("else", "elif"):
      if stack[-1] is None:
        if emit:
          out.append(line)
      else:
        if kind == "elif":
          raise SystemExit("#elif on a resolved symbol is not supported: %r" % line)
        stack[-1][0] = not stack[-1][0]
      continue
    top = stack.pop()      # endif
    if top is None and all(s is None or s[0] for s in stack):
      out.append(line)
  if stack:
    raise SystemExit("unterminated conditional")
  return "\n".join(out)


def main():
  defined = {a[2:] for a in sys.argv[1:] if a.startswith("-D")}
  for path in (a for a in sys.argv[1:] if not a.startswith("-D")):
    src = open(path).read()
    new = resolve(src, defined)
    if new != src:
      open(path, "w").write(new)
      print("%s: %d -> %d lines" % (path, src.count("\n"), new.count("\n")))


if __name__ == "__main__":
  main()
