#!/usr/bin/env python3
"""Carry a measured PMC traffic figure over to a build whose DEVICE KERNELS are instruction-identical to the build it
was measured on (host-only changes: ABI barrier, plan output, ...).  Runs scripts/kernel_isa_diff.py against the git
revision of the measurement; only when all kernels are identical does it write profiles/<round>_traffic.json = the
measured file re-keyed to the current source hash, with the provenance spelled out (bench.py prints it in
`traffic_unit`).  Any differing kernel: nothing is written, the figure must be re-measured (scripts/profile_round.sh).

usage: carry_traffic.py <measured json> <git revision it was measured at> <out json> [kernel_stats.csv of the measured runs ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
  src, rev, dst = sys.argv[1], sys.argv[2], sys.argv[3]
  stats = sys.argv[4:]      # the measured run's rocprofv3 kernel_stats.csv files: which kernels it launched
  r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "kernel_isa_diff.py"), rev], capture_output=True, text=True,
                     env=dict(os.environ, KERNEL_ISA_DIFF_LIST_DROPPED="1"))
  print("\n".join(l for l in r.stdout.splitlines() if "dropped:" not in l))
  line = [l for l in r.stdout.splitlines() if l.startswith("identical:")]
  if r.returncode != 0 or not line:
    raise SystemExit("kernel_isa_diff failed or kernels differ: the traffic figure is NOT carried over")
  f = dict(zip(["identical", "differing", "only_rev", "only_tree"], [int(x) for x in line[0].replace(":", " ").split() if x.isdigit()]))
  if f["differing"] or f["only_tree"]:
    raise SystemExit("kernels differ from %s: re-measure" % rev)
  dropped_note = ""
  if f["only_rev"]:
    # instantiations this build no longer carries (dispatch tables cut to what a plan can select): acceptable only if the
    # measured run launched none of them
    if not stats:
      raise SystemExit("%d kernels of %s are gone: pass the measured run's kernel_stats.csv files to show none was launched" % (f["only_rev"], rev))
    import csv
    launched = set()
    for path in stats:
      launched.update(row["Name"].replace("void ", "", 1) for row in csv.DictReader(open(path)))
    dropped = [l.split("dropped: ", 1)[1].strip() for l in r.stdout.splitlines() if "dropped:" in l]
    hit = sorted(d for d in dropped if d in launched)
    if hit:
      raise SystemExit("the measured run launched kernels this build no longer has: %s" % hit[:3])
    dropped_note = ("; %d instantiations of the measured build that no plan selects were dropped, none of them among the %d kernels "
                    "the measured runs launched" % (len(dropped), len(launched)))
  t = json.load(open(src))
  t["measured_on_csrc_sha"] = t["csrc_sha"]
  t["measured_at_revision"] = rev
  t["csrc_sha"] = bench.csrc_sha()
  t["carried_over"] = ("not re-measured: all %d device kernels of this build are instruction-identical to the build the "
                       "figures were measured on (scripts/kernel_isa_diff.py %s; host-side changes only)%s" % (f["identical"], rev, dropped_note))
  json.dump(t, open(dst, "w"), indent=1)
  print("wrote %s for csrc sha %s (measured on %s)" % (dst, t["csrc_sha"], t["measured_on_csrc_sha"]))


if __name__ == "__main__":
  main()
