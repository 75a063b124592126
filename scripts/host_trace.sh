#!/bin/bash
# copy + kernel timeline of the host-boundary modes: host_trace.sh sync|async|device
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
M=${1:-async}
rm -rf gpurun_out/ht_$M; mkdir -p gpurun_out
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/ht_$M -- python3 scripts/host_trace.py $M 8 > /dev/null 2>&1
python3 - $M <<'PY'
import csv, glob, sys
m = sys.argv[1]
kt = glob.glob("gpurun_out/ht_%s/**/*kernel_trace.csv" % m, recursive=True)[0]
ct = glob.glob("gpurun_out/ht_%s/**/*memory_copy_trace.csv" % m, recursive=True)[0]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(kt)) if "pclseg" in r["Kernel_Name"]]
cs = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"], 1 << 20) for r in csv.DictReader(open(ct))
      if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 20000]
ks.sort(); cs.sort()
# steady-state window: the middle half of the kernel span
t0, t1 = ks[0][0], ks[-1][1]
a, b = t0 + (t1 - t0) // 4, t1 - (t1 - t0) // 4
def busy(iv):
  iv = sorted((max(s, a), min(e, b)) for s, e in iv if e > a and s < b)
  tot, cur_s, cur_e = 0, None, None
  for s, e in iv:
    if cur_e is None or s > cur_e:
      if cur_e is not None: tot += cur_e - cur_s
      cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
  if cur_e is not None: tot += cur_e - cur_s
  return tot / (b - a)
print("mode %s: span %.2f ms, window %.2f ms" % (m, (t1 - t0) / 1e6, (b - a) / 1e6))
print("  any kernel running: %.1f %% of the window; mean kernels in flight %.2f" % (
  100 * busy([(s, e) for s, e, _, _ in ks]), sum(min(e, b) - max(s, a) for s, e, _, _ in ks if e > a and s < b) / (b - a)))
for q in sorted(set(k[3] for k in ks)):
  print("    queue %s: busy %.1f %%" % (q, 100 * busy([(s, e) for s, e, _, qq in ks if qq == q])))
for d in sorted(set(c[2] for c in cs)):
  sel = [c for c in cs if c[2] == d and c[3] > 100000]
  if not sel: continue
  dur = sorted((e - s) / 1e3 for s, e, _, _ in sel)
  gbs = sorted(sz / (e - s) for s, e, _, sz in sel)
  print("  %-28s n=%d  busy %.1f %%  median %.0f us  max %.0f us" % (d, len(sel), 100 * busy([(s, e) for s, e, _, _ in sel]), dur[len(dur) // 2], dur[-1]))
norm = [k for k in ks if "normalize" in k[2] and a <= k[0] <= b]
print("  micro-batches started in the window: %d -> %.0f scans/s (at 3.56 scans per micro-batch)" % (len(norm), len(norm) * 32 / 9 / ((b - a) / 1e9)))
PY
