"""Host-side logic that needs no GPU: config surface, registries, weight inventories, the C
ABI (loads, exports every declared symbol, plans graphs, fails loudly without a device)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import pclsegmentation_amd as P
from pclsegmentation_amd import configs as C
from pclsegmentation_amd import engine as E
from pclsegmentation_amd.nets import spec as S
from pclsegmentation_amd.nets import weights as Wt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_configs_match_reference_constants():
  """tests/golden/configs.json was dumped from the reference's own config functions."""
  g = json.load(open(os.path.join(ROOT, "tests", "golden", "configs.json")))
  assert set(g) == set(P.config_map)
  for key, ent in g.items():
    mc = P.config_map[key]()
    assert P.config_map[key].__name__ == ent["function"]
    ref = ent["fields"]
    assert set(ref) == set(mc)
    for k, v in ref.items():
      m = mc[k]
      if isinstance(v, dict) and v.get("__ndarray__"):
        assert str(m.dtype) == v["dtype"] and list(m.shape) == v["shape"], (key, k)
        assert np.array_equal(m.astype(np.float64).ravel(), np.array(v["data"])), (key, k)
      else:
        assert m == v and type(m) == type(v), (key, k)


def test_registry_keys_and_errors():
  assert set(P.model_map) == {"squeezesegv2", "darknet53", "darknet21"}
  mc, model = P.load_model_config("SqueezeSegV2", "SqueezeSegV2Kitti")     # case-insensitive
  assert mc.NUM_CLASS == 20 and mc.CLASSES.index("None") == 0 and model.arch_name() == "squeezesegv2"
  mc, model = P.load_model_config("darknet21", "darknet21", width=1024)
  assert (mc.ZENITH_LEVEL, mc.AZIMUTH_LEVEL) == (32, 1024) and model.num_blocks == [1, 1, 2, 2, 1]
  with pytest.raises(KeyError):
    P.load_model_config("resnet", "darknet21")
  with pytest.raises(KeyError):
    P.load_model_config("darknet21", "nope")


def test_weight_inventories():
  assert S.num_params(S.squeezesegv2_spec(20)) == 937080 and len(S.squeezesegv2_spec(20)) == 274
  assert S.num_params(S.squeezesegv2_spec(11)) == 931887
  assert S.num_params(S.darknet_spec(20, 53)) == 53042740
  assert S.num_params(S.darknet_spec(11, 53)) == 53040139
  assert S.num_params(S.darknet_spec(11, 21)) == 27352331
  sp = {w.path: w.shape for w in S.squeezesegv2_spec(20)}
  assert sp["fire10/upconv/kernel"] == (1, 4, 64, 64) and sp["cam2/squeeze/kernel"] == (1, 1, 128, 8)
  assert sp["conv14/kernel"] == (3, 3, 64, 20) and "fire10/upconv_bn/gamma" not in sp
  dn = {w.path: w.shape for w in S.darknet_spec(11, 53)}
  assert dn["enc3/residual_7/conv2/kernel"] == (3, 3, 128, 256) and "enc1/conv1/bias" not in dn
  assert dn["dec5/conv1/bias"] == (512,) and dn["dec4/upconv1/kernel"] == (1, 4, 256, 512)
  assert dn["dec5/block/conv1/kernel"] == (1, 1, 512, 1024)


def test_weight_file_round_trip(tmp_path):
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
  model.init_weights(1)
  path = str(tmp_path / "m.npz")
  model.save(path)
  m2 = P.load_model(path)
  assert m2.arch_name() == "squeezesegv2" and m2.mc.CLASSES == mc.CLASSES
  assert all(np.array_equal(model.weights[k], m2.weights[k]) for k in model.weights)
  bad = dict(model.weights)
  bad["conv1/kernel"] = np.zeros((3, 3, 5, 64), np.float32)
  with pytest.raises(ValueError):
    model.set_weights(bad)
  del bad["conv1/kernel"]
  with pytest.raises(ValueError):
    Wt.check_weights(model.weight_spec(), bad)


def test_library_exports_every_declared_symbol():
  lib = E.load_library()
  header = open(os.path.join(ROOT, "include", "pclseg.h")).read()
  declared = set(re.findall(r"\b(pclseg_[a-z0-9_]+)\s*\(", header))
  assert declared == set(E.EXPORTS)
  for name in declared:
    assert hasattr(lib, name), name
  assert lib.pclseg_version() == 300


PLAN_CASES = [
  # arch, config, H, W, params, GFLOP, ALG MB  (SURVEY.md §6 / BASELINE.md §2)
  ("squeezesegv2", C.SqueezeSegV2KittiConfig, 64, 2048, 937080, 26.089, 728.4),
  ("squeezesegv2", C.SqueezeSegV2Config, 32, 240, 931887, 1.449, 42.7),
  ("darknet53", C.Darknet53Kitti, 64, 2048, 53042740, 990.07, 1547.3),
  ("darknet21", C.Darknet21, 32, 1024, 27352331, 133.26, 227.4),
]


@pytest.mark.parametrize("arch,cfg,h,w,params,gflop,mb", PLAN_CASES, ids=[c[0] + "_%dx%d" % c[2:4] for c in PLAN_CASES])
def test_plan_reproduces_survey_figures(arch, cfg, h, w, params, gflop, mb):
  mc = cfg()
  d = E.make_desc(arch, h, w, mc.NUM_CLASS, mc.CLASSES.index("None"), mc.INPUT_MEAN, mc.INPUT_STD)
  p = E.plan(d)
  assert p["num_params"] == params
  assert abs(2 * p["alg_macs_per_scan"] / 1e9 - gflop) < 0.01
  assert abs(p["alg_bytes_per_scan"] / 1e6 - mb) < 0.1
  assert p["workspace_bytes"] > 0 and p["micro_batch"] >= 1


def test_plan_rejects_bad_descriptions():
  ok = dict(arch="squeezesegv2", height=64, width=2048, num_class=20, none_index=0, mean=[0] * 5, std=[1] * 5)
  E.plan(E.make_desc(**ok))
  for bad in (dict(width=2040), dict(height=0), dict(num_class=1), dict(none_index=20), dict(std=[1, 1, 0, 1, 1])):
    with pytest.raises(ValueError):
      E.plan(E.make_desc(**{**ok, **bad}))
  with pytest.raises(ValueError):
    E.plan(E.make_desc(**{**ok, "arch": "darknet21", "output_stride": 5}))
  assert "divisible" in E.load_library().pclseg_last_error(None).decode() or True


def test_no_cpu_fallback():
  """Without a GPU the product path must fail loudly, not compute on the CPU."""
  import torch
  if torch.cuda.is_available():
    pytest.skip("GPU present")
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
  model.init_weights()
  with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
    model([np.zeros((1, 32, 240, 6), np.float32), np.ones((1, 32, 240), bool)])


def test_product_code_never_imports_the_oracle():
  pkg = os.path.join(ROOT, "pclsegmentation_amd")
  for d, _, files in os.walk(pkg):
    for f in files:
      if f.endswith((".py", ".h", ".hip")):
        src = open(os.path.join(d, f)).read()
        assert "import oracle" not in src and "from oracle" not in src, f


def test_tf_exporter_walks_reference_attribute_paths(tmp_path):
  """tools/export_tf_weights.collect_weights reads a (stand-in) Keras object tree by the
  reference's attribute paths; the resulting file round-trips through load_model."""
  import types
  import pclsegmentation_amd as P
  from pclsegmentation_amd.tools.export_tf_weights import collect_weights

  class Var:                       # quacks like tf.Variable
    def __init__(self, a):
      self._a = a
    def numpy(self):
      return self._a

  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
  spec = model.weight_spec()
  from pclsegmentation_amd.nets.weights import synthetic_weights
  want = synthetic_weights(spec, seed=7)
  root = types.SimpleNamespace()
  for w in spec:
    *attrs, leaf = w.path.split("/")
    obj = root
    for a in attrs:
      if not hasattr(obj, a):
        setattr(obj, a, types.SimpleNamespace())
      obj = getattr(obj, a)
    setattr(obj, leaf, Var(want[w.path].astype(np.float64)))     # dtype is normalised to float32
  got = collect_weights(root, spec)
  assert set(got) == set(want) and all(np.array_equal(got[k], want[k]) for k in want)
  assert all(v.dtype == np.float32 for v in got.values())
  model.set_weights(got)
  model.save(str(tmp_path / "m.npz"))
  back = P.load_model(str(tmp_path / "m.npz"))
  assert back.arch_name() == "squeezesegv2" and np.array_equal(back.weights["conv14/kernel"], want["conv14/kernel"])
  del root.fire9.expand3x3_bn.moving_variance
  with pytest.raises(ValueError, match="fire9.expand3x3_bn.moving_variance"):
    collect_weights(root, spec)


def test_engine_rejects_mistyped_or_short_buffers():
  """The C ABI reads raw addresses: float64 / strided / short / wrong-side buffers must be refused in
  Python (ValueError) before they reach it.  Needs no GPU: the check runs before the call."""
  from pclsegmentation_amd.engine import _checked, MEM_HOST, MEM_DEVICE
  import torch
  ok = np.zeros((2, 4, 8, 5), np.float32)
  assert _checked(ok, np.float32, ok.size, "scans", MEM_HOST) is ok
  for bad in (ok.astype(np.float64), ok[:, :, ::2], ok[:1]):
    with pytest.raises(ValueError):
      _checked(bad, np.float32, ok.size, "scans", MEM_HOST)
  with pytest.raises(ValueError):
    _checked(ok, np.float32, ok.size, "scans", MEM_DEVICE)          # host array, device call
  t = torch.zeros((2, 4, 8), dtype=torch.int32)
  assert _checked(t, np.int32, t.numel(), "preds", MEM_HOST) is t
  with pytest.raises(ValueError):
    _checked(t.to(torch.int64), np.int32, t.numel(), "preds", MEM_HOST)
  with pytest.raises(ValueError):
    _checked(t.transpose(0, 2), np.int32, t.numel(), "preds", MEM_HOST)
  with pytest.raises(ValueError):
    _checked(t, np.int32, t.numel(), "preds", MEM_DEVICE)


@pytest.mark.parametrize("kh,cin,cout,sigma", [
  (3, 1024, 64, 0.0147),   # Darknet deep layer: He-scaled weights, fan-in 9*1024
  (3, 64, 32, 0.059),      # fan-in 576
  (1, 512, 64, 0.0625),    # a FIRE squeeze
  (3, 64, 20, 1e-4),       # a trained layer with very small weights
  (1, 32, 16, 300.0),      # and one with large ones (BatchNorm gamma/sqrt(var) can do that)
])
def test_split_f16_weights_keep_22_bits(kh, cin, cout, sigma):
  """hi = f16(w), lo = f16(w - hi) falls into f16 SUBNORMALS for |w| < 2^-3
  (p99 relative reconstruction error 8e-5 on sigma = 0.0147 weights).  With the power-of-two
  pre-scale per 16-output-channel tile the fragments the matrix cores multiply by reproduce every weight to
  2^-22 relative, except those below 2^-15 of their tile's maximum (absolute error 2^-37 of it)."""
  rng = np.random.default_rng(7)
  k = (rng.standard_normal((kh, kh, cin, cout)) * sigma).astype(np.float32)
  k[0, 0, 0, :4] = 0.0                       # exact zeros stay exact
  recon, exps = E.op_split_f16_roundtrip(k)
  assert recon.shape == k.shape and np.all(recon[0, 0, 0, :4] == 0.0)
  kd = k.astype(np.float64)
  cmax = np.abs(kd).reshape(-1, cout).max(0)
  tiles = [slice(t, min(t + 16, cout)) for t in range(0, cout, 16)]
  tmax = np.concatenate([np.full(sl.stop - sl.start, cmax[sl].max()) for sl in tiles])   # per 16-cout tile
  assert all(len(set(exps[sl].tolist())) == 1 for sl in tiles), "one exponent per 16-channel tile"
  scaled = tmax * np.exp2(exps.astype(np.float64))
  assert np.all((scaled >= 2.0 ** 12) & (scaled < 2.0 ** 13)), "tile maximum must land in [2^12, 2^13)"
  nz = kd != 0
  rel = np.abs(recon - kd)[nz] / np.abs(kd)[nz]
  assert np.percentile(rel, 99) <= 2e-7, np.percentile(rel, 99)
  assert (np.abs(recon - kd) <= np.maximum(np.abs(kd) * 2.0 ** -21, tmax * 2.0 ** -36)).all()
  # the unscaled split, emulated: this is what round 2 shipped
  hi = k.astype(np.float16)
  lo = (k - hi.astype(np.float32)).astype(np.float16)
  old = np.abs(hi.astype(np.float64) + lo.astype(np.float64) - kd)[nz] / np.abs(kd)[nz]
  if sigma < 0.02:
    assert np.percentile(old, 99) > 1e-5      # (documents the defect the scale removes)


@pytest.mark.parametrize("arch,cfg,h,w", [("squeezesegv2", C.SqueezeSegV2KittiConfig, 64, 2048),
                                         ("darknet21", C.Darknet21, 32, 1024), ("darknet53", C.Darknet53Kitti, 64, 2048)])
def test_plan_ops_lists_every_launch_with_its_macs(arch, cfg, h, w):
  """pclseg_plan_ops (profile labels): one line per kernel launch of a micro-batch, and the per-launch
  multiply-accumulates add up to the plan's ALG figure exactly — fused pieces (next squeeze, up-convolution,
  skip branch, head) are counted in the launch that computes them."""
  mc = cfg()
  d = E.make_desc(arch, h, w, mc.NUM_CLASS, mc.CLASSES.index("None"), mc.INPUT_MEAN, mc.INPUT_STD,
                  output_stride=mc.get("OUTPUT_STRIDE", 16))
  info = E.plan(d)
  ops = E.plan_op_macs(d)
  assert len(ops) == info["num_ops"] and E.plan_ops(d) == [n for n, _ in ops]
  assert sum(m for _, m in ops) == info["alg_macs_per_scan"]
  names = [n for n, _ in ops]
  if arch == "squeezesegv2":
    assert names[0] == "conv1" and names[-1] == "up+fire13/expand+conv14+head" and "cam2+fire3/squeeze" in names
    assert len(names) == 19                      # + the pre-processing launch = 20 per micro-batch
  else:
    assert names[-1] == "head+head" and "enc5/residual_0/conv2" in names
  # static launch resources (round 4: what decides which kernels can share a CU): every block fits a CU's 160 KB of
  # LDS, has 256 or 512 threads and at least one block per scan; the figures the three-lane analysis quotes
  res = E.plan_op_resources(d)
  assert [r[0] for r in res] == names and [r[1] for r in res] == [m for _, m in ops]
  for name, _, lds, threads, blocks in res:
    assert 0 <= lds <= 160 * 1024 and threads in (256, 512) and blocks >= 1, (name, lds, threads, blocks)
  if arch == "squeezesegv2":
    by = {r[0]: r for r in res}
    # shipped kernels: fire8/9's 136 KB partial-sum slab (one block per CU); the candidate build (make candidates,
    # -DPCLSEG_CAND_SLAB, not yet run on an MI355X) passes it through LDS in two halves: 72 KB, two blocks fit a CU
    slab = by["fire8/expand+fire9/squeeze"][2:]
    assert slab == (139264, 512, 128) or (E.DEBUG_LIB and slab == (73728, 512, 128)), slab
    assert by["up+fire13/expand+conv14+head"][2:] == (71424, 256, 1024)    # two 70 KB blocks per CU
    assert by["conv1"][4] == 512 and by["cam1"][3] == 512


_OOM_CHILD = r"""
import ctypes, os, resource, sys
import numpy as np
sys.path.insert(0, %r)
from pclsegmentation_amd import engine as E
lib = E.load_library()
vm = [int(l.split()[1]) for l in open("/proc/self/status") if l.startswith("VmSize")][0] * 1024
resource.setrlimit(resource.RLIMIT_AS, (vm + (512 << 20), vm + (512 << 20)))   # 512 MB of head room
kernel = np.zeros(16, np.float32)          # never read: the first allocation (2 x 2 GB of fold scales) fails
recon = np.zeros(16, np.float64)
rc = lib.pclseg_op_split_f16_roundtrip(E._ptr(kernel), 1, 1, 4, 1 << 28, E._ptr(recon), None)
msg = lib.pclseg_last_error(None).decode()
print("RC", rc, "|", msg)
# the process is alive and the library still works
rec, exps = E.op_split_f16_roundtrip(np.ones((1, 1, 4, 4), np.float32))
print("ALIVE", float(rec.sum()))
"""


def test_allocation_failure_is_a_status_not_an_abort():
  """include/pclseg.h: "no exceptions cross the boundary".  A std::bad_alloc inside the library (forced here
  with RLIMIT_AS and a fold of 2^28 output channels, in a child process) comes back as PCLSEG_ERR_OOM with a
  message; the process is not terminated and the next call works."""
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  r = subprocess.run([sys.executable, "-c", _OOM_CHILD % root], capture_output=True, text=True, timeout=300)
  assert r.returncode == 0, (r.returncode, r.stdout[-1000:], r.stderr[-2000:])
  line = [l for l in r.stdout.splitlines() if l.startswith("RC")][0]
  assert line.split()[1] == str(E.ERR_OOM) and "bad_alloc" in line, line
  assert "ALIVE 16.0" in r.stdout
  with pytest.raises(MemoryError):
    E.check(E.ERR_OOM)


def test_every_entry_point_is_behind_the_exception_barrier():
  """Source-level: each extern "C" definition with a body of more than one line is a function-try-block."""
  src = open(os.path.join(os.path.dirname(E.__file__), "csrc", "pclseg_api.hip")).read()
  body = src[src.index('extern "C" {'):src.index('}  // extern "C"')]
  defs = re.findall(r"^(?:int|void\*|const char\*) (pclseg_\w+)\([^;{]*\)( try)? \{$", body, re.M)
  assert len(defs) >= 28 and all(t for _, t in defs), [n for n, t in defs if not t]
  assert body.count("PCLSEG_CATCH(") + body.count("catch (...)") >= len(defs)


def test_binary_carries_the_hash_of_the_sources_it_was_built_from():
  """Makefile bakes sha256(csrc/* + include/pclseg.h)[:16] into the library; bench.py's csrc_sha() reads the
  same files in the same order.  (After editing a source without rebuilding this fails: rebuild.)"""
  import importlib.util
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
  bench = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(bench)
  assert re.fullmatch(r"[0-9a-f]{16}", E.build_sha()), E.build_sha()
  assert E.build_sha() == bench.csrc_sha()


def test_committed_traffic_figure_belongs_to_these_sources():
  """bench.py quotes roofline.traffic from the newest profiles/rNN_traffic.json only when its csrc sha is the loaded
  library's (attach_traffic); a source edit without `scripts/carry_traffic.py` (kernels unchanged) or a re-measurement
  (kernels changed) would silently turn the field into null on the next bench line."""
  import importlib.util
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  spec = importlib.util.spec_from_file_location("bench_mod3", os.path.join(root, "bench.py"))
  bench = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(bench)
  t = json.load(open(bench._latest_traffic_json()))
  assert t["csrc_sha"] == bench.csrc_sha(), (bench._latest_traffic_json(), t["csrc_sha"], bench.csrc_sha())
  assert set(t["workloads"]) >= {"ssv2_64x2048", "darknet53_64x2048", "darknet21_32x1024"}


def test_shipped_kernels_are_the_set_verified_on_an_mi355x():
  """The last hardware run of this repository's kernels was round 3's (GPUTEST_r03 / BENCH_r03: 127 tests green,
  6 549 scans/s).  Until an MI355X runs the suite again, the library that SHIPS (plain `make`) may only contain device
  kernels whose instruction streams are those of that tree: tests/golden/verified_kernels_r03.json holds a hash of
  every kernel's gfx950 instructions (scripts/kernel_isa_diff.py --write-manifest ad2e081).  Instantiations may be
  dropped (dispatch tables cut to what a plan selects), none may change or appear.  Kernel work that has not met the
  hardware lives behind -DPCLSEG_CAND (`make candidates`); this test is what keeps it there.  When a GPU run has
  verified a new kernel set, regenerate the manifest from that commit."""
  import importlib.util
  import shutil
  if not shutil.which("/opt/rocm/bin/hipcc"):
    pytest.skip("needs hipcc")
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  spec = importlib.util.spec_from_file_location("kernel_isa_diff", os.path.join(root, "scripts", "kernel_isa_diff.py"))
  kid = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(kid)
  want = json.load(open(os.path.join(root, "tests", "golden", "verified_kernels_r03.json")))
  hipcc = kid.subprocess.check_output([kid.HIPCC[0], "--version"], text=True).splitlines()[0]
  if hipcc != want["hipcc"]:
    pytest.skip("manifest was written with %s, this is %s" % (want["hipcc"], hipcc))
  from concurrent.futures import ThreadPoolExecutor
  with ThreadPoolExecutor(2) as pool:      # two hipcc device compiles side by side (about 40 s)
    f_have, f_cand = pool.submit(kid.manifest_of_tree), pool.submit(kid.manifest_of_tree, ["-DPCLSEG_CAND"])
    have, cand = f_have.result(), f_cand.result()
  assert len(have) >= 100
  new = sorted(k for k in have if k not in want["kernels"])
  changed = sorted(k for k in have if k in want["kernels"] and have[k] != want["kernels"][k])
  assert not new and not changed, "kernels that never ran on an MI355X in the shipped build: new %s changed %s" % (new[:3], changed[:3])
  # and the candidate switch really changes kernels (the manifest check is not vacuous)
  assert any(k not in want["kernels"] or cand[k] != want["kernels"][k] for k in cand)
