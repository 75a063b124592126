"""The drop-in boundary used from plain C: tests/c_abi/consumer.c is compiled with gcc -std=c99
against include/pclseg.h and linked with libpclseg.so — no Python, torch or C++ on the caller side."""
import os
import subprocess

import pytest

from conftest import SIM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "pclsegmentation_amd")
LIBNAME = "libpclseg.so"
if SIM:      # simulator mode (tests/simlib.py): the same C program, linked with the x86-64 build of the same sources
  LIBDIR, LIBNAME = os.path.split(os.environ["PCLSEG_LIB"])


@pytest.fixture(scope="module")
def consumer(tmp_path_factory):
  if SIM in ("asan", "ubsan"):
    pytest.skip("the sanitizer builds of the simulator need their runtime linked into the C program")
  if not os.path.isfile(os.path.join(LIBDIR, LIBNAME)):
    pytest.fail("libpclseg.so is not built (run `make`)")
  exe = str(tmp_path_factory.mktemp("c_abi") / "consumer")
  subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O1",
                         "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "consumer.c"),
                         "-L", LIBDIR, "-l:" + LIBNAME, "-lm", "-Wl,-rpath," + LIBDIR, "-o", exe])
  return exe


def test_c_consumer_plan_and_error_codes(consumer):
  out = subprocess.run([consumer, "plan"], capture_output=True, text=True, timeout=120)
  assert out.returncode == 0, out.stderr
  assert "plan ok" in out.stdout


@pytest.mark.gpu
def test_c_consumer_forward(consumer):
  out = subprocess.run([consumer, "forward"], capture_output=True, text=True, timeout=300)
  assert out.returncode == 0, out.stderr + out.stdout
  assert "forward ok" in out.stdout
