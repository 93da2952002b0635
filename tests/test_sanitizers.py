"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU builds (GPU sanitizers are not available on the pool):
the oracle's C restatement and the CPU model of the GPU pipeline, which compiles the product's lane_math.h -- so the
bit tricks (shifts, carries, popcounts) that run on the device are checked for undefined behaviour here."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.timeout(600)
def test_asan_ubsan_clean(tmp_path):
    model_so, oracle_so = str(tmp_path / "libfused_model_san.so"), str(tmp_path / "liblatok_oracle_san.so")
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared", "-fPIC"]
    subprocess.check_call(["g++", "-std=c++17", "-w", *san, "-I" + os.path.join(ROOT, "latok_amd", "csrc"),
                           os.path.join(ROOT, "oracle", "fused_model.cpp"), "-o", model_so])
    subprocess.check_call(["gcc", "-std=c11", *san, os.path.join(ROOT, "oracle", "latok_oracle.c"), "-o", oracle_so])
    libs = [subprocess.check_output(["gcc", "-print-file-name=" + n], text=True).strip() for n in ("libasan.so", "libubsan.so")]
    env = dict(os.environ, LD_PRELOAD=" ".join(libs), ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "sanitized_model_run.py"), model_so, oracle_so],
                         env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0 and "sanitized run ok" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
