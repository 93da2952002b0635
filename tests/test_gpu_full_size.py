"""BASELINE configs[3] and configs[4] at their stated sizes on one GPU: 100 M strings (51.2 GB of
UTF-32 resident) and 10 K documents x 1 M chars (40 GB), with the size-independent property checks and sampled oracle
parity of tools/full_size_check.py.  Needs the 288 GB of an MI355X; skipped on a smaller device."""
import ctypes as C
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _hbm_bytes(lib):
    n_cu, hbm = C.c_int(0), C.c_int64(0)
    name = C.create_string_buffer(128)
    assert lib.latok_device_props(C.byref(n_cu), C.byref(hbm), name, 128) == 0
    return hbm.value


@pytest.mark.parametrize("workload", ["C4", "C5"])
def test_full_size_config(gpu, oracle, workload):
    if _hbm_bytes(gpu) < 200 * (1 << 30):
        pytest.skip("needs > 200 GB of HBM")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import full_size_check
    res = full_size_check.run(workload, verbose=False)
    assert res["ok"] and res["strings"] == {"C4": 100_000_000, "C5": 10_000}[workload]
    assert res["chars"] > 1.0e10 - 1 and res["boundaries"] > res["tokens"] > 0
