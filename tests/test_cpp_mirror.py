"""include/eccx.hpp, the C++ mirror of eccoxide's Point/Scalar surface: it compiles against
the C ABI (CPU), and its batched mul_base / `*` agree on the GPU."""
import os
import subprocess

import pytest

from tests.oracle_lib import ROOT

CPP = os.path.join(ROOT, "tests", "cpp")


def test_mirror_header_compiles():
    subprocess.check_call(["make", "-C", CPP, "-s"])
    assert os.path.exists(os.path.join(CPP, "mirror_check"))


@pytest.mark.gpu
def test_mirror_runs_on_gpu():
    subprocess.check_call(["make", "-C", CPP, "-s"])
    r = subprocess.run([os.path.join(CPP, "mirror_check")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mirror_check ok" in r.stdout
