"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol the
header declares, and refuses to run without a device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from gfalign_amd import build as gbuild
from gfalign_amd import scorer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    gbuild.build_scorer()
    return scorer.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gfalign_scorer.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gfal_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(scorer.EXPORTS)


def test_every_declared_symbol_is_exported(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_abi_version_and_strerror(lib):
    assert lib.gfal_abi_version() == 4
    assert lib.gfal_strerror(0) == b"ok"
    assert b"device" in lib.gfal_strerror(-3)


def test_binary_matches_the_sources(lib):
    """build.py stamps every artefact with the hash of its sources: the library
    that loads here (and travels to the GPU box) is the one HEAD describes."""
    assert lib.gfal_build_id().decode() == gbuild.scorer_build_id()
    cli = gbuild.build_cli()
    import subprocess
    out = subprocess.run([cli, "--build-id"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == gbuild.cli_build_id()


def test_argument_validation_needs_no_device(lib):
    h = ctypes.c_void_p()
    off = np.array([1, 2], np.int32)   # aln_off[0] != 0
    st = np.array([0, 0], np.int32)
    rc = lib.gfal_scorer_create(off.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                st.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                1, 4, 0, ctypes.byref(h))
    assert rc == -1 and not h


def test_no_cpu_fallback(lib):
    """Without a HIP device the product path must fail loudly."""
    if scorer.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(scorer.ScorerError) as e:
        scorer.Scorer([0, 2], [0, 2], 4)
    assert e.value.code == -3


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "gfalign_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), f
