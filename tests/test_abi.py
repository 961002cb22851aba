"""CPU suite: the C-ABI library loads and exports every symbol include/zkg.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_in_header():
    src = open(os.path.join(ROOT, "include", "zkg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zkg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_header_symbols():
    from zklaim_amd import build
    so = build.build()
    lib = ctypes.CDLL(so)
    names = declared_in_header()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libzkg.so does not export {n}"
    import zklaim_amd
    assert sorted(zklaim_amd.DECLARED_SYMBOLS) == names
    for n in zklaim_amd.COMPAT_SYMBOLS:                      # the reference's own seam names (zklaim.h:257-259)
        assert hasattr(lib, n), n


def test_no_cpu_fallback():
    """Without a GPU every compute entry point must fail loudly (never route through a CPU path)."""
    import numpy as np
    import zklaim_amd
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(zklaim_amd.ZkgError):
        zklaim_amd.init(0)
    with pytest.raises(zklaim_amd.ZkgError):
        zklaim_amd.ntt(np.zeros((4, 4), np.uint64))
    with pytest.raises(zklaim_amd.ZkgError):
        zklaim_amd.msm_g1(np.zeros((1, 8), np.uint64), np.zeros((1, 4), np.uint64))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "zklaim_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "zkoracle" not in txt and "pyref" not in txt and "oracle/" not in txt, f


def test_domain_rule_matches_the_definition_without_a_gpu():
    """zkg_evaluation_domain_size is pure host code: libfqfft's get_evaluation_domain rule (basic_radix2 / step_radix2) against the
    table tests/golden/step_domain.json holds (from oracle/pyref.py's restatement of the rule)"""
    import zklaim_amd as zkg
    from util import golden
    for k, kind, m in golden("step_domain.json")["rule"]:
        assert zkg.evaluation_domain_size(k) == (m, kind == "step"), k
    for bad in (0, 1, (1 << 28) + 1):
        with pytest.raises(zkg.ZkgError):
            zkg.evaluation_domain_size(bad)
