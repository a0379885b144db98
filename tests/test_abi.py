"""The C-ABI library loads without a GPU and exports every symbol include/smplpp_hip.h declares (no compute calls)."""
import ctypes as C
import os

import pytest


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build()
    from smplpp_amd import _lib

    return _lib


def test_every_declared_symbol_is_exported(built):
    L = built.load()
    names = built.declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_product_fails_loudly_without_gpu(built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert built.device_count() == 0
    with pytest.raises(built.SmplppError):
        built.require_gpu()
    from smplpp_amd import model_io
    from smplpp_amd.smpl import SMPL

    s = SMPL()
    s.setDevice("cuda:0")
    with pytest.raises(built.SmplppError):
        s.init(model_io.tiny_model(8))
    with pytest.raises(built.SmplppError):
        s.setDevice("cpu")


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under smplpp_amd/ may reference it."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "smplpp_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("# oracle", ""), os.path.join(dp, f)
