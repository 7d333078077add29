"""The C-ABI library loads (no GPU needed) and exports every symbol that
include/genpc_hip.h declares; the ctypes table in genpc_amd/_lib.py lists exactly
those symbols with matching arity.  No compute call is made here."""
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def header_prototypes():
    txt = open(os.path.join(ROOT, "include", "genpc_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|float|const char \*)\s*(genpc_\w+)\s*\(([^)]*)\)\s*;", txt):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args == "void" else len(args.split(","))
    return protos


def test_header_has_the_path():
    p = header_prototypes()
    for name in ("genpc_chamfer_forward", "genpc_chamfer_backward", "genpc_emd_forward", "genpc_emd_backward",
                 "genpc_nm_distance"):
        assert name in p
    assert p["genpc_emd_forward"] == 20        # b,n,m + 14 buffers + eps, iters + stream
    assert p["genpc_chamfer_forward"] == 10
    assert p["genpc_chamfer_backward"] == 12


def test_library_exports_every_declared_symbol():
    from genpc_amd import build
    build.build(verbose=False)
    from genpc_amd import _lib
    protos = header_prototypes()
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    for name, nargs in protos.items():
        fn = getattr(_lib.lib, name)           # AttributeError if not exported
        assert len(_lib.SIGNATURES[name][1]) == nargs, name
        assert fn is not None
    assert _lib.lib.genpc_abi_version() == _lib.ABI_VERSION


def test_arith_mode_switch():
    from genpc_amd import _lib
    prev = _lib.lib.genpc_set_arith(0)
    assert _lib.lib.genpc_get_arith() == 0
    _lib.lib.genpc_set_arith(prev)
    assert _lib.lib.genpc_get_arith() == prev


def test_cpu_tensors_are_rejected_loudly():
    import torch
    from genpc_amd.loss_functions import chamfer_3DDist, emdModule
    with pytest.raises(RuntimeError, match="GPU tensors only"):
        chamfer_3DDist()(torch.zeros(1, 8, 3), torch.zeros(1, 8, 3))
    with pytest.raises(RuntimeError, match="GPU tensors only"):
        emdModule()(torch.zeros(1, 256, 3), torch.zeros(1, 256, 3), 0.005, 2)


def test_reference_api_names():
    import loss_functions
    from genpc_amd.utils.loss_util import Completionloss
    assert hasattr(loss_functions, "chamfer_3DDist") and hasattr(loss_functions, "emdModule")
    for m in ("chamfer_l1", "chamfer_l2", "chamfer_partial_l1", "chamfer_partial_l2", "emd_loss", "get_loss"):
        assert hasattr(Completionloss, m)
    with pytest.raises(Exception):
        Completionloss("nope")
