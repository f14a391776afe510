"""CPU: the C-ABI library loads, exports every symbol include/dlsg.h declares, and the ctypes mirror of every
argument struct has the size the compiler gave it (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from dlsg_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, 'include', 'dlsg.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(?:int|int64_t)\s+(dlsg_\w+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    lib = hip.load_library()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(hip.SYMBOLS) == names
    assert lib.dlsg_abi_version() == hip.ABI_VERSION == 8


def test_struct_layouts_match_the_compiler():
    lib = hip.load_library()
    for i, st in enumerate(hip.STRUCTS):
        assert lib.dlsg_struct_size(i) == ctypes.sizeof(st), st.__name__
    assert lib.dlsg_struct_size(99) == -1


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError):
        hip.HipOps()
    import dlsg_amd
    from helpers import small_args
    net = dlsg_amd.CapGnnModel(small_args(), dlsg_amd.make_vocab(50))
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 26, 112), torch.zeros(1, 26, 16, 32), torch.zeros(1, 26, dtype=torch.long))


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md is the reference-side binding guide: every symbol include/dlsg.h declares must appear in it."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, 'include', 'dlsg.h')).read()
    doc = open(os.path.join(root, 'INTEGRATION.md')).read()
    names = sorted(set(re.findall(r'\b(dlsg_[a-z0-9_]+)\s*\(', header)))
    missing = []
    for n in names:
        stem = re.sub(r'_(fwd|bwd)$', '', n)
        if n not in doc and (stem + '_fwd/bwd') not in doc:
            missing.append(n)
    assert not missing, missing
