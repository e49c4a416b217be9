"""CPU-only: the C-ABI library is built, loads, and exports every symbol include/ctrlhair_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'ctrlhair_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(ch_[a-z_0-9]+)\s*\(', txt)))


def test_header_symbols_exported():
    from ctrlhair_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    l = lib.load()
    names = _declared()
    assert 'ch_sean_generate' in names and len(names) >= 10
    for n in names:
        assert hasattr(l, n), f'{n} declared in include/ctrlhair_hip.h but not exported'
    assert set(names) == set(lib.SYMBOLS), 'ctypes prototype table out of sync with the header'
    assert l.ch_abi_version() == 1


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from ctrlhair_amd import lib
    with pytest.raises(RuntimeError):
        lib.Handle(0)
