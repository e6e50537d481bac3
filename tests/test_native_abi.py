"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/dsgcn.h declares (no compute)."""
import ctypes
import os
import re
import subprocess

from dsgcn_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(ROOT, 'include', 'dsgcn.h')) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    return sorted(set(re.findall(r'\b(?:int|size_t)\s+(dsgcn_\w+)\s*\(', text)))


def test_library_builds_and_exports_header_symbols():
    path = native.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/dsgcn.h but not exported'
    assert native.lib().dsgcn_version() >= 100
    # the product library exports exactly the header: no measurement-only entry points (those live in
    # libdsgcn_lab.so, include/dsgcn_lab.h) and no internal cross-file helpers
    out = subprocess.run(['nm', '-D', '--defined-only', path], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r' T (dsgcn_\w+)', out)))
    assert exported == names, (set(exported) ^ set(names))
    for n in native.LAB_SIGNATURES:
        assert not hasattr(lib, n), f'{n} is a lab entry point and must not be in the product ABI'


def test_python_binding_covers_header():
    assert set(declared_symbols()) <= set(native.SIGNATURES), set(declared_symbols()) - set(native.SIGNATURES)


def test_argument_rejection_without_gpu():
    lib = native.lib()
    # NULL pointers / bad sizes are rejected before any launch (DSGCN_EINVAL = -1)
    assert lib.dsgcn_aggregate_fwd(None, None, None, 1, None, None, 1, 1, 1, 25, None) == -1
    assert lib.dsgcn_pwconv_fwd(None, None, None, None, None, None, 0, None, None, None, None, None, 1, 1, 1, 1, 25, 1, 0,
                                0, None) == -1
    assert lib.dsgcn_colsum(None, 1, 1, None, None) == -1
