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


# ---- no scratch on the kernels a BASELINE step launches --------------------------------------------
# Allow-list = every kernel named in the committed step sequences of the latest round (profiles/rNN/step_sequence*.txt:
# the replayed steps of DS-STGCN, ST-GCN and CTR-GCN in launch order, written by tools/step_sequence.py from a rocprofv3
# kernel trace).  Checked against the code objects of the library that build() just produced (tools/codeobj_report.py):
# `.vgpr_spill_count` of the metadata notes AND the scratch_* instructions of the disassembly.  The notes' spill count also
# counts VGPR -> AGPR moves (k_tspw2<64> runs one wave per SIMD with 512 registers: its 42 "spills" are v_accvgpr moves, the
# disassembly has no scratch access), so the instruction count is the bar and the spill count is pinned where it is not 0.
PINNED = {
    # kernel: (max scratch instructions, max vgpr_spill_count) — a ratchet, not a licence: lower when a kernel improves
    'k_tspw2<64>': (0, 42),                 # AGPR moves only (one wave per SIMD)
    'k_tms_dgrad<5, 4, 2>': (22, 7),        # CTR-GCN's stride-2 5-tap data gradient: 7 loop-invariant values, outside its matrix loops
}


def _step_sequence_kernels():
    import glob
    rounds = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]')))
    for rdir in reversed(rounds):
        files = sorted(glob.glob(os.path.join(rdir, 'step_sequence*.txt')))
        if files:
            names = {}
            for f in files:
                for line in open(f):
                    m = re.match(r'\s*[\d.]+ us\s+(k_\S.*)$', line)
                    if m:
                        names.setdefault(m.group(1).strip(), os.path.basename(f))
            return rdir, names
    raise AssertionError('no profiles/rNN/step_sequence*.txt committed')


def test_no_scratch_on_the_kernels_of_the_baseline_steps():
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import codeobj_report
    native.build()
    ks = codeobj_report.kernels(native.LIB_PATH)
    rdir, names = _step_sequence_kernels()
    assert len(names) >= 100, (rdir, len(names))
    bad = []
    for name, where in sorted(names.items()):
        assert name in ks, f'{name} ({where}) is in a committed step sequence but not in the library: stale sequence?'
        k = ks[name]
        lim_i, lim_s = PINNED.get(name, (0, 0))
        if k.get('scratch_instructions', 0) > lim_i or k.get('vgpr_spill_count', 0) > lim_s:
            bad.append((name, where, k.get('scratch_instructions', 0), k.get('vgpr_spill_count', 0)))
    assert not bad, f'kernels of a BASELINE step with scratch / spilled registers: {bad}'
    for name, (lim_i, lim_s) in PINNED.items():          # the pins must not outlive their reason
        if name in ks:
            k = ks[name]
            assert k.get('scratch_instructions', 0) == lim_i and k.get('vgpr_spill_count', 0) == lim_s, \
                f'{name} changed ({k.get("scratch_instructions")}, {k.get("vgpr_spill_count")}): tighten its pin'
