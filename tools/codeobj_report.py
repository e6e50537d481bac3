#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of libdsgcn.so, read from the gfx950 code objects' metadata notes.

    python tools/codeobj_report.py [--lib PATH] [--spills] [--json]

The library is a host ELF with one offload bundle per translation unit; ``llvm-objdump --offloading`` unpacks them (into
a temporary directory — it writes next to its input) and ``llvm-readelf --notes`` prints the AMDGPU metadata
(`.vgpr_count`, `.vgpr_spill_count`, `.private_segment_fixed_size`, `.group_segment_fixed_size` per kernel).
tests/test_native_abi.py uses ``kernels()`` to keep every kernel a BASELINE step launches free of scratch."""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
DEFAULT_LIB = os.path.join(ROOT, 'ds-gcn_amd', 'lib', 'libdsgcn.so')
FIELDS = ('vgpr_count', 'agpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count', 'private_segment_fixed_size',
          'group_segment_fixed_size', 'max_flat_workgroup_size')


def short_name(demangled):
    """'(anonymous namespace)::k_pwg3<2, 1, 2>(Args...)' -> 'k_pwg3<2, 1, 2>' (what rocprof's kernel trace prints)."""
    s = demangled.replace('(anonymous namespace)::', '')
    s = re.sub(r'^void ', '', s)
    depth = 0
    for i, ch in enumerate(s):
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            return s[:i]
    return s


def kernels(lib=DEFAULT_LIB):
    """-> {short name: {field: int}} for every kernel of every code object in the library."""
    tmp = tempfile.mkdtemp(prefix='dsgcn_co_')
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', local], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = {}
        for f in sorted(os.listdir(tmp)):
            if 'hipv4-amdgcn' not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', os.path.join(tmp, f)], check=True,
                                   capture_output=True, text=True).stdout
            cur = None
            for line in notes.splitlines():
                m = re.match(r'\s*-?\s*\.(\w+):\s*(.*)$', line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip().strip("'")
                if key == 'name' and val.startswith('_Z') and not val.endswith('.kd'):
                    cur = {}
                    out[val] = cur
                elif cur is not None and key in FIELDS:
                    cur[key] = int(val)
            # scratch INSTRUCTIONS per kernel (the spill count of the notes also counts VGPR -> AGPR moves, which touch no
            # memory; a private segment can be reserved without a single access): what actually goes to scratch memory
            dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', os.path.join(tmp, f)], check=True,
                                 capture_output=True, text=True).stdout
            sym = None
            for line in dis.splitlines():
                m = re.match(r'^[0-9a-f]+ <(\w+)>:', line)
                if m:
                    sym = m.group(1)
                    if sym in out:
                        out[sym].setdefault('scratch_instructions', 0)
                elif sym in out and 'scratch_' in line:
                    out[sym]['scratch_instructions'] += 1
        names = list(out)
        dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True,
                             check=True).stdout.splitlines()
        return {short_name(d): out[n] for n, d in zip(names, dem)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', default=DEFAULT_LIB)
    ap.add_argument('--spills', action='store_true', help='only kernels with spilled registers or a private segment')
    ap.add_argument('--json', action='store_true')
    a = ap.parse_args()
    ks = kernels(a.lib)
    if a.spills:
        ks = {k: v for k, v in ks.items() if v.get('vgpr_spill_count', 0) or v.get('private_segment_fixed_size', 0) or
              v.get('scratch_instructions', 0)}
    if a.json:
        json.dump(ks, sys.stdout, indent=1, sort_keys=True)
        return
    print(f'{len(ks)} kernels')
    for k in sorted(ks):
        v = ks[k]
        print(f"{v.get('vgpr_count', 0):4d} v {v.get('agpr_count', 0) or 0:4d} a  spill {v.get('vgpr_spill_count', 0):4d}  "
              f"scratch {v.get('private_segment_fixed_size', 0):5d} B / {v.get('scratch_instructions', 0):3d} instr  {k}")


if __name__ == '__main__':
    main()
