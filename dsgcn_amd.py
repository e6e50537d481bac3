"""Import shim: the package directory is named ``ds-gcn_amd`` (not a Python identifier), so
``import dsgcn_amd`` lands here and this file swaps itself for the real package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ds-gcn_amd')
_spec = importlib.util.spec_from_file_location(
    'dsgcn_amd', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules['dsgcn_amd'] = _pkg
_spec.loader.exec_module(_pkg)
