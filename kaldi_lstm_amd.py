"""Import shim: the package directory is named `kaldi-lstm_amd/` (not a valid Python
identifier), so `import kaldi_lstm_amd` loads it from there under this module name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kaldi-lstm_amd")
_spec = importlib.util.spec_from_file_location(
    "kaldi_lstm_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["kaldi_lstm_amd"] = _mod
_spec.loader.exec_module(_mod)
