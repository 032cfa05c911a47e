"""CPU checks of the C-ABI library: it loads without a GPU, exports every symbol that
include/klstm.h declares, and refuses (loudly) to create an engine when no device exists."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "klstm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(klstm_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import kaldi_lstm_amd as k
    lib = k.load_library()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libklstm.so does not export {n}"
    assert b"gfx950" in lib.klstm_version()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    import kaldi_lstm_amd as k
    with pytest.raises(k.KlstmError) as ei:
        k.Engine(40, 800, 512, 4)
    assert ei.value.status == 5          # KLSTM_ERR_NOGPU
    assert "no CPU fallback" in str(ei.value)


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "kaldi-lstm_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dp, fn), errors="ignore").read()
                assert not re.search(r"import\s+oracle|from\s+oracle|#include[^\n]*oracle|liblstmp_oracle|lstmp_oracle_",
                                     src), f"{fn} references the oracle"
