"""CPU: the C-ABI library loads and exports every symbol include/nvllm_amd.h declares; failures are loud."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nvllm_amd.h")              # the boundary a maintainer binds
DEBUG_HEADER = os.path.join(ROOT, "include", "nvllm_amd_debug.h")  # profiling / parity-debug / tuning exports


def header_functions(paths=(HEADER, DEBUG_HEADER)):
    names = set()
    for path in paths:
        src = open(path).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names.update(re.findall(r"\b(nvllm_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_boundary_header_holds_no_debug_or_tuning_exports():
    names = header_functions((HEADER,))
    assert not [n for n in names if "_debug_" in n or "_profile_" in n], names


def test_library_exports_every_header_symbol():
    import ctypes

    import nano_vllm_candle_amd as pkg

    lib = ctypes.CDLL(pkg._lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nvllm_amd.h but not exported"


def test_python_binding_covers_the_header():
    import nano_vllm_candle_amd as pkg

    assert sorted(pkg._lib.EXPORTED_SYMBOLS) == header_functions()


def test_library_is_in_tree_and_no_cpu_fallback():
    import nano_vllm_candle_amd as pkg

    assert os.path.dirname(pkg._lib.LIB_PATH).endswith("nano-vllm-candle_amd")
    # the product never imports the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "nano-vllm-candle_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "qwen3_oracle" not in txt, f


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_context_creation_fails_loudly_without_a_gpu():
    import nano_vllm_candle_amd as pkg

    with pytest.raises(pkg._lib.NvllmError):
        pkg.Context(0)


def test_missing_library_raises(monkeypatch, tmp_path):
    import nano_vllm_candle_amd as pkg

    monkeypatch.setattr(pkg._lib, "_lib", None)
    monkeypatch.setattr(pkg._lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(pkg._lib.NvllmLibraryMissing):
        pkg._lib.lib()
