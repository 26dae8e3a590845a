"""Import shim: the package directory is named after the reference repo (`nano-vllm-candle_amd`, with
hyphens), which Python cannot import by name.  `import nano_vllm_candle_amd` loads that directory as
a regular package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nano-vllm-candle_amd")
_spec = importlib.util.spec_from_file_location("nano_vllm_candle_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nano_vllm_candle_amd"] = _mod
_spec.loader.exec_module(_mod)
