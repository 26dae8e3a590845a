"""nano-vllm-candle_amd: MI355X-native (gfx950) Qwen3 forward path behind the reference's
ModelRunner / layers surface.  All arithmetic runs in libnvllm_amd.so (hand-written HIP); this
package is the host-side mirror of the reference interface.  Import name: nano_vllm_candle_amd."""
from . import _lib  # noqa: F401
from .context import Context, DeviceArray  # noqa: F401
from .engine import (LLMEngine, ModelRunner, Qwen3ModelRunner, SamplingParams, Scheduler, SchedulerConfig,  # noqa: F401
                     Sequence)
from .qwen3 import Qwen3Config, Qwen3ForCausalLM  # noqa: F401
from .tp import TPConfig, get_tp  # noqa: F401

__all__ = ["Context", "DeviceArray", "Qwen3Config", "Qwen3ForCausalLM", "Qwen3ModelRunner", "LLMEngine", "Scheduler",
           "SchedulerConfig", "SamplingParams", "Sequence", "ModelRunner", "TPConfig", "get_tp"]
