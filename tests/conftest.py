import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def oracle_0_6b(oracle):
    """(config, oracle model) of Qwen3-0.6B with synthetic weights, seed 0: 3 GB of f32 and several seconds of
    generator time, so every GPU parity test at the real shapes shares one"""
    import nano_vllm_candle_amd as pkg
    from tests.util import oracle_config

    cfg = pkg.Qwen3Config.qwen3_0_6b()
    return cfg, oracle.Model(oracle_config(oracle, cfg)).fill_synthetic(0)
