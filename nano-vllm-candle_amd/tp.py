"""tp::TPConfig mirror (src/tp.rs:4-70).  The reference's struct has no communication; here the same
{size, rank, dim} record also names the RCCL group a Context joins (one process per rank)."""
import os
from dataclasses import dataclass, replace


@dataclass(frozen=True)
class TPConfig:
    size: int = 1
    rank: int = 0
    dim: int = 0

    @classmethod
    def from_env(cls):
        # src/tp.rs:21-31: TP_SIZE default 1; TP_RANK >= size folds to 0
        def _int(name):
            try:
                return int(os.environ[name])
            except (KeyError, ValueError):
                return None

        size = _int("TP_SIZE") or 1
        rank = _int("TP_RANK")
        rank = 0 if rank is None or rank >= size or rank < 0 else rank
        return cls(size, rank, 0)

    @classmethod
    def single(cls):
        return cls(1, 0, 0)

    def with_size(self, size):
        return replace(self, size=size)

    def with_rank(self, rank):
        return replace(self, rank=rank)

    def with_dim(self, dim):
        return replace(self, dim=dim)

    def is_distributed(self):
        return self.size > 1

    def shard_size(self, total):
        return total // self.size

    def shard_offset(self, total):
        return self.rank * self.shard_size(total)


def get_tp():
    """re-reads the environment on every call, like src/tp.rs:68-70"""
    return TPConfig.from_env()


def shard_region(cfg, tp_size, tp_rank, hf_name):
    """(row0, col0, rows, cols) of the full HF tensor owned by this rank -- from the C ABI (host-only call)"""
    import ctypes as C

    from . import _lib

    out = (C.c_int64 * 4)()
    cc = cfg.to_c()
    rc = _lib.lib().nvllm_tp_shard(C.byref(cc), tp_size, tp_rank, hf_name.encode(), out)
    if rc != 0:
        raise ValueError(f"no TP shard for {hf_name!r} at tp_size={tp_size}")
    return tuple(int(v) for v in out)
