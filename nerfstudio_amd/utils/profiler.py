"""Profiling hooks of the hot path (reference: nerfstudio/utils/profiler.py:52-115 `time_function`, :133-178 `Profiler`;
SURVEY.md §5 row 1).

The reference decorates `VanillaPipeline.get_train_loss_dict` and `Trainer.train_iteration` with `@profiler.time_function`:
a running mean of the HOST wall time per qualified name, printed at exit. On MI355X the interesting clock is the device's,
so the same hook additionally opens a **roctx range** (`roctxRangePushA` / `roctxRangePop` of libroctx64 — what
`rocprofv3 --marker-trace` records and what groups the kernels of `--kernel-trace` under a named phase):

  * `time_function` here has the reference's two forms (decorator on a function, context manager around a block) and is
    applied to the phases of the explicit kernel schedule (train_step.py), the eval chunk loop (eval_render.py) and the
    optimiser / exchange calls (arena.py);
  * `hook_reference_profiler()` wraps the reference's OWN `_TimeFunction.__enter__ / __exit__` when nerfstudio is
    importable (plugin.nerfacto_hip calls it), so that the hooks the reference already has emit ranges too — no edit of
    the reference.

Ranges are off by default (two ctypes calls per hook); `NSAMD_ROCTX=1` or `enable_ranges()` turns them on. Host timing
needs `setup_profiler()` as in the reference. With both off the decorator calls straight through."""
from __future__ import annotations

import ctypes
import functools
import os
import time
from contextlib import ContextDecorator
from typing import Callable, Dict, List, Optional, Union

_ROCTX = None
_RANGES = os.environ.get("NSAMD_ROCTX", "0") == "1"
PROFILER: List["Profiler"] = []


def _roctx():
    """libroctx64 (ROCm's marker API); None when it cannot be loaded (CPU-only containers): ranges become no-ops."""
    global _ROCTX
    if _ROCTX is None:
        _ROCTX = False
        for name in ("libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _ROCTX = lib
                break
            except OSError:
                continue
    return _ROCTX or None


def enable_ranges(on: bool = True) -> bool:
    """Turn roctx ranges on / off at run time; returns whether the marker library is available."""
    global _RANGES
    _RANGES = bool(on)
    return _roctx() is not None


def ranges_enabled() -> bool:
    return _RANGES and _roctx() is not None


def range_push(name: str) -> None:
    lib = _roctx() if _RANGES else None
    if lib is not None:
        lib.roctxRangePushA(name.encode())


def range_pop() -> None:
    lib = _roctx() if _RANGES else None
    if lib is not None:
        lib.roctxRangePop()


class Profiler:
    """Running mean of the host wall time per name (utils/profiler.py:133-178)."""

    def __init__(self) -> None:
        self.profiler_dict: Dict[str, Dict[str, float]] = {}

    def update_time(self, func_name: str, start_time: float, end_time: float) -> None:
        val = end_time - start_time
        entry = self.profiler_dict.get(func_name, {"val": 0.0, "step": 0})
        entry = {"val": (entry["val"] * entry["step"] + val) / (entry["step"] + 1), "step": entry["step"] + 1}
        self.profiler_dict[func_name] = entry

    def print_profile(self) -> None:
        print("Printing profiling stats, from longest to shortest duration in seconds")
        for k, v in sorted(self.profiler_dict.items(), key=lambda kv: kv[1]["val"], reverse=True):
            print(f"{k:<60}: {v['val']:0.6f}  ({int(v['step'])} calls)")


def setup_profiler() -> Profiler:
    """Start accumulating host times (the reference's `--logging.profiler basic`)."""
    if not PROFILER:
        PROFILER.append(Profiler())
    return PROFILER[0]


def flush_profiler() -> None:
    if PROFILER:
        PROFILER[0].print_profile()


class _TimeFunction(ContextDecorator):
    def __init__(self, name: str) -> None:
        self.name = name
        self.start: Optional[float] = None

    def __enter__(self):
        self.start = time.time()
        range_push(self.name)
        return self

    def __exit__(self, *exc):
        range_pop()
        if PROFILER:
            PROFILER[0].update_time(self.name, self.start, time.time())
        return False


def time_function(name_or_func: Union[Callable, str]):
    """`@time_function` on a function / method, or `with time_function("name"):` around a block — the reference's two forms."""
    if isinstance(name_or_func, str):
        return _TimeFunction(name_or_func)
    if not callable(name_or_func):
        raise ValueError(f"Argument func of type {type(name_or_func)} is not a string or a callable.")
    func, name = name_or_func, name_or_func.__qualname__

    @functools.wraps(func)
    def inner(*args, **kwargs):
        if not (_RANGES or PROFILER):  # nothing listening: straight through
            return func(*args, **kwargs)
        with _TimeFunction(name):
            return func(*args, **kwargs)

    return inner


def hook_reference_profiler() -> bool:
    """Make the reference's own `@profiler.time_function` hooks (pipelines/base_pipeline.py:289, engine/trainer.py:486, ...)
    open roctx ranges: wraps `nerfstudio.utils.profiler._TimeFunction.__enter__/__exit__` once. False when nerfstudio is not
    importable."""
    try:
        from nerfstudio.utils import profiler as ref
    except Exception:  # noqa: BLE001
        return False
    cls = ref._TimeFunction
    if getattr(cls, "_nsamd_ranges", False):
        return True
    enter, exit_ = cls.__enter__, cls.__exit__

    def __enter__(self):
        range_push(str(self.name))
        return enter(self)

    def __exit__(self, *args, **kwargs):
        out = exit_(self, *args, **kwargs)
        range_pop()
        return out

    cls.__enter__, cls.__exit__, cls._nsamd_ranges = __enter__, __exit__, True
    return True
