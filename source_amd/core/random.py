"""
Host MT19937-64 stream, API-compatible with raysect.core.math.random (seed / uniform / probability,
random.pyx:215-333). The generator itself is librsx's rsx_mt (C++); this module owns the process-global state
exactly like the reference's module-level ``mt[]`` array. Used for RSX_RNG_STREAM renders (bit parity with the
reference's SerialEngine on primary-ray scenes) — bulk draws go through uniform_block().
"""
import ctypes as C
import os

import numpy as np

from .. import _lib

_state = None


def _get():
    global _state
    if _state is None:
        seed(None)
    return _state


def seed(d=None):
    """random.pyx:215-243: the integer is split into 312 big-endian 64-bit words -> init_by_array64."""
    global _state
    b = int(d).to_bytes(8 * 312, "big") if d else os.urandom(8 * 312)
    words = np.frombuffer(b, dtype=">u8").astype(np.uint64)
    st = _lib.MT()
    _lib.lib().rsx_mt_seed_words(C.byref(st), _lib.ptr(words), 312)
    _state = st


def uniform_block(n):
    out = np.empty(int(n), dtype=np.float64)
    _lib.lib().rsx_mt_uniform(C.byref(_get()), int(n), _lib.ptr(out))
    return out


_override = None      # a stream object with next(): while set, uniform() draws from it (source_amd/optical/hybrid.py: per-path Philox streams)


def set_stream(stream):
    """Routes uniform()/probability() to `stream.next()` (None restores the process-global MT19937-64 state). Returns the previous one."""
    global _override
    previous, _override = _override, stream
    return previous


def uniform():
    if _override is not None:
        return _override.next()
    return float(uniform_block(1)[0])


def probability(p):
    return uniform() < p
