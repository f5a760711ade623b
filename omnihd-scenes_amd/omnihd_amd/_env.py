"""Environment switches read in the hot path of a training step.

A step of the detector asks ``os.environ`` about a thousand times (policy switches inside every convolution, BatchNorm and
pooling wrapper: 0.4-0.8 ms of host time per step, measured with scripts/cpu_profile.py on a step that is bound by the host's
enqueue rate).  Between ``epoch_begin()`` and ``epoch_end()`` — the harness brackets every training step with them — a switch is
looked up once and then served from a dict; outside an epoch every call goes to ``os.environ`` (tests that flip a switch between
two calls see it at once)."""
import os

_CACHE = {}
_LIVE = [0]


def env(name, default=None):
    if _LIVE[0]:
        v = _CACHE.get(name, _CACHE)
        if v is _CACHE:
            v = _CACHE[name] = os.environ.get(name)
        return default if v is None else v
    return os.environ.get(name, default)


def epoch_begin():
    _CACHE.clear()
    _LIVE[0] += 1


def epoch_end():
    _LIVE[0] = max(0, _LIVE[0] - 1)
    if not _LIVE[0]:
        _CACHE.clear()
