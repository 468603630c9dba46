"""Device-time spans of the stage-2 loops (round 5: where SparseGPT's ~20 s go).  Off by default
(no events, no cost); `PhaseTimer.enable()` — tools/run_sparsegpt.py --phases — makes every
`with PhaseTimer.span(name):` record a torch event pair on the current stream, `report()` adds
them up per name after one synchronize.  Spans nest (an inner span's time is also part of the
outer one) and measure the stream between their two events: the device work queued inside plus
whatever gap the host leaves there."""
import contextlib

import torch


class PhaseTimer:
    enabled = False
    pairs = {}

    @classmethod
    def enable(cls, on=True):
        cls.enabled = bool(on) and torch.cuda.is_available()
        cls.pairs = {}

    @classmethod
    @contextlib.contextmanager
    def span(cls, name):
        if not cls.enabled:
            yield
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        try:
            yield
        finally:
            b.record()
            cls.pairs.setdefault(name, []).append((a, b))

    @classmethod
    def report(cls):
        if not cls.enabled:
            return None
        torch.cuda.synchronize()
        return {name: {"spans": len(v), "seconds": sum(a.elapsed_time(b) for a, b in v) * 1e-3}
                for name, v in cls.pairs.items()}
