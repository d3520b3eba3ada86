"""Wall-clock stage breakdown of the drop-in functions (off unless asked for).

``PCH_STAGE_TIMINGS=1`` (or ``stages.enable()``) makes ``run_voxel_downsampling`` / ``extract_towers`` record how
long every stage took - the device is drained at each stage edge, so the numbers add up to the call's wall clock
but the call itself gets a little slower.  The last call's table is ``stages.last(name)``: an ordered dict
{stage: seconds}.  bench.py reports it for the 100 M-point run.
"""
from __future__ import annotations

import os
import time

_enabled = bool(os.environ.get("PCH_STAGE_TIMINGS"))
_last = {}


def enable(on=True):
    global _enabled
    _enabled = bool(on)


def last(name):
    return _last.get(name)


class Clock:
    """``c = Clock("extract_towers"); ...; c.mark("read")`` - no-ops when timings are off."""

    def __init__(self, name):
        self.name, self.on = name, _enabled
        self.table = {}
        if self.on:
            self.t = time.perf_counter()
            _last[name] = self.table

    def mark(self, stage, sync=True):
        if not self.on:
            return
        if sync:
            try:
                import torch
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
            except Exception:
                pass
        now = time.perf_counter()
        self.table[stage] = self.table.get(stage, 0.0) + (now - self.t)
        self.t = now

    def move(self, src, dst, seconds):
        """books `seconds` that were measured inside a callback of stage `src` under `dst` instead"""
        if self.on:
            self.table[src] = self.table.get(src, 0.0) - seconds
            self.table[dst] = self.table.get(dst, 0.0) + seconds
