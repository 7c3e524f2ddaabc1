"""When to re-sort: mirror of the reference's caller-side policy.

Reference: `SortConfig { period_ms: 1000 }` (src/sort/mod.rs:76-86), `SortTrigger`
(src/sort/mod.rs:143-150), `update_sort_trigger` (src/sort/mod.rs:153-194) and the CPU sorts'
self-throttle `period_ms = max(period_ms, 4 * last sort duration)` (src/sort/rayon.rs:124-129).
The device sort here costs ~0.1 ms per million splats, so a GPU caller can simply sort every
frame (`bgs_render` does); this policy object exists for callers that keep the reference's
throttled behaviour. Time is injected so the logic is testable.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np


@dataclass
class SortConfig:
    """src/sort/mod.rs:76-86."""
    period_ms: int = 1000


@dataclass
class SortTrigger:
    """src/sort/mod.rs:143-150 (one per GaussianCamera)."""
    camera_index: int = 0
    needs_sort: bool = False
    last_camera_position: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    last_sort_time: Optional[float] = None


def update_sort_trigger(trigger: SortTrigger, camera_position, camera_order: int, config: SortConfig,
                        now: Callable[[], float] = time.monotonic) -> SortTrigger:
    """src/sort/mod.rs:164-193, for one camera."""
    t = now()
    if trigger.last_sort_time is None:
        if camera_order < 0:
            raise ValueError("camera order must be a non-negative index into gaussian cameras")
        trigger.camera_index = int(camera_order)
        trigger.needs_sort = True
        trigger.last_sort_time = t
        return trigger
    if (t - trigger.last_sort_time) * 1000.0 < config.period_ms:
        return trigger
    pos = np.asarray(camera_position, np.float32)
    if not np.array_equal(trigger.last_camera_position, pos):
        trigger.needs_sort = True
        trigger.last_sort_time = t
        trigger.last_camera_position = pos.copy()
    return trigger


def after_cpu_sort(config: SortConfig, sort_duration_s: float) -> SortConfig:
    """src/sort/rayon.rs:124-129 / std_sort.rs: the CPU sort paths stretch the period to at least
    4x the measured sort time (`max(period, period*4/5)` is a no-op kept for fidelity)."""
    config.period_ms = max(config.period_ms, config.period_ms * 4 // 5, 4 * int(sort_duration_s * 1000.0))
    return config
