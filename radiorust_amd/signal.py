"""Signal messages — host-side mirror of src/signal.rs:19-46,170-215."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


class Event:
    """signal.rs:19-31"""

    def is_interrupt(self) -> bool:
        return False

    def is_flush(self) -> bool:
        return False


class Disconnection(Event):
    """signal.rs:37-46"""

    def is_interrupt(self) -> bool:
        return True


class SamplesLost(Event):
    """chunks.rs:20-28"""

    def is_interrupt(self) -> bool:
        return True


@dataclass
class Samples:
    """Signal::Samples { sample_rate, chunk } (signal.rs:172-178)."""

    sample_rate: float
    chunk: np.ndarray

    def is_event(self) -> bool:
        return False

    def duration(self) -> float:
        return len(self.chunk) / self.sample_rate


@dataclass
class EventSignal:
    """Signal::Event(Arc<dyn Event>) (signal.rs:179-182)."""

    event: Event

    def is_event(self) -> bool:
        return True

    def duration(self) -> float:
        return 0.0
