"""Host side of "frames in flight" (DESIGN 14k).  Query frames are independent (inference.py:97-182 walks them one by one), so a
serving process overlaps them: every frame in flight on a HIP stream of its own.  Two facts of the HIP runtime decide how many pay:

* streams are dealt onto a pool of GPU_MAX_HW_QUEUES hardware queues -- 4 unless the variable is set -- and two streams on one queue run
  strictly one behind the other (with the default pool a fourth stream adds nothing);
* the variable is read when the runtime initialises, i.e. at the first HIP call of the process, not when torch is imported
  (profiles/r04_ab_live_four_in_flight.txt).

Measured optimum (profiles/r04_sweep_hw_queues.txt): FOUR frames in flight on >= 5 queues; a fifth concurrently running frame loses 5-20 %.
"""
import os
import warnings

import torch

HW_QUEUES = 8
FRAMES_IN_FLIGHT = 4


def configure_hip_queues(n=HW_QUEUES):
    """OPT-IN: export GPU_MAX_HW_QUEUES=n unless the caller already did; returns the value in force (str) or None when it is too late
    (the runtime of this process is initialised and the variable was not set: the pool stays at the runtime's default) or when the
    caller opted out.  Nothing calls this at import time; StreamRing() calls it (so does bench.py, explicitly, before `import torch`).

    This is a PROCESS-WIDE setting of the HIP runtime: it changes the queue pool of every other HIP user in the process.  Opt out with
    ONEPOSE_AMD_NO_HIP_QUEUE_EXPORT=1 (this function then never touches the variable; StreamRing still works, its fourth stream shares
    a hardware queue).  The call must come before the first HIP call of the process to take effect -- that includes
    torch.cuda.is_available() / device_count(), which bring the runtime up without torch's own lazy init noticing: in that case the
    value is exported and returned but ignored by the runtime (it cannot be detected from here; tests/conftest.py calls this right
    after importing the package for that reason)."""
    if os.environ.get("ONEPOSE_AMD_NO_HIP_QUEUE_EXPORT", "0") not in ("", "0"):
        return None
    cur = os.environ.get("GPU_MAX_HW_QUEUES")
    if cur is not None:
        return cur
    if torch.cuda.is_initialized():   # (no torch.cuda.is_available() in front of the export: it is itself a HIP call)
        warnings.warn("onepose_amd: the HIP runtime is already initialised with its default pool of 4 hardware queues; export "
                      f"GPU_MAX_HW_QUEUES={n} before the first HIP call to keep {FRAMES_IN_FLIGHT} frames in flight (DESIGN 14k)", stacklevel=2)
        return None
    os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    return str(n)


class StreamRing:
    """n HIP streams dealt round-robin: ``with ring.next(): pred, conf = model(data)`` keeps n frames in flight (the modules cache
    their workspaces per stream; every frame in flight needs outputs of its own, which the modules allocate per call)."""

    def __init__(self, device, n=FRAMES_IN_FLIGHT, streams=None):
        """streams: optional list of existing torch.cuda.Stream objects to deal frames over instead of creating n new ones (a process that
        already owns per-frame streams should reuse them: the runtime maps streams onto its hardware queues in creation order, and two busy
        streams that land on one queue serialise)."""
        self.hw_queues = configure_hip_queues() if n >= 4 else os.environ.get("GPU_MAX_HW_QUEUES")   # before the first HIP call below, if it still can be
        if not torch.cuda.is_available():
            raise RuntimeError("StreamRing needs a ROCm GPU (there is no CPU path)")
        self.device = torch.device(device)
        self.streams = list(streams) if streams is not None else [torch.cuda.Stream(self.device) for _ in range(n)]
        self._i = 0

    _shared = {}

    @classmethod
    def shared(cls, device, n=FRAMES_IN_FLIGHT):
        """The process-wide ring of `device`: every component that overlaps frames should deal them over ONE set of streams (see __init__)."""
        key = (str(torch.device(device)), n)
        if key not in cls._shared:
            cls._shared[key] = cls(device, n)
        return cls._shared[key]

    def next(self):
        s = self.streams[self._i % len(self.streams)]
        self._i += 1
        return torch.cuda.stream(s)

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
