"""Per-(device, stream) scratch blocks shared by the host-side wrappers (split-K slabs, batch-norm partials).
Scenes in flight run on several host threads, so the table is guarded by a lock and evicts least-recently-used."""
import threading
from collections import OrderedDict

import torch


class StreamScratch(object):
    def __init__(self, max_entries=16):
        self._lock = threading.Lock()
        self._table = OrderedDict()
        self._max = max_entries

    def get(self, device, nbytes, min_bytes=0, zero=False):
        """A uint8 block of at least `nbytes` owned by (device, current stream); grown on demand.  zero: a NEW block is
        zero-filled (workspaces that hold counters their kernels leave at zero: csrc/bnorm.hip)."""
        alloc = torch.zeros if zero else torch.empty
        # (the raw C entry points: torch.cuda.current_stream() builds a Stream object per call, ~7 us of the wrappers' ~20)
        if torch._C._cuda_isCurrentStreamCapturing():
            # inside a HIP-graph capture: a cached block would belong to THIS graph's private pool and be handed to the
            # next capture on torch's shared capture stream (use after free once this graph is released): allocate per call
            return alloc(max(int(nbytes), int(min_bytes)), dtype=torch.uint8, device=device)
        index = device.index if isinstance(device, torch.device) and device.index is not None else torch._C._cuda_getDevice()
        key = (index, torch._C._cuda_getCurrentRawStream(index))
        with self._lock:
            ws = self._table.get(key)
            if ws is not None and ws.numel() >= nbytes:
                self._table.move_to_end(key)
                return ws
            # an evicted block goes back to the pool of the stream it was allocated on, which orders any reuse
            while len(self._table) >= self._max and key not in self._table:
                self._table.popitem(last=False)
            ws = alloc(max(int(nbytes), int(min_bytes)), dtype=torch.uint8, device=device)
            self._table[key] = ws
            self._table.move_to_end(key)
            return ws
