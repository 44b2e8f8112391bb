"""Host I/O either side of the path: the reference decodes JPEGs in one DataLoader worker
(src/inference.py:75-78) and writes every PNG serially from the frame loop (src/utils/utils.py:34-42,97-100); once a frame takes
about a millisecond on the GPU that host work is the bottleneck.  Two pieces, no change of behaviour:
  * `make_loader`: the reference's batch-1, in-order DataLoader with N decode workers and a deeper prefetch queue;
  * `AsyncMaskWriter`: masks of a finished video leave the GPU with a non-blocking copy into pinned memory, and a small thread pool
    waits for that copy and encodes the mode-P PNGs (PIL releases the GIL while it compresses) while the loop runs the next video.
File names, palette and pixel values are exactly those of `save_predictions`."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import torch.utils.data

from .utils import save_predictions


def default_io_workers():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(8, n - 1))


def make_loader(dataset, io_workers=1, pin=False):
    """batch_size=1, shuffle=False - the order contract of the reference's loader - with `io_workers` decode processes.
    pin stays off: measured on the MI355X box, the loader's pinning thread (a pinned allocation per frame) contends with the
    main thread's HIP calls and a frame's H2D copy goes from 0.09-0.17 ms to 6.7 ms; copying the workers' shared-memory
    tensors directly costs 0.17 ms per 480p uint8 frame and overlaps with the GPU work already queued."""
    io_workers = max(0, int(io_workers))
    kw = dict(batch_size=1, shuffle=False, num_workers=io_workers, pin_memory=bool(pin) and torch.cuda.is_available())
    if io_workers > 0:
        kw.update(prefetch_factor=8, persistent_workers=False)
    return torch.utils.data.DataLoader(dataset, **kw)


class AsyncMaskWriter:
    """submit(video, palette, masks) returns at once; close() waits for every PNG and re-raises the first failure.
    masks: list of (H,W) uint8 tensors (any device) or an (n,H,W) array."""

    def __init__(self, save, workers=2):
        self.save = save
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(workers)), thread_name_prefix='vosprop-png')
        self.jobs = []

    def submit(self, video, palette, masks):
        if self.save is None or masks is None or len(masks) == 0:
            return
        if isinstance(masks, np.ndarray):
            host, done = masks, None
        else:
            stack = torch.stack(list(masks)) if not torch.is_tensor(masks) else masks
            if stack.is_cuda:
                host_t = torch.empty(stack.shape, dtype=stack.dtype, pin_memory=True)
                host_t.copy_(stack, non_blocking=True)
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(stack.device))
                host = host_t
            else:
                host, done = stack, None
        self.jobs.append(self.pool.submit(self._write, video, list(palette) if palette is not None else None, host, done))

    def _write(self, video, palette, host, done):
        if done is not None:
            done.synchronize()
        # one pass out of the pinned staging buffer (pinned host memory is uncached for the CPU: PIL would crawl over it)
        arr = np.array(host.numpy(), copy=True) if torch.is_tensor(host) else host
        save_predictions(arr, palette, self.save, video)
        return len(arr)

    def close(self):
        err = None
        n = 0
        for j in self.jobs:
            try:
                n += j.result()
            except Exception as e:     # keep draining: every job must finish before the pool goes away
                err = err or e
        self.jobs.clear()
        self.pool.shutdown(wait=True)
        if err is not None:
            raise err
        return n


class ShmFrameLoader:
    """In-order frame source for the single-tensor strategies with ZERO host copies between the JPEG decoder and the DMA engine:
    N decode processes (forked: they inherit the dataset's preloaded bytes) write raw uint8 frames into the slots of one
    shared-memory ring, the ring is registered with HIP once as pinned memory (hipHostRegister), and the loop's H2D copies read it
    directly.  torch's DataLoader instead ships every frame through a fresh shared-memory file and (optionally) a pinned
    allocation; under load the main thread then spends 1.5 ms per 480p frame on that copy alone (tools/cli_timing.py).

    Protocol: iterate -> (uint8 tensor (1,H,W,3) viewing a slot, (video_name,)), the DataLoader's batch-1 item.  The consumer
    calls recycle(tensors, event) once the asynchronous copies out of those tensors have been enqueued; a slot goes back to the
    decoders when that event has completed.  Frames whose byte size exceeds a slot (a larger video) are returned as ordinary
    tensors through the result queue."""

    def __init__(self, dataset, workers=4, slots=128, register=True):
        import multiprocessing as mp
        from multiprocessing import shared_memory
        self.ds = dataset
        self.n = len(dataset)
        self.workers = max(1, int(workers))
        probe, _ = dataset[0] if self.n else (torch.zeros(1, 1, 3, dtype=torch.uint8), '')
        if isinstance(probe, (tuple, list)) or probe.dtype != torch.uint8:
            raise ValueError('ShmFrameLoader needs a single-tensor raw_uint8 dataset')
        self.slot_bytes = int(probe.numel())
        self.slots = max(2 * self.workers, int(slots))
        self.shm = shared_memory.SharedMemory(create=True, size=self.slot_bytes * self.slots)
        self.buf = np.ndarray((self.slots, self.slot_bytes), dtype=np.uint8, buffer=self.shm.buf)
        self.base = torch.from_numpy(self.buf)
        self.registered = False
        if register and torch.cuda.is_available():
            rc = torch.cuda.cudart().cudaHostRegister(self.base.data_ptr(), self.base.numel(), 0)
            self.registered = int(rc) == 0
        ctx = mp.get_context('fork')
        self.tasks = ctx.Queue()
        self.results = ctx.Queue()
        self.procs = [ctx.Process(target=self._work, daemon=True) for _ in range(self.workers)]
        for p in self.procs:
            p.start()
        self.free = list(range(self.slots))
        self.pending = []            # (event | None, [slots]) waiting for their copies to finish
        self.closed = False

    # ---- decode process ----
    def _work(self):
        torch.set_num_threads(1)
        while True:
            job = self.tasks.get()
            if job is None:
                return
            seq, idx, slot = job
            try:
                x, name = self.ds[idx]
                a = x.numpy().reshape(-1)
                if a.size <= self.slot_bytes:
                    self.buf[slot, :a.size] = a
                    self.results.put((seq, slot, tuple(x.shape), name, None))
                else:
                    self.results.put((seq, slot, tuple(x.shape), name, x))
            except Exception as e:      # surfaced in the consumer
                self.results.put((seq, slot, None, None, repr(e)))

    # ---- consumer side ----
    def _reclaim(self, block):
        while self.pending and (block or self.pending[0][0] is None or self.pending[0][0].query()):
            ev, slots = self.pending.pop(0)
            if ev is not None:
                ev.synchronize()
            self.free.extend(slots)
            block = False

    def recycle(self, tensors, event=None):
        slots = [t._vosprop_slot for t in tensors if hasattr(t, '_vosprop_slot')]
        if slots:
            self.pending.append((event, slots))

    def __len__(self):
        return self.n

    def __iter__(self):
        submitted = 0
        ready = {}
        for want in range(self.n):
            self._reclaim(block=False)
            while submitted < self.n and self.free:            # keep every free slot busy
                self.tasks.put((submitted, submitted, self.free.pop()))
                submitted += 1
            if submitted <= want:                               # the wanted frame has no slot yet: wait for copies to finish
                self._reclaim(block=True)
                if not self.free:
                    raise RuntimeError('ShmFrameLoader: every slot is held by the consumer (recycle() not called?)')
                self.tasks.put((submitted, submitted, self.free.pop()))
                submitted += 1
            while want not in ready:
                seq, slot, shape, name, extra = self.results.get()
                ready[seq] = (slot, shape, name, extra)
            slot, shape, name, extra = ready.pop(want)
            if shape is None:
                raise RuntimeError(f'frame {want}: decode failed in a worker: {extra}')
            if extra is not None:          # did not fit a slot
                self.free.append(slot)
                yield extra[None], (name,)
                continue
            n = int(np.prod(shape))
            t = self.base[slot, :n].view(shape)[None]
            t._vosprop_slot = slot
            yield t, (name,)

    def close(self):
        if self.closed:
            return
        self.closed = True
        for _ in self.procs:
            self.tasks.put(None)
        for p in self.procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        if self.registered:
            torch.cuda.synchronize()
            torch.cuda.cudart().cudaHostUnregister(self.base.data_ptr())
        del self.base, self.buf
        self.shm.close()
        self.shm.unlink()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
