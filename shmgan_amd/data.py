"""Dataset loader: the caller side of `train_step` for real data.

Mirrors /root/reference/datasetLoader.py:19-170 (`datasetLoad(self)`): five sibling directories of the
polarimetric views are listed in sorted order (image_dataset_from_directory(labels=None, shuffle=False)),
zipped, every file decoded to RGB, resized to image_size x image_size with tf.image.resize's bilinear kernel,
scaled by 1/255 and flipped top-to-bottom.  As executed the flip is unconditional: the `map` lambda
`x if self.random_flip else flip_up_down(x)` (datasetLoader.py:61) is traced once with the constructor's
`self.random_flip = 0.0` (SHM.py:203); `flip_ud=` makes it explicit.

Decode is PIL on the host, in a WORKER THREAD: the decoded bytes land in pinned uint8 staging buffers, go to the GPU
with non-blocking copies on the loader's side stream, and the rest is one kernel (shm_resize_bilinear_u8) per image on
that stream.  The training thread only enqueues the next batch and, when it takes a batch, waits for the worker's
future and makes its stream wait for the batch's event -- the 5 B decodes of batch j+1 run under step j.

Under torch.distributed the loader shards by rank: global batch i of rank r is images [(i*world + r)*B, +B), so N ranks
consume N*B distinct samples per step (the data-parallel identity of shmgan_amd/dist.py) and len() = n // (B*world).
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

from . import ops

PSD_SUBDIRS = ("I0", "I60", "I90", "I150", "ED")          # datasetLoader.py:30-34 (PSD polar dataset)
SHMGAN_SUBDIRS = ("I0", "I45", "I90", "I135", "ED")       # datasetLoader.py:23-27 (commented alternative)
_EXT = (".bmp", ".gif", ".jpeg", ".jpg", ".png")          # Keras' ALLOWLIST_FORMATS


def list_images(directory):
    """Sorted file list as image_dataset_from_directory(shuffle=False) yields it."""
    d = Path(directory)
    return sorted(str(p) for p in d.iterdir() if p.suffix.lower() in _EXT)


class PolarDataset:
    """Iterable of 5-tuples of [B,S,S,3] float32 device tensors in [0,1]."""

    def __init__(self, data_dir, image_size, batch_size=1, subdirs=PSD_SUBDIRS, flip_ud=True, device=None, epochs=1,
                 rank=None, world=None):
        self.S, self.B, self.flip_ud, self.epochs = image_size, batch_size, flip_ud, epochs
        self.files = [list_images(os.path.join(data_dir, s)) for s in subdirs]
        n = len(self.files[0])
        if any(len(f) != n for f in self.files):
            raise ValueError(f"the five view directories hold different numbers of images: {[len(f) for f in self.files]}")
        self.n = n
        if rank is None or world is None:
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized()
            rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
        self.rank, self.world = int(rank), int(world)
        self._dev, self._stream = device, None
        self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="shm-loader")
        # pinned staging, two generations (a batch in preparation + the one just handed over): {(gen, view, b): uint8 [H,W,3]}
        self._pin = {}
        self._gen_event = [None, None]
        self._prepared = 0

    @property
    def dev(self):
        if self._dev is None:
            self._dev = torch.device("cuda", torch.cuda.current_device())
        return torch.device(self._dev)

    @property
    def stream(self):
        """The loader's side stream (created on first use: listing and sharding need no GPU)."""
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.dev)
        return self._stream

    def __len__(self):
        return self.n // (self.B * self.world)

    def image_index(self, index, b):
        """Dataset position of sample b of this rank's batch `index`."""
        return (index * self.world + self.rank) * self.B + b

    def _decode(self, path, key):
        from PIL import Image
        with Image.open(path) as im:
            a = np.asarray(im.convert("RGB"), dtype=np.uint8)
        buf = self._pin.get(key)
        if buf is None or tuple(buf.shape) != a.shape:
            buf = torch.empty(a.shape, dtype=torch.uint8).pin_memory()
            self._pin[key] = buf
        buf.numpy()[...] = a
        return buf

    def _prepare_worker(self, index, gen):
        """Runs on the loader thread (torch's current stream is per thread): decode into this generation's pinned buffers,
        then enqueue copy + resize per image on the loader stream and record the batch's event."""
        if self._gen_event[gen] is not None:         # the copies that last read this generation's staging buffers
            self._gen_event[gen].synchronize()       # (a host wait, but on the loader thread)
        staged = [[self._decode(self.files[v][self.image_index(index, b)], (gen, v, b)) for b in range(self.B)] for v in range(5)]
        with torch.cuda.device(self.dev), torch.cuda.stream(self.stream):
            # allocated, filled and consumed on the loader stream: the caching allocator hands a block back to loader-stream
            # allocations only, which are ordered behind the resize kernel that read it
            outs = [torch.empty((self.B, self.S, self.S, 3), device=self.dev) for _ in range(5)]
            for v in range(5):
                for b in range(self.B):
                    src = staged[v][b].to(self.dev, non_blocking=True)
                    ops.resize_bilinear_u8(src, outs[v][b], 1.0 / 255.0, self.flip_ud)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._gen_event[gen] = ev
        return tuple(outs), ev

    def prepare(self, index):
        """Start batch `index` (0-based, of this rank) on the loader thread / stream; returns a future of
        (five [B,S,S,3] tensors, ready event).  The outputs are ALLOCATED on the loader stream: a block the consumer has
        dropped is then only reused after the consumer stream's work recorded by `take()` has finished (train_step is
        fully asynchronous and reads its inputs late in the step, so allocating them on the consumer stream would let the
        next batch's resize kernels overwrite images that queued step kernels still read)."""
        gen = self._prepared & 1
        self._prepared += 1
        return self._pool.submit(self._prepare_worker, index, gen)

    def take(self, prepared):
        """Hand a prepared batch to the current stream (waits for the loader thread's host work, not for the GPU)."""
        outs, ev = prepared.result() if hasattr(prepared, "result") else prepared
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for t in outs:
            t.record_stream(cur)
        return outs

    def batch(self, index):
        """Batch `index` (0-based) as five [B,S,S,3] tensors; prepared on the loader's stream."""
        return self.take(self.prepare(index))

    def __iter__(self):
        """One batch is always in preparation on the loader thread while the previous one is consumed."""
        order = [i for _ in range(self.epochs) for i in range(len(self))]
        nxt = self.prepare(order[0]) if order else None
        for j in range(len(order)):
            cur = nxt
            nxt = self.prepare(order[j + 1]) if j + 1 < len(order) else None
            yield self.take(cur)


def datasetLoad(trainer, subdirs=PSD_SUBDIRS, flip_ud=True):
    """Reference signature (datasetLoader.py:19): returns (length_dataset, loadedDataset) and sets the same
    attributes on the trainer object."""
    ds = PolarDataset(trainer.data_dir, trainer.image_size, trainer.batch_size, subdirs, flip_ud, trainer.device,
                      epochs=trainer.num_epochs)
    trainer.stddev_arr, trainer.mean_arr, trainer.variance_arr = [], [], []
    # per-rank length: batches_per_epoch = length // batch_size (SHM.py:957) then counts this rank's batches
    trainer.length_dataset, trainer.loadedDataset = ds.n // ds.world, ds
    return trainer.length_dataset, ds
