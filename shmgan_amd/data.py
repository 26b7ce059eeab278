"""Dataset loader: the caller side of `train_step` for real data.

Mirrors /root/reference/datasetLoader.py:19-170 (`datasetLoad(self)`): five sibling directories of the
polarimetric views are listed in sorted order (image_dataset_from_directory(labels=None, shuffle=False)),
zipped, every file decoded to RGB, resized to image_size x image_size with tf.image.resize's bilinear kernel,
scaled by 1/255 and flipped top-to-bottom.  As executed the flip is unconditional: the `map` lambda
`x if self.random_flip else flip_up_down(x)` (datasetLoader.py:61) is traced once with the constructor's
`self.random_flip = 0.0` (SHM.py:203); `flip_ud=` makes it explicit.

Decode is PIL on the host; the decoded bytes go to the GPU as uint8 and the rest is one kernel
(shm_resize_bilinear_u8) per image, on a side stream so the next batch is prepared under the current step.
"""
from __future__ import annotations

import os
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

    def __init__(self, data_dir, image_size, batch_size=1, subdirs=PSD_SUBDIRS, flip_ud=True, device=None, epochs=1):
        self.S, self.B, self.flip_ud, self.epochs = image_size, batch_size, flip_ud, epochs
        self.files = [list_images(os.path.join(data_dir, s)) for s in subdirs]
        n = len(self.files[0])
        if any(len(f) != n for f in self.files):
            raise ValueError(f"the five view directories hold different numbers of images: {[len(f) for f in self.files]}")
        self.n = n
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.dev = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.dev)

    def __len__(self):
        return self.n // self.B

    def _load(self, path, out):
        from PIL import Image
        with Image.open(path) as im:
            a = np.array(im.convert("RGB"), dtype=np.uint8)             # own, writable copy
        # allocated, filled and consumed on the loader stream: the caching allocator hands the block back to
        # loader-stream allocations only, which are ordered behind the resize kernel -- no host sync needed
        src = torch.from_numpy(a).to(self.dev, non_blocking=False)
        ops.resize_bilinear_u8(src, out, 1.0 / 255.0, self.flip_ud)

    def prepare(self, index):
        """Start batch `index` (0-based) on the loader's stream; returns (five [B,S,S,3] tensors, ready event).
        The outputs are ALLOCATED on the loader stream: a block the consumer has dropped is then only reused
        after the consumer stream's work recorded by `take()` has finished (train_step is fully asynchronous and
        reads its inputs late in the step, so allocating them on the consumer stream would let the next batch's
        resize kernels overwrite images that queued step kernels still read)."""
        with torch.cuda.stream(self.stream):
            outs = [torch.empty((self.B, self.S, self.S, 3), device=self.dev) for _ in range(5)]
            for v in range(5):
                for b in range(self.B):
                    self._load(self.files[v][index * self.B + b], outs[v][b])
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return tuple(outs), ev

    def take(self, prepared):
        """Hand a prepared batch to the current stream."""
        outs, ev = prepared
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for t in outs:
            t.record_stream(cur)
        return outs

    def batch(self, index):
        """Batch `index` (0-based) as five [B,S,S,3] tensors; prepared on the loader's stream."""
        return self.take(self.prepare(index))

    def __iter__(self):
        """One batch is always in preparation on the loader stream while the previous one is consumed."""
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
    trainer.length_dataset, trainer.loadedDataset = ds.n, ds
    return ds.n, ds
