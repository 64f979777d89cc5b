"""Image data path of the train step: LMDB -> decoded fp32 batch -> device, off the critical path.

Reference: utils/dataset.py:9-45 (`MultiResolutionDataset`: LMDB environment, key f'{resolution}-{index:05d}' -> encoded
image bytes, 'length' -> sample count, a failed decode falls back to a random other sample) and the loader set-up of
train_spatial_query.py:511-525 (RandomHorizontalFlip -> ToTensor -> Normalize(0.5, 0.5), `num_workers=0`, so decode and
the host->device copy sit inside the training iteration there).  Here:

* `MultiResolutionDataset` keeps the reference's name, constructor and key format (a dataset prepared for the reference
  is read unchanged).  `lmdb` is imported when the dataset is opened (it is not a dependency of the kernels); an already
  opened environment can be passed instead of a path.
* `image_transform` is the torchvision-free equivalent of the reference's transform pipeline.
* `DevicePrefetcher` moves batches to the GPU through pinned memory on a side HIP stream, one batch ahead, so the copy of
  batch i+1 overlaps the kernels of iteration i (the MI355X train step is GPU-bound: 201 ms per iteration, profiles/).
"""
import random
from io import BytesIO

import numpy as np
import torch
from torch.utils.data import Dataset


def image_transform(flip_probability=0.5):
    """PIL image -> fp32 [3, H, W] in [-1, 1]: RandomHorizontalFlip -> ToTensor -> Normalize((.5,.5,.5), (.5,.5,.5))."""
    def run(img):
        a = np.asarray(img.convert('RGB'), dtype=np.float32)            # [H, W, 3] in 0..255
        if flip_probability > 0 and random.random() < flip_probability:
            a = a[:, ::-1]
        t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1)
        return t.div_(255.0).sub_(0.5).div_(0.5)
    return run


def _open_lmdb(path):
    try:
        import lmdb
    except ImportError as e:          # same failure point as the reference's module-level import
        raise ImportError('MultiResolutionDataset reads an LMDB environment: the `lmdb` package is required') from e
    env = lmdb.open(path, max_readers=32, readonly=True, lock=False, readahead=False, meminit=False)
    if not env:
        raise IOError('Cannot open lmdb dataset', path)
    return env


class MultiResolutionDataset(Dataset):
    def __init__(self, path, transform, resolution=256):
        # An LMDB environment must not cross a fork / be pickled into DataLoader workers: the path is kept and every
        # process opens its own handle on first use (the reference runs num_workers=0, train_spatial_query.py:520-525, and
        # never meets the problem; here decode runs in worker processes, see `data_loader`).
        self._path = path if isinstance(path, (str, bytes)) else None
        self._pid = None
        self._env = None if self._path is not None else path       # an opened environment (.begin(write=False) -> txn.get(key))
        with self.env.begin(write=False) as txn:
            self.length = int(txn.get('length'.encode('utf-8')).decode('utf-8'))
        if self._path is not None:
            # the handle that read `length` must not survive into fork-started workers (a child would hold the inherited
            # environment AND open a second one on the same path - both unsupported by LMDB): every process, this one
            # included, opens lazily on its first __getitem__
            close = getattr(self._env, 'close', None)
            if close is not None:
                close()
            self._env, self._pid = None, None
        self.resolution = resolution
        self.transform = transform

    @property
    def env(self):
        import os
        if self._path is not None and (self._env is None or self._pid != os.getpid()):
            self._env, self._pid = _open_lmdb(self._path), os.getpid()
        return self._env

    def __getstate__(self):                   # (spawned workers: the handle is re-opened on the other side)
        st = dict(self.__dict__)
        if self._path is not None:
            st['_env'], st['_pid'] = None, None
        return st

    def __len__(self):
        return self.length

    def key(self, index):
        return f'{self.resolution}-{str(index).zfill(5)}'.encode('utf-8')      # b'256-00140'

    def __getitem__(self, index):
        from PIL import Image
        with self.env.begin(write=False) as txn:
            img_bytes = txn.get(self.key(index))
        try:
            img = Image.open(BytesIO(img_bytes))
            return self.transform(img)
        except Exception as e:                # reference behaviour: report, then serve a random other sample
            print(e)
            return self.__getitem__(random.randint(0, self.length - 1))


def data_loader(dataset, batch_size, sampler=None, num_workers=4, drop_last=True):
    """The reference's loader (train_spatial_query.py:520-525: batch_size, sampler, drop_last=True) with the decode moved
    OFF the training process: `num_workers` worker processes (kept alive between epochs, two batches each in flight) decode
    and normalise, pinned batches come back through shared memory; wrap the result in `DevicePrefetcher` for the
    asynchronous host->device copy.  num_workers=0 reproduces the reference's in-process behaviour."""
    from torch.utils.data import DataLoader
    kw = dict(batch_size=batch_size, sampler=sampler, shuffle=sampler is None, drop_last=drop_last, num_workers=num_workers,
              pin_memory=torch.cuda.is_available())
    if num_workers > 0:
        kw.update(persistent_workers=True, prefetch_factor=2)
    return DataLoader(dataset, **kw)


def sample_data(loader):
    """train_spatial_query.py:64-67: cycle over the loader forever."""
    while True:
        for batch in loader:
            yield batch


class DevicePrefetcher:
    """Iterate `batches` (an iterable of CPU tensors) as device tensors, one batch ahead: pinned staging buffer ->
    asynchronous copy on a side stream -> the consumer's stream waits on the copy's event only."""

    def __init__(self, batches, device):
        self.it = iter(batches)
        self.device = torch.device(device)
        self.cuda = self.device.type == 'cuda'
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None
        self._next = None
        self._preload()

    def _preload(self):
        try:
            cpu = next(self.it)
        except StopIteration:
            self._next = None
            return
        if not self.cuda:
            self._next = (cpu, None)
            return
        pinned = cpu if cpu.is_pinned() else cpu.pin_memory()
        with torch.cuda.stream(self.stream):
            dev = pinned.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._next = (dev, ev, pinned)        # the pinned buffer stays alive until the copy has been consumed

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        cur = self._next
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_event(cur[1])
            cur[0].record_stream(torch.cuda.current_stream(self.device))
        self._preload()
        return cur[0]
