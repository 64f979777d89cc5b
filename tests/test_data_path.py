"""(f.4) data path + checkpoint layout on CPU: the LMDB key / length contract of the reference's MultiResolutionDataset
(utils/dataset.py:9-45) against an in-memory environment, the torchvision-free transform, the prefetcher's pass-through."""
import io

import numpy as np
import pytest
import torch
from PIL import Image

from transeditor_amd.utils.dataset import DevicePrefetcher, MultiResolutionDataset, image_transform, sample_data


class _Txn:
    def __init__(self, store):
        self.store = store

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def get(self, key):
        return self.store.get(key)


class FakeEnv:
    """what lmdb.open(...) returns, reduced to the two calls the dataset makes"""

    def __init__(self, store):
        self.store = store

    def begin(self, write=False):
        assert write is False
        return _Txn(self.store)


def _png(arr):
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, format='PNG')
    return buf.getvalue()


def _store(n, res):
    rng = np.random.RandomState(0)
    imgs = [rng.randint(0, 256, size=(res, res, 3), dtype=np.uint8) for _ in range(n)]
    store = {f'{res}-{str(i).zfill(5)}'.encode(): _png(a) for i, a in enumerate(imgs)}
    store[b'length'] = str(n).encode()
    return store, imgs


def test_dataset_keys_length_and_transform():
    store, imgs = _store(5, 16)
    ds = MultiResolutionDataset(FakeEnv(store), image_transform(flip_probability=0.0), resolution=16)
    assert len(ds) == 5 and ds.key(140) == b'16-00140'
    for i in (0, 3):
        t = ds[i]
        assert t.shape == (3, 16, 16) and t.dtype == torch.float32
        want = torch.from_numpy(imgs[i].astype(np.float32)).permute(2, 0, 1) / 255.0
        assert torch.allclose(t, (want - 0.5) / 0.5, atol=1e-6)
        assert float(t.min()) >= -1.0 and float(t.max()) <= 1.0
    flipped = MultiResolutionDataset(FakeEnv(store), image_transform(flip_probability=1.0), 16)[2]
    assert torch.allclose(flipped, ds[2].flip(2))


def test_dataset_corrupt_sample_falls_back_like_the_reference(capsys):
    store, _ = _store(4, 8)
    store[b'8-00001'] = b'not an image'
    ds = MultiResolutionDataset(FakeEnv(store), image_transform(0.0), 8)
    t = ds[1]                                   # decode fails -> message printed, a random other sample is served
    assert t.shape == (3, 8, 8)
    assert capsys.readouterr().out.strip() != ''


def test_dataset_without_lmdb_package_fails_loudly():
    try:
        import lmdb  # noqa: F401
        pytest.skip('lmdb is installed here')
    except ImportError:
        pass
    with pytest.raises(ImportError, match='lmdb'):
        MultiResolutionDataset('/nonexistent/path', image_transform(), 256)


def test_sample_data_cycles_and_prefetcher_passthrough_on_cpu():
    store, _ = _store(6, 8)
    ds = MultiResolutionDataset(FakeEnv(store), image_transform(0.0), 8)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, drop_last=True)
    gen = sample_data(loader)
    got = [next(gen) for _ in range(5)]          # 3 batches per epoch: wraps around
    assert all(b.shape == (2, 3, 8, 8) for b in got) and torch.equal(got[0], got[3])
    pre = list(DevicePrefetcher([got[0], got[1]], 'cpu'))
    assert len(pre) == 2 and torch.equal(pre[1], got[1])


class _PidTransform:
    """image_transform + the decoding process's pid in element [0, 0, 0] (picklable: worker processes get a copy)"""

    def __init__(self):
        self.base = image_transform(0.0)

    def __call__(self, img):
        import os
        t = self.base(img)
        t[0, 0, 0] = float(os.getpid() % 100000)
        return t


def test_decode_runs_in_worker_processes():
    """(f.4) decode off the training process: `data_loader(num_workers=2)` decodes in DataLoader workers (the reference keeps
    num_workers=0, train_spatial_query.py:520-525), batches keep the reference's shape / drop_last semantics."""
    import os
    from transeditor_amd.utils.dataset import data_loader
    store, imgs = _store(12, 8)
    ds = MultiResolutionDataset(FakeEnv(store), _PidTransform(), 8)
    loader = data_loader(ds, batch_size=4, sampler=torch.utils.data.SequentialSampler(ds), num_workers=2)
    batches = list(loader)
    assert len(batches) == 3 and all(b.shape == (4, 3, 8, 8) for b in batches)
    pids = {int(b[i, 0, 0, 0]) for b in batches for i in range(4)}
    assert os.getpid() % 100000 not in pids and 1 <= len(pids) <= 2
    want = (torch.from_numpy(imgs[5].astype(np.float32)).permute(2, 0, 1) / 255.0 - 0.5) / 0.5
    got = batches[1][1].clone()
    got[0, 0, 0] = want[0, 0, 0]
    assert torch.allclose(got, want, atol=1e-6)
    same = data_loader(ds, batch_size=4, sampler=torch.utils.data.SequentialSampler(ds), num_workers=0)
    assert {int(b[0, 0, 0, 0]) for b in same} == {os.getpid() % 100000}
    del loader
