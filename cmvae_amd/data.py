"""Episode data for the three CMVAE variants (omniglot/data.py:5-55, mimgnet/data.py:5-54,
celeba/data.py:7-91), kept resident on the GPU.

What stays as in the reference: file names and array layouts, the shuffle of the meta-train pool at load
time, and the order in which `generate_test_episode` consumes numpy's global RNG (one `choice` of classes
per episode, then one `choice` of instances per class) -- a seeded run draws the same episodes.
What changes: the pools are uploaded once; an episode batch is a host-side index draw
followed by ONE device gather instead of a Python double loop of numpy copies and a fresh H2D copy per call;
class counts come from the arrays instead of being hard-coded (4800/400/1292, 64/16/20), so small synthetic
pools work.
"""
import csv
import os

import numpy as np
import torch


def _group_by_label(csv_path):
    """filename,label rows -> list of group sizes in first-appearance order (celeba/data.py:46-59)."""
    sizes, order = {}, []
    with open(csv_path) as f:
        rd = csv.reader(f, delimiter=',')
        next(rd, None)
        for row in rd:
            if row[1] not in sizes:
                sizes[row[1]] = 0
                order.append(row[1])
            sizes[row[1]] += 1
    return [sizes[k] for k in order]


class EpisodeData:
    """variant: 'omniglot' | 'mimgnet' | 'celeba'."""

    def __init__(self, data_dir, variant='omniglot', device=None, instances_per_class=None):
        self.variant = variant
        self.device = torch.device(device) if device is not None else None
        ld = lambda n: np.load(os.path.join(data_dir, n))  # noqa: E731   (allow_pickle stays False)
        if variant == 'omniglot':
            x_mtr = np.transpose(ld('train.npy'), (0, 1, 4, 2, 3))                 # [K,20,1,28,28]
            self.N = x_mtr.shape[1]
            x_mtr = np.reshape(x_mtr, [-1, 1, 28, 28])
            np.random.shuffle(x_mtr)
            pools = {False: np.transpose(ld('val.npy'), (0, 1, 4, 2, 3)),
                     True: np.transpose(ld('test.npy'), (0, 1, 4, 2, 3))}
            self._groups = {k: [v.shape[1]] * v.shape[0] for k, v in pools.items()}
            pools = {k: v.reshape(-1, 1, 28, 28) for k, v in pools.items()}
            self.input_shape = [1, 28, 28]
        elif variant == 'mimgnet':
            x_mtr = ld('train_features.npy')
            np.random.shuffle(x_mtr)
            self.N = instances_per_class or 600
            pools = {False: ld('val_features.npy'), True: ld('test_features.npy')}
            for k, v in pools.items():
                if v.shape[0] % self.N:
                    raise ValueError('%d feature rows are not a multiple of %d per class' % (v.shape[0], self.N))
            self._groups = {k: [self.N] * (v.shape[0] // self.N) for k, v in pools.items()}
            self.input_shape = int(x_mtr.shape[-1])
        elif variant == 'celeba':
            x_mtr = ld('train_features.npy')
            np.random.shuffle(x_mtr)
            pools = {False: ld('val_features.npy'), True: ld('test_features.npy')}
            self._groups = {False: _group_by_label(os.path.join(data_dir, 'val.csv')),
                            True: _group_by_label(os.path.join(data_dir, 'test.csv'))}
            self.input_shape = int(x_mtr.shape[-1])
        else:
            raise ValueError('unknown variant %r' % variant)
        self.x_mtr = x_mtr
        # class -> (first row, size); celeba keeps only identities with more than 20 images (celeba/data.py:16,25)
        self._classes = {}
        for k, sizes in self._groups.items():
            starts = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
            keep = [(int(s), int(n)) for s, n in zip(starts, sizes) if variant != 'celeba' or n > 20]
            self._classes[k] = keep
        self._pool_host = pools
        self._pool_dev = {}
        self._train_dev = None
        if variant == 'celeba':
            self.generate_test_episode(5, 5, 15)          # the reference burns one episode here (celeba/data.py:44)

    # ------------------------------------------------------------------ pools
    def n_classes(self, test):
        return len(self._classes[bool(test)])

    def _pool(self, test):
        test = bool(test)
        if self.device is None:
            return torch.from_numpy(np.ascontiguousarray(self._pool_host[test])).float()
        if test not in self._pool_dev:
            self._pool_dev[test] = torch.from_numpy(np.ascontiguousarray(self._pool_host[test])).float().to(self.device)
        return self._pool_dev[test]

    def train_pool(self):
        if self._train_dev is None:
            t = torch.from_numpy(np.ascontiguousarray(self.x_mtr)).float()
            self._train_dev = t if self.device is None else t.to(self.device)
        return self._train_dev

    # ------------------------------------------------------------------ evaluation episodes
    def draw_episode_indices(self, way, shot, query, n_episodes=1, test=False):
        """Row indices into the val/test pool, [n_episodes, way, shot+query]; numpy RNG order of the reference:
        per episode one choice of `way` classes, then per class one choice of shot+query instances."""
        classes = self._classes[bool(test)]
        out = np.empty((n_episodes, way, shot + query), dtype=np.int64)
        for t in range(n_episodes):
            cls = np.random.choice(range(len(classes)), size=way, replace=False)
            for i, k in enumerate(list(cls)):
                start, n = classes[k]
                out[t, i] = start + np.random.choice(range(n), size=shot + query, replace=False)
        return out

    def generate_test_episode(self, way, shot, query, n_episodes=1, test=False):
        """-> [X_tr [E, way*shot, ...], y_tr [E, way*shot] int64, X_te [E, way*query, ...], y_te] as torch
        tensors on `device` (the reference returns numpy arrays that the trainer uploads)."""
        idx = self.draw_episode_indices(way, shot, query, n_episodes, test)
        pool = self._pool(test)
        flat = torch.from_numpy(idx).to(pool.device)
        tr = pool.index_select(0, flat[:, :, :shot].reshape(-1)).view(n_episodes, way * shot, *pool.shape[1:])
        te = pool.index_select(0, flat[:, :, shot:].reshape(-1)).view(n_episodes, way * query, *pool.shape[1:])
        y = torch.arange(way, device=pool.device)
        return [tr, y.repeat_interleave(shot)[None].expand(n_episodes, -1).contiguous(),
                te, y.repeat_interleave(query)[None].expand(n_episodes, -1).contiguous()]


class TrainStream:
    """The reference's DataLoader(TensorDataset(x_mtr), batch_size=B*S, shuffle=True, drop_last=True) that is
    re-created on exhaustion (omniglot/trainer.py:23-32, 82-86), as a device gather: one host permutation per
    epoch, one index_select per step.  Same sampling law (uniform without replacement per epoch); the
    permutation comes from torch.randperm on the global CPU generator, not from DataLoader's private one."""

    def __init__(self, pool, items_per_step):
        self.pool, self.n = pool, items_per_step
        if pool.shape[0] < items_per_step:
            raise ValueError('meta-train pool of %d items is smaller than one step (%d)' % (pool.shape[0], items_per_step))
        self._perm, self._pos = None, 0

    def next(self):
        if self._perm is None or self._pos + self.n > self._perm.numel():
            self._perm = torch.randperm(self.pool.shape[0]).to(self.pool.device)
            self._pos = 0
        idx = self._perm[self._pos:self._pos + self.n]
        self._pos += self.n
        return self.pool.index_select(0, idx)
