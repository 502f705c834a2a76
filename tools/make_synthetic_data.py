"""Synthetic data files in the layouts the reference's loaders expect (omniglot/data.py:12-21,
mimgnet/data.py:12-20, celeba/data.py:13-38), at any size.  SURVEY.md §8d describes the full-size shapes.

  python tools/make_synthetic_data.py omniglot /tmp/omni --train-classes 60 --val-classes 30 --test-classes 30
"""
import argparse
import os

import numpy as np


def omniglot(out, ktr, kva, kte, rng):
    for name, k in (('train', ktr), ('val', kva), ('test', kte)):
        proto = rng.random((k, 1, 28, 28, 1)) < 0.15                     # one stroke pattern per class
        flip = rng.random((k, 20, 28, 28, 1)) < 0.03
        np.save(os.path.join(out, name + '.npy'), np.logical_xor(proto, flip).astype(np.float32))


def features(out, ktr, kva, kte, per_class, dim, rng, csv_labels):
    for name, k in (('train', ktr), ('val', kva), ('test', kte)):
        centers = rng.normal(size=(k, 1, dim))
        x = np.maximum(centers + 0.5 * rng.normal(size=(k, per_class, dim)), 0).astype(np.float32)
        np.save(os.path.join(out, name + '_features.npy'), x.reshape(k * per_class, dim))
        if csv_labels and name != 'train':
            with open(os.path.join(out, name + '.csv'), 'w') as f:
                f.write('filename,label\n')
                for c in range(k):
                    for i in range(per_class):
                        f.write('%s_%d_%d.jpg,id%d\n' % (name, c, i, c))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('variant', choices=['omniglot', 'mimgnet', 'celeba'])
    ap.add_argument('out')
    ap.add_argument('--train-classes', type=int, default=64)
    ap.add_argument('--val-classes', type=int, default=32)
    ap.add_argument('--test-classes', type=int, default=32)
    ap.add_argument('--per-class', type=int, default=600)
    ap.add_argument('--dim', type=int, default=256)
    ap.add_argument('--seed', type=int, default=10086)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    rng = np.random.default_rng(a.seed)
    if a.variant == 'omniglot':
        omniglot(a.out, a.train_classes, a.val_classes, a.test_classes, rng)
    else:
        features(a.out, a.train_classes, a.val_classes, a.test_classes, a.per_class, a.dim, rng, a.variant == 'celeba')
