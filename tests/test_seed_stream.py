"""The EM seed stream of the library (cmvae_em_seed_draw_host, cmvae_amd/dist.py: SeedStream) against numpy itself:
the reference draws `np.random.choice(N, K, replace=False)` per episode from numpy's global generator
(omniglot/model.py:351-356); the restated algorithm must return the same indices AND leave the generator in the same
state, in every interleaving with other consumers of the stream."""
import numpy as np
import pytest

from cmvae_amd import dist as cdist


@pytest.mark.parametrize('n,k,b', [(9600, 20, 4), (1280, 5, 16), (6400, 5, 32), (7, 7, 3), (1, 1, 2), (300, 1, 5),
                                   (70000, 3, 2)])
def test_native_draw_equals_numpy_choice(n, k, b):
    for seed in (10086, 1, 2 ** 31 - 5):
        np.random.seed(seed)
        np.random.rand(seed % 7)                    # start somewhere inside a block of the generator
        want = cdist._draw_em_init_numpy(n, k, b)
        after = np.random.get_state()
        np.random.seed(seed)
        np.random.rand(seed % 7)
        got = cdist.SeedStream().draw(n, k, b)
        now = np.random.get_state()
        assert got.dtype == np.int64 and np.array_equal(got, want)
        assert now[2] == after[2] and np.array_equal(now[1], after[1])
    # the gaussian cache of the legacy generator survives the draw
    np.random.seed(3)
    np.random.randn()                               # leaves one cached normal
    a = np.random.get_state()
    cdist.SeedStream().draw(50, 3, 2)
    b_ = np.random.get_state()
    assert a[3] == b_[3] == 1 and a[4] == b_[4]


def test_prefetched_stream_is_the_sequential_stream():
    """Prefetching one step ahead, with an evaluation-like consumer of numpy's stream in between at some steps."""
    n, k, b = 640, 5, 8

    def sequential():
        np.random.seed(77)
        out = []
        for step in range(12):
            out.append(cdist._draw_em_init_numpy(n, k, b))
            if step % 4 == 3:                        # an epoch boundary: evaluation draws episodes from the stream
                out.append(np.random.choice(range(100), size=5, replace=False))
        out.append(np.random.rand())
        return out

    want = sequential()
    np.random.seed(77)
    ss = cdist.SeedStream()
    got = []
    for step in range(12):
        # the caller may or may not know that the epoch ends: both must give the sequential stream
        got.append(ss.next(n, k, b, prefetch=(step % 8 != 3)))
        if step % 4 == 3:
            got.append(np.random.choice(range(100), size=5, replace=False))
    got.append(np.random.rand())
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert np.array_equal(np.asarray(g), np.asarray(w))


def test_shape_change_voids_the_prefetch():
    np.random.seed(5)
    want = [cdist._draw_em_init_numpy(100, 4, 3), cdist._draw_em_init_numpy(200, 2, 5)]
    np.random.seed(5)
    ss = cdist.SeedStream()
    got = [ss.next(100, 4, 3, prefetch=True), ss.next(200, 2, 5)]
    assert all(np.array_equal(g, w) for g, w in zip(got, want))


def test_rejects_bad_shapes():
    with pytest.raises(RuntimeError):
        cdist.SeedStream().draw(5, 6, 1)
