"""Host logic of the episode pipeline (cmvae_amd/data.py) on the CPU: file layouts, RNG consumption order and
label structure of omniglot/data.py:27-55, mimgnet/data.py:26-54, celeba/data.py:61-91."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(tmp_path, variant, *extra):
    out = str(tmp_path / variant)
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'make_synthetic_data.py'), variant, out,
                           '--train-classes', '6', '--val-classes', '7', '--test-classes', '9', *extra])
    return out


def _episode_loop(pool_by_class, way, shot, query, n_episodes):
    """Literal double loop: per episode choose classes, per class choose instances (the reference's order)."""
    xtr, xte = [], []
    for _ in range(n_episodes):
        cls = np.random.choice(range(len(pool_by_class)), size=way, replace=False)
        a, b = [], []
        for k in list(cls):
            idx = np.random.choice(range(len(pool_by_class[k])), size=shot + query, replace=False)
            xk = pool_by_class[k][idx]
            a.append(xk[:shot]); b.append(xk[shot:])
        xtr.append(np.concatenate(a, 0)); xte.append(np.concatenate(b, 0))
    return np.stack(xtr, 0), np.stack(xte, 0)


def test_omniglot_episodes_follow_the_numpy_stream(tmp_path):
    from cmvae_amd.data import EpisodeData
    d = _make(tmp_path, 'omniglot')
    np.random.seed(5)
    data = EpisodeData(d, 'omniglot')
    assert data.x_mtr.shape == (6 * 20, 1, 28, 28) and data.n_classes(False) == 7 and data.n_classes(True) == 9
    state = np.random.get_state()
    X_tr, y_tr, X_te, y_te = data.generate_test_episode(way=5, shot=1, query=3, n_episodes=3, test=True)
    np.random.set_state(state)
    pool = np.transpose(np.load(os.path.join(d, 'test.npy')), (0, 1, 4, 2, 3))
    rtr, rte = _episode_loop(list(pool), 5, 1, 3, 3)
    assert X_tr.shape == (3, 5, 1, 28, 28) and X_te.shape == (3, 15, 1, 28, 28)
    assert np.array_equal(X_tr.numpy(), rtr) and np.array_equal(X_te.numpy(), rte)
    assert y_tr.dtype == torch.int64 and y_tr[0].tolist() == [0, 1, 2, 3, 4]
    assert y_te[1].tolist() == [c for c in range(5) for _ in range(3)]


def test_feature_pools_and_celeba_grouping(tmp_path):
    from cmvae_amd.data import EpisodeData
    d = _make(tmp_path, 'mimgnet', '--per-class', '30', '--dim', '64')
    np.random.seed(1)
    data = EpisodeData(d, 'mimgnet', instances_per_class=30)
    assert data.input_shape == 64 and data.n_classes(True) == 9
    X_tr, y_tr, X_te, y_te = data.generate_test_episode(5, 5, 15, n_episodes=2, test=False)
    assert X_tr.shape == (2, 25, 64) and X_te.shape == (2, 75, 64)
    with pytest.raises(ValueError):
        EpisodeData(d, 'mimgnet', instances_per_class=600)

    d = _make(tmp_path, 'celeba', '--per-class', '25', '--dim', '64')
    np.random.seed(2)
    data = EpisodeData(d, 'celeba')             # burns one episode at construction, as the reference does
    state = np.random.get_state()
    X_tr, _, X_te, _ = data.generate_test_episode(5, 5, 15, n_episodes=2, test=True)
    np.random.set_state(state)
    pool = np.load(os.path.join(d, 'test_features.npy')).reshape(9, 25, 64)
    rtr, rte = _episode_loop(list(pool), 5, 5, 15, 2)
    assert np.array_equal(X_tr.numpy(), rtr) and np.array_equal(X_te.numpy(), rte)


def test_celeba_drops_small_identities(tmp_path):
    from cmvae_amd.data import EpisodeData
    d = _make(tmp_path, 'celeba', '--per-class', '20', '--dim', '64')      # 20 images: not "> 20"
    np.random.seed(3)
    with pytest.raises(ValueError):
        EpisodeData(d, 'celeba')                 # no identity is large enough to sample from


def test_train_stream_is_a_permutation_per_epoch():
    from cmvae_amd.data import TrainStream
    torch.manual_seed(0)
    pool = torch.arange(10.)[:, None]
    st = TrainStream(pool, 4)
    a, b = st.next(), st.next()
    assert len(set(a.flatten().tolist() + b.flatten().tolist())) == 8
    c = st.next()                                 # 2 items left < 4: new epoch (drop_last)
    assert c.shape == (4, 1)
    with pytest.raises(ValueError):
        TrainStream(pool, 11)


def test_cli_defaults_match_the_reference_flags():
    from cmvae_amd.main import build_parser
    a = build_parser('omniglot').parse_args(['--data-dir', 'x', '--save-dir', 'y'])
    assert (a.batch_size, a.sample_size, a.way, a.train_mc_sample_size, a.lr, a.weight_decay, a.train_iters,
            a.unsupervised_em_iters, a.hidden_size, a.query, a.eval_episodes) == (4, 200, 20, 32, 1e-3, 5e-5, 60000, 10,
                                                                                  64, 15, 1000)
    a = build_parser('mimgnet').parse_args(['--data-dir', 'x', '--save-dir', 'y'])
    assert (a.batch_size, a.sample_size, a.way, a.train_mc_sample_size, a.lr, a.train_iters, a.shot, a.seed) == \
        (16, 5, 5, 256, 1e-4, 100000, 1, 15)


def test_dual_ascent_schedule():
    """The augmented-Lagrangian schedule of omniglot/trainer.py:171-177 (commented out in the reference; offered
    behind --dual-ascent): h_c grows tenfold while h does not shrink to a quarter, the multiplier takes h_c * h."""
    from cmvae_amd.trainer import dual_ascent_update
    hl, hc, ho = 0.1, 0.1, float('inf')
    hl, hc, ho = dual_ascent_update(0.08, hl, hc, ho)              # first epoch: h_old = inf, no growth
    assert (hc, ho) == (0.1, 0.08) and abs(hl - (0.1 + 0.1 * 0.08)) < 1e-12
    hl2, hc2, ho2 = dual_ascent_update(0.05, hl, hc, ho)           # 0.05 > 0.25 * 0.08: penalty grows, then the step
    assert hc2 == 1.0 and abs(hl2 - (hl + 1.0 * 0.05)) < 1e-12 and ho2 == 0.05
    hl3, hc3, ho3 = dual_ascent_update(0.01, hl2, hc2, ho2)        # shrank below a quarter: penalty kept
    assert hc3 == 1.0 and abs(hl3 - (hl2 + 0.01)) < 1e-12
    assert dual_ascent_update(1e-9, 0.3, 10.0, 0.01) == (0.3, 10.0, 1e-9)     # below h_tol: nothing moves
    assert dual_ascent_update(1.0, 0.0, 1e11, 1.0)[1] == 1e11                  # capped at h_c_max
