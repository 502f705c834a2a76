"""`python -m cmvae_amd.main {omniglot,mimgnet,celeba} --data-dir D --save-dir S [flags]`

Flags and defaults are those of the reference's omniglot/main.py:23-49, mimgnet/main.py:24-50 and
celeba/main.py:24-52 (seed 10086 for numpy / torch / random).  wandb is not required."""
import argparse
import os
import random

import numpy as np
import torch

DEFAULTS = {
    'omniglot': dict(mc=32, batch=4, sample=200, iters=60000, way=20, lr=1e-3),
    'mimgnet': dict(mc=256, batch=16, sample=5, iters=100000, way=5, lr=1e-4),
    'celeba': dict(mc=256, batch=16, sample=5, iters=100000, way=5, lr=1e-4),
}


def build_parser(variant):
    d = DEFAULTS[variant]
    p = argparse.ArgumentParser('CMVAE for %s (MI355X)' % variant)
    p.add_argument('--data-dir', type=str, required=True)
    p.add_argument('--save-dir', type=str, required=True)
    p.add_argument('--unsupervised-em-iters', type=int, default=10)
    p.add_argument('--semisupervised-em-iters', type=int, default=10)
    p.add_argument('--fix-pi', action='store_true')
    if variant == 'omniglot':
        p.add_argument('--hidden-size', type=int, default=64)
    p.add_argument('--latent-size', type=int, default=64)
    p.add_argument('--train-mc-sample-size', type=int, default=d['mc'])
    p.add_argument('--test-mc-sample-size', type=int, default=d['mc'])
    p.add_argument('--batch-size', type=int, default=d['batch'])
    p.add_argument('--sample-size', type=int, default=d['sample'])
    p.add_argument('--train-iters', type=int, default=d['iters'])
    p.add_argument('--freq-iters', type=int, default=1000)
    p.add_argument('--eval-episodes', type=int, default=1000)
    p.add_argument('--eval-batch-size', type=int, default=None,
                   help='episodes per prediction call during evaluation (default: --batch-size, as the reference; '
                        '32 doubles the evaluation throughput on one MI355X)')
    p.add_argument('--dual-ascent', action='store_true',
                   help='run the augmented-Lagrangian schedule on (h_lambda, h_c) after every epoch: the block the '
                        'reference carries commented out (omniglot/trainer.py:171-177); off = the reference\'s live code')
    p.add_argument('--graph-step', choices=('auto', 'on', 'off'), default='auto',
                   help='replay the training step from a HIP graph (new; auto: the feature variants in a single '
                        'process, whose step is bound by the host\'s launch rate; same arithmetic as the eager step)')
    p.add_argument('--eval-streams', type=int, default=8,
                   help='HIP streams the independent prediction calls of an evaluation are dealt to (new; 1 = the '
                        'reference\'s sequential loop, same results)')
    if variant != 'omniglot':
        p.add_argument('--shot', type=int, default=1)
        p.add_argument('--seed', type=int, default=15)
    p.add_argument('--way', type=int, default=d['way'])
    p.add_argument('--query', type=int, default=15)
    p.add_argument('--lr', type=float, default=d['lr'])
    p.add_argument('--weight-decay', type=float, default=5e-5)
    p.add_argument('--debug', action='store_true')
    p.add_argument('--gpu-id', type=int, default=0)
    p.add_argument('--log-interval', type=int, default=50, help='steps between loss read-backs (new)')
    return p


def main(argv=None):
    top = argparse.ArgumentParser(add_help=False)
    top.add_argument('variant', choices=sorted(DEFAULTS))
    ns, rest = top.parse_known_args(argv)
    args = build_parser(ns.variant).parse_args(rest)
    seed = 10086                                   # the reference seeds with 10086 whatever --seed says
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    local = int(os.environ.get('LOCAL_RANK', args.gpu_id))
    args.device = torch.device('cuda', local)
    try:
        torch.cuda.set_device(args.device)
        torch.zeros(1, device=args.device)          # create the device context before the HIP library is loaded
    except Exception as e:
        raise SystemExit('cmvae_amd needs an MI355X GPU (there is no CPU path): %s' % e)
    os.makedirs(args.save_dir, exist_ok=True)
    from .trainer import Trainer
    return Trainer(args, ns.variant).train()


if __name__ == '__main__':
    main()
