"""CPU: `python bench.py --gpus N` without a torchrun environment starts its own ranks (the driver calls
it exactly like that) — the command it builds, the refusal when fewer GPUs are visible, and that nothing touches
the GPU in the parent."""
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_builds_a_torchrun_child(monkeypatch):
    import torch
    import bench
    calls = {}
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    monkeypatch.setattr(bench.subprocess, 'call', lambda cmd, env=None: calls.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '7', '--warmup', '2'])
    rc = bench.self_launch(types.SimpleNamespace(gpus=4))
    assert rc == 0
    cmd = calls['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '4', '--steps', '7', '--warmup', '2']
    assert os.path.samefile(cmd[-7], os.path.join(ROOT, 'bench.py'))
    assert calls['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_self_launch_refuses_more_ranks_than_gpus(monkeypatch, capsys):
    import torch
    import bench
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 1)
    monkeypatch.setattr(bench.subprocess, 'call', lambda *a, **k: pytest.fail('must not spawn'))
    assert bench.self_launch(types.SimpleNamespace(gpus=2)) == 2
    assert 'only 1 GPU' in capsys.readouterr().err


def test_main_routes_multi_gpu_requests_to_the_launcher(monkeypatch):
    import bench
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2'])
    monkeypatch.setattr(bench, 'self_launch', lambda args: 17)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 17


def test_strong_scaling_flags_reach_the_ranks(monkeypatch):
    """--scaling strong --global-batch 32 is part of the command line every rank re-parses."""
    import torch
    import bench
    calls = {}
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    monkeypatch.setattr(bench.subprocess, 'call', lambda cmd, env=None: calls.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--scaling', 'strong', '--global-batch', '32'])
    assert bench.self_launch(types.SimpleNamespace(gpus=8)) == 0
    assert calls['cmd'][-6:] == ['--gpus', '8', '--scaling', 'strong', '--global-batch', '32']
    assert 'GPU_MAX_HW_QUEUES' not in calls['env'] or calls['env']['GPU_MAX_HW_QUEUES'] == os.environ.get('GPU_MAX_HW_QUEUES')


def test_rank_count_mismatch_is_refused(monkeypatch):
    import bench
    monkeypatch.setenv('WORLD_SIZE', '1')
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '3', '--scaling', 'strong', '--global-batch', '32'])
    with pytest.raises(SystemExit):
        bench.main()                     # 3 ranks requested in a 1-rank job: refused before anything is built
