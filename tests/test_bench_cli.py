"""bench.py's launcher logic on a box without a GPU: the --gpus flag is honoured (it tries to start ranks, i.e. it needs GPUs and says
so), and under a launcher (RANK / WORLD_SIZE set) it does not spawn again."""
import os
import subprocess
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_flag_without_gpu_fails_loudly():
    if torch.cuda.is_available():
        return
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'needs a GPU' in (r.stderr + r.stdout)


def test_no_respawn_under_a_launcher_or_for_one_gpu(monkeypatch):
    b = _load_bench()
    called = []
    monkeypatch.setattr(subprocess, 'call', lambda *a, **k: called.append(a) or 0)
    monkeypatch.setenv('RANK', '0')
    monkeypatch.setenv('WORLD_SIZE', '2')
    b.spawn_ranks_if_needed(types.SimpleNamespace(gpus=2))       # torchrun already started us
    monkeypatch.delenv('RANK')
    monkeypatch.delenv('WORLD_SIZE')
    b.spawn_ranks_if_needed(types.SimpleNamespace(gpus=1))       # single GPU: this process is the job
    assert not called


def test_respawn_command_is_one_rank_per_gpu_on_loopback(monkeypatch):
    b = _load_bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    monkeypatch.delenv('RANK', raising=False)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3'])
    try:
        b.spawn_ranks_if_needed(types.SimpleNamespace(gpus=4))
        raise AssertionError('expected SystemExit')
    except SystemExit as e:
        assert e.code == 7                                       # the children's exit code is passed on
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=4' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-4:] == ['--gpus', '4', '--steps', '3']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and seen['env']['ARCN_BENCH_SELF_SPAWNED'] == '1'
