"""CPU test of `python bench.py --gpus N` starting its own ranks (no torchrun around it): two gloo
ranks through the same entry the driver uses, with `--dry-run` standing in for the device work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ, **env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=e, capture_output=True,
                          text=True, timeout=300)


def test_bench_starts_its_own_ranks_gloo_world2():
    r = _run(['--gpus', '2', '--steps', '5', '--warmup', '1', '--dry-run'], CHS_DIST_BACKEND='gloo')
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                       # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['steps'] == 5
    assert d['energies_last_step'] == [[0.0, 2.0], [1.0, 2.0]]   # all_gather: one record per rank, in rank order
    assert 'dry-run' in d['data'] and d['value'] is None          # never mistaken for a measurement


def test_bench_fails_when_a_rank_fails():
    # the RCCL backend cannot come up without GPUs: every rank dies, the launcher must say so
    r = _run(['--gpus', '2', '--steps', '5', '--dry-run'], CHS_DIST_BACKEND='nccl')
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]


def test_bench_as_ranks_of_torch_distributed_run():
    """The driver's N>1 invocation: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
    (RANK/WORLD_SIZE come from the launcher: bench.py must not start ranks of its own then)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    e = dict(os.environ, CHS_DIST_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'),
                        '--gpus', '2', '--steps', '4', '--warmup', '1', '--dry-run'], env=e, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['n_gpus'] == 2


def test_one_dead_rank_ends_the_launch_quickly(tmp_path):
    """A rank that dies early (here: rank 1, before the rendezvous) must not leave its sibling waiting for the
    collective's own timeout: the launcher notices the first non-zero exit, ends the others and fails, and it
    shows every rank's stdout."""
    import time
    t0 = time.time()
    r = _run(['--gpus', '2', '--steps', '5', '--dry-run'], CHS_DIST_BACKEND='gloo', CHS_BENCH_TEST_DIE_RANK='1')
    assert r.returncode != 0
    assert time.time() - t0 < 120
    assert 'rank 1 exited with' in r.stderr


def test_ensemble_baseline_cpu_leg_runs_the_protocol():
    """`--ensemble-baseline` (CPU leg only under --dry-run): P oracle processes, 1 warm-up + 3 repetitions."""
    r = _run(['--ensemble-baseline', '--dry-run', '--ens-procs', '2', '--ens-steps', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    c = d['cpu_ensemble']
    assert c['cores'] == 2 and c['kind'] == 'port' and c['value'] > 0 and c['best'] >= c['value']
    assert '1 warm-up + 3 timed repetitions' in c['sample'] and 'gpu_ensemble' not in d
