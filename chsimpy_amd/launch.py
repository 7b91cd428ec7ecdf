"""Starting and placing the ranks of an ensemble: one process per GPU.

The reference starts its own worker pool, sized to the physical cores (chsimpy/experiment.py:197-216).  The
counterpart here is a parent process that has NOT touched the GPU and starts one ordinary child process per GPU
(`spawn_ranks`; never an exec from a process that has initialised the device), and ranks that restrict themselves
to their own share of the host cores before their first GPU call (`pin_rank_to_cores`): eight ranks whose runtime
threads all roam over every core of a node -- or sit inside one CPU quota -- get in the way of the threads that
issue the kernel launches.
"""
import os
import socket
import subprocess
import sys
import tempfile
import time


def allowed_cores():
    try:
        return sorted(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover - not Linux
        return list(range(os.cpu_count() or 1))


def core_slices(cores, world):
    """Deal `cores` (sorted ids) to `world` ranks in contiguous, near-equal slices; with fewer cores than ranks every
    rank gets one core (shared round-robin)."""
    cores = list(cores)
    n = len(cores)
    if n == 0 or world <= 0:
        return [[] for _ in range(max(world, 0))]
    if n < world:
        return [[cores[r % n]] for r in range(world)]
    base, extra = divmod(n, world)
    out, at = [], 0
    for r in range(world):
        k = base + (1 if r < extra else 0)
        out.append(cores[at:at + k])
        at += k
    return out


def pin_rank_to_cores(local_rank, world, cores=None):
    """Restrict this process to its slice of the host cores.  Call it BEFORE anything touches the GPU (the HIP
    runtime starts its helper threads with the affinity of the thread that initialises it).  Returns the core set,
    or None when pinning is switched off (CHS_PIN_CORES=0), pointless (one rank) or not permitted."""
    if world <= 1 or os.environ.get('CHS_PIN_CORES', '1') == '0':
        return None
    mine = core_slices(allowed_cores() if cores is None else cores, world)[local_rank % world]
    if not mine:
        return None
    try:
        os.sched_setaffinity(0, mine)
    except (AttributeError, OSError):  # pragma: no cover - platform / container policy
        return None
    return mine


def quiet_host_threads():
    """One intra-op thread for torch: the collectives move a few scalars, and a pool sized for every core of the host,
    spinning inside a CPU quota, throttles the thread that issues the kernel launches."""
    import torch
    torch.set_num_threads(1)
    os.environ.setdefault('OMP_NUM_THREADS', '1')


def free_port():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def spawn_ranks(argv, nranks, timeout_s=1500.0, env_extra=None):
    """Start `nranks` child processes running `argv` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
    MASTER_PORT set, poll them all, end the siblings of the first one that fails, bound the whole launch.
    Returns (failed, codes, texts): `failed` is None or a reason, `texts[r]` rank r's stdout (kept in a file: no
    pipe to fill up).  The caller must not have initialised the GPU."""
    port = free_port()
    procs, outs = [], []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if env_extra:
            env.update(env_extra)
        outs.append(tempfile.TemporaryFile(mode='w+'))
        procs.append(subprocess.Popen(list(argv), env=env, stdout=outs[-1], text=True))
    deadline = time.time() + float(timeout_s)
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        bad = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad:
            failed = f'rank {bad[0]} exited with {procs[bad[0]].returncode}'
        elif time.time() > deadline:
            failed = 'timeout'
        else:
            time.sleep(0.05)
    if failed is not None:
        for p in procs:          # our own children, by handle
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    codes = [p.returncode for p in procs]
    if failed is None and any(codes):
        failed = 'a rank failed'
    texts = []
    for f in outs:
        f.seek(0)
        texts.append(f.read())
        f.close()
    return failed, codes, texts


def report_failure(prog, failed, codes, texts):
    sys.stderr.write(f'{prog}: {failed}; ranks exited with {codes}\n')
    for r, t in enumerate(texts):
        if t:
            sys.stderr.write(f'--- stdout of rank {r} ---\n{t}')
