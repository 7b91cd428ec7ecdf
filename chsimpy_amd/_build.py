"""Building ``lib/libchs_hip.so`` and knowing what it was built from.

Provenance: the library carries, as a string constant, the sha256 of every file under ``csrc/`` and ``include/``
it was compiled from and the extra compiler flags of the build (``chs_version()`` returns them; the marker
``CHS_SRC_HASH=`` can be read from the file without loading it).  `build_hip()` rebuilds whenever that differs
from the tree -- modification times play no part -- and `_lib.load()` refuses (or rebuilds) a product library that
does not match, so an experiment variant left behind in ``lib/`` or a library older than its sources cannot ship
as the product.  Variants for A/B timing live under ``lib/variants/`` and are selected through ``CHS_LIB_PATH``
(tools/ab.sh), never by overwriting the product library.
"""
import glob
import hashlib
import os
import re
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
INCLUDE = os.path.join(ROOT, 'include')
LIBDIR = os.path.join(PKG, 'lib')
LIB = os.path.join(LIBDIR, 'libchs_hip.so')

# -DCHS_TEST_HOOKS=1: the environment-variable test hooks of the library (include/chs_hip.h) are compiled in; the
# GPU suite exercises them on this very library.
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
               '-Wno-unused-value', '-Wno-unused-result', '-Wno-pass-failed', '-DCHS_TEST_HOOKS=1',
               '-I' + INCLUDE, '-I' + CSRC]

_MARK = re.compile(rb'CHS_SRC_HASH=([0-9a-f]{64});CHS_FLAGS=([^;\x00]*);')


def source_files():
    return (sorted(glob.glob(os.path.join(CSRC, '*.hip'))) + sorted(glob.glob(os.path.join(CSRC, '*.h')))
            + sorted(glob.glob(os.path.join(INCLUDE, '*.h'))))


def source_hash():
    """sha256 over (relative name, content) of every source the library is compiled from, plus the base flags;
    None when the sources are not there (an installed copy without them cannot be checked)."""
    files = source_files()
    if not files:
        return None
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode() + b'\0')
        with open(f, 'rb') as fh:
            h.update(fh.read())
        h.update(b'\0')
    h.update(' '.join(f for f in HIPCC_FLAGS if not f.startswith('-I')).encode())
    return h.hexdigest()


def embedded_provenance(path):
    """(source hash, extra flags) a built library carries, or (None, None)."""
    try:
        with open(path, 'rb') as fh:
            m = _MARK.search(fh.read())
    except OSError:
        return None, None
    if not m:
        return None, None
    return m.group(1).decode(), m.group(2).decode()


def is_current(path=LIB, extra=''):
    """True when `path` was built from the tree as it stands, with exactly the extra flags `extra`."""
    want = source_hash()
    have, flags = embedded_provenance(path)
    return want is not None and have == want and (flags or '') == extra.strip()


def build_hip(force=False, verbose=False, out=None, extra=None):
    """hipcc --offload-arch=gfx950: every .hip translation unit to an object file, in parallel (the kernel
    instantiations of the two element types are translation units of their own), then one shared library.
    Rebuilds when the library does not carry the hash of the present sources and flags."""
    from concurrent.futures import ThreadPoolExecutor
    out = out or LIB
    extra = (extra or '').strip()   # (experiment variants only: tools/build_variant.sh; the product has none)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not force and is_current(out, extra):
        return out
    # One builder at a time: several ranks of one launch may find a stale library at once (torch.distributed.run starts
    # them together); the others wait for the lock and then find the library current.
    import fcntl
    with open(os.path.join(LIBDIR, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and is_current(out, extra):
                return out
            return _build_locked(out, extra, force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(out, extra, force, verbose):
    from concurrent.futures import ThreadPoolExecutor
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    hipcc = os.environ.get('HIPCC', 'hipcc')
    objdir = os.path.join(LIBDIR, 'obj' if out == LIB else 'obj_' + os.path.basename(out))
    os.makedirs(objdir, exist_ok=True)
    stamp = f'CHS_SRC_HASH={source_hash()};CHS_FLAGS={extra};'
    cflags = [f for f in HIPCC_FLAGS if f != '-shared'] + extra.split()
    # object cache: a translation unit is recompiled when its own text, any header or the flags changed (only
    # chs_api.hip, which defines chs_version(), sees the provenance string, so the others survive unrelated edits)
    hh = hashlib.sha256(' '.join(cflags).encode())
    for f in sorted(glob.glob(os.path.join(CSRC, '*.h'))) + sorted(glob.glob(os.path.join(INCLUDE, '*.h'))):
        with open(f, 'rb') as fh:
            hh.update(fh.read())
    # (.hip files that other translation units include, e.g. the stamps build of chs_fast.hip)
    for f in srcs:
        with open(f, 'rb') as fh:
            txt = fh.read()
        if b'#include "' + os.path.basename(f).encode() in b''.join(open(g, 'rb').read() for g in srcs if g != f):
            hh.update(txt)
    hdr_hash = hh.hexdigest()

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + '.o')
        with open(src, 'rb') as fh:
            key = hashlib.sha256(hdr_hash.encode() + fh.read()).hexdigest()
        own = cflags
        if os.path.basename(src) == 'chs_api.hip':
            own = cflags + ['-DCHS_PROVENANCE="' + stamp + '"']
            key = hashlib.sha256((key + stamp).encode()).hexdigest()
        keyfile = obj + '.key'
        if not force and os.path.exists(obj) and os.path.exists(keyfile) and open(keyfile).read() == key:
            return obj
        cmd = [hipcc] + own + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
        with open(keyfile, 'w') as fh:
            fh.write(key)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as pool:
        objs = list(pool.map(compile_one, srcs))
    tmp = out + '.tmp%d' % os.getpid()
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', tmp]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(tmp, out)   # never a half-written product library
    return out
