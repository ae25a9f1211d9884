"""Build the HIP library in-tree (the .so travels to the GPU box with the snapshot).

    python -m arpeggio_amd.build        # or arpeggio_amd.build.build()

hipcc cross-compiles gfx950 without a GPU.  -ffp-contract=off is part of the numerics
contract (csrc/arp_numerics.h): the only fused operations are explicit fma() calls.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libarpeggio_hip.so')
SOURCES = ['arp_api.hip']
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith('.h'))
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
         '-Wall', '-Wno-unused-function']


def hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found')


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(HERE, '..', 'include', 'arpeggio_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not is_stale():
        build_pyexport()
        return LIB
    # ARP_EXTRA_HIPCC_FLAGS: developer builds only (e.g. -DARP_SEARCH_TRACE, -DHOME_CELLS=1)
    cmd = [hipcc()] + FLAGS + os.environ.get('ARP_EXTRA_HIPCC_FLAGS', '').split() + [os.path.join(CSRC, s) for s in SOURCES] + ['-o', LIB]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    build_pyexport(force)
    build_host(force)
    return LIB


HOST_LIB = os.path.join(CSRC, 'libarpeggio_host.so')
HOST_DEPS = ['arp_host.cpp', 'arp_cif.h', 'arp_cif_api.h', 'arp_json.h']


def build_host(force=False):
    """libarpeggio_host.so: the host-only entry points (mmCIF reader, JSON writer) with g++ — no hipcc, no GPU.  What
    tests/golden/make_golden*.py and the file reader need on a machine where the HIP library cannot be built."""
    deps = [os.path.join(CSRC, f) for f in HOST_DEPS] + [os.path.join(HERE, '..', 'include', 'arpeggio_hip.h')]
    if not force and os.path.exists(HOST_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_LIB) for d in deps):
        return HOST_LIB
    cxx = shutil.which('g++') or shutil.which('c++') or shutil.which('clang++')
    if not cxx:
        raise RuntimeError('no C++ compiler for libarpeggio_host.so')
    subprocess.run([cxx, '-O2', '-std=c++17', '-fPIC', '-shared', '-Wall', os.path.join(CSRC, 'arp_host.cpp'), '-o', HOST_LIB, '-lpthread'], check=True)
    return HOST_LIB


PYEXPORT = os.path.join(HERE, '_pyexport.so')


def build_pyexport(force=False):
    """The optional CPython helper of core/export.py (host code, gcc): get_contacts() records built in C."""
    import sysconfig
    src = os.path.join(CSRC, 'arp_pyexport.c')
    if not force and os.path.exists(PYEXPORT) and os.path.getmtime(PYEXPORT) >= os.path.getmtime(src):
        return PYEXPORT
    inc = sysconfig.get_paths()['include']
    cc = shutil.which('gcc') or shutil.which('cc')
    if not cc or not os.path.exists(os.path.join(inc, 'Python.h')):
        return None          # no compiler / no headers: export.py keeps its Python loop
    subprocess.run([cc, '-O2', '-shared', '-fPIC', '-I', inc, src, '-o', PYEXPORT], check=True)
    return PYEXPORT


if __name__ == '__main__':
    print(build_pyexport(force=True))
    print(build(force=True, verbose=True))
