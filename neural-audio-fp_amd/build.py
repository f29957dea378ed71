"""Build libnafp.so (the C-ABI HIP library) in-tree for gfx950.

    python neural-audio-fp_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The .so lands next to this file
(git-ignored; it travels to the GPU box with the repo snapshot).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libnafp.so')
STAMP = os.path.join(HERE, '.libnafp.stamp')
SOURCES = ['api.hip', 'melspec.hip', 'conv.hip', 'tail.hip', 'ntxent.hip', 'optim.hip', 'specaug.hip', 'backward.hip', 'search.hip', 'augment.hip']
HEADERS = ['nafp_common.h', os.path.join('..', '..', 'include', 'nafp.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
         '-Wall', '-Wno-unused-result', '-fno-gpu-rdc']


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    want = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == want:
                return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ['-o', LIB]
    if verbose:
        print('[nafp build]', ' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, 'w') as fh:
        fh.write(want)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
