"""Build libkimg.so (HIP, gfx950 only) in-tree with hipcc.

Usage: ``python -m katsdpimager_amd.build`` or :func:`build_lib`.  The shared
library is written next to this file so that it travels with the source tree.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libkimg.so')

SOURCES = ['api.hip', 'grid.hip', 'grid_mfma.hip', 'grid_binned.hip', 'degrid_mfma.hip', 'image.hip', 'fft.hip', 'weight.hip',
           'clean.hip', 'clean_multi.hip', 'preprocess.hip', 'ktable.hip', 'store.hip']

# -ffp-contract=off: a*b+c is fused only where the source says fmaf(); the image/CLEAN
# kernels must round exactly like the reference's numpy host path.
# -munsafe-fp-atomics: atomicAdd(float*/double*) lowers to global_atomic_add_f32/f64.
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-ffp-contract=off',
         '-munsafe-fp-atomics', '-Wall', '-Wno-unused-function']


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found; libkimg.so cannot be built')
    return exe


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), 'include', 'kimg.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        cmd = [hipcc()] + FLAGS + list(extra_flags) + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on ' + src)
    cmd = [hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-lrocfft']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build_lib(force='--force' in sys.argv, verbose=True))
