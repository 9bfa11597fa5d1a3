"""Build a variant of libkimg.so for A/B experiments: recompiles the named csrc files with extra
flags and links them with the objects of the regular build.

    python tools/build_variant.py NAME 'grid_mfma.hip degrid_mfma.hip' -DKIMG_SOMETHING ...

-> build_variants/libkimg_NAME.so (git-ignored; travels to the GPU box).  Experiment scripts pick
it with KIMG_VARIANT_LIB=NAME (tools only: the package itself reads nothing from the environment)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from katsdpimager_amd import build

name, files, flags = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
build.build_lib()
out_dir = os.path.join(ROOT, 'build_variants')
os.makedirs(out_dir, exist_ok=True)
objs = []
procs = []
for src in build.SOURCES:
    obj = os.path.join(build.CSRC, src.replace('.hip', '.o'))
    if src in files:
        obj = os.path.join(out_dir, '%s_%s.o' % (name, src.replace('.hip', '')))
        procs.append(subprocess.Popen([build.hipcc()] + build.FLAGS + flags + ['-c', os.path.join(build.CSRC, src), '-o', obj]))
    objs.append(obj)
for p in procs:
    assert p.wait() == 0
lib = os.path.join(out_dir, 'libkimg_%s.so' % name)
subprocess.check_call([build.hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs + ['-lrocfft'])
print(lib)
