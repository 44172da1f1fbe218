"""In-tree builds of the native code (no JIT cache: the .so files travel with the repo snapshot).

* :func:`build_hip`  -- ``igmc_amd/lib/libigmc_hip.so``: the gfx950 product library (hipcc cross-compiles
  without a GPU).
* :func:`build_emu`  -- ``tests/emu/libigmc_emu.so``: the SAME sources compiled for the host against
  ``tools/hipemu/hipemu.h``; test infrastructure for kernel-logic checks on GPU-less machines, never loaded
  by the product package.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (debug hooks of the same-box A/B tooling: an experimental copy of the sources built into libraries of its own --
#  IGMC_CSRC_DIR / IGMC_HIP_LIB_OUT / IGMC_EMU_LIB_OUT; the product never sets them)
CSRC = os.environ.get('IGMC_CSRC_DIR') or os.path.join(ROOT, 'igmc_amd', 'csrc')
SOURCES = ['extract.hip', 'model.hip', 'graphstep2.hip', 'sortpool.hip', 'capi.hip']
HEADERS = sorted(h for h in os.listdir(CSRC) if h.endswith('.h')) + ['../../include/igmc_hip.h', '../../include/igmc_rng.h']
HIP_LIB = os.environ.get('IGMC_HIP_LIB_OUT') or os.path.join(ROOT, 'igmc_amd', 'lib', 'libigmc_hip.so')
EMU_LIB = os.environ.get('IGMC_EMU_LIB_OUT') or os.path.join(ROOT, 'tests', 'emu', 'libigmc_emu.so')


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError('build failed: ' + ' '.join(cmd))
    return r.stdout


def _hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    return None


def build_hip(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS]
    if not force and not _newer(HIP_LIB, deps):
        return HIP_LIB
    hipcc = _hipcc()
    if hipcc is None:
        if os.path.exists(HIP_LIB):
            return HIP_LIB      # GPU box without a compiler on PATH: use the prebuilt library
        raise RuntimeError('hipcc not found and no prebuilt libigmc_hip.so')
    os.makedirs(os.path.dirname(HIP_LIB), exist_ok=True)
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-o', HIP_LIB] + srcs
    if verbose:
        cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
    out = _run(cmd)
    if verbose:
        print(out)
    return HIP_LIB


def build_emu(force=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    emu_h = os.path.join(ROOT, 'tools', 'hipemu', 'hipemu.h')
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS] + [emu_h]
    if not force and not _newer(EMU_LIB, deps):
        return EMU_LIB
    cxx = None
    for c in ('/opt/rocm/lib/llvm/bin/clang++', shutil.which('amdclang++'), shutil.which('clang++')):
        if c and os.path.exists(c):
            cxx = c
            break
    if cxx is None:
        raise RuntimeError('clang++ not found (needed for ext_vector_type in the emulation build)')
    os.makedirs(os.path.dirname(EMU_LIB), exist_ok=True)
    cmd = [cxx, '-x', 'c++', '-std=c++17', '-O2', '-g', '-fPIC', '-shared', '-DIGMC_HIPEMU', '-Wno-unused-value',
           '-include', emu_h, '-o', EMU_LIB] + srcs
    _run(cmd)
    return EMU_LIB


if __name__ == '__main__':
    print(build_hip(force='--force' in sys.argv, verbose='-v' in sys.argv))
    if '--emu' in sys.argv:
        print(build_emu(force=True))
