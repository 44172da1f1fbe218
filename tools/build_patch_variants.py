"""Build experimental variants of the HIP library from patches, each into a library of its own beside the product's, for a
same-box A/B on the GPU (tools/gpu_ab_variants.sh).  hipcc cross-compiles here; the libraries travel with the snapshot.

    python tools/build_patch_variants.py prologue=profiles/r04_experiments/prologue_scalar_seq_and_kernarg_lines.patch \\
        transpose=profiles/r04_experiments/g2_register_blocked_relm_transpose.patch \\
        both=profiles/r04_experiments/prologue_scalar_seq_and_kernarg_lines.patch+profiles/r04_experiments/g2_register_blocked_relm_transpose.patch

A variant is name=patch[+patch...]; the patched sources live under .scratch/variants/<name>/ (git- and gpurun-ignored), the
library is igmc_amd/lib/libigmc_hip_<name>.so (git-ignored; remove it when the A/B is over).  Only the csrc part of a patch
is applied (hunks for files outside igmc_amd/csrc are skipped)."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    for spec in sys.argv[1:]:
        name, patches = spec.split('=', 1)
        work = os.path.join(ROOT, '.scratch', 'variants', name)
        shutil.rmtree(work, ignore_errors=True)
        os.makedirs(os.path.join(work, 'igmc_amd'))
        shutil.copytree(os.path.join(ROOT, 'igmc_amd', 'csrc'), os.path.join(work, 'igmc_amd', 'csrc'))
        shutil.copytree(os.path.join(ROOT, 'include'), os.path.join(work, 'include'))
        for p in patches.split('+'):
            r = subprocess.run(['git', 'apply', '--include=igmc_amd/csrc/*', '--include=include/*', os.path.join(ROOT, p)], cwd=work,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
            if r.returncode != 0:
                sys.exit('%s: %s does not apply:\n%s' % (name, p, r.stdout))
        out = os.path.join(ROOT, 'igmc_amd', 'lib', 'libigmc_hip_%s.so' % name)
        env = dict(os.environ, IGMC_CSRC_DIR=os.path.join(work, 'igmc_amd', 'csrc'), IGMC_HIP_LIB_OUT=out)
        r = subprocess.run([sys.executable, '-c', 'from igmc_amd import build; print(build.build_hip(force=True))'], cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        if r.returncode != 0:
            sys.exit('%s: build failed:\n%s' % (name, r.stdout[-3000:]))
        print('%-12s %s' % (name, out))


if __name__ == '__main__':
    main()
