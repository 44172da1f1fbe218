"""Per-kernel register / LDS / scratch figures of the gfx950 code objects (no GPU needed).

    python tools/kernel_resources.py [> profiles/rNN_kernel_resources.txt]

Compiles every igmc_amd/csrc/*.hip device-only (hipcc --cuda-device-only), unbundles the gfx950 code object and prints
the amdhsa kernel metadata.  `vgpr` is the unified count (architectural + accumulation registers, `agpr` of them AGPRs);
waves/SIMD = floor(512 / vgpr rounded up to 8), capped at 8: gfx950 has a unified 512-entry register file per SIMD lane.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.environ.get('IGMC_CSRC_DIR') or os.path.join(ROOT, 'igmc_amd', 'csrc')      # (a variant's sources: tools/build_variant.sh)
LLVM = '/opt/rocm/lib/llvm/bin'


def main():
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    print('%-62s %5s %5s %5s %6s %8s %7s %10s' % ('kernel', 'vgpr', 'agpr', 'sgpr', 'spill', 'scratch', 'lds', 'waves/SIMD'))
    with tempfile.TemporaryDirectory() as tmp:
        for src in srcs:
            co, elf = os.path.join(tmp, src + '.co'), os.path.join(tmp, src + '.elf')
            subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-c',
                            os.path.join(CSRC, src), '-o', co], check=True, stderr=subprocess.DEVNULL)
            subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + co,
                            '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + elf], check=True)
            notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', elf], check=True,
                                   stdout=subprocess.PIPE, universal_newlines=True).stdout
            for blk in notes.split('  - .agpr_count:')[1:]:
                def g(key):
                    mm = re.search(r'\.' + key + r':\s*(\S+)', blk)
                    return mm.group(1) if mm else '?'
                agpr = blk.split()[0]
                name = g('name')
                try:
                    name = subprocess.run(['c++filt', name], stdout=subprocess.PIPE, universal_newlines=True).stdout.strip()
                except OSError:
                    pass
                name = re.sub(r'\(.*', '', name)
                tot = (int(g('vgpr_count')) + 7) // 8 * 8
                waves = min(8, 512 // max(tot, 1))
                print('%-62s %5s %5s %5s %6s %8s %7s %10d' % ((src[:-4] + ':' + name)[:62], g('vgpr_count'), agpr,
                      g('sgpr_count'), g('vgpr_spill_count'), g('private_segment_fixed_size'),
                      g('group_segment_fixed_size'), waves))


if __name__ == '__main__':
    sys.exit(main())
