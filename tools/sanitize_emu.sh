#!/bin/bash
# Sanitizer run of the host layer + kernel logic (no GPU): the same .hip sources compiled for the CPU emulation with
# clang's UndefinedBehaviorSanitizer (signed overflow, shifts, bounds of static arrays, null / misaligned derefs are
# fatal), then the emulator test files run against that build.  Usage: bash tools/sanitize_emu.sh [asan]
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
KIND=${1:-ubsan}
CXX=/opt/rocm/lib/llvm/bin/clang++
RTDIR=$($CXX -print-resource-dir)/lib/linux
OUT=/tmp/igmc_san
mkdir -p $OUT
SRCS="igmc_amd/csrc/extract.hip igmc_amd/csrc/model.hip igmc_amd/csrc/graphstep2.hip igmc_amd/csrc/sortpool.hip igmc_amd/csrc/capi.hip"
if [ "$KIND" = asan ]; then
  FLAGS="-fsanitize=address -fno-omit-frame-pointer"
  RT=$RTDIR/libclang_rt.asan-x86_64.so
  export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1
  TESTS="tests/test_emu_extract.py tests/test_emu_ctrl.py tests/test_emu_model.py"
else
  FLAGS="-fsanitize=undefined,bounds -fno-sanitize=alignment -fno-sanitize-recover=undefined"
  RT=$RTDIR/libclang_rt.ubsan_standalone-x86_64.so
  export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
  TESTS="tests/test_emu_extract.py tests/test_emu_ctrl.py tests/test_emu_model.py tests/test_emu_extract_random.py"
fi
$CXX -x c++ -std=c++17 -O1 -g -fPIC -shared -DIGMC_HIPEMU -Wno-unused-value $FLAGS -include tools/hipemu/hipemu.h \
     -o $OUT/libigmc_emu_$KIND.so $SRCS
echo "built $OUT/libigmc_emu_$KIND.so ($(nm -D $OUT/libigmc_emu_$KIND.so | grep -c -i "$KIND") sanitizer references)"
LD_PRELOAD=$RT IGMC_EMU_LIB=$OUT/libigmc_emu_$KIND.so python -m pytest $TESTS -x -q
