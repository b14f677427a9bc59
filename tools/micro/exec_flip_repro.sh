#!/bin/bash
# Builds tools/micro/exec_flip_repro.hip as compiled and with its misplaced FLOW-block copies rewritten (isa_exec.repair), and — on a
# GPU — runs both on the same inputs.  usage: tools/micro/exec_flip_repro.sh [outdir] [-DWAVES=1 -DEXTRA=40 …]; exit 1 = results differ
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$HERE/../..
OUT=${1:-/tmp/exec_flip_repro}; shift || true
DEFS=${*:--DWAVES=1 -DEXTRA=40}
LLVM=/opt/rocm/lib/llvm/bin
mkdir -p $OUT
hipcc -w --cuda-device-only --offload-arch=gfx950 -O3 $DEFS -S $HERE/exec_flip_repro.hip -o $OUT/repro.s
python3 $ROOT/tools/isa_exec_check.py $OUT/repro.s --repair $OUT/repro_fixed.s | tail -3
for v in repro repro_fixed; do
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $OUT/$v.s -o $OUT/$v.o
  $LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $OUT/$v.hsaco $OUT/$v.o
done
hipcc -w --offload-arch=gfx950 -O1 $DEFS $HERE/exec_flip_repro.hip -o $OUT/exec_flip_repro
if [ "$NO_RUN" = 1 ]; then exit 0; fi
$OUT/exec_flip_repro $OUT/repro.hsaco $OUT/repro_fixed.hsaco
