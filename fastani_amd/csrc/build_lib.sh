#!/bin/bash
# Builds libfastani_amd.so for gfx950 from its units (host/engine.hpp lists them), one hipcc per unit, in parallel.
# (engine_l2 is compiled with its own scheduling strategy, see unit_flags below.)
#
# ANI_ASM_PEEPHOLE=1 routes the device code of engine_map.hip (the L2 simulation lives there) through a one-line assembly peephole
# between hipcc's code generation and the assembler:
#     v_cndmask_b32_e32 vD, a, vB, vcc   ->   v_cndmask_b32_e64 vD, a, vB, vcc
# On MI355X two VOP2-encoded v_cndmask_b32 issued back to back on one SIMD cost ~18-22 cycles each instead of 4
# (tools/ubench/valu.hip, profiles/r02_ubench_valu.txt: "1 v_cmp + 7 v_cndmask_e32" 16 cycles per instruction, the VOP3 encoding
# 4.3).  Measured on the product kernels (profiles/r02d_ab.txt) it is worth 1.5 % of k_l2_sim and nothing elsewhere — with
# several waves per SIMD other waves' instructions separate the selects — so it is NOT the default build.
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
LLVM=${LLVM_BIN:-/opt/rocm/lib/llvm/bin}
OUT=${1:-$HERE/libfastani_amd.so}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC ${ANI_EXTRA_FLAGS:-}"
UNITS="engine_core engine_ingest engine_sketch engine_index engine_map engine_l2 sort_device"
# per-unit flags: the L2 codes / simulation kernels with the max-ILP scheduler (engine_l2.hip says why); ANI_L2_UNIT_FLAGS="" builds them like the rest
unit_flags() { if [ "$1" = "engine_l2" ]; then echo "${ANI_L2_UNIT_FLAGS--mllvm -amdgpu-sched-strategy=max-ilp}"; fi; }
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
pids=()
for u in $UNITS; do
  if [ "${ANI_ASM_PEEPHOLE:-0}" = "1" ] && [ "$u" = "engine_map" ]; then continue; fi
  ( $HIPCC $FLAGS $(unit_flags $u) -c -o "$T/$u.o" "$HERE/$u.hip" 2> "$T/$u.err" || { cat "$T/$u.err" >&2; exit 1; } ) &
  pids+=($!)
done
if [ "${ANI_ASM_PEEPHOLE:-0}" = "1" ]; then
  u=engine_map
  $HIPCC $FLAGS -S --cuda-device-only -o "$T/dev.s" "$HERE/$u.hip" 2> "$T/dev.err" || { cat "$T/dev.err" >&2; exit 1; }
  sed -E 's/v_cndmask_b32_e32 (v[0-9]+), ([^,]+), (v[0-9]+), vcc/v_cndmask_b32_e64 \1, \2, \3, vcc/' "$T/dev.s" > "$T/dev_pp.s"
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$T/dev_pp.s" -o "$T/dev.o"
  $LLVM/ld.lld -shared "$T/dev.o" -o "$T/dev.hsaco"
  $LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
    -input=/dev/null -input="$T/dev.hsaco" -output="$T/dev.hipfb"
  $HIPCC $FLAGS -c --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$T/dev.hipfb" -o "$T/$u.o" "$HERE/$u.hip"
fi
for p in "${pids[@]}"; do wait "$p"; done
objs=""
for u in $UNITS; do objs="$objs $T/$u.o"; done
$HIPCC --offload-arch=gfx950 -fPIC -shared -o "$OUT" $objs
