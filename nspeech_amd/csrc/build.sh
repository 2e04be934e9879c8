#!/bin/bash
# Builds libnspeech_hip.so for gfx950 (cross-compiles without a GPU).
#   NS_ASM_DIR=<dir> build.sh   instead writes the device assembly of every source, compiled with the flags of the real
#                               build, to <dir>/<name>.s (tests/test_isa_guard_cpu.py reads it) and builds nothing.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value"
# per-file flags.  audio.hip, attn_cluster.hip, attn_gru.hip: no SLP vectorisation - the packed-fp32 instructions it forms out of scalar
# code pick their operand selects freely, including the src1 high-half select that MI355X misreads beside the MFMA waves of
# another kernel (profiles/tools/pk_opsel_probe.hip; tests/test_isa_guard_cpu.py keeps every kernel of the library free of
# that form)
extra_flags() {
  case "$1" in audio.hip|attn_cluster.hip|attn_gru.hip) echo "-fno-slp-vectorize" ;; *) echo "" ;; esac
}
if [ -n "$NS_ASM_DIR" ]; then
  mkdir -p "$NS_ASM_DIR"
  pids=()
  for f in *.hip; do
    [ "$f" = "flac.hip" ] && continue          # host code only
    $HIPCC $FLAGS $(extra_flags "$f") -S --cuda-device-only "$f" -o "$NS_ASM_DIR/${f%.hip}.s" &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p; done
  echo "wrote device assembly to $NS_ASM_DIR"
  exit 0
fi
OBJS=""
pids=()
for f in *.hip; do
  o="${f%.hip}.o"
  OBJS="$OBJS $o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ build.sh -nt "$o" ] || [ -n "$(find . -name "*.h" -newer "$o")" ] || [ ../../include/nspeech_hip.h -nt "$o" ]; then
    $HIPCC $FLAGS $(extra_flags "$f") -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libnspeech_hip.so $OBJS
echo "built $(pwd)/libnspeech_hip.so"
