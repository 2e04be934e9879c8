#!/bin/bash
# Builds libnspeech_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value"
OBJS=""
pids=()
for f in *.hip; do
  o="${f%.hip}.o"
  OBJS="$OBJS $o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find . -name "*.h" -newer "$o")" ] || [ ../../include/nspeech_hip.h -nt "$o" ]; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libnspeech_hip.so $OBJS
echo "built $(pwd)/libnspeech_hip.so"
