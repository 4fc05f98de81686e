#!/bin/bash
# Host-side AddressSanitizer + UBSan build of libmsn_hip.so (the C-ABI shim: argument validation, launch planning,
# workspace sizing, descriptor tables).  Device code is compiled as usual (-fno-gpu-sanitize: GPU ASan is not available);
# the result is for CPU-only runs of the entry points that return before any launch (tests/test_host_cpu.py does that
# in a subprocess with the ASan runtime preloaded).   usage: bash tools/build_host_sanitized.sh -> build_asan/libmsn_hip_asan.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/multimodal_supernovae_amd/build_asan
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=fast -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer"
pids=""
# (objects are rebuilt when their source OR any header is newer; objects without a source -- a removed translation unit -- are dropped)
NEWEST_H=$(ls -t "$ROOT"/multimodal_supernovae_amd/csrc/*.h "$ROOT"/include/msn_hip.h | head -1)
for o in "$OUT"/*.o; do [ -f "$ROOT/multimodal_supernovae_amd/csrc/$(basename "${o%.o}").hip" ] || rm -f "$o"; done
for f in "$ROOT"/multimodal_supernovae_amd/csrc/*.hip; do
    o=$OUT/$(basename "${f%.hip}").o
    if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$NEWEST_H" -nt "$o" ]; then /opt/rocm/bin/hipcc $FLAGS -I"$ROOT/include" -c "$f" -o "$o" & pids="$pids $!"; fi
    if [ $(jobs -r | wc -l) -ge 6 ]; then wait -n; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -o "$OUT/libmsn_hip_asan.so" "$OUT"/*.o
echo "built $OUT/libmsn_hip_asan.so"
