#!/bin/bash
# gpurun with the commit stamped into the snapshot (a GPU box has no .git; tools/summarize_pmc.py and bench.py read .msn_commit).
#   bash tools/gpu.sh [--timeout S] -- '<command>'
ROOT=$(cd "$(dirname "$0")/.." && pwd)
c=$(git -C "$ROOT" rev-parse --short HEAD)
if [ -n "$(git -C "$ROOT" status --porcelain --untracked-files=no)" ]; then c="$c+dirty"; fi
echo "$c" > "$ROOT/.msn_commit"
exec /usr/local/graft/bin/gpurun "$@"
