#!/bin/bash
# tools/probes/variants/build_commit.sh <commit> <name> -- the PRODUCT library of an earlier commit of this repository, built from
# its own history into libmemc_hip_<name>.so next to this script (git-ignored; travels to the GPU box with the snapshot), for
# same-process A/Bs against the working tree's library (tools/ab_libs.py).  Round 6: round5 = 86e2dab (the round's last
# commit), round4_kernels is built by build_variants.sh.
set -eu
HERE=$(cd "$(dirname "$0")" && pwd)
REPO=$(cd "$HERE/../../.." && pwd)
COMMIT=$1
NAME=$2
WT=$(mktemp -d)
git -C "$REPO" worktree add -f "$WT" "$COMMIT" >/dev/null
trap 'git -C "$REPO" worktree remove --force "$WT"' EXIT
cd "$WT/memc-net_amd/csrc"
SR=$(sed -n 's/^SRCS *:= *//p' Makefile)
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -I../../include"
mkdir -p obj
pids=""
for s in $SR; do
    /opt/rocm/bin/hipcc $FL -c -o "obj/${s%.*}.o" "$s" &
    pids="$pids $!"
done
for p in $pids; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$HERE/libmemc_hip_${NAME}.so" obj/*.o
ls -la "$HERE/libmemc_hip_${NAME}.so"
