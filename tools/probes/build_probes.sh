#!/bin/bash
# tools/probes/build_probes.sh -- the two stand-alone HIP probes of round 5 (binaries are not tracked: they are built from the
# sources beside them, for gfx950, and travel to the GPU box with the snapshot like the libraries do).
#   scratch_streams            private-scratch stress over many streams (tools/sessions/r05_s1.sh, _s4, _s5)
#   r4_cache/r4_cache_probe    round 4's scratch cache, stand-alone, against its symptom
set -eu
HERE=$(cd "$(dirname "$0")" && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -mllvm -disable-promote-alloca-to-vector -mllvm -disable-promote-alloca-to-lds \
    -o "$HERE/scratch_streams" "$HERE/scratch_streams.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I"$HERE/r4_cache" -o "$HERE/r4_cache/r4_cache_probe" "$HERE/r4_cache/r4_cache_probe.hip"
ls -la "$HERE/scratch_streams" "$HERE/r4_cache/r4_cache_probe"
