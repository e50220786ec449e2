#!/bin/sh
# build the HIP library; only if that succeeds, run "$@" on the GPU box
set -e
make -s -j8 -C /root/repo/patolette_amd/csrc 2>&1 | grep -v "int-to-pointer-cast" | grep -E "error|warning:" && { echo "BUILD FAILED"; exit 1; } || true
test -f /root/repo/patolette_amd/lib/libpatolette_amd.so
exec /usr/local/graft/bin/gpurun "$@"
