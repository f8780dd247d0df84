#!/bin/bash
# gpurun for runs that need neither the golden fixtures nor the object files (ubench binaries only): a 15 MB push instead of 71 MB.
# The ignore file exists only for the duration of the call (the round-end driver run must see the whole tree).
set -u
cd "$(dirname "$0")/.."
printf 'tests/golden\nfish_diffusion_amd/csrc/*.o\nfish_diffusion_amd/csrc/libfishdx_trace.so\n' > .gpurunignore
/usr/local/graft/bin/gpurun "$@"
rc=$?
rm -f .gpurunignore
exit $rc
