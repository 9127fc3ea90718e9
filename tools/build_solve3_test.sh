#!/bin/bash
# builds tools/test_solve3.bin against libmpqr.so and tools/test_solve3_kt.bin against an MPQR_KTRACE build (in-kernel stamps)
set -e
cd "$(dirname "$0")/.."
C=mixedprecisionblockqr_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -ffp-contract=off $XFL"
make -C $C -j4 >/dev/null
hipcc $FL -I$C tools/test_solve3.hip -o tools/test_solve3.bin -Lmixedprecisionblockqr_amd -lmpqr -Wl,-rpath,'$ORIGIN/../mixedprecisionblockqr_amd'
mkdir -p /tmp/ktobj
for f in driver.hip kernels_gemm.hip kernels_gemm2.hip kernels_fp8.hip kernels_panel.hip kernels_solve.hip kernels_misc.hip; do
  hipcc $FL -DMPQR_KTRACE -c $C/$f -o /tmp/ktobj/$f.o &
done
hipcc $FL -DMPQR_KTRACE -x hip -c $C/host_util.cpp -o /tmp/ktobj/host_util.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libmpqr_kt.so /tmp/ktobj/*.o -L/opt/rocm/lib -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib
hipcc $FL -I$C tools/test_solve3.hip -o tools/test_solve3_kt.bin -Ltools -lmpqr_kt -Wl,-rpath,'$ORIGIN'
echo built
