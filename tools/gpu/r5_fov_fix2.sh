#!/bin/bash
# test_fov_gate_ranges_on_their_bound on the shipped library (must pass) and on a scratch build with the 1-ulp square root put
# back (must fail: shows the test sees the defect it was written for)
mkdir -p gpurun_out
{
echo "== shipped library"
timeout 900 python -m pytest tests/test_global_init.py -m gpu -q -x --tb=line -p no:cacheprovider 2>&1 | tail -3
echo "== scratch build with __fsqrt_rn (bare v_sqrt_f32) in store_fov_kernel"
cd sonar_slam_amd/csrc
sed 's/sqrtf(__fadd_rn/__fsqrt_rn(__fadd_rn/' sfe_store.hip > sfe_store_old_tmp.hip
grep -c "__fsqrt_rn(__fadd_rn" sfe_store_old_tmp.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -c sfe_store_old_tmp.hip -o /tmp/sfe_store_old.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsonarfe_old.so $(ls _obj/*.o | grep -v sfe_store.o) /tmp/sfe_store_old.o
rm -f sfe_store_old_tmp.hip
cd ../..
SONARFE_LIB=/tmp/libsonarfe_old.so timeout 600 python -m pytest tests/test_global_init.py -m gpu -q -x --tb=line -p no:cacheprovider -k ranges_on_their_bound 2>&1 | tail -4
} 2>&1 | tee gpurun_out/r05_fov_range_test.txt
