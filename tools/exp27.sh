#!/bin/bash
O=gpurun_out/exp27; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?
tail -3 $O/pytest.txt
if grep -q "Memory access fault" $O/pytest.txt; then echo "GPU FAULT"; exit 99; fi
[ $rc -ne 0 ] && exit $rc
echo "--- unclamped reciprocal (expected to FAIL):"
P3D_LIB=$PWD/u_4a_2s_p3d_raytracer_template2_amd/libp3d_hip_x.so timeout -k 10 300 python -m pytest tests/test_gpu_random_scenes.py -m gpu -q -k zero_direction > $O/unclamped.txt 2>&1
tail -4 $O/unclamped.txt
timeout -k 10 300 python tools/perf_probe.py mount_low --n 300 2>&1 | grep -E "wavefront lds/lane"
exit 0
