#!/bin/bash
# rebuild pt_kernels with a given waves-per-SIMD register budget and time the path tracer workload (GPU box)
for w in 3 4 5 6; do
  rm -f u_4a_2s_p3d_raytracer_template2_amd/csrc/build/pt_kernels.o
  make -s -C u_4a_2s_p3d_raytracer_template2_amd/csrc -j8 all KFLAGS_EXTRA=-DPT_WAVES_PER_EU=$w 2>&1 | grep error
  echo "waves_per_eu $w"
  python bench.py --workload pathtracer --spp 64 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value'], 1), d['unit'], round(d['ms_per_step'], 2), 'ms/step')"
done
