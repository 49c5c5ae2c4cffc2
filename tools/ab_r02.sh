#!/bin/bash
# A/B of blend variants in one gpurun call: base / diet / diet + hardware exp.  bench lines -> gpurun_out/ab_*.json
R=$(pwd); O=$R/gpurun_out; P=$R/3dgs.cpp_amd
for v in "" _diet _fast; do
  L=$P/libgs3d_hip$v.so
  for rep in 1 2; do
    GS3D_HIP_LIB=$L python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/ab${v}_$rep.json 2>/dev/null
    python - $O/ab${v}_$rep.json "base$v" <<'PY'
import json,sys
b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], b["value"], "serial", b["frames_per_s_one_in_flight"], "blend serial ms", b["passes_serial_ms"]["render"], "blend timed", b["passes"]["render"]["ms"], "spread", b["timed"]["spread"])
PY
  done
done
for v in _diet _fast; do GS3D_HIP_LIB=$P/libgs3d_hip$v.so python tools/ab_image_check.py 2>&1 | grep -v amdgpu.ids; done
