#!/bin/bash
# A/B of library builds in one gpurun call: tools/ab_libs.sh tagA tagB ...  (3dgs.cpp_amd/libgs3d_hip_<tag>.so; "cur" = the default library)
R=$(pwd); O=$R/gpurun_out; P=$R/3dgs.cpp_amd; exec < /dev/null
for rep in 1 2; do for v in "$@"; do
  L=$P/libgs3d_hip_$v.so; [ "$v" = cur ] && L=$P/libgs3d_hip.so
  GS3D_HIP_LIB=$L timeout 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/ab_${v}_$rep.json 2>/dev/null
  python - $O/ab_${v}_$rep.json $v <<'PY'
import json,sys
b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], b["value"], "one-in-flight", b["frames_per_s_one_in_flight"], "blend serial", b["passes_serial_ms"]["render"], "pre", b["passes_serial_ms"]["preprocess"], "hwexp", b["frames_per_s_hw_exp"])
PY
done; done
