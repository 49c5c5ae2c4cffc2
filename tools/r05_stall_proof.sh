#!/bin/bash
# The stall's cause shown both ways (profiles/r05_stall_hunt.txt): gpurun -- bash tools/r05_stall_proof.sh
set -u
O=$PWD/gpurun_out/stall
mkdir -p $O
export GS_DEBUG_STALLS=2
P="python tools/stall_probe.py"
$P --frames 8000 --torch --legacy-pointers --label "torch imported, round-4 binding (ndarray.ctypes.data_as)" > $O/proof_legacy.json 2> $O/proof_legacy.err
$P --frames 42000 --torch --label "torch imported, round-5 binding, 10 s" > $O/proof_fixed_10s.json 2> $O/proof_fixed_10s.err
$P --frames 42000 --label "no torch, round-5 binding, 10 s" > $O/proof_notorch_10s.json 2> $O/proof_notorch_10s.err
for f in proof_legacy proof_fixed_10s proof_notorch_10s; do echo "== $f: $(cat $O/$f.json)"; grep -c "stall:" $O/$f.err; done
python bench.py --steps 20 --warmup 5 --no-other-configs > $O/bench_20.json 2> $O/bench_20.err
python - <<'PY'
import json
b = json.load(open("gpurun_out/stall/bench_20.json"))
print("bench: value", b["value"], "sustained", b["sustained_frames_per_s"], "batches", b["timed"]["batches"], "outliers", b["timed"]["outliers"],
      "max batch ms", b["timed"]["batch_ms"]["max"], "median", b["timed"]["batch_ms"]["median"], "one_in_flight", b["frames_per_s_one_in_flight"])
PY
