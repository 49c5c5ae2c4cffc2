#!/bin/bash
# blend experiments of round 5 (one gpurun call)
R=$PWD; O=$R/gpurun_out/blend5; mkdir -p $O; exec < /dev/null
L=$R/3dgs.cpp_amd/libgs3d_hip_both.so
Q="python tools/tune_sweep.py --quick --frames 300 --batches 3"
echo "== B: plain order";          GS3D_HIP_LIB=$L $Q 2>&1 | grep "fif 1" 
python tools/tile_order_experiment.py write /tmp/costB.bin | tail -1
echo "== B: heaviest first";       GS3D_HIP_LIB=$L GS_TILE_COST_FILE=/tmp/costB.bin $Q 2>&1 | grep "fif"
T="--gaussians 6000000 --scene T --frames 100"
echo "== T: plain order";          GS3D_HIP_LIB=$L $Q $T 2>&1 | grep "fif"
python tools/tile_order_experiment.py write /tmp/costT.bin --gaussians 6000000 --scene T | tail -1
echo "== T: heaviest first";       GS3D_HIP_LIB=$L GS_TILE_COST_FILE=/tmp/costT.bin $Q $T 2>&1 | grep "fif"
# counters of the new loop
cd /tmp && export TMPDIR=/tmp
PMC="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
GS3D_HIP_LIB=$L timeout 120 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc -o p -- python $R/tools/tune_sweep.py --no-prime --batches 1 --fif 1 --frames 3 --warm 2 > /dev/null 2>&1
PMC2="SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
GS3D_HIP_LIB=$L timeout 120 rocprofv3 --kernel-trace --pmc $PMC2 --output-format csv -d $O/pmc2 -o p -- python $R/tools/tune_sweep.py --no-prime --batches 1 --fif 1 --frames 3 --warm 2 > /dev/null 2>&1
cd $R
find $O -name '*_kernel_trace.csv' -delete
python - <<'PY'
import csv, glob, collections
for d in ("pmc", "pmc2"):
    for f in glob.glob(f"gpurun_out/blend5/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "k_blend" in k or "k_bin_fast" in k or "k_preprocess" in k:
                print(d, k, {c: f"{x/5:.4g}" for c, x in v.items()})
PY
