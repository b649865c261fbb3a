#!/bin/bash
# round 3, GPU job 2: VALU issue micro-benchmark (v2, with the SQ counter calibration) + the new long-fixture GPU tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
./scratch/micro/valu_rate > gpurun_out/r03a/valu_rate.txt 2>&1
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03a/valu_pmc -- $GRAFT_REPO_ROOT/scratch/micro/valu_rate > /dev/null 2>&1)
python - <<'PY' > gpurun_out/r03a/valu_rate_pmc.txt
import csv, glob, collections
acc = collections.OrderedDict()
for f in glob.glob('gpurun_out/r03a/valu_pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:60], r['Dispatch_Id'])
        acc.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])
for (k, d), v in acc.items():
    iv = v.get('SQ_INSTS_VALU', 0)
    if iv:
        print('%-62s %5s INSTS_VALU %.4g ACTIVE_INST_VALU %.4g (%.3f quad-cycles per instr) GRBM/8 %.4g BUSY_CYCLES %.4g WAVE_CYCLES %.4g WAIT_INST_ANY %.4g' % (
            k, d, iv, v.get('SQ_ACTIVE_INST_VALU', 0), v.get('SQ_ACTIVE_INST_VALU', 0) / iv, v.get('GRBM_GUI_ACTIVE', 0) / 8, v.get('SQ_BUSY_CYCLES', 0), v.get('SQ_WAVE_CYCLES', 0), v.get('SQ_WAIT_INST_ANY', 0)))
PY
rm -rf gpurun_out/r03a/valu_pmc
DS_PARITY_LOG=$GRAFT_REPO_ROOT/gpurun_out/r03a/parity_new.jsonl timeout 1500 python -m pytest tests -m gpu -x -q -k "long_recording or an101 or chain_full_batch" > gpurun_out/r03a/gpu_tests_new.txt 2>&1
tail -5 gpurun_out/r03a/gpu_tests_new.txt
