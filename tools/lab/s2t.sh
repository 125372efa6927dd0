#!/bin/bash
# BatchNorm statistics finished inside the split-K finishing launch (last workgroup per column block): conv / resnet tests, config-2 A/B by switch
mkdir -p gpurun_out/s2t
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "resnet or convgemm or reference_run or abi" > gpurun_out/s2t/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/s2t/tests.log)"; grep -n "^E  \|FAILED\|Error" gpurun_out/s2t/tests.log | head -10 | cut -c1-300
bash tools/lab/ab.sh s2t_c2 3 --config 2 --steps 100 --warmup 10 -- separate=MEMEHIP_BN_FINISH_IN_SPLITK=0 -- head=
