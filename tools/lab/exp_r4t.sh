#!/bin/bash
# round 4, t: config 2 -- BatchNorm statistics as fixed-point integer atomics (MEMEHIP_BN_ATOMIC): no partial buffers, no finishing launches
mkdir -p gpurun_out/r4t
timeout -k 10 600 python -m pytest tests/test_resnet_gpu.py tests/test_convgemm_gpu.py "tests/test_reference_run_gpu.py::test_organizers_train_test_evaluate_match_the_reference_run" tests/test_abi.py -x -q > gpurun_out/r4t/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/r4t/tests.log)"; grep -n "^E  \|Error" gpurun_out/r4t/tests.log | head -8 | cut -c1-300
show() { tail -1 $1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"].get("final_loss"))'; }
for rep in 1 2 3; do
  for f in 0 1; do
    MEMEHIP_BN_ATOMIC=$f timeout -k 10 200 python bench.py --config 2 --steps 100 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/r4t/c2_a${f}_$rep.log 2>&1; echo "bn_atomic=$f rep $rep: $(show gpurun_out/r4t/c2_a${f}_$rep.log)"
  done
done
