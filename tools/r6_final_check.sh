#!/bin/bash
# round 6 (GPU box): the final library — GPU suite, stress runs beyond the suite's sizes (mismatch counts must be 0)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6_final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -14 | tee $O/tests.txt
(timeout 600 python tools/stress_parity.py; timeout 600 python tools/stress_mesh.py 4; timeout 600 python tools/stress_world.py 40 200000; timeout 600 python tools/stress_csg.py 40 200000) 2>&1 | grep -v "^$" | tail -60 | tee $O/stress.txt
