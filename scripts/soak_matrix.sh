#!/bin/bash
# small soaks over a matrix of settings (on the GPU box, from the repo root); any difference makes soak_parity exit non-zero
O=gpurun_out/soak_matrix.txt
: > $O
fail=0
# (the exit status is soak_parity's own -- not the grep's -- through a log file; a timeout counts as a failure too)
run() {
  echo "== $*" >> $O
  local log=gpurun_out/soak_matrix_last.log rc=0
  timeout -k 10 500 python tests/soak_parity.py "$@" > $log 2>&1 || rc=$?
  grep -E "soak|Traceback|Error|differ" $log | tail -2 >> $O || true
  if [ $rc -ne 0 ]; then echo "FAILED (exit $rc): $*" >> $O; fail=1; fi
  echo "done (exit $rc): $*"
}
run --pairs 12 --workers 12 --seed 101 --pano-width 720 --features 100 --kp-cap 128 --iters 50 --median 3
run --pairs 12 --workers 12 --seed 102 --pano-width 1200 --features 300 --kp-cap 512 --iters 300 --median 5
run --pairs 12 --workers 12 --seed 103 --pano-width 1440 --features 1000 --kp-cap 512 --iters 2000 --median 0
run --pairs 12 --workers 12 --seed 104 --pano-width 960 --features 50 --kp-cap 64 --iters 17 --median 11 --solver GP3P
run --pairs 12 --workers 12 --seed 105 --pano-width 2880 --features 1000 --kp-cap 1024 --iters 100 --median 11 --frame-cap 8192
run --pairs 12 --workers 12 --seed 106 --detector ORB --median 0 --features 60 --kp-cap 256 --iters 200
run --pairs 12 --workers 12 --seed 107 --detector ORB --median 3 --features 500 --kp-cap 1280 --iters 200 --pano-width 1200
run --pairs 12 --workers 12 --seed 108 --detector FAST --kp-cap 256 --median 0 --iters 100
run --pairs 12 --workers 12 --seed 109 --detector AGAST --kp-cap 4096 --median 5 --iters 100 --pano-width 720
run --pairs 12 --workers 12 --seed 110 --rgbd EPNP --features 300 --iters 100
run --pairs 12 --workers 12 --seed 111 --rgbd KNEIP --features 3000 --iters 500
run --pairs 12 --workers 12 --seed 112 --pano-width 1440 --features 1000 --kp-cap 512 --iters 500 --median 11 --solver GP3P
cat $O
exit $fail
