#!/bin/bash
# Runs the GPU test groups one after another; stops at the first group that ends abnormally
# (anything but "all passed" = 0 or "some tests failed" = 1), so a fault never gets a follow-up run.
set -u
mkdir -p gpurun_out
run() {
    name=$1; shift
    echo "=== $name ==="
    timeout -k 10 "$TMO" python -u -m pytest "$@" -m gpu -v --timeout 120 --timeout-method thread -p no:cacheprovider \
        > "gpurun_out/$name.log" 2>&1
    rc=$?
    tail -n 25 "gpurun_out/$name.log"
    echo "=== $name rc=$rc ==="
    if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
}
TMO=${TMO:-300}
for g in "$@"; do
    case $g in
        kernels) run kernels tests/test_gpu_kernels.py -k "not flash" ;;
        flash) run flash tests/test_gpu_kernels.py -k "flash" ;;
        models) run models tests/test_gpu_models.py ;;
        pipeline) run pipeline tests/test_gpu_pipeline.py ;;
        shells) run shells tests/test_gpu_matcher_clip_pipeline.py ;;
        sam) run sam tests/test_gpu_sam.py ;;
        io) run io tests/test_gpu_io_runner.py ;;
        full) run full tests/test_gpu_fullsize.py ;;
    esac
done
exit 0
