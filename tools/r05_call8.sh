#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05; mkdir -p $O
# whole test files under the buffer guard (RAPIDNET_GUARD=1: red zones + NaN poison), one process: the round-5 kernels among them
# (tests/test_gpu_slab_kernels.py forces the LDS-staged and the register-resident slab products)
timeout -k 10 1000 python tools/guard_suite.py -m gpu tests/test_gpu_slab_kernels.py tests/test_gpu_sharded_batched.py tests/test_gpu_fbe_nama.py tests/test_gpu_lazy_dual.py \
    tests/test_gpu_random_shapes.py tests/test_gpu_parity.py tests/test_golden_synthetic.py tests/test_gpu_closed_loop.py tests/test_reference_barcelona30.py \
    tests/test_nonuniform_trees.py tests/test_gpu_fullsize.py tests/test_gpu_oneshot.py tests/test_gpu_comm_timeout.py 2>&1 | tee $O/guard_suite.log | tail -8
