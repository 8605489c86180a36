#!/bin/bash
# Round 3, trip 1: power / clock matrix, interleaved A/B of the existing switches, DP at one rank (both backends), K sweep
# with and without the epilogue.  Everything lands under gpurun_out/r3t1_*.
export TMPDIR=/tmp
O=gpurun_out
python tools/power.py > $O/r3t1_power_src.log 2>&1
(timeout 400 python tools/power_matrix.py --seconds 7 --out $O/r3t1_power_matrix.json > $O/r3t1_power_matrix.md 2> $O/r3t1_power_matrix.err; echo "rc=$?" >> $O/r3t1_power_matrix.err)
(timeout 400 python tools/abab.py --power --rounds 8 --steps 5 --out $O/r3t1_abab1.json \
  --arms "base;4w:gemm_4w=1;4wfwd:gemm_fwd_flags=256;4wdgrad:gemm_dgrad_flags=256;tn:wgrad_tn=1;lanes2:wgrad_lanes=2;nofwdov:overlap_fwd=0;serial:no_overlap=1" \
  > $O/r3t1_abab1.md 2> $O/r3t1_abab1.err; echo "rc=$?" >> $O/r3t1_abab1.err)
(timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r3t1_bench.json 2> $O/r3t1_bench.err; echo "rc=$?" >> $O/r3t1_bench.err)
(VJ_FORCE_DP=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-pass > $O/r3t1_dp1_torch.json 2> $O/r3t1_dp1_torch.err; echo "rc=$?" >> $O/r3t1_dp1_torch.err)
(VJ_FORCE_DP=1 VJ_COMM_BACKEND=capi timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-pass > $O/r3t1_dp1_capi.json 2> $O/r3t1_dp1_capi.err; echo "rc=$?" >> $O/r3t1_dp1_capi.err)
(timeout 200 python tools/gemm_ksweep.py > $O/r3t1_ksweep.log 2>&1)
(timeout 300 python -m pytest tests/test_round3_gpu.py tests/test_round2_gpu.py -q -x -p no:cacheprovider -k "prefetcher or runtime_options or c_chain" > $O/r3t1_tests.log 2>&1; echo "tests rc=$?" >> $O/r3t1_tests.log)
tail -3 $O/r3t1_tests.log; cat $O/r3t1_power_matrix.md; cat $O/r3t1_abab1.md; tail -2 $O/r3t1_bench.err | cut -c1-250; tail -3 $O/r3t1_dp1_torch.err | cut -c1-250; tail -3 $O/r3t1_dp1_capi.err | cut -c1-250
