#!/bin/bash
# Round 5, trip 5: non-temporal cache hints in the persistent GEMM -- bit-identity, fabric bytes per shape (PMC FETCH_SIZE with the hint off / on
# the streaming operand / on the other), interleaved A/B in the step; the non-temporal-STORE build variant in a second process on the same box
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 300 python -m pytest tests/test_round5_gpu.py -q -p no:cacheprovider -x -k "non_temporal" > $O/r5t5_tests.log 2>&1; echo "tests rc=$?" >> $O/r5t5_tests.log); tail -3 $O/r5t5_tests.log
for nt in 0 1 2; do
  mkdir -p $O/pmc_nt$nt
  (VJ_GEMM_NT=$nt timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_nt$nt/f -o s -- python tools/gemm_traffic.py run > $O/pmc_nt$nt.f.log 2>&1)
  (VJ_GEMM_NT=$nt timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_nt$nt/w -o s -- python tools/gemm_traffic.py run > $O/pmc_nt$nt.w.log 2>&1)
  python tools/gemm_traffic.py summary $O/pmc_nt$nt/f $O/pmc_nt$nt/w $O/r5t5_traffic_nt$nt.md > /dev/null 2>&1
  echo "=== gemm_nt=$nt"; head -30 $O/r5t5_traffic_nt$nt.md
  find $O/pmc_nt$nt -name "*.csv" -size +4M -delete
done
(timeout 600 python tools/abab.py --arms "base;nt1:gemm_nt=1;nt2:gemm_nt=2" --rounds 8 --steps 6 --out $O/r5t5_abab.json > $O/r5t5_abab.md 2> $O/r5t5_abab.err; echo "rc=$?" >> $O/r5t5_abab.err)
cat $O/r5t5_abab.md; tail -2 $O/r5t5_abab.err
(VJ_LIB_VARIANT=nts timeout 600 python tools/abab.py --arms "base;nt1:gemm_nt=1" --rounds 6 --steps 6 --out $O/r5t5_abab_nts.json > $O/r5t5_abab_nts.md 2> $O/r5t5_abab_nts.err; echo "rc=$?" >> $O/r5t5_abab_nts.err)
echo "=== build variant with non-temporal epilogue stores"; cat $O/r5t5_abab_nts.md; tail -2 $O/r5t5_abab_nts.err
