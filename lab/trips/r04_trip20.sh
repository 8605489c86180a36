#!/bin/bash
# Round 4, trip 20: tile order of the persistent GEMM (option gemm_raster): bit identity, isolated rates, fabric bytes (PMC, counters only), step A/B
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 600 python -m pytest tests/test_round4_gpu.py -q -p no:cacheprovider -x -k "tile_orders" > $O/r4t20_tests.log 2>&1; echo "tests rc=$?" >> $O/r4t20_tests.log)
grep -E "passed|failed|FAILED|Error|rc=|assert" $O/r4t20_tests.log | tail -6
(timeout 400 python tools/gemm_bench.py --reps 20 --cfgs 8.4 --no-wgrad --toggle gemm_raster=0,4,16,260,264,258 --only tgt,ctx,prd > $O/r4t20_gemm.txt 2>&1; echo "rc=$?" >> $O/r4t20_gemm.txt)
grep -v amdgpu.ids $O/r4t20_gemm.txt | cut -c1-200
for r in 0 260 4; do
  mkdir -p $O/pmc_raster_$r
  VJ_GEMM_RASTER=$r timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_raster_$r/f -o s -- python tools/gemm_traffic.py run > $O/r4t20_pmc_f_$r.log 2>&1
  VJ_GEMM_RASTER=$r timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_raster_$r/w -o s -- python tools/gemm_traffic.py run > $O/r4t20_pmc_w_$r.log 2>&1
  python tools/gemm_traffic.py summary $O/pmc_raster_$r/f $O/pmc_raster_$r/w $O/r4t20_traffic_raster_$r.md > /dev/null 2>&1
  echo "== raster $r"; grep -E "^\| (tgt|ctx|prd)" $O/r4t20_traffic_raster_$r.md | cut -d'|' -f2,7,8,10,11,12
done
find $O/pmc_raster_0 $O/pmc_raster_260 $O/pmc_raster_4 -name "*.csv" -size +4M -delete
(timeout 500 python tools/abab.py --arms "base;g4:gemm_raster=4;g16:gemm_raster=16;c4:gemm_raster=260;c8:gemm_raster=264" --rounds 6 --steps 6 --out $O/r4t20_abab.json > $O/r4t20_abab.md 2> $O/r4t20_abab.err; echo "rc=$?" >> $O/r4t20_abab.err)
cat $O/r4t20_abab.md; tail -2 $O/r4t20_abab.err
