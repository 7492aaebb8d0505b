# round 5, call c: the placement effect with counters (one PMC pass per counter group)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r7c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/scripts/r5/placement_pmc.py > $OUT/plain.log 2>&1 || { tail -20 $OUT/plain.log; exit 1; }
grep PLACEMENT $OUT/plain.log
i=0
for CTRS in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
            "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
            "GRBM_UTCL2_BUSY GRBM_EA_BUSY GRBM_TC_BUSY TCP_UTCL1_STALL_INFLIGHT_MAX_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc$i -- python3 $R/scripts/r5/placement_pmc.py > $OUT/pmc$i.log 2>&1 || { tail -5 $OUT/pmc$i.log; continue; }
  grep PLACEMENT $OUT/pmc$i.log | cut -c1-60
  python3 $R/scripts/r5/placement_condense.py $OUT/pmc$i | tee $OUT/pmc${i}_condensed.txt
  find $OUT/pmc$i -name "*.csv" -size +2M -delete
done
du -sh $OUT
