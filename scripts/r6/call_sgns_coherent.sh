# round 6: the SGNS kernel with agent-scope (sc1) row accesses against the product build: rows that differ from the
# ordered run, link AUC on cfg 2 by hub_rows, rate on a 10^8 x 128 model.  Run on the GPU box.
R=$GRAFT_REPO_ROOT
V=$R/build_variants/libn2v_sgns_coherent.so
O=$R/gpurun_out/r10m_sgns_coherent.log
: > $O
for lib in "" $V; do
  export N2V_HIP_LIB=$lib N2V_VARIANT_LIB=$lib
  echo "=== library: ${lib:-product}" >> $O
  for a in "10000000 top 0" "10000000 spread 0"; do timeout -k 10 200 python $R/scripts/r6/diag_hogwild_rows.py $a 2>&1 | grep -E "whole|token rows|negative-only|hub_rows" >> $O || exit 1; done
  for h in 0 auto; do HUB_ROWS=$h timeout -k 10 300 python $R/scripts/r3/hogwild_auc_runs.py 3 x 0 2>&1 | grep -E "batched=|max_waves=0:" >> $O || exit 1; done
  timeout -k 10 300 python $R/scripts/time_sgns_scale.py 1e8 128 2>&1 | tail -1 >> $O || exit 1
done
cat $O
