# round 6: product build (agent-scope row accesses, every vector width) against -DN2V_SGNS_COHERENT=0: rates by dim
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r10n_sgns_coherent_rates.log
: > $O
for args in "1e8 128" "5e7 256" "2e7 64" "1e7 512" "471785 128"; do
  for lib in "" $R/build_variants/libn2v_sgns_plain.so; do
    N2V_VARIANT_LIB=$lib timeout -k 10 300 python $R/scripts/time_sgns_scale.py $args 2>&1 | tail -1 >> $O || exit 1
  done
done
cat $O
