#!/bin/bash
# same-box A/B of builds of the library (ab_libs/lib_<tag>.so; "new" = the tree's) on a TitaNet-M / -L step:  bash tools/ab_ml.sh m 10 new old B
SIZE=${1:-m}; NB=${2:-10}; shift; shift
cp titanet_amd/libtitanet_amd.so /tmp/lib_new.so
for v in "$@"; do
  if [ $v = new ]; then cp /tmp/lib_new.so titanet_amd/libtitanet_amd.so; else cp ab_libs/lib_$v.so titanet_amd/libtitanet_amd.so; fi
  echo "== $v"; bash tools/quick_stats_ml.sh ab_$v $SIZE $NB 2>&1 | grep -E "pgemm_nt|pgemm_tn|bn_bwd_apply|dw_bwd_slab_kernel<7, 7|dw_fwd_slab_kernel<7, 7|combine_fwd|total" | cut -c1-50,100-160
done
cp /tmp/lib_new.so titanet_amd/libtitanet_amd.so
