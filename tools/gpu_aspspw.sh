for v in 1 2 3 6; do echo "TN_ASP_SPW=$v"; TN_ASP_SPW=$v bash tools/prof_headline.sh r05_aspspw "asp_v2" 2>&1 | tail -3; done
echo "TN_ASP_FUSED=0"; TN_ASP_FUSED=0 bash tools/prof_headline.sh r05_aspspw "asp_|wide_out_v2_kernel<128, 0>" 2>&1 | tail -4
