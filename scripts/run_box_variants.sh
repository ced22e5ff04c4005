for v in a b c d; do echo "== variant $v";
TRICOLOUR_AMD_LIB=$PWD/tricolour_amd/variants/lib_$v.so python scripts/boxfilter_bench.py --win 252 --stage 0 --radii 17,21,43,54 --variants 2 2>&1 | grep "r="
TRICOLOUR_AMD_LIB=$PWD/tricolour_amd/variants/lib_$v.so python scripts/boxfilter_bench.py --win 252 --stage 1 --radii 8,10,17,25,34,43 --variants 2 2>&1 | grep "r="
done
