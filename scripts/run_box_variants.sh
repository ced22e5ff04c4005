for v in w2 w1; do echo "== $v"; TRICOLOUR_AMD_LIB=$PWD/tricolour_amd/variants/lib_$v.so python scripts/boxfilter_bench.py --win 252 --stage 1 --radii 17,21,25,30 --variants 2 2>&1 | grep "r="; done
