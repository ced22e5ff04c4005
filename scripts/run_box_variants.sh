for v in u2 u4 u8; do echo "== variant $v"; TRICOLOUR_AMD_LIB=$PWD/tricolour_amd/variants/lib_$v.so python scripts/median_bench.py --variants 5 2>&1 | grep variant; done
