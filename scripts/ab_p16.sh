cd $GRAFT_REPO_ROOT
for v in main p16; do
  if [ $v = main ]; then unset TRICOLOUR_AMD_LIB; else export TRICOLOUR_AMD_LIB=$PWD/tricolour_amd/variants/lib_$v.so; fi
  echo "== $v"
  python scripts/boxfilter_bench.py --stage 1 --variants 0 --radii 43,34,25,17 --win 1008 --rounds 2 2>&1 | grep -v amdgpu | tail -6
done
