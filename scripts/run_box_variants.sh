for v in bits bitsneither; do echo -n "$v: ";
  TRICOLOUR_AMD_LIB=$PWD/tricolour_amd/variants/lib_$v.so python bench.py --roofline-only 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['roofline'][0]; print(d['ms_per_launch'], 'ms', d['frac'])"
done
