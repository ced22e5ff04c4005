for v in b512 b1024; do for blk in 64 128 256 512 1024; do echo -n "$v blk=$blk: ";
  TRI_ST_BLK=$blk TRICOLOUR_AMD_LIB=$PWD/tricolour_amd/variants/lib_$v.so python bench.py --roofline-only 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['roofline'][0]; print(d['ms_per_launch'], 'ms', d['frac'])"
done; done
