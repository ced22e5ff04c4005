python scripts/boxfilter_bench.py --win 252 --stage 0 --radii 17,21,32,43,54
python scripts/boxfilter_bench.py --win 252 --stage 1 --radii 5,8,10,17,25,34,43
