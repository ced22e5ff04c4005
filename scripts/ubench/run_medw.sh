cd $GRAFT_REPO_ROOT/scripts/ubench
for b in medwave_dev.bin; do echo "== $b"; timeout -k 10 120 ./$b 252 1024 4096 10 || true; timeout -k 10 120 ./$b 252 4096 1024 1 || true; done
