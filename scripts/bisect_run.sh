cd $GRAFT_REPO_ROOT
for k in NONE TRI_NO_FUSED_MEDREJ TRI_ST_NO_PANEL TRI_FILTER_NO_BOXW TRI_MEDREJ_FORCE_FALLBACK; do
  echo "== $k"
  env $k=1 python scripts/bisect_case.py ${1:-1} 2>&1 | grep -v amdgpu | tail -2
done
