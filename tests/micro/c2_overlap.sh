cd $GRAFT_REPO_ROOT
for cfg in "TCFD_GRAPH=1 TCFD_OVERLAP=0" "TCFD_GRAPH=0 TCFD_OVERLAP=0" "TCFD_GRAPH=0 TCFD_OVERLAP=1"; do
  echo "$cfg: $(env $cfg python tests/micro/c2_rate.py 2>/dev/null | tail -1)"
done
