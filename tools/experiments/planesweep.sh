cd $GRAFT_REPO_ROOT
for v in probe1 probe2 probe3 probe4; do echo "== $v"; ASPIRE_HIP_LIB=build/variants/$v/libaspire_hip.so ASPIRE_HIP_GRAM_TILE=256256 ASPIRE_HIP_GRAM_PP=1 python tools/planebench.py 2>&1 | grep "l2max fp16 planes"; done
