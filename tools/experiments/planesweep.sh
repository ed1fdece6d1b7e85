cd $GRAFT_REPO_ROOT
echo "== product"; python tools/planebench.py 2>&1 | grep "l2max fp16 planes"
for v in gp5 gp6 gp7; do echo "== $v (5 no epilogue, 6 no MFMAs, 7 no LDS-DMA after the prologue)"; ASPIRE_HIP_LIB=build/variants/$v/libaspire_hip.so python tools/planebench.py 2>&1 | grep "l2max fp16 planes"; done
