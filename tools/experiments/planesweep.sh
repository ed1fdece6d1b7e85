cd $GRAFT_REPO_ROOT
for lib in product s0 s1p0 s1p8; do
for tile in 128128 128256 256256; do
  echo "== $lib tile $tile"; L=build/variants/$lib/libaspire_hip.so; [ $lib = product ] && L=aspire_amd/lib/libaspire_hip.so
  ASPIRE_HIP_LIB=$L ASPIRE_HIP_GRAM_TILE=$tile python tools/planebench.py 2>&1 | grep "l2max fp16 planes\|max"
done; done
echo "== product tile 256256 ring 4"; ASPIRE_HIP_GRAM_TILE=256256 ASPIRE_HIP_GRAM_RING=4 python tools/planebench.py 2>&1 | grep "l2max fp16 planes\|max"
