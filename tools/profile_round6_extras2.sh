cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out
for shp in cfg2 cfg4; do
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $out/r06_xf$shp -o f -- python tools/pmc_ipattn.py $shp > /dev/null 2>&1; echo "ipattn $shp fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $out/r06_xw$shp -o w -- python tools/pmc_ipattn.py $shp > /dev/null 2>&1; echo "ipattn $shp write rc=$?"
python tools/pmc_summary.py $(find $out/r06_xf$shp -name "*results.db" | head -1) $(find $out/r06_xw$shp -name "*results.db" | head -1) $out/r06_pmc_hbm_ipattn_$shp.json $out/r06_pmc_hbm_ipattn_$shp.md > /dev/null
rm -rf $out/r06_xf$shp $out/r06_xw$shp
cat $out/r06_pmc_hbm_ipattn_$shp.md | grep -v "^$" | head -12
done
python tools/forward_ab.py --rounds 5 --configs base,pf_off,pf_cap4,pf_cap8,pf_chunk8,pf_chunk13,pf_chunk4 > $out/r06_forward_ab_prefetch.json 2> $out/r06_forward_ab_prefetch.log
grep "==\|ff.geglu\|cross.to_out\|self.to_out \|ff.out\|cross.fused \|self.to_qkv" $out/r06_forward_ab_prefetch.log
