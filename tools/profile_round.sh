#!/bin/bash
# tools/profile_round.sh <tag>: the measurement set committed under profiles/ each round (run on the MI355X box through gpurun):
# bench line, rocprofv3 kernel-trace summary of the same command, HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate
# runs, never combined with tracing), SQ counter passes on the GEMM / attention micro-workload.
tag=${1:-rXX}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
timeout 600 python bench.py --steps 5 --warmup 2 > $out/${tag}_bench_n1.json 2> $out/${tag}_bench_n1.err; echo "bench rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_kt -o k -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --in-flight 1 --stacked 1 > $out/${tag}_prof_bench.json 2> $out/${tag}_prof.err; echo "trace rc=$?"
python tools/rocpd_summary.py $(find $out/${tag}_kt -name "*results.db" | head -1) $out/${tag}_bench_kernel_stats.md "${tag}: python bench.py --steps 3 --warmup 1 under rocprofv3 --kernel-trace --stats" > /dev/null
# (the PMC passes run the byte-count model's XCD cells only: the engine's pick would put three more candidate plans in the counters)
export IMH_XCD_AUTOTUNE=0
B="python bench.py --steps 1 --warmup 0 --denoise-steps 4 --no-cpu-baseline --in-flight 1 --stacked 1"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $out/${tag}_pf -o f -- $B > /dev/null 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $out/${tag}_pw -o w -- $B > /dev/null 2>&1; echo "write rc=$?"
python tools/pmc_summary.py $(find $out/${tag}_pf -name "*results.db" | head -1) $(find $out/${tag}_pw -name "*results.db" | head -1) $out/${tag}_pmc_hbm_traffic.json $out/${tag}_pmc_hbm_traffic.md
unset IMH_XCD_AUTOTUNE
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS -d $out/${tag}_s1 -o s -- python tools/pmc_gemm.py > /dev/null 2>&1; echo "sq1 rc=$?"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d $out/${tag}_s2 -o s -- python tools/pmc_gemm.py > /dev/null 2>&1; echo "sq2 rc=$?"
python tools/pmc_sq_summary.py $(find $out/${tag}_s1 -name "*results.db" | head -1) $(find $out/${tag}_s2 -name "*results.db" | head -1) > $out/${tag}_pmc_sq_gemm_attn.md
# the north-star call on its own (xattn_kernel + the to_out launch that reads its output), batch 2 and batch 8
for shp in cfg2 cfg4; do
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS -d $out/${tag}_x1$shp -o s -- python tools/pmc_ipattn.py $shp > /dev/null 2>&1; echo "ipattn $shp sq1 rc=$?"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d $out/${tag}_x2$shp -o s -- python tools/pmc_ipattn.py $shp > /dev/null 2>&1; echo "ipattn $shp sq2 rc=$?"
python tools/pmc_sq_summary.py $(find $out/${tag}_x1$shp -name "*results.db" | head -1) $(find $out/${tag}_x2$shp -name "*results.db" | head -1) | sed "s/^# SQ counters per launch.*/# SQ counters per launch of the north-star IP-attention call, $shp (tools\/pmc_ipattn.py: xattn_kernel + to_out gemm_ws 64x160)/" > $out/${tag}_pmc_sq_ipattn_$shp.md
done
# the headline configuration end to end against the fp32 CPU oracle (~5 min of host time), recorded with every other measured parity number
IMH_SLOW=1 timeout 1500 python -m pytest tests/test_gpu_parity_fullsize.py -x -q -m gpu -k "configs1_1024" > $out/${tag}_trajectory30.log 2>&1; echo "trajectory rc=$?"
cp $out/parity_measured.json $out/${tag}_parity.json 2>/dev/null
rm -rf $out/${tag}_kt $out/${tag}_pf $out/${tag}_pw $out/${tag}_s1 $out/${tag}_s2 $out/${tag}_x1cfg2 $out/${tag}_x2cfg2 $out/${tag}_x1cfg4 $out/${tag}_x2cfg4
ls -la $out | grep ${tag}_
