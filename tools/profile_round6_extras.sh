#!/bin/bash
# tools/profile_round6_extras.sh: the round-6 probes behind profiles/r06_* (run on the MI355X box through gpurun AFTER
# `python tools/ws_phase_probe.py build`, `python tools/xattn_phase_probe.py build` and `bash tools/build_variant.sh w16_timing gemm_w16.hip -DW16_TIMING=1`)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out
mkdir -p $out
python tools/ws_timeline_probe.py > $out/r06_ws_timeline.txt 2>&1
python tools/xattn_wide_probe.py > $out/r06_xattn_wide_probe.txt 2>&1
python tools/xattn_phase_probe.py run > $out/r06_xattn_phase_probe.txt 2>&1
python tools/w16_probe.py > $out/r06_w16_probe.txt 2>&1
W16_TIMING=1 IMH_LIB_PATH=tools/tmp_libs/lib_w16_timing.so python tools/w16_probe.py 2>&1 | grep -B2 "2048x10240x1280\|8192x10240x1280" >> $out/r06_w16_probe.txt
python tools/vae_time.py > $out/r06_vae_decode_modes.txt 2>&1
# ff.net.0 on 23256 x 160 (rounds 2-5) vs 26256 x 320 (round 6) in alternating PROCESSES (steady-state clocks, unlike forward_ab's short replays)
OV='{"2048,10240,1280,0,1": [23256, 160, 1], "8192,5120,640,0,1": [23256, 160, 1]}'
for i in 1 2; do
  IMH_TUNING_OVERRIDE="$OV" python bench.py --steps 4 --warmup 2 --no-cpu-baseline --in-flight 1 --stacked 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ff.net.0 on 23256 x 160:', d['value'], 'images/sec', d['ms_per_step'], 'ms per denoise', d['config']['clocks_under_load'])"
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline --in-flight 1 --stacked 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ff.net.0 on 26256 x 320:', d['value'], 'images/sec', d['ms_per_step'], 'ms per denoise', d['config']['clocks_under_load'])"
done > $out/r06_bench_ab_geglu_w16.txt 2>&1
python tools/forward_ab.py --rounds 4 --configs base,x1 --stacked 4 > $out/r06_forward_ab_xattn_wide_s4.json 2> $out/r06_forward_ab_xattn_wide_s4.log
ls -la $out | grep r06_
