#!/bin/bash
# Round 5: C1 decode - the fused pixel norm forces conv_halo<256> (78 tiles = 0.3 rounds) on the 256-channel stage; A/B against the unfused form.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5k; mkdir -p $O
cd $R
J=$O/ab.jsonl; : > $J
run() { tag=$1; shift; env "$@" python3 $R/bench.py --config $CFG --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = {n.split(' (')[0].split('/')[0]: {'ms': round(v['ms_total'], 2), 'avg_us': round(1e3 * v['avg_ms'], 1)} for n, v in d.get('kernels', {}).items()}
print(json.dumps({'arm': '$tag', 'config': '$CFG', 'frames_per_s': round(d['value'], 2), 'ms_per_video': round(d['ms_per_step'], 2), 'dit_step_ms': round(d.get('dit_step_ms', 0), 3), 'vae_decode_ms': round(d.get('vae_decode_ms', 0), 2), 'kernels': k}))" >> $J; }
CFG=c1
run default A=1
run "vae_fuse_norm=0" LTX_OPTIONS=vae_fuse_norm=0
run default-again A=1
run "vae_fuse_norm=0 again" LTX_OPTIONS=vae_fuse_norm=0
cut -c1-600 $J
