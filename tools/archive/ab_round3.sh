#!/bin/bash
# Same-box A/B of round 3's switches (one gpurun call): writes gpurun_out/r3_ab_switches.jsonl, one bench line per arm.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3_ab_switches.jsonl; : > $O
run() { tag=$1; shift; env "$@" python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = {n.split(' (')[0].split('/')[0]: {'ms': round(v['ms_total'], 1), 'rate': round(v.get('TFLOP/s', v.get('GB/s', 0)))} for n, v in d.get('kernels', {}).items()}
print(json.dumps({'arm': '$tag', 'config': '$CFG', 'frames_per_s': round(d['value'], 2), 'ms_per_video': round(d['ms_per_step'], 1), 'dit_step_ms': round(d.get('dit_step_ms', 0), 2), 'vae_decode_ms': round(d.get('vae_decode_ms', 0), 1), 'kernels': k}))" >> $O; }
CFG=c2
run default A=1
run LTX_GEMM_ASM16=0 LTX_GEMM_ASM16=0
run LTX_CONV_HALO_PIPE=0 LTX_CONV_HALO_PIPE=0
run "round-2 kernels (both off)" LTX_GEMM_ASM16=0 LTX_CONV_HALO_PIPE=0
run default-again A=1
CFG=c5
run default A=1
run LTX_ATTN_Q128=0 LTX_ATTN_Q128=0
CFG=c1
run default A=1
run "LTX_GEMM_SPLIT_SMALLM=0 LTX_GEMM_BIG_CONV_MINM=1024" LTX_GEMM_SPLIT_SMALLM=0 LTX_GEMM_BIG_CONV_MINM=1024
cat $O | cut -c1-230
